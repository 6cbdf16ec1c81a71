// inst_pp8.hip -- instantiations of the position-parallel 8 bit encoder (hsrle_encode8p.hip.h): rle8_multi, rle8_packed_multi
#include "hsrle_launch.h"
#include "hsrle_encode8p.hip.h"

namespace hsrle {

template <int FAM>
static hipError_t pp_launch(const PpArgs &a, int phase, hipStream_t st)
{
  PpScratch sc;
  sc.recs = (uint32_t *)a.scratch;
  sc.recStride = pp_record_stride(a.B);
  sc.recCount = sc.recs + (uint64_t)sc.recStride * a.nBlocks + 64u;   // (+ 64: the last block's lanes read 64 words from its first record on)
  sc.stamps = a.scratch + ((pp_scratch_bytes(a.nBlocks, a.B) + 255ull) & ~255ull);   // (diagnostic builds: 128 bytes per block behind the scratch, in the caller's slot area)
  if (phase == 0)
    hipLaunchKernelGGL((k_encode8_pp<FAM, 0>), dim3((a.nBlocks + kPp8Bpw - 1u) / kPp8Bpw), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.sizes, a.offsets, a.payload, sc);
  else
    hipLaunchKernelGGL((k_encode8_pp<FAM, 1>), dim3((a.nBlocks + kPp8Bpw - 1u) / kPp8Bpw), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.sizes, a.offsets, a.payload, sc);
  return hipGetLastError();
}

void register_pp8(PpLaunch *pp)
{
  pp[0] = pp_launch<PLAIN>;
  pp[1] = pp_launch<PACKED>;
}

} // namespace hsrle
