// inst_pp8s.hip -- instantiations of the position-parallel 8 bit Single encoder (hsrle_encode8sp.hip.h): rle8_single, rle8_packed_single, rle8_single_short
#include "hsrle_launch.h"
#include "hsrle_encode8sp.hip.h"

namespace hsrle {

template <int CODEC>
static hipError_t pps_launch(const PpArgs &a, int phase, hipStream_t st)
{
  PpScratch sc;
  sc.recs = (uint32_t *)a.scratch;
  sc.recStride = pp_record_stride(a.B);
  sc.recCount = sc.recs + (uint64_t)sc.recStride * a.nBlocks + 64u;   // (+ 64: the last block's lanes read 64 words from its first record on)
  sc.stamps = nullptr;
  if (phase == 0)
    hipLaunchKernelGGL((k_encode8s_pp<CODEC, 0>), dim3(a.nBlocks), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.sizes, a.offsets, a.payload, sc);
  else
    hipLaunchKernelGGL((k_encode8s_pp<CODEC, 1>), dim3(a.nBlocks), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.sizes, a.offsets, a.payload, sc);
  return hipGetLastError();
}

void register_pp8s(PpLaunch *pp)
{
  pp[4] = pps_launch<0>;               // rle8_single
  pp[5] = pps_launch<1>;               // rle8_packed_single
  pp[kSingleShort] = pps_launch<2>;    // rle8_single_short
}

} // namespace hsrle
