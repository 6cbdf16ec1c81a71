// inst_pp128.hip -- instantiations of the position-parallel 128 bit encoder (hsrle_encode128p.hip.h): rle128_sym, rle128_sym_packed, rle128_byte, rle128_byte_packed
#include "hsrle_launch.h"
#include "hsrle_encode128p.hip.h"

namespace hsrle {

template <bool PK, int AL>
static hipError_t pp128_launch(const PpArgs &a, int phase, hipStream_t st)
{
  PpScratch sc;
  sc.recs = (uint32_t *)a.scratch;
  sc.recStride = pp_record_stride(a.B);
  sc.recCount = sc.recs + (uint64_t)sc.recStride * a.nBlocks + 64u;   // (+ 64: the last block's lanes read 64 words from its first record on)
  sc.stamps = nullptr;
  if (phase == 0)
    hipLaunchKernelGGL((k_encode128_pp<PK, AL, 0>), dim3(a.nBlocks), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.sizes, a.offsets, a.payload, sc);
  else
    hipLaunchKernelGGL((k_encode128_pp<PK, AL, 1>), dim3(a.nBlocks), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.sizes, a.offsets, a.payload, sc);
  return hipGetLastError();
}

void register_pp128(PpLaunch *pp)
{
  pp[46] = pp128_launch<false, 1>;   // rle128_sym
  pp[47] = pp128_launch<true, 1>;    // rle128_sym_packed
  pp[48] = pp128_launch<false, 0>;   // rle128_byte
  pp[49] = pp128_launch<true, 0>;    // rle128_byte_packed
}

} // namespace hsrle
