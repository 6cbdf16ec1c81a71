// inst_ppSw.hip -- instantiations of the windowed position-parallel encoder of 1 .. 8 byte symbols (hsrle_encodeSpw.hip.h): the codecs of inst_ppS.hip, blocks above 4 KiB
#include "hsrle_launch.h"
#include "hsrle_encodeSpw.hip.h"

namespace hsrle {

template <int FAM, int S, int AL>
static hipError_t ppS_launch(const PpwArgs &a, int phase, hipStream_t st)
{
  if (phase == 0)
    hipLaunchKernelGGL((k_encodeS_ppw_scan<FAM, S, AL>), dim3(a.nUnits), dim3(64), 0, st, a);
  else
    hipLaunchKernelGGL((k_encodeS_ppw_emit<FAM, S, AL>), dim3(a.nWindows), dim3(64), 0, st, a);
  return hipGetLastError();
}

// codec ids 6 + 8 * w + v: w = width index (16, 24, 32, 48, 64 bit), v = 0 sym, 1 sym_packed, 4 byte, 5 byte_packed
template <int S, int W>
static void reg_width(PpwLaunch *pp)
{
  pp[6 + 8 * W + 0] = ppS_launch<PLAIN, S, 1>;
  pp[6 + 8 * W + 1] = ppS_launch<PACKED, S, 1>;
  pp[6 + 8 * W + 4] = ppS_launch<PLAIN, S, 0>;
  pp[6 + 8 * W + 5] = ppS_launch<PACKED, S, 0>;
}

// ... v = 2 3symlut_sym, 6 3symlut_byte: symbols of 3 bytes and more (hsrle_encodeSp.hip.h)
template <int S, int W>
static void reg_lut3(PpwLaunch *pp)
{
  pp[6 + 8 * W + 2] = ppS_launch<LUT3, S, 1>;
  pp[6 + 8 * W + 6] = ppS_launch<LUT3, S, 0>;
}

// Short family, ids 54 + 8 * w + v: v = 0 sym_short, 1 1symlut_sym_short, 4 byte_short, 5 1symlut_byte_short (no list / a one-symbol list: the chain of the
// emit decisions runs through (lastRLE, last stored symbol) as it does for plain / Packed)
template <int S, int W>
static void reg_short(PpwLaunch *pp)
{
  pp[54 + 8 * W + 0] = ppS_launch<SHORT0, S, 1>;
  pp[54 + 8 * W + 1] = ppS_launch<SHORT1, S, 1>;
  pp[54 + 8 * W + 4] = ppS_launch<SHORT0, S, 0>;
  pp[54 + 8 * W + 5] = ppS_launch<SHORT1, S, 0>;
}

// ... v = 2 3symlut_sym_short, 6 3symlut_byte_short: 6 and 8 byte symbols (every run stored: hsrle_encodeSp.hip.h)
template <int S, int W>
static void reg_short3(PpwLaunch *pp)
{
  pp[54 + 8 * W + 2] = ppS_launch<SHORT3, S, 1>;
  pp[54 + 8 * W + 6] = ppS_launch<SHORT3, S, 0>;
}

void register_ppSw(PpwLaunch *pp)
{
  reg_short3<6, 3>(pp); reg_short3<8, 4>(pp);
  pp[50] = ppS_launch<SHORT0, 1, 0>;      // rle8_multi_short
  pp[51] = ppS_launch<SHORT1, 1, 0>;      // rle8_1symlut_short
  reg_short<2, 0>(pp); reg_short<3, 1>(pp); reg_short<4, 2>(pp); reg_short<6, 3>(pp); reg_short<8, 4>(pp);
  reg_width<2, 0>(pp); reg_width<3, 1>(pp); reg_width<4, 2>(pp); reg_width<6, 3>(pp); reg_width<8, 4>(pp);
  reg_lut3<3, 1>(pp); reg_lut3<4, 2>(pp); reg_lut3<6, 3>(pp); reg_lut3<8, 4>(pp);
}

} // namespace hsrle
