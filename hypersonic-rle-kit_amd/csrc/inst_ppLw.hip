// inst_ppLw.hip -- instantiations of the windowed position-parallel LUT encoder (hsrle_encodeLpw.hip.h): the codecs of inst_ppL.hip, blocks above 4 KiB
#include "hsrle_launch.h"
#include "hsrle_encodeLpw.hip.h"

namespace hsrle {

template <int FAM, int S, int AL>
static hipError_t ppL_launch(const PpwArgs &a, int phase, hipStream_t st)
{
  if (phase == 0)
    hipLaunchKernelGGL((k_encodeL_ppw_scan<FAM, S, AL>), dim3(a.nUnits), dim3(64), 0, st, a);
  else
    hipLaunchKernelGGL((k_encodeL_ppw_emit<FAM, S, AL>), dim3(a.nWindows), dim3(64), 0, st, a);
  return hipGetLastError();
}

// codec ids 6 + 8 * w + v: w = width index (16, 24, 32, 48, 64 bit), v = 3 7symlut_sym, 7 7symlut_byte
template <int S, int W>
static void reg_lut7(PpwLaunch *pp)
{
  pp[6 + 8 * W + 3] = ppL_launch<LUT7, S, 1>;
  pp[6 + 8 * W + 7] = ppL_launch<LUT7, S, 0>;
}

// Short family, ids 54 + 8 * w + v: v = 2 3symlut_sym_short, 3 7symlut_sym_short, 6 3symlut_byte_short, 7 7symlut_byte_short (the three-symbol list of 6 / 8 byte
// symbols has its closed form in hsrle_encodeSp.hip.h)
template <int S, int W>
static void reg_short37(PpwLaunch *pp)
{
  if constexpr (S <= 4) { pp[54 + 8 * W + 2] = ppL_launch<SHORT3, S, 1>; pp[54 + 8 * W + 6] = ppL_launch<SHORT3, S, 0>; }
  pp[54 + 8 * W + 3] = ppL_launch<SHORT7, S, 1>;
  pp[54 + 8 * W + 7] = ppL_launch<SHORT7, S, 0>;
}

void register_ppLw(PpwLaunch *pp)
{
  // (rle8_3symlut_short / rle8_7symlut_short stay with the ring / run list encoders: with 8 bit symbols every PAIR of equal bytes is a candidate -- a second round for
  //  a handful of them on run data, 280 candidates per block on video-shaped data -- measured 1 355 / 632 and 956 / 705 GiB/s against 1 134 / 1 221 and 1 126 / 1 173)
  reg_short37<2, 0>(pp); reg_short37<3, 1>(pp); reg_short37<4, 2>(pp); reg_short37<6, 3>(pp); reg_short37<8, 4>(pp);
  pp[2] = ppL_launch<LUT3, 1, 0>;         // rle8_3symlut
  pp[3] = ppL_launch<LUT7, 1, 0>;         // rle8_7symlut
  pp[6 + 2] = ppL_launch<LUT3, 2, 1>;     // rle16_3symlut_sym
  pp[6 + 6] = ppL_launch<LUT3, 2, 0>;     // rle16_3symlut_byte
  reg_lut7<2, 0>(pp); reg_lut7<3, 1>(pp); reg_lut7<4, 2>(pp); reg_lut7<6, 3>(pp); reg_lut7<8, 4>(pp);
}

} // namespace hsrle
