// inst_pp8w.hip -- instantiations of the windowed position-parallel 8 bit encoder (hsrle_encode8pw.hip.h): rle8_multi, rle8_packed_multi with blocks above
// 4 KiB, and the chunks of their monolithic streams
#include "hsrle_launch.h"
#include "hsrle_encode8pw.hip.h"

namespace hsrle {

template <int FAM>
static hipError_t ppw_launch(const PpwArgs &a, int phase, hipStream_t st)
{
  if (phase == 0)
    hipLaunchKernelGGL((k_encode8_ppw_scan<FAM>), dim3(a.nUnits), dim3(64), 0, st, a);
  else
    hipLaunchKernelGGL((k_encode8_ppw_emit<FAM>), dim3(a.nWindows), dim3(64), 0, st, a);
  return hipGetLastError();
}

void register_pp8w(PpwLaunch *ppw)
{
  ppw[0] = ppw_launch<PLAIN>;
  ppw[1] = ppw_launch<PACKED>;
}

} // namespace hsrle
