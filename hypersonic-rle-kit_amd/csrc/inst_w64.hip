// 64 bit symbols: rle64_{sym,byte}[_packed], rle64_{3,7}symlut_{sym,byte}  (reference: src/rle.h)
#define HSRLE_W 64
#define HSRLE_S 8
#define HSRLE_BASE 38
#include "hsrle_inst_generic.inc"

#ifdef HSRLE_RL_STAMPS
// diagnostic builds: phase cycles of the run list encoders of this width (hsrle_encodeSr.hip.h); reset = 1 zeroes the counters
extern "C" int hsrle_debug_rl_stamps(unsigned long long *out8, int reset)
{
  unsigned long long z[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(hsrle::g_rl_stamps), sizeof(z)) != hipSuccess) return -1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(hsrle::g_rl_stamps), z, sizeof(z)) != hipSuccess) return -1;
  return 0;
}
#endif
