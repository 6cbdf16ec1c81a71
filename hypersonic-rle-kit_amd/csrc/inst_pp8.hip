// inst_pp8.hip -- instantiations of the position-parallel 8 bit encoder (hsrle_encode8p.hip.h): rle8_multi, rle8_packed_multi
#include "hsrle_launch.h"
#include "hsrle_encode8p.hip.h"

namespace hsrle {

// phase 3: the single-pass launch (a.ctrl: the look-back words, zeroed by the caller -- pp_ctrl_bytes());  phases 0 / 1: sizes, then emission at a.offsets
template <int FAM>
static hipError_t pp_launch(const PpArgs &a, int phase, hipStream_t st)
{
  const uint32_t groups = (a.nBlocks + kPpGroup - 1u) / kPpGroup;
  PpLookBack lb;
  lb.status = a.ctrl;
  lb.grpAcc = (unsigned long long *)(a.ctrl + kPpCtrlWords);
  lb.grpPref = lb.grpAcc + groups;
  lb.sizeW = (uint32_t *)(lb.grpPref + groups);
  if (phase == 3)
    hipLaunchKernelGGL((k_encode8_pp<FAM, 3>), dim3((a.nBlocks + 8u * kPpGroup - 1u) / (8u * kPpGroup) * (8u * kPpGroup)), dim3(64), 0, st,   // (whole rounds of 8 groups: see the kernel's block mapping)
                       a.in, a.U, a.B, a.nBlocks, a.sizes, a.offsets, a.payload, lb);
  else if (phase == 0)
    hipLaunchKernelGGL((k_encode8_pp<FAM, 0>), dim3(a.nBlocks), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.sizes, a.offsets, a.payload, lb);
  else
    hipLaunchKernelGGL((k_encode8_pp<FAM, 1>), dim3(a.nBlocks), dim3(64), 0, st, a.in, a.U, a.B, a.nBlocks, a.sizes, a.offsets, a.payload, lb);
  return hipGetLastError();
}

void register_pp8(PpLaunch *pp)
{
  pp[0] = pp_launch<PLAIN>;
  pp[1] = pp_launch<PACKED>;
}

} // namespace hsrle
