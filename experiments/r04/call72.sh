#!/bin/bash
# packet list decode (walk + expand) on BIG containers against the per-lane block decoder: is the two-kernel path faster where packets are dense?
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for c in "rle64_3symlut_byte video" "rle8_packed_multi video" "rle16_7symlut_byte video" "rle48_7symlut_byte video" "rle32_sym video" "rle8_packed_multi runs" "rle64_3symlut_byte runs" "rle24_7symlut_byte runs"; do
  set -- $c
  timeout 300 python tools/split_bench.py --codec $1 --synth $2 --size $((2<<30)) --block 4096 --subs 1,4096 --reps 5 2>&1 | grep -v amdgpu.ids | cut -c1-150
done
