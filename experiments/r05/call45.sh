#!/bin/bash
# round 5 call 45: index walks with the first touches of memory lines made together (every E hops every lane asks for the next L lines): 1 GiB rle8_packed mono decode
cd /root/repo
for v in waoff default wae4l2 wae8l4 wae16l4 wae16l6; do
  if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=/root/repo/variants/libhsrle_$v.so; fi
  echo "== $v"; python tools/mono_bench.py --cases packed8_runs_1g,packed8_video_88m --reps 5 2>&1 | grep -v amdgpu | cut -c1-200
done
unset HSRLE_LIB
ROWS=8 bash tools/prof_script.sh mono_dec_ahead tools/mono_bench.py --cases packed8_runs_1g --reps 5 | grep -i "index\|decode"
