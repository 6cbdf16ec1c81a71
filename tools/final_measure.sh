#!/bin/bash
# usage (GPU box, via gpurun): bash tools/final_measure.sh <tag>     e.g. r04   -> gpurun_out/<tag>_final/
# The judged measurements of one build in one go: PMC traffic of the headline kernels (tools/traffic.sh), the default bench line, rocprofv3
# kernel stats of the same command, kernel stats of the BASELINE config-3 frame (encode, split decode), the small-container table.  Nothing
# under profiles/ is touched: copy what is to be committed with `python tools/collect_profiles.py gpurun_out/<tag>_final <tag>`.
set -u
tag=${1:?usage: final_measure.sh <tag>}
R="${GRAFT_REPO_ROOT:?run this on the GPU box (gpurun)}"
cd "$R" || exit 1
export TMPDIR=/tmp
O="$R/gpurun_out/${tag}_final"; mkdir -p "$O"
{
  bash tools/traffic.sh "${tag}final" > "$O/traffic.log" 2>&1; cp "gpurun_out/traffic_${tag}final/traffic.json" "$O/traffic.json"
  cp "$O/traffic.json" "profiles/${tag}_traffic.json"   # (on the box's copy of the tree: bench.py takes `roofline.traffic` from the PMC file of THIS build)
  timeout 900 python bench.py > "$O/bench.json" 2> "$O/bench.err"; tail -c 400 "$O/bench.json"
  cd /tmp || exit 1
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -o k -- python3 "$R/bench.py" --no-extras > "$O/stats.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/frame" -o f -- python3 "$R/tools/frame_prof.py" rle64_3symlut_byte > "$O/frame.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/split" -o s -- python3 "$R/tools/split_bench.py" > "$O/split.log" 2>&1
  cd "$R" || exit 1
  timeout 1500 python tools/small_container_sweep.py > "$O/small_containers.md" 2> "$O/small.err"
  tail -3 "$O/small_containers.md"
} > "$O/log.txt" 2>&1
tail -12 "$O/log.txt"
