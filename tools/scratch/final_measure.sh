cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3_final; mkdir -p $O
( bash tools/traffic.sh r3final > $O/traffic.log 2>&1; cp gpurun_out/traffic_r3final/traffic.json $O/traffic.json; cp $O/traffic.json profiles/r03_traffic.json
  timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- python3 $GRAFT_REPO_ROOT/bench.py --no-extras > $O/stats.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/frame -o f -- python3 $GRAFT_REPO_ROOT/tools/frame_prof.py rle64_3symlut_byte > $O/frame.log 2>&1
  cd $GRAFT_REPO_ROOT
  timeout 1500 python tools/small_container_sweep.py > $O/small_containers.md 2> $O/small.err
  tail -3 $O/small_containers.md
) > $O/log.txt 2>&1
tail -12 $O/log.txt
