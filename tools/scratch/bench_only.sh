cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3_final
timeout 900 python bench.py > gpurun_out/r3_final/bench.json 2> gpurun_out/r3_final/bench.err
tail -c 1500 gpurun_out/r3_final/bench.json
