cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3w
timeout 2400 python tools/sweep.py 8192 4096 video > gpurun_out/r3w/sweep.md 2> gpurun_out/r3w/sweep.err
tail -5 gpurun_out/r3w/sweep.md | cut -c1-200; tail -3 gpurun_out/r3w/sweep.err
