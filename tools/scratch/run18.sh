cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3D
O=$PWD/gpurun_out/r3D
( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/tools/scratch/frame_enc.py > $O/prof.log 2>&1; cd $GRAFT_REPO_ROOT
  python3 - <<PY
import csv,glob
for f in glob.glob("$O/prof/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:80], r['Calls'], round(float(r['AverageNs'])/1e3,1), "us  total", round(float(r['TotalDurationNs'])/1e3/10,1), "us per call")
PY
) > gpurun_out/r3D/log.txt 2>&1
cat gpurun_out/r3D/log.txt | head -24
