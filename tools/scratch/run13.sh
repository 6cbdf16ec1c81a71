cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r3u
O=$PWD/gpurun_out/r3u
( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $GRAFT_REPO_ROOT/tools/mono_enc_bench.py rle8_single 1 > $O/prof.log 2>&1; cd $GRAFT_REPO_ROOT
  python3 - <<PY
import csv,glob
for f in glob.glob("$O/prof/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e6,3), round(float(r['TotalDurationNs'])/1e6,2))
PY
) > gpurun_out/r3u/log.txt 2>&1
cat gpurun_out/r3u/log.txt | head -30
