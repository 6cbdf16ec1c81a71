#!/bin/bash
# feasibility: the encode kernel with an in-kernel copy of every wave's 64 streams (fused compaction) -- what does the kernel cost then?
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4fc
cd /tmp && export TMPDIR=/tmp
for v in default fcopy; do
  if [ $v = default ]; then unset HSRLE_LIB; else export HSRLE_LIB=$R/variants/libhsrle_$v.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/r4fc/$v" -o f -- python3 "$R/bench.py" --no-cpu --steps 6 --warmup 2 > "$R/gpurun_out/r4fc/$v.log" 2>&1
  echo "== $v"; grep '^{' "$R/gpurun_out/r4fc/$v.log" | python3 -c "
import sys,json
for l in sys.stdin:
    j=json.loads(l); print('dec %.3f ms | enc %.3f ms %.0f GiB/s | ok %s'%(j['ms_per_step'], j['encode']['ms'], j['encode']['value'], j['bit_exact']))"
  python3 - "$R/gpurun_out/r4fc/$v" <<'PY'
import csv,glob,sys
for f in glob.glob(sys.argv[1]+'/**/f_kernel_stats.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('k_encode8_blocks','k_compact','k_tile')): print('   %-70s calls %4s avg %10.1f us'%(r['Name'][:70], r['Calls'], float(r['AverageNs'])/1000))
PY
done
