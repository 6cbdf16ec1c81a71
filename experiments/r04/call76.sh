#!/bin/bash
# kernel stats of the 1 GiB monolithic decode + encode (tools/mono_bench.py), for profiles/r04_mono_1GiB_kernel_stats.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_mono_r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o mono -- python3 $R/tools/mono_bench.py --cases packed8_runs_1g --reps 5 > $O/bench.txt 2>&1
cd $R; tail -2 $O/bench.txt | cut -c1-250
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/prof_mono_r04/**/mono_kernel_stats.csv',recursive=True):
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:-float(r['TotalDurationNs']))
    with open('gpurun_out/prof_mono_r04/summary.txt','w') as o:
        o.write('rocprofv3 --kernel-trace --stats -- python3 tools/mono_bench.py --cases packed8_runs_1g --reps 5   (1 GiB rle8_packed monolithic stream: decode and many-lane encode)\n')
        o.write('%-90s %6s %12s\n'%('kernel','calls','avg us'))
        for r in rows[:24]:
            line='%-90s %6s %12.1f'%(r['Name'][:90], r['Calls'], float(r['AverageNs'])/1000); o.write(line+'\n'); print(line)
PY
