#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
bash tools/prof_mono.sh packed8_runs_1g
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_mono/packed8_runs_1g/**/*kernel_stats.csv', recursive=True):
    rows=list(csv.DictReader(open(f)))
    for r in rows[:14]: print('%-90s calls %5s avg %10.1f us total %9.3f ms' % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
