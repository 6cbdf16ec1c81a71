#!/bin/bash
# what the driver runs at round end: smoke(), the default bench line (with the committed traffic of this build), and its N=1 distributed leg
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -3 ) 2>&1 | tail -6
( time python bench.py 2>/dev/null | tail -1 | python -c "
import sys,json
j=json.loads(sys.stdin.read()); r=j['roofline']
print('value',j['value'],j['unit'],'ms',j['ms_per_step'],'frac',r['frac'],'traffic',r['traffic'],'cpu',j['cpu_baseline']['value'],j['cpu_baseline']['kind'],'enc',j['encode']['value'],j['encode']['roofline']['frac'],'traffic enc',j['encode']['roofline'].get('traffic'),'build',j['library_build_id'])
print('extras', list(j['extras'].keys()), 'config5 rows', len(j['extras']['config5']['rows']))
" ) 2>&1 | tail -6
