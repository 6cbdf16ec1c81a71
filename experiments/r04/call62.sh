#!/bin/bash
# compiler scheduling flags for the 8 bit unit (headline decode + encode kernels): same-box A/B
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
REPS=2 bash tools/ab.sh ilp memc bias100 bias0 nopost o2 nocluster 2>&1 | grep -v "^$"
REPS=1 SKIP_DEFAULT= bash tools/ab.sh 2>&1 | grep "dec "
