#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
/opt/rocm/bin/hipcc -O2 -pthread tools/ubench/host_copy.hip -o /tmp/host_copy && timeout 300 /tmp/host_copy
timeout 600 python tools/dropin_bench.py 2>&1 | head -4
