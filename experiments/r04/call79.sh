#!/bin/bash
# differential stress with the split decode and the drop-in paths added (tools/gpu_stress.py), fresh seeds
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 500 python tools/gpu_stress.py 300 901 2>&1 | grep -v amdgpu.ids | tail -6
timeout 500 python tools/gpu_stress.py 300 902 2>&1 | grep -v amdgpu.ids | tail -6
