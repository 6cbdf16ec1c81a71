#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for m in eager_zero_dst graph_keep_dst graph_zero_dst; do timeout 120 python experiments/r04/graph_single_repro2.py rle8_single $m 2>&1 | grep -v "amdgpu.ids\|^Extension" | tail -4 | cut -c1-150; done
