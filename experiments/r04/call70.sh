#!/bin/bash
# GPU suite with poisoned workspaces (split decode, rle8m, mono encode / decode, graph capture)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( time timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^Extension modules" | tail -12 ) 2>&1 | tail -16
