#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
( timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 )
