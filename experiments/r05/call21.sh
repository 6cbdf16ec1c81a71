#!/bin/bash
# the position-parallel encoder as the encoder of every rle8_multi / rle8_packed_multi container of <= 4 KiB blocks: the whole GPU suite
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
