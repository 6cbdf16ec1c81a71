#!/bin/bash
# position-parallel encoder of 2 .. 8 byte symbols (plain / Packed, 20 codecs): parity, then 8 GiB times
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1500 python -m pytest tests/test_gpu_pp.py -x -q 2>&1 | tail -12
for c in rle16_byte_packed rle32_byte_packed rle64_byte_packed rle64_sym; do timeout 300 python tools/enc_time.py $c 0 8 2>&1 | tail -1; done
timeout 300 python tools/enc_time.py rle64_byte_packed 1 8 2>&1 | tail -1
