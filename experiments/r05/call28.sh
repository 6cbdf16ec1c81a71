#!/bin/bash
# 3 symbol LUT codecs of 3 .. 8 byte symbols on the position-parallel encoder: parity, 8 GiB times, the config-3 frame
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1500 python -m pytest tests/test_gpu_pp.py -x -q -k "lut3" 2>&1 | tail -12
for c in rle64_3symlut_byte rle24_3symlut_byte; do for k in 0 1; do timeout 300 python tools/enc_time.py $c $k 8 2>&1 | tail -1; done; done
timeout 300 python tools/frame_enc_time.py 2>&1 | tail -6
