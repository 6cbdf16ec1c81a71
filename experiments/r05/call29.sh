#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for c in rle64_3symlut_byte rle64_byte_packed rle8_packed_multi; do timeout 300 python tools/frame_enc_time.py $c 2>&1 | tail -2; done
