#!/bin/bash
# monolithic (rle.h) encode of rle8_multi / rle8_packed_multi with one WAVE per chunk (position-parallel encoder, CHUNK mode): parity, then 1 GiB times
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 1500 python -m pytest tests/test_gpu_mono.py tests/test_gpu_big.py -x -q -k "rle8_multi or rle8_packed_multi or mono or packed8" 2>&1 | tail -6
timeout 600 python tools/mono_enc_bench.py rle8_packed_multi,rle8_multi 1 2>&1 | tail -6
