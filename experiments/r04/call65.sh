#!/bin/bash
# where the many-lane monolithic encode stands per family (1 GiB, device side), and the monolithic decode of the same streams
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
timeout 900 python tools/mono_enc_bench.py rle8_multi,rle8_packed_multi,rle8_3symlut,rle8_7symlut,rle8_single,rle8_packed_single,rle8_single_short,rle16_sym,rle24_byte_packed,rle32_3symlut_sym,rle48_7symlut_byte,rle64_3symlut_byte,rle64_byte_short,rle128_sym,rle128_byte_packed 1 2>&1 | grep -v amdgpu.ids
timeout 600 python tools/mono_bench.py 2>&1 | grep -v amdgpu.ids | tail -12
