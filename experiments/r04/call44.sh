#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for lib in variants/libhsrle_cur.so variants/libhsrle_base.so; do echo "== $lib"; env HSRLE_LIB=$lib timeout 300 python tools/enc_time.py rle8_packed_multi 2>&1 | grep -v amdgpu.ids | tail -2; done
echo "== default (global_window16 as before)"; timeout 300 python tools/enc_time.py rle8_packed_multi 2>&1 | grep -v amdgpu.ids | tail -2
