#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export HSRLE_LIB=$PWD/variants/libhsrle_old.so
bash tools/pmc_kernel.sh enc_old k_encode8_blocks -- tools/enc_time.py rle8_packed_multi > /dev/null 2>&1
unset HSRLE_LIB
bash tools/pmc_kernel.sh enc_new k_encode8_blocks -- tools/enc_time.py rle8_packed_multi > /dev/null 2>&1
cd "${GRAFT_REPO_ROOT:-.}"
paste gpurun_out/pmc_enc_old/summary.txt gpurun_out/pmc_enc_new/summary.txt | awk '{printf "%-26s old %14.0f new %14.0f  %+.1f%%\n", $1, $5, $12, ($12/$5-1)*100}'
