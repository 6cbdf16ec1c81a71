#!/bin/bash
# round 6, call 5: headline decode with 64-byte tile rows / steps (12 waves per CU) against the shipped 128-byte form, same box
mkdir -p gpurun_out/r06_c05
REPS=3 bash tools/ab.sh t64 2>&1 | tee gpurun_out/r06_c05/ab_t64.log
