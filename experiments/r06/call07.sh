#!/bin/bash
# round 6, call 7: headline decode with 64-byte steps into a 64-byte ring, 128-byte tile rows (12 waves per CU, whole-line flushes) against the shipped form, same box
mkdir -p gpurun_out/r06_c07
REPS=3 bash tools/ab.sh q64r64 2>&1 | tee gpurun_out/r06_c07/ab_q64r64.log
