#!/bin/bash
# round 6, call 15: headline encode, pass 1 with the candidates' symbols from an LDS copy of the block instead of a global gather
mkdir -p gpurun_out/r06_c15
REPS=3 bash tools/ab.sh bpw2 bpw4 2>&1 | tee gpurun_out/r06_c15/ab_bpw.log
