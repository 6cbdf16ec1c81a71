#!/bin/bash
# round 6, call 14: headline decode as parse (lane per block) + expand (wave per output chunk), experiment build, against the shipped kernel on the same box
mkdir -p gpurun_out/r06_c14
REPS=3 bash tools/ab.sh pe 2>&1 | tee gpurun_out/r06_c14/ab_pe.log
