#!/bin/bash
# round 6, call 21: first run of the windowed position-parallel encoder (blocks above 4 KiB, monolithic streams) against the oracle
mkdir -p gpurun_out/r06_c21
timeout 900 python tools/probe_ppw.py all 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c21/probe.log | tail -120
