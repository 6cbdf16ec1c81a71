#!/bin/bash
# round 6, call 29: first run of the windowed encoder of the 1 .. 8 byte symbol codecs (blocks above 4 KiB) against the oracle
mkdir -p gpurun_out/r06_c29
timeout 1700 python tools/probe_ppws.py "" 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_c29/probe.log | grep -v " done, " | head -80
tail -3 gpurun_out/r06_c29/probe.log
