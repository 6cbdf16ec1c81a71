#!/bin/bash
# builds the micro-benchmarks of this directory for gfx950 (binaries are git-ignored; they travel to the GPU box with gpurun)
cd "$(dirname "$0")"
for f in *.hip; do /opt/rocm/bin/hipcc -O2 --offload-arch=${HSRLE_ARCH:-gfx950} "$f" -o "${f%.hip}" & done
wait
