#!/bin/bash
# Build the in-tree libraries (they travel with the snapshot), then hand the command to gpurun.
#   tools/gpu.sh 1150 'bash tools/r06_k10_check.sh'
set -e
cd "$(dirname "$0")/.."
make -s -j2 -C ursabench_amd/csrc libursa_hip.so libursa_hip_knobs.so
make -s -C oracle liboracle.so
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
