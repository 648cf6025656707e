#!/bin/bash
# knobs libraries with other values of U (members whose loads are in flight per lane) for tools/exp/k5_waves_ab.py
set -e
cd "$(dirname "$0")/../../ursabench_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fPIC -shared -DURSA_DEBUG_KNOBS"
for u in 1 3 4; do
  /opt/rocm/bin/hipcc $F -DURSA_BMA_U_EPL8=$u -o ../../tools/exp/_libs/libursa_hip_knobs_U$u.so ursa_kernels.hip ursa_bn.hip &
done
wait
