#!/bin/bash
# compile-time variants of the K6 held forms for tools/exp/bn_held_fwd_ab.py: threads per workgroup, float4 per thread held
# in registers (EPT) and in LDS (LX), forward (F) and backward (B)
set -e
cd "$(dirname "$0")/../../ursabench_amd/csrc"
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fPIC -shared"
L=../../tools/exp/_libs
mkdir -p $L
rm -f $L/libursa_hip_heldfwd_*.so
v() { name=$1; shift; /opt/rocm/bin/hipcc $F "$@" -o $L/libursa_hip_heldfwd_$name.so ursa_kernels.hip ursa_bn.hip; }
# (with the knobs compiled in: URSA_BN_HELD_MIN_MIB forces the size from which the held form is taken)
K=-DURSA_DEBUG_KNOBS
v F256e16l9_4waves $K -DURSA_HELD_FWD_BLOCK=256 -DURSA_HELD_FWD_EPT=16 -DURSA_HELD_FWD_LX=9 -DURSA_HELD_FWD_MIN_WAVES=4 &
v F512e16l9 $K -DURSA_HELD_FWD_EPT=16 &
v F512e32l16 $K -DURSA_HELD_FWD_LX=16 &
v F1024e16l9 $K -DURSA_HELD_FWD_BLOCK=1024 -DURSA_HELD_FWD_EPT=16 &
wait
v B512e12l6 $K -DURSA_HELD_BWD_EPT=12 -DURSA_HELD_BWD_LX=6 &
v B512e16l4 $K -DURSA_HELD_BWD_EPT=16 &
v B512e8l2 $K -DURSA_HELD_BWD_LX=2 &
v B1024e8l4 $K -DURSA_HELD_BWD_BLOCK=1024 &
wait
