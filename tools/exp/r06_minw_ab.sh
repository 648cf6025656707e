#!/bin/bash
# A/B: K8 / K10 / pair kernels compiled with __launch_bounds__(256, 2) (knobs library built with KNOBS_EXTRA=-DURSA_MINW=2)
# against the shipped library, same box, same process order.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_minw_ab
mkdir -p "$out"
for k in 0 1; do
  echo "knobs_lib=$k (1 = URSA_MINW=2)" | tee -a "$out/ab.txt"
  URSA_K10_KNOBS=$k timeout -k 10 300 python3 tools/k10_bench.py "$out/k10_lib$k.json" 2>/dev/null | tee -a "$out/ab.txt"
  URSA_K10_KNOBS=$k timeout -k 10 300 python3 tools/k10_eval_bench.py "$out/k10_eval_lib$k.json" 2>/dev/null | tee -a "$out/ab.txt"
done
