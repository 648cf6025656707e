#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_eval
mkdir -p "$out"
timeout -k 10 300 python3 tools/k10_eval_bench.py "$out/k10_eval_bench.json" 2>/dev/null | tee "$out/k10_eval_bench.txt"
for ipw in 2 4 8 16; do
  echo "URSA_K8_EVAL_IPW=$ipw" | tee -a "$out/k10_eval_bench.txt"
  URSA_K10_KNOBS=1 URSA_K8_EVAL_IPW=$ipw timeout -k 10 300 python3 tools/k10_eval_bench.py "$out/k10_eval_bench_ipw$ipw.json" 2>/dev/null | tee -a "$out/k10_eval_bench.txt"
done
