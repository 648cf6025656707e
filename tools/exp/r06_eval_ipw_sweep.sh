#!/bin/bash
# images per workgroup of the evaluation-form units re-swept under the two-waves-per-SIMD register cap (4,096 and 10,000 rows)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_eval_ipw
mkdir -p "$out"
for rows in 4096 10000; do
  for ipw in 0 2 4 8 16 32; do
    echo "rows=$rows ipw=$ipw (0 = the plan's own)" | tee -a "$out/sweep.txt"
    if [ $ipw -eq 0 ]; then
      URSA_EVAL_ROWS=$rows URSA_K10_KNOBS=1 timeout -k 10 200 python3 tools/k10_eval_bench.py "$out/r${rows}_ipw$ipw.json" 2>/dev/null | tee -a "$out/sweep.txt"
    else
      URSA_EVAL_ROWS=$rows URSA_K10_KNOBS=1 URSA_K8_EVAL_IPW=$ipw timeout -k 10 200 python3 tools/k10_eval_bench.py "$out/r${rows}_ipw$ipw.json" 2>/dev/null | tee -a "$out/sweep.txt"
    fi
  done
done
