#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_bwd_ipw
mkdir -p "$out"
for v in 1 2 4; do
  echo "URSA_BWD_IPW=$v" | tee -a "$out/ab.txt"
  URSA_K10_KNOBS=1 URSA_BWD_IPW=$v timeout -k 10 300 python3 tools/k10_bench.py "$out/k10_ipw$v.json" 2>/dev/null | tee -a "$out/ab.txt"
done
