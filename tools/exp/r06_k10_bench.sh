#!/bin/bash
# K10 per-launch costs (tools/k10_bench.py), with the knobs build's attribution variants, then the unit tests.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_k10
mkdir -p "$out"
timeout -k 10 300 python -m pytest tests/test_fused_block_gpu.py -x -q > "$out/pytest_block.log" 2>&1; rc=$?
echo "pytest_block rc=$rc"; tail -5 "$out/pytest_block.log"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/k10_bench.py "$out/k10_bench.json" 2>&1 | tee "$out/k10_bench.txt"
for dbg in 1 2 3 4; do
  echo "URSA_K10_DBG=$dbg (16x16x32 k10_fwd only: 1 = prologue only, 2 = sums only, 3 = sums accumulated but not handed over, 4 = slots stored, no final add)" | tee -a "$out/k10_bench.txt"
  URSA_K10_KNOBS=1 URSA_K10_DBG=$dbg timeout -k 10 300 python3 tools/k10_bench.py "$out/k10_bench_dbg$dbg.json" 2>/dev/null | grep 16x16 | tee -a "$out/k10_bench.txt"
done
