#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
bash tools/exp/r06_k10_check.sh || exit $?
timeout -k 10 300 python3 tools/k10_bench.py gpurun_out/r06_k10/k10_bench.json 2>/dev/null | tee gpurun_out/r06_k10/k10_bench.txt
