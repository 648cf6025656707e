#!/bin/bash
# C5 under rocprofv3 --kernel-trace --stats (3 proposals per chain) + the C-ABI host and HMC tests
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_c5_prof
mkdir -p "$out"
timeout -k 10 600 python -m pytest tests/test_c_abi_host.py tests/test_samplers_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 "$out/pytest.log"
[ $rc -ne 0 ] && exit $rc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 "$R/bench.py" --config c5 --c5-batch 1024 --steps 3 --warmup 0 --detail-out "$R/$out/prof_detail.json" > "$R/$out/prof_line.json" 2> "$R/$out/prof.err"; echo "c5 prof rc=$?"
python3 "$R/tools/prof_summary.py" /tmp/prof_c5 "$R/$out/r06_c5_kernel_stats.csv" | head -32
