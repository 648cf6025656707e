#!/bin/bash
# K13 / K14 tests, then C5 (PreResNet-164 HMC, 4 chains, 1,024 rows) with K14 on / off, alternating, and a kernel table
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_k14
mkdir -p "$out"
timeout -k 10 600 python -m pytest tests/test_fused_bottleneck_gpu.py tests/test_samplers_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 "$out/pytest.log"
[ $rc -ne 0 ] && exit $rc
for v in 1 0 1 0; do
  URSA_K14=$v timeout -k 10 500 python3 bench.py --config c5 --c5-batch 1024 --steps 8 --warmup 1 --detail-out "$out/detail_$v.json" > "$out/line_$v.json" 2> "$out/err_$v.txt"
  echo "URSA_K14=$v rc=$?"; python3 -c "
import json
d=json.loads(open('$out/line_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d.get('errors'))"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 "$R/bench.py" --config c5 --c5-batch 1024 --steps 3 --warmup 0 --detail-out "$R/$out/prof_detail.json" > "$R/$out/prof_line.json" 2> "$R/$out/prof.err"; echo "c5 prof rc=$?"
python3 "$R/tools/prof_summary.py" /tmp/prof_c5 "$R/$out/r06_c5_kernel_stats.csv" | head -16
