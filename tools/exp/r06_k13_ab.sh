#!/bin/bash
# K13 tests, then C5 (PreResNet-164 HMC, 4 chains, 1,024 rows) with K13 on / off, alternating
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_k13
mkdir -p "$out"
timeout -k 10 600 python -m pytest tests/test_fused_bottleneck_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 "$out/pytest.log"
[ $rc -ne 0 ] && exit $rc
for v in 1 0 1 0; do
  URSA_K13=$v timeout -k 10 500 python3 bench.py --config c5 --c5-batch 1024 --steps 8 --warmup 1 --detail-out "$out/detail_$v.json" > "$out/line_$v.json" 2> "$out/err_$v.txt"
  echo "URSA_K13=$v rc=$?"; python3 -c "
import json
d=json.loads(open('$out/line_$v.json').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d.get('acceptance'), d.get('errors'))"
done
