#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_k12
mkdir -p "$out"
timeout -k 10 600 python -m pytest tests/test_fused_conv_gpu.py -x -q -k "k12" > "$out/pytest.log" 2>&1; rc=$?
echo "pytest k12 rc=$rc"; tail -4 "$out/pytest.log"
[ $rc -ne 0 ] && exit $rc
python3 tools/k12_bench.py "$out/k12_bench.json" 2>/dev/null
timeout -k 10 400 python3 bench.py --config c5 --c5-batch 1024 --steps 3 --warmup 0 --detail-out "$out/c5_detail.json" > "$out/c5.json" 2> "$out/c5.err"
python3 -c "
import json
d=json.loads(open('$out/c5.json').read().strip().splitlines()[-1])
print('c5 value', d['value'], d['unit'], 'acceptance', d.get('acceptance_rate_rank0'), 'errors', d.get('errors'))"
