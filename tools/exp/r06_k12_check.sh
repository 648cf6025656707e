#!/bin/bash
# K12 (1x1 stride-1 convolutions): its tests, the samplers' HMC tests, the C5 configuration with and without it
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_k12
mkdir -p "$out"
timeout -k 10 600 python -m pytest tests/test_fused_conv_gpu.py -x -q -k "k12 or K12 or hand_written or large_batches" > "$out/pytest.log" 2>&1; rc=$?
echo "pytest k12 rc=$rc" | tee "$out/rc.txt"; tail -20 "$out/pytest.log"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_samplers_gpu.py -x -q -k "hmc or HMC or 164" > "$out/pytest_hmc.log" 2>&1; rc=$?
echo "pytest hmc rc=$rc" | tee -a "$out/rc.txt"; tail -6 "$out/pytest_hmc.log"
[ $rc -ne 0 ] && exit $rc
for conv in 1 0; do
  URSA_K12=$conv timeout -k 10 400 python3 bench.py --config c5 --c5-batch 1024 --steps 3 --warmup 0 --detail-out "$out/c5_detail_k12_$conv.json" > "$out/c5_k12_$conv.json" 2> "$out/c5_k12_$conv.err"; rc=$?
  python3 -c "
import json
d=json.loads(open('$out/c5_k12_$conv.json').read().strip().splitlines()[-1])
print('URSA_K12=$conv rc=$rc value', d['value'], d['unit'], 'acceptance', d.get('acceptance_rate_rank0'), 'errors', d.get('errors'))" | tee -a "$out/rc.txt"
done
