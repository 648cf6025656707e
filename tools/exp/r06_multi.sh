#!/bin/bash
# multi-chain sweep with the paired backward launch on / off (and K10 off)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_k10
mkdir -p "$out"
for cfg in "1 1" "1 0" "0 0"; do
  set -- $cfg
  URSA_FUSED_BLOCK=$1 URSA_BWD_PAIR=$2 timeout -k 10 400 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-parity --ref-style-steps 0 --multi-chain-sweep "4,8" \
      --detail-out "$out/multi_detail_$1$2.json" > "$out/multi_$1$2.json" 2> "$out/multi_$1$2.err"; rc=$?
  python3 -c "
import json
d=json.loads(open('$out/multi_$1$2.json').read().strip().splitlines()[-1])
print('block=$1 pair=$2 rc=$rc value', d['value'], 'multi', json.dumps(d.get('multi_chain_per_gpu')), 'bma', d.get('bma_preds_per_s'))"
done
