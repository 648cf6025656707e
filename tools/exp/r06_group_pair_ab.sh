#!/bin/bash
# ChainGroup (8 chains per GPU): the units' backward launches separate (shipped) against paired (URSA_GROUP_PAIR=1)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_group_pair
mkdir -p "$out"
for v in 0 1 0 1; do
  URSA_GROUP_PAIR=$v timeout -k 10 400 python3 bench.py --no-parity --no-cpu-baseline --ref-style-steps 0 --no-full-size-legs --steps 2 --warmup 1 --detail-out "$out/detail_$v.json" > "$out/line_$v.json" 2> "$out/err_$v.txt"
  echo "URSA_GROUP_PAIR=$v rc=$?"; python3 -c "
import json,sys
d=json.loads(open('$out/line_$v.json').read().strip().splitlines()[-1]); print(d['value'], json.dumps(d['multi_chain_per_gpu']))"
done
