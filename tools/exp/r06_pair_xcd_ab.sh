#!/bin/bash
# the paired backward launch with its workgroups in plain order (URSA_PAIR_XCD=0, knobs build) against eighths of each role per XCD
# (shipped): us per launch (tools/k10_bench.py) and HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over tools/exp/pair_only.py)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=$R/gpurun_out/r06_pair_xcd
mkdir -p "$out"
timeout -k 10 300 python -m pytest tests/test_fused_block_gpu.py -x -q -m gpu > "$out/pytest.log" 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 "$out/pytest.log"
[ $rc -ne 0 ] && exit $rc
for v in 0 1 0 1; do
  echo "URSA_PAIR_XCD=$v" | tee -a "$out/ab.txt"
  URSA_K10_KNOBS=1 URSA_PAIR_XCD=$v timeout -k 10 300 python3 tools/k10_bench.py "$out/k10_xcd$v.json" 2>/dev/null | grep -o '^[0-9x]*s[12] \|"bwd_pair": [0-9.]*' | paste -sd' ' | tee -a "$out/ab.txt"
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_x${v}_$c
    URSA_PAIR_XCD=$v rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_x${v}_$c -- python3 "$R/tools/exp/pair_only.py" > /dev/null 2> "$out/pmc.err"; echo "pmc xcd=$v $c rc=$?"
    python3 - /tmp/pmc_x${v}_$c $c $v <<'PY' | tee -a "$out/ab.txt"
import csv, glob, sys
d, c, v = sys.argv[1:4]
tot, n = 0.0, 0
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_bwd_pair' in r.get('Kernel_Name', '') and r.get('Counter_Name') == c:
            tot += float(r['Counter_Value']); n += 1
print(f'xcd={v} {c}: {tot / max(n, 1):.1f} KiB per launch over {n} launches')
PY
  done
done
