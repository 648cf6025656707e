#!/bin/bash
# Copies what one part of tools/r06_evidence.sh produced from gpurun_out/r06_evidence/ into profiles/ (tracked).
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
E=gpurun_out/r06_evidence
part=$1
mkdir -p profiles/r06
case $part in
a) cp $E/r06_k10_bench.json $E/r06_k10_bench_dbg1.json $E/r06_k10_bench_dbg2.json $E/r06_k10_bench_dbg3.json $E/r06_k10_bench_dbg4.json $E/r06_kbench.json $E/r06_g16_gate_parity.json profiles/ ;;
b) cp $E/r06_bench_kernel_stats.csv $E/r06_bench_kernel_stats.meta.json $E/r06_pmc.json profiles/ ;;
c) cp $E/r06_bench_line.json $E/r06_bench_detail.json $E/r06_bench_line_driver_cmd.json $E/r06_bench_detail_driver_cmd.json $E/r06_bench_driver_cmd_wall_time.txt \
      $E/r06_bench_line_k6_k8.json $E/r06_bench_line_unpaired.json profiles/ ;;
f) cp $E/r06_step_timeline.json profiles/ ;;
g) cp $E/r06_parity_full_sample.json profiles/ ;;
esac
cp $E/r06_sha256_part_$part.txt profiles/r06/
python3 - <<'PY'
import json, glob
for p in glob.glob('profiles/r06_bench_line*.json'):
    l = open(p).read().strip().splitlines()[-1]
    json.loads(l)
    open(p, 'w').write(l + '\n')
PY
echo copied "$part"
