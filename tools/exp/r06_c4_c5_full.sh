#!/bin/bash
# BASELINE configs[3] / configs[4] at full size, one GPU: the WideResNet-28-10 SWAG 30-member ensemble and the PreResNet-164 HMC run
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_c4_c5
mkdir -p "$out"
sha256sum ursabench_amd/csrc/libursa_hip.so bench.py > "$out/r06_sha256_c4_c5.txt"
timeout -k 10 500 python3 bench.py --config c5 --c5-batch 1024 --detail-out "$out/r06_c5_bench_detail.json" > "$out/r06_c5_bench_line.json" 2> "$out/c5.err"; echo "c5 rc=$?"; tail -c 400 "$out/r06_c5_bench_line.json"; echo
timeout -k 10 900 python3 bench.py --config c4 --detail-out "$out/r06_c4_bench_detail.json" > "$out/r06_c4_bench_line.json" 2> "$out/c4.err"; echo "c4 rc=$?"; tail -c 400 "$out/r06_c4_bench_line.json"; echo
