#!/bin/bash
# Regenerates the round-6 files under profiles/ at the current HEAD. Parts (one gpurun call each, all at the same commit:
# r06_sha256_part_<p>.txt is written by every part and must agree):
#   for p in a b c f; do tools/gpu.sh 1190 "bash tools/r06_evidence.sh $p"; done
# Outputs: gpurun_out/r06_evidence/ (copy into profiles/ with tools/r06_copy_evidence.sh <part>).
set -u
part=${1:-a}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$R/gpurun_out/r06_evidence
mkdir -p "$out"
cd "$R"
sha256sum ursabench_amd/csrc/libursa_hip.so ursabench_amd/csrc/*.hip bench.py > "$out/r06_sha256_part_$part.txt"
case $part in
a)  # per-launch costs of the K10 / K11 launches, attribution variants, the G16 gate-parity report (now with MIOpen's convolutions paired)
  python3 tools/k10_bench.py "$out/r06_k10_bench.json" > "$out/k10_bench.log" 2>&1; echo "k10_bench rc=$?"
  for dbg in 1 2 3 4; do
    URSA_K10_KNOBS=1 URSA_K10_DBG=$dbg python3 tools/k10_bench.py "$out/r06_k10_bench_dbg$dbg.json" > /dev/null 2>&1; echo "k10_bench dbg $dbg rc=$?"
  done
  python3 tools/kbench.py > "$out/kbench.log" 2>&1; echo "kbench rc=$?"; cp gpurun_out/kbench.json "$out/r06_kbench.json"
  python3 -m pytest tests/test_gate_parity_gpu.py -q -m gpu > "$out/g16_pytest.log" 2>&1; echo "g16 gate parity rc=$?"; cp gpurun_out/g16_gate_parity.json "$out/r06_g16_gate_parity.json"
  ;;
b)  # rocprofv3: kernel stats of the default command, PMC passes of the K10 launches
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 "$R/bench.py" --steps 3 --warmup 1 --detail-out "$out/bench_detail_under_rocprof.json" > "$out/bench_line_under_rocprof.json" 2> "$out/bench_under_rocprof.err"; echo "bench under rocprof rc=$?"
  python3 "$R/tools/prof_summary.py" /tmp/prof_bench "$out/r06_bench_kernel_stats.csv" > /dev/null
  python3 - "$R" "$out" <<'PY'
import hashlib, json, sys
R, out = sys.argv[1:3]
json.dump({'libursa_hip_sha256': hashlib.sha256(open(R + '/ursabench_amd/csrc/libursa_hip.so', 'rb').read()).hexdigest(),
           'command': 'rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1'}, open(out + '/r06_bench_kernel_stats.meta.json', 'w'), indent=1)
PY
  for grp in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
    d=/tmp/pmc_$(echo $grp | tr ' ' '_')
    rm -rf "$d"
    rocprofv3 --pmc $grp --output-format csv -d "$d" -- python3 "$R/tools/pmc_only.py" "$out/pmc_manifest.json" > /dev/null 2> "$out/pmc_last.err"; echo "pmc [$grp] rc=$?"
  done
  cd "$R"
  python3 tools/pmc_summary2.py "$out/pmc_manifest.json" "$out/r06_pmc.json" /tmp/pmc_* > /dev/null; echo "pmc summary rc=$?"
  ;;
c)  # the bench lines: plain, the driver's flags (with wall time), and the A/Bs of this round's launches
  python3 bench.py --detail-out "$out/r06_bench_detail.json" > "$out/r06_bench_line.json" 2> "$out/bench.err"; echo "plain bench rc=$?"
  ( time python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out "$out/r06_bench_detail_driver_cmd.json" > "$out/r06_bench_line_driver_cmd.json" 2> "$out/bench_driver.err" ) 2> "$out/r06_bench_driver_cmd_wall_time.txt"; echo "bench with the driver's flags rc=$?"
  URSA_FUSED_BLOCK=0 python3 bench.py --no-parity --no-cpu-baseline --ref-style-steps 0 --no-full-size-legs --detail-out "$out/r06_bench_detail_k6_k8.json" > "$out/r06_bench_line_k6_k8.json" 2> "$out/bench_k6k8.err"; echo "bench with the K6 / K8 launches (round 5's step) rc=$?"
  URSA_BWD_PAIR=0 python3 bench.py --no-parity --no-cpu-baseline --ref-style-steps 0 --no-full-size-legs --detail-out "$out/r06_bench_detail_unpaired.json" > "$out/r06_bench_line_unpaired.json" 2> "$out/bench_unpaired.err"; echo "bench with separate backward launches rc=$?"
  ;;
g)  # one WHOLE posterior sample (391 steps at the workload batch) against the CPU port, the port's gates given at every step
  python3 bench.py --parity-full-sample --steps 1 --warmup 0 --no-parity --no-cpu-baseline --ref-style-steps 0 --multi-chain-sweep "" --no-full-size-legs \
      --detail-out "$out/r06_parity_full_sample_detail.json" > "$out/r06_parity_full_sample_line.json" 2> "$out/parity_full.err"; echo "parity full sample rc=$?"
  python3 - "$out" <<'PY'
import json, sys
d = json.load(open(sys.argv[1] + '/r06_parity_full_sample_detail.json'))
json.dump({'what': 'bench.py --parity-full-sample: one whole posterior sample of the workload (391 minibatch steps, 128 rows each) on the GPU, every step a '
                   'hipGraph replay, against the torch-CPU port of the reference path on identical init / inputs / noise; predictive on 128 test rows',
           'parity_full_sample': d.get('parity_full_sample'), 'errors': d.get('errors')}, open(sys.argv[1] + '/r06_parity_full_sample.json', 'w'), indent=1)
print(json.dumps(d.get('parity_full_sample')))
PY
  ;;
f)  # one minibatch step dispatch by dispatch
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-parity --no-cpu-baseline --ref-style-steps 0 --multi-chain-sweep "" --no-full-size-legs --detail-out "$out/tl_detail.json" > "$out/tl_line.json" 2> "$out/tl.err"; echo "trace rc=$?"
  cd "$R"
  python3 tools/step_timeline.py /tmp/tl "$out/r06_step_timeline.json" > "$out/r06_step_timeline.txt"; echo "step_timeline rc=$?"
  ;;
esac
ls "$out" | head -60
