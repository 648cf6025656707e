#!/bin/bash
# Regenerates every round-3 file under profiles/ at the current HEAD — INCLUDING the rocprofv3 passes of the C4 / C5
# configurations (round 2's script re-ran only their bench lines: VERDICT r2 weak #5). A gpurun call is limited to 20
# minutes, so the work is cut into parts; each part is one call on a fresh box, all of them at the same commit
# (r03_sha256.txt is written by every part and must agree):
#   for p in a b c d; do /usr/local/graft/bin/gpurun --timeout 1200 -- "bash tools/r03_evidence.sh $p"; done
# Outputs: gpurun_out/r03_evidence/ (copy into profiles/).
part=${1:-a}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r03_evidence
mkdir -p $out
cd $R
sha256sum ursabench_amd/csrc/libursa_hip.so ursabench_amd/csrc/ursa_kernels.hip ursabench_amd/csrc/ursa_bn.hip bench.py > $out/r03_sha256_part_$part.txt
case $part in
a)  # kernel micro-benchmarks and diagnostics
  python3 tools/kbench.py > $out/kbench.log 2>&1; echo "kbench rc=$?"; cp gpurun_out/kbench.json $out/r03_kbench.json
  python3 tools/k1_ctl_bench.py > $out/k1_ctl_bench.log 2>&1; echo "k1_ctl_bench rc=$?"; cp gpurun_out/k1_ctl_bench.json $out/r03_k1_ctl_bench.json
  (/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ticket_probe tools/exp/ticket_probe.hip 2>/dev/null && /tmp/ticket_probe > $out/r03_ticket_probe.txt); echo "ticket_probe rc=$?"
  python3 tools/exp/k3_spread.py > $out/k3_spread.log 2>&1; echo "k3_spread rc=$?"; cp gpurun_out/k3_spread.json $out/r03_k3_spread.json
  python3 tools/exp/bn_fused_bench.py > $out/r03_bn_fused_bench.json 2> $out/bn_fused_bench.err; echo "bn_fused_bench rc=$?"
  python3 tools/exp/bn_mask_flips.py > $out/r03_bn_same_input_vs_cpu.json 2> $out/bn_mask_flips.err; echo "bn_mask_flips rc=$?"
  python3 tools/exp/bn_gate_diag.py > $out/r03_bn_gate_diag.txt 2> $out/bn_gate_diag.err; echo "bn_gate_diag rc=$?"
  python3 tools/exp/g9_gate_diag.py > $out/r03_g9_gate_diag.txt 2> $out/g9_gate_diag.err; echo "g9_gate_diag rc=$?"

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kb -- python3 $R/tools/kbench.py > /dev/null 2>&1; echo "kbench under rocprof rc=$?"
  python3 $R/tools/prof_summary.py /tmp/prof_kb $out/r03_kbench_kernel_stats.csv > /dev/null
  ;;
b)  # rocprofv3 passes: the default bench, C4, C5; PMC counters
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 3 --warmup 1 > $out/bench_line_under_rocprof.json 2> $out/bench_under_rocprof.err; echo "bench under rocprof rc=$?"
  python3 $R/tools/exp/k1_in_workload.py /tmp/prof_bench $out/r03_k1_in_workload.json > /dev/null; echo "k1_in_workload rc=$?"
  python3 $R/tools/prof_summary.py /tmp/prof_bench $out/r03_bench_kernel_stats.csv > /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -- python3 $R/bench.py --config c4 --steps 8 --warmup 1 --c4-epochs 2 --c4-train 5120 > $out/c4_line_under_rocprof.json 2> $out/c4_under_rocprof.err; echo "c4 under rocprof rc=$?"
  python3 $R/tools/prof_summary.py /tmp/prof_c4 $out/r03_c4_kernel_stats.csv > /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 $R/bench.py --config c5 --c5-batch 1024 --steps 3 --warmup 0 > $out/c5_line_under_rocprof.json 2> $out/c5_under_rocprof.err; echo "c5 under rocprof rc=$?"
  python3 $R/tools/prof_summary.py /tmp/prof_c5 $out/r03_c5_kernel_stats.csv > /dev/null
  for grp in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    d=/tmp/pmc_$(echo $grp | tr ' ' '_')
    rm -rf $d
    rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $R/tools/pmc_only.py $out/pmc_manifest.json > /dev/null 2> $out/pmc_last.err; echo "pmc [$grp] rc=$?"
  done
  cd $R
  python3 tools/pmc_summary2.py $out/pmc_manifest.json $out/r03_pmc.json /tmp/pmc_* > /dev/null; echo "pmc summary rc=$?"
  ;;
c)  # the bench lines and the harness drivers
  python3 bench.py > $out/r03_bench_line.json 2> $out/bench.err; echo "plain bench rc=$?"
  URSA_FUSED_BN=0 python3 bench.py --no-cpu-baseline --ref-style-steps 0 --multi-chain-probe 0 > $out/r03_bench_line_stock_bn.json 2> $out/bench_stock.err; echo "bench with stock BatchNorm launches rc=$? (its parity leg may land on either side: reported)"
  python3 bench.py --chains-per-gpu 4 --no-cpu-baseline --ref-style-steps 0 --multi-chain-probe 0 > $out/r03_bench_line_4chains.json 2> $out/bench4.err; echo "4-chain bench rc=$?"
  python3 -m ursabench_amd.time_script --dataset CIFAR10 --model PreResNet20 --save_path $out/r03_time_script_preresnet20 --samples 3 --trials 10 --discard_first \
      --methods SGLD SGHMC cSGLD cSGHMC SWAG MCdropout SGD > $out/time_script.log 2>&1; echo "time_script rc=$?"
  python3 tools/exp/bn_grad_diag.py > $out/r03_bn_grad_diag.txt 2> $out/bn_grad_diag.err; echo "bn_grad_diag rc=$?"
  python3 -m ursabench_amd.experiment --dataset CIFAR10 --model PreResNet20 --inference_method SGHMC --hyperparams_path tools/hyperparams/preresnet20_sghmc.json \
      --save_path $out/r03_experiment_ --num_trials 2 > $out/experiment.log 2>&1; echo "experiment rc=$?"
  ;;
d)  # C4 and C5 at full size
  python3 bench.py --config c4 > $out/r03_c4_bench_line.json 2> $out/c4.err; echo "c4 rc=$?"
  python3 bench.py --config c5 --c5-batch 1024 > $out/r03_c5_bench_line.json 2> $out/c5.err; echo "c5 rc=$?"
  ;;
esac
ls $out | head -80
