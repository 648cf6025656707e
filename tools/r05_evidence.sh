#!/bin/bash
# Regenerates every round-5 file under profiles/ at the current HEAD, including the rocprofv3 passes of the C4 / C5
# configurations. A gpurun call is limited to 20
# minutes, so the work is cut into parts; each part is one call on a fresh box, all of them at the same commit
# (r05_sha256.txt is written by every part and must agree):
#   for p in a b c d e f; do /usr/local/graft/bin/gpurun --timeout 1200 -- "bash tools/r05_evidence.sh $p"; done
# Outputs: gpurun_out/r05_evidence/ (copy into profiles/).
part=${1:-a}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r05_evidence
mkdir -p $out
cd $R
sha256sum ursabench_amd/csrc/libursa_hip.so ursabench_amd/csrc/ursa_kernels.hip ursabench_amd/csrc/ursa_bn.hip ursabench_amd/csrc/ursa_conv.hip bench.py > $out/r05_sha256_part_$part.txt
case $part in
a)  # kernel micro-benchmarks and diagnostics
  python3 tools/kbench.py > $out/kbench.log 2>&1; echo "kbench rc=$?"; cp gpurun_out/kbench.json $out/r05_kbench.json
  python3 tools/k1_ctl_bench.py > $out/k1_ctl_bench.log 2>&1; echo "k1_ctl_bench rc=$?"; cp gpurun_out/k1_ctl_bench.json $out/r05_k1_ctl_bench.json
  (/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/ticket_probe tools/exp/ticket_probe.hip 2>/dev/null && /tmp/ticket_probe > $out/r05_ticket_probe.txt); echo "ticket_probe rc=$?"
  python3 tools/exp/k3_spread.py > $out/k3_spread.log 2>&1; echo "k3_spread rc=$?"; cp gpurun_out/k3_spread.json $out/r05_k3_spread.json
  python3 tools/exp/bn_fused_bench.py > $out/r05_bn_fused_bench.json 2> $out/bn_fused_bench.err; echo "bn_fused_bench rc=$?"
  python3 tools/exp/bn_held_ab.py > $out/r05_bn_held_ab.json 2> $out/bn_held_ab.err; echo "bn_held_ab rc=$?"
  python3 tools/exp/gate_probe_overwrite2.py 2 plain_product,plain_product@stock,plain,plain@stock,plain_sum_out,plain_persist_tmp,plain_two_stage_tmp,plain_kernel_copy,plain_hold_tmp@spy,plain@eager > $out/r05_gate_probe_bisect.log 2>&1; echo "gate probe bisect rc=$?"; cp gpurun_out/gate_probe_overwrite2.json $out/r05_gate_probe_bisect.json
  (/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -o /tmp/bn_tl tools/exp/bn_held_timeline.hip 2>/dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w -DTL_BLOCK_THREADS=256 -DTL_EPT=16 -o /tmp/bn_tl256 tools/exp/bn_held_timeline.hip 2>/dev/null && timeout -k 5 60 /tmp/bn_tl > $out/r05_bn_held_timeline.txt && timeout -k 5 60 /tmp/bn_tl256 >> $out/r05_bn_held_timeline.txt); echo "bn_held_timeline rc=$?"
  python3 tools/exp/grouped_conv_ab.py > $out/r05_grouped_conv_ab.json 2> $out/grouped_conv_ab.err; echo "grouped_conv_ab rc=$?"
  K5_ONLY=prefetch,no_prefetch python3 tools/k5_bench.py 30 10000 100 30 10000 64 30 10000 128 30 10000 256 > $out/r05_k5_prefetch_ab.txt 2>&1; echo "k5 prefetch A/B rc=$?"
  # K7 / K8 / K9 against MIOpen's launches for the same calls, kernel durations under rocprofv3 (tools/exp/conv_*_probe.sh)
  VARIANTS=default bash tools/exp/conv_wgrad_probe.sh > $out/conv_wgrad_probe.txt 2>&1; echo "conv_wgrad_probe rc=$?"
  cp gpurun_out/conv_wgrad_probe_default.json $out/r05_conv_wgrad_probe.json; cp gpurun_out/conv_wgrad_probe_default_kernels.txt $out/r05_conv_wgrad_probe_kernels.txt
  bash tools/exp/conv_fwd_probe.sh > $out/conv_fwd_probe.txt 2>&1; echo "conv_fwd_probe rc=$?"
  cp gpurun_out/conv_fwd_probe.json $out/r05_conv_fwd_probe.json; cp gpurun_out/conv_fwd_probe_kernels.txt $out/r05_conv_fwd_probe_kernels.txt
  cd $R
  # the gate-conditioned parity report (G16, eight seeds, K6 vs stock launches paired): written by the GPU test itself
  python3 -m pytest tests/test_gate_parity_gpu.py -q -m gpu > $out/g16_pytest.log 2>&1; echo "g16 gate parity rc=$?"; cp gpurun_out/g16_gate_parity.json $out/r05_g16_gate_parity.json

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kb -- python3 $R/tools/kbench.py > /dev/null 2>&1; echo "kbench under rocprof rc=$?"
  python3 $R/tools/prof_summary.py /tmp/prof_kb $out/r05_kbench_kernel_stats.csv > /dev/null
  ;;
b)  # rocprofv3 passes: the default bench, C4, C5; PMC counters
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 3 --warmup 1 --detail-out $out/bench_detail_under_rocprof.json > $out/bench_line_under_rocprof.json 2> $out/bench_under_rocprof.err; echo "bench under rocprof rc=$?"
  python3 $R/tools/exp/k1_in_workload.py /tmp/prof_bench $out/r05_k1_in_workload.json > /dev/null; echo "k1_in_workload rc=$?"
  python3 $R/tools/prof_summary.py /tmp/prof_bench $out/r05_bench_kernel_stats.csv > /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c4 -- python3 $R/bench.py --config c4 --steps 8 --warmup 1 --c4-epochs 2 --c4-train 5120 > $out/c4_line_under_rocprof.json 2> $out/c4_under_rocprof.err; echo "c4 under rocprof rc=$?"
  python3 $R/tools/prof_summary.py /tmp/prof_c4 $out/r05_c4_kernel_stats.csv > /dev/null
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -- python3 $R/bench.py --config c5 --c5-batch 1024 --steps 3 --warmup 0 > $out/c5_line_under_rocprof.json 2> $out/c5_under_rocprof.err; echo "c5 under rocprof rc=$?"
  python3 $R/tools/prof_summary.py /tmp/prof_c5 $out/r05_c5_kernel_stats.csv > /dev/null
  for grp in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
    d=/tmp/pmc_$(echo $grp | tr ' ' '_')
    rm -rf $d
    rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $R/tools/pmc_only.py $out/pmc_manifest.json > /dev/null 2> $out/pmc_last.err; echo "pmc [$grp] rc=$?"
  done
  cd $R
  python3 tools/pmc_summary2.py $out/pmc_manifest.json $out/r05_pmc.json /tmp/pmc_* > /dev/null; echo "pmc summary rc=$?"
  ;;
c)  # the bench lines and the harness drivers
  # (stdout = the ONE compact line the driver parses; --detail-out = the full record of the same run)
  python3 bench.py --detail-out $out/r05_bench_detail.json > $out/r05_bench_line.json 2> $out/bench.err; echo "plain bench rc=$?"
  ( time python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail-out $out/r05_bench_detail_driver_cmd.json > $out/r05_bench_line_driver_cmd.json 2> $out/bench_driver.err ) 2> $out/r05_bench_driver_cmd_wall_time.txt; echo "bench with the driver's flags rc=$?"
  URSA_FUSED_CONV=0 python3 bench.py --no-cpu-baseline --ref-style-steps 0 --multi-chain-sweep "" --detail-out $out/r05_bench_detail_stock_conv.json > $out/r05_bench_line_stock_conv.json 2> $out/bench_stock_conv.err; echo "bench with MIOpen's convolution launches rc=$?"
  URSA_FUSED_BN=0 python3 bench.py --no-parity --no-cpu-baseline --ref-style-steps 0 --multi-chain-sweep "" --detail-out $out/r05_bench_detail_stock_bn.json > $out/r05_bench_line_stock_bn.json 2> $out/bench_stock.err; echo "bench with stock BatchNorm launches rc=$?"
  python3 -m ursabench_amd.time_script --dataset CIFAR10 --model PreResNet20 --save_path $out/r05_time_script_preresnet20 --samples 3 --trials 10 --discard_first \
      --methods SGLD SGHMC cSGLD cSGHMC SWAG MCdropout SGD > $out/time_script.log 2>&1; echo "time_script rc=$?"
  python3 -m ursabench_amd.experiment --dataset CIFAR10 --model PreResNet20 --inference_method SGHMC --hyperparams_path tools/hyperparams/preresnet20_sghmc.json \
      --save_path $out/r05_experiment_ --num_trials 2 > $out/experiment.log 2>&1; echo "experiment rc=$?"
  ;;
d)  # C4 and C5 at full size
  # the product default (two-launch K6 on large activations) and, beside it, the OPT-IN held form (URSA_BN_HELD=1)
  python3 bench.py --config c5 --c5-batch 1024 --detail-out $out/r05_c5_bench_detail.json > $out/r05_c5_bench_line.json 2> $out/c5.err; echo "c5 rc=$?"
  URSA_BN_HELD=1 python3 bench.py --config c5 --c5-batch 1024 --detail-out $out/r05_c5_bench_detail_held_opt_in.json > $out/r05_c5_bench_line_held_opt_in.json 2> $out/c5_held.err; echo "c5 held opt-in rc=$?"
  python3 bench.py --config c4 --detail-out $out/r05_c4_bench_detail.json > $out/r05_c4_bench_line.json 2> $out/c4.err; echo "c4 rc=$?"
  ;;
f)  # one minibatch step of configs[1] dispatch by dispatch (tools/step_timeline.py on a kernel trace of the sampling leg)
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/bench.py --steps 2 --warmup 1 --no-parity --no-cpu-baseline --ref-style-steps 0 --multi-chain-sweep "" --detail-out $out/tl_detail.json > $out/tl_line.json 2> $out/tl.err; echo "trace rc=$?"
  cd $R
  python3 tools/step_timeline.py /tmp/tl $out/r05_step_timeline.json > /dev/null; echo "step_timeline rc=$?"
  ;;
e)  # C4 with the held form opted in (A/B of part d's C4 line)
  URSA_BN_HELD=1 python3 bench.py --config c4 --detail-out $out/r05_c4_bench_detail_held_opt_in.json > $out/r05_c4_bench_line_held_opt_in.json 2> $out/c4_held.err; echo "c4 held opt-in rc=$?"
  ;;
esac
ls $out | head -80
