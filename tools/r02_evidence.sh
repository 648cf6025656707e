#!/bin/bash
# Regenerates every round-2 file under profiles/ at the current HEAD, on one GPU box.
#   /usr/local/graft/bin/gpurun --timeout 4200 -- 'bash tools/r02_evidence.sh'
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r02_evidence
mkdir -p $out
cd $R
python3 tools/kbench.py > $out/kbench.log 2>&1; cp gpurun_out/kbench.json $out/r02_kbench.json; echo "kbench rc=$?"
python3 tools/k5_bench.py 50 10000 10 20 10000 10 3 10000 10 3 128 10 30 10000 16 30 10000 32 30 10000 64 30 10000 100 30 10000 256 > $out/k5_bench.log 2>&1; cp gpurun_out/k5_bench.json $out/r02_k5_bench.json; echo "k5_bench rc=$?"
python3 tools/exp/graph_stress.py > $out/r02_graph_stress.txt 2>&1; echo "graph_stress rc=$?"; cat $out/r02_graph_stress.txt | grep -v amdgpu
python3 tools/exp/parity_sweep.py > $out/parity_sweep.log 2>&1; cp gpurun_out/parity_sweep.json $out/r02_parity_sweep.json; echo "parity_sweep rc=$?"
python3 tools/exp/bma_probe.py 4 > $out/r02_bma_probe.log 2>&1; cp gpurun_out/bma_probe.json $out/r02_bma_probe.json; echo "bma_probe rc=$?"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $R/bench.py --steps 3 --warmup 1 > $out/bench_line_under_rocprof.json 2> $out/bench_under_rocprof.err; echo "bench under rocprof rc=$?"
python3 $R/tools/prof_summary.py /tmp/prof_bench $out/r02_bench_kernel_stats.csv > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_kb -- python3 $R/tools/kbench.py > /dev/null 2>&1; echo "kbench under rocprof rc=$?"
python3 $R/tools/prof_summary.py /tmp/prof_kb $out/r02_kbench_kernel_stats.csv > /dev/null
for grp in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  d=/tmp/pmc_$(echo $grp | tr ' ' '_')
  rm -rf $d
  rocprofv3 --pmc $grp --output-format csv -d $d -- python3 $R/tools/pmc_only.py $out/pmc_manifest.json > /dev/null 2> $out/pmc_last.err; echo "pmc [$grp] rc=$?"
done
cd $R
python3 tools/pmc_summary2.py $out/pmc_manifest.json $out/r02_pmc.json /tmp/pmc_* > /dev/null; echo "pmc summary rc=$?"
python3 bench.py > $out/r02_bench_line.json 2> $out/bench.err; echo "plain bench rc=$?"
python3 -m ursabench_amd.time_script --dataset CIFAR10 --model PreResNet20 --save_path $out/r02_time_script_preresnet20 --samples 3 --trials 10 --discard_first \
    --methods SGLD SGHMC cSGLD cSGHMC SWAG MCdropout SGD > $out/time_script.log 2>&1; echo "time_script rc=$?"
python3 -m ursabench_amd.experiment --dataset CIFAR10 --model PreResNet20 --inference_method SGHMC --hyperparams_path tools/hyperparams/preresnet20_sghmc.json \
    --save_path $out/r02_experiment_ --num_trials 2 > $out/experiment.log 2>&1; echo "experiment rc=$?"
python3 bench.py --config c4 > $out/r02_c4_bench_line.json 2> $out/c4.err; echo "c4 rc=$?"
python3 bench.py --config c5 --c5-batch 1024 > $out/r02_c5_bench_line.json 2> $out/c5.err; echo "c5 rc=$?"
ls $out
