#!/bin/bash
# Builds the MIOpen user databases shipped under ursabench_amd/miopen_db/: MIOpen's own exhaustive per-layer tuning
# (MIOPEN_FIND_ENFORCE=3: every applicable solver is benchmarked, tunable ones are tuned) recorded while the benchmark
# configurations run once. Stock MIOpen mechanism, stock kernels: the databases only tell MIOpen which of ITS solvers
# (and which of their tuning parameters) to pick for the layer shapes of PreResNet-20 / WideResNet-28-10 /
# PreResNet-164 at the benchmark's batch sizes — measured +4.6 % posterior-samples/s on configs[1]
# (tools/exp/miopen_find_tune.sh). Keyed by MIOpen build + gfx950: a mismatching box simply ignores them.
#   /usr/local/graft/bin/gpurun --timeout 5400 -- 'bash tools/miopen_tune.sh'
R=$GRAFT_REPO_ROOT
db=$R/gpurun_out/miopen_db_tuned
rm -rf $db; mkdir -p $db
export MIOPEN_USER_DB_PATH=$db
export MIOPEN_FIND_ENFORCE=3
export URSA_NO_SHIPPED_MIOPEN_DB=1
cd $R
T0=$(date +%s)
python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 > $db.c2.json 2> $db.c2.err; echo "c2 rc=$? t=$(( $(date +%s) - T0 ))s"
python3 -m ursabench_amd.time_script --dataset CIFAR10 --model PreResNet20 --save_path $db.ts --samples 1 --trials 1 --methods SWAG MCdropout SGD > $db.ts.log 2>&1; echo "time_script rc=$? t=$(( $(date +%s) - T0 ))s"
timeout 1500 python3 bench.py --config c5 --c5-batch 1024 --steps 1 --warmup 0 > $db.c5.json 2> $db.c5.err; echo "c5 rc=$? t=$(( $(date +%s) - T0 ))s"
timeout 2400 python3 bench.py --config c4 --steps 4 --warmup 1 --c4-epochs 2 --c4-train 2560 > $db.c4.json 2> $db.c4.err; echo "c4 rc=$? t=$(( $(date +%s) - T0 ))s"
ls -la $db; wc -c $db/*
