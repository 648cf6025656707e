#!/bin/bash
# HIP runtime switches that touch kernel-launch latency, on the headline leg only (same box, one after another):
#   HIP_FORCE_DEV_KERNARG (kernel arguments in device memory), GPU_MAX_HW_QUEUES, HSA_ENABLE_SDMA
out=gpurun_out/runtime_env_ab.txt
: > $out
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-sanity-legs --ref-style-steps 0 --multi-chain-sweep 4 --bma-members 3"
run() { name=$1; shift; env "$@" $B 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
mc=d.get('multi_chain_per_gpu',{}).get('sweep',[{}])
print('$name', 'samples/s', d['value'], 'ms/step', round(1000/d['minibatch_steps_per_s'],4), 'K1 us', d['roofline']['us_per_launch'], '4 chains', mc[0].get('value'), 'errors', d['errors'])" >> $out; }
run default A=1
run kernarg_dev HIP_FORCE_DEV_KERNARG=1
run kernarg_host HIP_FORCE_DEV_KERNARG=0
run hwq2 GPU_MAX_HW_QUEUES=2
run hwq8 GPU_MAX_HW_QUEUES=8
run sdma_off HSA_ENABLE_SDMA=0
run default_again A=1
cat $out
