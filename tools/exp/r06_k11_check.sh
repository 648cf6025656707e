#!/bin/bash
# Round 6, K11 (fused head): its tests, the K10 tests again, the headline config, a kernel trace of the step.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_k11
mkdir -p "$out"
timeout -k 10 500 python -m pytest tests/test_fused_head_gpu.py tests/test_fused_block_gpu.py -x -q > "$out/pytest.log" 2>&1; rc=$?
echo "pytest rc=$rc" | tee "$out/rc.txt"; tail -25 "$out/pytest.log"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --ref-style-steps 0 --multi-chain-sweep "8" \
    --detail-out "$out/bench_detail.json" > "$out/bench.json" 2> "$out/bench.err"; rc=$?
echo "bench rc=$rc" | tee -a "$out/rc.txt"
python3 -c "
import json
d=json.loads(open('$out/bench.json').read().strip().splitlines()[-1])
print('value', d['value'], d['unit'], 'ms_per_step', d['ms_per_step'], 'multi', json.dumps(d.get('multi_chain_per_gpu')), 'errors', d.get('errors'))" | tee -a "$out/rc.txt"
[ $rc -ne 0 ] && exit $rc
URSA_FUSED_EVAL=0 timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --ref-style-steps 0 --multi-chain-sweep "" --no-full-size-legs \
    --detail-out "$out/bench_detail_noeval.json" > "$out/bench_noeval.json" 2> "$out/bench_noeval.err"
python3 -c "
import json
for f in ('bench', 'bench_noeval'):
    d=json.loads(open('$out/' + f + '.json').read().strip().splitlines()[-1])
    print(f, 'value', d['value'], 'bma_preds_per_s', d.get('bma_preds_per_s'), 'bma_member_forwards_per_s', d.get('bma_member_forwards_per_s'), 'errors', d.get('errors'))" | tee -a "$out/rc.txt"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 "$R/bench.py" --steps 2 --warmup 1 \
    --no-cpu-baseline --no-parity --ref-style-steps 0 --multi-chain-sweep "" --detail-out /tmp/tl_detail.json > /tmp/tl.log 2>&1; rc=$?
echo "trace rc=$rc" | tee -a "$R/$out/rc.txt"
cd "$R" && python3 tools/step_timeline.py /tmp/tl "$out/step_timeline.json" | tee "$out/step_timeline.txt"
timeout -k 10 300 python3 tools/k10_bench.py "$out/k10_bench.json" 2>/dev/null | tee "$out/k10_bench.txt"
