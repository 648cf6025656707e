#!/bin/bash
# Round 6, K11 (fused head): its tests, the K10 tests again, the headline config, a kernel trace of the step.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
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
URSA_BWD_PAIR=0 timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --ref-style-steps 0 --multi-chain-sweep "8" \
    --detail-out "$out/bench_detail_nopair.json" > "$out/bench_nopair.json" 2> "$out/bench_nopair.err"
python3 -c "
import json
d=json.loads(open('$out/bench_nopair.json').read().strip().splitlines()[-1])
print('URSA_BWD_PAIR=0: value', d['value'], d['unit'], 'ms_per_step', d['ms_per_step'], 'multi', json.dumps(d.get('multi_chain_per_gpu')), 'errors', d.get('errors'))" | tee -a "$out/rc.txt"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 "$R/bench.py" --steps 2 --warmup 1 \
    --no-cpu-baseline --no-parity --ref-style-steps 0 --multi-chain-sweep "" --detail-out /tmp/tl_detail.json > /tmp/tl.log 2>&1; rc=$?
echo "trace rc=$rc" | tee -a "$R/$out/rc.txt"
cd "$R" && python3 tools/step_timeline.py /tmp/tl "$out/step_timeline.json" | tee "$out/step_timeline.txt"
timeout -k 10 300 python3 tools/k10_bench.py "$out/k10_bench.json" 2>/dev/null | tee "$out/k10_bench.txt"
for ipw in 2 4; do
  echo "URSA_CONV_IPW=$ipw (knobs build: images per K7 workgroup)" | tee -a "$out/k10_bench.txt"
  URSA_K10_KNOBS=1 URSA_CONV_IPW=$ipw timeout -k 10 300 python3 tools/k10_bench.py "$out/k10_bench_ipw$ipw.json" 2>/dev/null | tee -a "$out/k10_bench.txt"
done
