#!/bin/bash
# Round 6, K10 (fused pre-activation unit): the new parity tests, the regression tests of the launches it shares code with,
# then the headline config with and without it, and a kernel trace of the step.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/r06_k10_check.sh'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
out=gpurun_out/r06_k10
mkdir -p "$out"
timeout -k 10 500 python -m pytest tests/test_fused_block_gpu.py -x -q > "$out/pytest_block.log" 2>&1; rc=$?
echo "pytest_block rc=$rc" | tee "$out/rc.txt"; tail -15 "$out/pytest_block.log"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python -m pytest tests/test_fused_conv_gpu.py tests/test_fused_bn_gpu.py -x -q -m gpu > "$out/pytest_conv_bn.log" 2>&1; rc=$?
echo "pytest_conv_bn rc=$rc" | tee -a "$out/rc.txt"; tail -4 "$out/pytest_conv_bn.log"
[ $rc -ne 0 ] && exit $rc
for mode in 1 0; do
  URSA_FUSED_BLOCK=$mode timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-parity --ref-style-steps 0 --multi-chain-sweep "" \
      --detail-out "$out/bench_detail_fused$mode.json" > "$out/bench_fused$mode.json" 2> "$out/bench_fused$mode.err"; rc=$?
  echo "bench fused=$mode rc=$rc" | tee -a "$out/rc.txt"
  python3 -c "
import json,sys
d=json.loads(open('$out/bench_fused$mode.json').read().strip().splitlines()[-1])
print('fused=$mode value', d['value'], d['unit'], 'ms_per_step', d['ms_per_step'], 'errors', d.get('errors'))" | tee -a "$out/rc.txt"
  [ $rc -ne 0 ] && exit $rc
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 "$R/bench.py" --steps 2 --warmup 1 \
    --no-cpu-baseline --no-parity --ref-style-steps 0 --multi-chain-sweep "" --detail-out /tmp/tl_detail.json > /tmp/tl.log 2>&1; rc=$?
echo "trace rc=$rc" | tee -a "$R/$out/rc.txt"
cd "$R" && python3 tools/step_timeline.py /tmp/tl "$out/step_timeline.json" | tee "$out/step_timeline.txt"
