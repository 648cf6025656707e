#!/bin/bash
# Exactly what the driver runs at round end, in its order, on a fresh GPU box:
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/preflight.sh [tag]'
# Outputs land in gpurun_out/preflight_<tag>/ (copy what should be judged into profiles/).
tag=${1:-run}
out=gpurun_out/preflight_$tag
mkdir -p "$out"
python -c "import __graft_entry__ as g; g.build()" > "$out/build.log" 2>&1; echo "build rc=$?" | tee "$out/rc.txt"
python -m pytest tests/ -x -q -m gpu > "$out/pytest_gpu.log" 2>&1; echo "pytest_gpu rc=$?" | tee -a "$out/rc.txt"
tail -3 "$out/pytest_gpu.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$out/rc.txt"
tail -2 "$out/smoke.log"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$out/bench.json" 2> "$out/bench.err"; echo "bench rc=$?" | tee -a "$out/rc.txt"
cp gpurun_out/bench_detail_c2.json "$out/bench_detail.json" 2>/dev/null
# what the driver does with stdout: the LAST line must parse by itself (round 4's 27.7 KB line did not fit its tail buffer)
python3 - "$out/bench.json" <<'PY' | tee -a "$out/rc.txt"
import json, sys
last = open(sys.argv[1]).read().rstrip("\n").splitlines()[-1]
d = json.loads(last)
print(f"bench line: {len(last.encode())} bytes, parses alone; value={d['value']} {d['unit']} ms_per_step={d['ms_per_step']} "
      f"roofline.frac={d['roofline']['frac']} cpu_baseline.value={d['cpu_baseline']['value']} parity.pass={d['parity']['pass']} errors={d['errors']}")
PY
