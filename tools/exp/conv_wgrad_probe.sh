#!/bin/bash
# K7 probe under rocprofv3 (kernel durations), default plan and URSA_CONV_IPW variants of the knobs build.
#   bash tools/exp/conv_wgrad_probe.sh -> gpurun_out/conv_wgrad_probe_*.{json,txt}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$R/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for v in ${VARIANTS:-default 2 4}; do
  rm -rf /tmp/prof_k7_$v
  if [ $v = default ]; then unset URSA_CONV_IPW; export URSA_PROBE_KNOBS=0; else export URSA_CONV_IPW=$v URSA_PROBE_KNOBS=1; fi
  export URSA_PROBE_OUT=conv_wgrad_probe_$v.json
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k7_$v -- python3 $R/tools/exp/conv_wgrad_probe.py > $out/conv_wgrad_probe_$v.log 2>&1 || { echo "probe $v failed"; tail -5 $out/conv_wgrad_probe_$v.log; exit 1; }
  f=$(find /tmp/prof_k7_$v -name '*kernel_stats.csv' | head -1)
  python3 - "$f" > $out/conv_wgrad_probe_${v}_kernels.txt <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_conv', 'igemm_wrw', 'transpose', 'SubTensor')):
        print(r['Name'][:80], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
P
  echo "== $v"; cat $out/conv_wgrad_probe_${v}_kernels.txt
done
