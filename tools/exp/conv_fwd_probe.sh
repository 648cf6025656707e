#!/bin/bash
# K8 probe under rocprofv3 (kernel durations).   bash tools/exp/conv_fwd_probe.sh -> gpurun_out/conv_fwd_probe*.{json,txt,log}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
out=$R/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k8
[ -n "$URSA_CONV_FWD_VARIANT" ] && export URSA_PROBE_KNOBS=1
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_k8 -- python3 $R/tools/exp/conv_fwd_probe.py > $out/conv_fwd_probe.log 2>&1 || { echo "probe failed"; tail -20 $out/conv_fwd_probe.log; exit 1; }
f=$(find /tmp/prof_k8 -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $out/conv_fwd_probe_kernels.txt <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('k_conv', 'miopenSp3', 'igemm', 'transpose', 'SubTensor', 'naive', 'Cijk')):
        print(r['Name'][:80], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
P
cat $out/conv_fwd_probe_kernels.txt; grep -h "fwd_err" $out/conv_fwd_probe.log | cut -c1-460
