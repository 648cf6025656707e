#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/bn_cfirst_ab.jsonl
: > $out
for s in -1 100000 16; do
  URSA_BN_CFIRST_MAX_MIB=$s timeout -k 10 200 python3 $R/tools/exp/bn_cfirst_ab.py >> $out 2>/dev/null; echo "cfirst_max_mib=$s rc=$?"
done
python3 - $out <<'PY'
import json, sys
runs = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')]
for i, r0 in enumerate(runs[0]['rows']):
    print(r0['shape'], r0['mbytes'], 'MB')
    for r in runs:
        x = r['rows'][i]
        print('   cfirst_max_mib', r['setting'], '| fwd', x['fwd_us'], x['fwd_frac'], '| bwd', x['bwd_us'], x['bwd_frac'], '| fwd+res', x['fwd_residual_us'], '| bwd+res', x['bwd_residual_us'], '| eval', x.get('eval_us'))
PY
