R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/bn_target_wgs_cfirst.jsonl
: > $out
for t in 1024 512 2048 256; do
  URSA_BN_TARGET_WGS=$t timeout -k 10 200 python3 $R/tools/exp/bn_cfirst_ab.py >> $out 2>/dev/null; echo "target_wgs=$t rc=$?"
done
python3 - $out <<'PY'
import json, sys
runs = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{')]
tags = [1024, 512, 2048, 256]
for i, r0 in enumerate(runs[0]['rows'][:5]):
    print(r0['shape'], r0['mbytes'], 'MB')
    for t, r in zip(tags, runs):
        x = r['rows'][i]
        print('   target_wgs', t, '| fwd', x['fwd_us'], '| bwd', x['bwd_us'], '| fwd+res', x['fwd_residual_us'], '| bwd+res', x['bwd_residual_us'], '| eval', x.get('eval_us'))
PY
