rm -rf gpurun_out/miopen_db; T0=$(date +%s)
mkdir -p gpurun_out/miopen_db
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db
export MIOPEN_FIND_ENFORCE=3
python bench.py --no-parity --no-cpu-baseline --ref-style-steps 0 --multi-chain-probe 0 --steps 2 --warmup 1 > gpurun_out/tune1.json 2> gpurun_out/tune1.err
echo "tune rc=$? seconds=$(( $(date +%s) - T0 ))"; tail -3 gpurun_out/tune1.err
python -c "
import json;d=json.loads(open('gpurun_out/tune1.json').read().strip().splitlines()[-1]);print('tuned run', d['value'], d['ms_per_step'], d['errors'])"
unset MIOPEN_FIND_ENFORCE
python bench.py --no-parity --no-cpu-baseline --ref-style-steps 0 --multi-chain-probe 0 --steps 3 --warmup 1 > gpurun_out/tune2.json 2>/dev/null
python -c "
import json;d=json.loads(open('gpurun_out/tune2.json').read().strip().splitlines()[-1]);print('reuse tuned db', d['value'], d['ms_per_step'], d['bma_preds_per_s'])"
ls -la gpurun_out/miopen_db | head; du -sh gpurun_out/miopen_db
unset MIOPEN_USER_DB_PATH
python bench.py --no-parity --no-cpu-baseline --ref-style-steps 0 --multi-chain-probe 0 --steps 3 --warmup 1 > gpurun_out/tune3.json 2>/dev/null
python -c "
import json;d=json.loads(open('gpurun_out/tune3.json').read().strip().splitlines()[-1]);print('default', d['value'], d['ms_per_step'], d['bma_preds_per_s'])"
