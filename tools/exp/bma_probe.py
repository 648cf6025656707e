"""BMA member-forward throughput on the bank -> twin -> hipGraph path, by member count and lane count;
with torch.profiler kernel breakdown of one evaluation. python tools/exp/bma_probe.py [lanes ...]"""
import os, sys, tempfile, time, json
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='ursa_probe_miopen_'))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import inference, models, tasks, util
from ursabench_amd.data import synthetic
from ursabench_amd.tasks.task_base import EnsembleAccumulator
dev = torch.device('cuda', 0)
train = synthetic(1024, (3, 32, 32), 10, seed=0, device=dev, batch_size=128)
test = synthetic(10000, (3, 32, 32), 10, seed=1, device=dev, batch_size=128)
util.set_random_seed(0)
s = inference.SGHMC({'lr': 0.1, 'prior_std': 0.5, 'num_samples': 20, 'alpha': 0.5, 'burn_in_epochs': 0},
                    models.PreResNet(10, 20).to(dev), train, device=dev)
ens = s.sample()
res = []
ROWS = [int(v[5:]) for v in sys.argv[1:] if v.startswith('rows=')] or [EnsembleAccumulator.EVAL_ROWS]
for lanes, rows in [(int(v), r) for v in ([a for a in sys.argv[1:] if not a.startswith('rows=')] or ['4']) for r in ROWS]:
    EnsembleAccumulator.LANES = lanes
    EnsembleAccumulator.EVAL_ROWS = rows
    for S in (3, 4, 8, 20):
        pred = tasks.Prediction({'in_distribution_test': test}, 10, dev, 'ALL')
        pred.update_statistics(ens[:S], output_performance=False)
        pred.reset()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pred.update_statistics(ens[:S], output_performance=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        r = dict(lanes=lanes, eval_rows=rows, members=S, seconds=round(dt, 4), preds_per_s=round(10000 / dt), member_forwards_per_s=round(S * 10000 / dt))
        print(r, flush=True); res.append(r)
EnsembleAccumulator.LANES = 4
EnsembleAccumulator.EVAL_ROWS = 1024
from torch.profiler import profile, ProfilerActivity
pred = tasks.Prediction({'in_distribution_test': test}, 10, dev, 'ALL')
pred.update_statistics(ens[:8], output_performance=False); pred.reset()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    pred.update_statistics(ens[:8], output_performance=False)
    torch.cuda.synchronize()
rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:14]
tot = sum(e.device_time_total for e in prof.key_averages())
print('total device ms', tot / 1e3)
for e in rows:
    print(f'   {e.key[:80]:80s} calls {e.count:6d}  avg {e.device_time_total / e.count:8.1f} us  {100 * e.device_time_total / tot:5.1f}%')
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/bma_probe.json', 'w'), indent=1)
