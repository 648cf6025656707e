"""BMA evaluation of WideResNet-28-10 members (BASELINE configs[3]): how many twin lanes (parallel graph branches),
and does merging loader batches help? Random members, 2,048 test rows."""
import os, sys, tempfile, time, json
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='ursa_wrn_'))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import models, tasks, util
from ursabench_amd.arena import FlatArena, MemberBank
from ursabench_amd.data import synthetic
from ursabench_amd.tasks.task_base import EnsembleAccumulator
dev = torch.device('cuda', 0)
util.set_random_seed(0)
net = models.WideResNet(100, 28, 10).to(dev)
arena = FlatArena(net.parameters(), module=net)
bank = MemberBank(arena)
ens = []
for k in range(4):
    arena.theta.add_(torch.randn_like(arena.theta) * 1e-3)
    ens.append(bank.snapshot(net))
test = synthetic(2048, (3, 32, 32), 100, seed=1, device=dev, batch_size=128)
res = []
for lanes, rows, maxp in ((1, 0, 0), (2, 0, 0), (4, 0, 0), (1, 512, 10**9), (4, 512, 10**9), (1, 256, 10**9)):
    EnsembleAccumulator.LANES, EnsembleAccumulator.EVAL_ROWS, EnsembleAccumulator.MERGE_MAX_PARAMS = lanes, rows, maxp
    pred = tasks.Prediction({'in_distribution_test': test}, 100, dev, 'ALL')
    pred.update_statistics(ens, output_performance=False)
    pred.reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pred.update_statistics(ens, output_performance=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    r = dict(lanes=lanes, eval_rows=rows or 128, seconds=round(dt, 3), member_forwards_per_s=round(4 * 2048 / dt))
    print(r, flush=True); res.append(r)
    del pred
json.dump(res, open('gpurun_out/wrn_lanes.json', 'w'), indent=1)
