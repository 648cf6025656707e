"""BMA predictive (30 members x 10,000 rows, PreResNet-20) by member forwards run concurrently (EnsembleAccumulator.LANES) and rows per
evaluation forward (EVAL_ROWS_SMALL), with the round-6 evaluation units. python3 tools/exp/bma_lanes_rows_ab.py [out.json]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from ursabench_amd.tuning import use_shipped_miopen_db  # noqa: E402
use_shipped_miopen_db('ursa_bma_ab_miopen_')
from ursabench_amd import inference, models, tasks, util  # noqa: E402
from ursabench_amd.data import synthetic  # noqa: E402
from ursabench_amd.tasks import task_base  # noqa: E402

dev = torch.device('cuda', 0)
train = synthetic(128 * 40, (3, 32, 32), 10, seed=0, device=dev, batch_size=128)
test = synthetic(10000, (3, 32, 32), 10, seed=1, device=dev, batch_size=128)
util.set_random_seed(0)
s = inference.SGHMC({'lr': 0.1, 'prior_std': 0.5, 'num_samples': 30, 'alpha': 0.5, 'burn_in_epochs': 0}, models.PreResNet(10, 20).to(dev), train, device=dev)
ens = s.sample()
out = {}
for lanes in (1, 2, 4, 8):
    for rows in (1024, 4096, 10000):
        task_base.EnsembleAccumulator.LANES = lanes
        task_base.EnsembleAccumulator.EVAL_ROWS_SMALL = rows
        pred = tasks.Prediction({'in_distribution_test': test}, 10, dev, 'ALL')
        pred._acc.accumulate(ens)
        pred.reset()
        pred._acc.reset(entropy_too=True)
        best = 1e9
        for _ in range(3):
            pred.reset()
            pred._acc.reset(entropy_too=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pred.update_statistics(ens, output_performance=False)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        out[f'lanes{lanes}_rows{rows}'] = round(10000 / best, 1)
        print(f'lanes {lanes} rows {rows}: {10000 / best:.0f} BMA-preds/s', flush=True)
        del pred
        torch.cuda.empty_cache()
if len(sys.argv) > 1:
    json.dump(dict(what='BMA-preds/s, 30 members x 10,000 rows, PreResNet-20, by (LANES, EVAL_ROWS_SMALL)', results=out), open(sys.argv[1], 'w'), indent=1)
