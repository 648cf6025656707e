"""Runs ONLY the BMA evaluation of configs[1] (PreResNet-20, 10,000 test rows, bank-resident members through the twin: member
forwards over merged batches of 4,096 + 4,096 + 1,808 rows) so that MIOpen, started with MIOPEN_FIND_ENFORCE=3 and a copy of
the shipped user databases, tunes exactly the evaluation shapes those databases lack (they were recorded when the twin merged
up to 1,024 rows: batch sizes 1,024 / 784), and then a few 32-row training steps (the small trials of bench.py's parity leg).
See tools/miopen_tune_bma.sh."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ursabench_amd import inference, models, tasks, util
from ursabench_amd.data import synthetic
dev = torch.device('cuda', 0)
util.set_random_seed(0)
train = synthetic(1280, (3, 32, 32), 10, seed=0, device=dev, batch_size=128)
test = synthetic(10000, (3, 32, 32), 10, seed=1, device=dev, batch_size=128)
s = inference.SGHMC({'lr': 0.1, 'prior_std': 0.5, 'alpha': 0.5, 'burn_in_epochs': 0, 'num_samples': 5}, models.PreResNet(10, 20).to(dev), train, device=dev, seed=0)
ens = s.sample()
pred = tasks.Prediction({'in_distribution_test': test}, 10, dev, 'ALL')
pred.update_statistics(ens, output_performance=False)
torch.cuda.synchronize()
print('members', pred.num_samples_collected, 'engine', pred._acc.stats)
# ... and the 32-row training shapes of bench.py's parity leg (its small trials): forward, backward-data, weight gradient
small = synthetic(32 * 6, (3, 32, 32), 10, seed=2, device=dev, batch_size=32)
s32 = inference.SGHMC({'lr': 0.1, 'prior_std': 0.5, 'alpha': 0.5, 'burn_in_epochs': 0, 'num_samples': 1}, models.PreResNet(10, 20).to(dev), small, device=dev, seed=1)
s32.sample()
torch.cuda.synchronize()
print('32-row steps', s32.engine.stats)
