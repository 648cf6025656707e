"""CPU: the torch-CPU port used as bench.py's cpu_baseline is bitwise equal to the reference
(golden trajectories captured with the same global-generator seeds)."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('torch_cpu_path', os.path.join(ROOT, 'oracle', 'torch_cpu_path.py'))
port = importlib.util.module_from_spec(spec)
spec.loader.exec_module(port)

K1_CASES = ['sghmc_wd_noise', 'sghmc_nowd_mixed', 'sghmc_sched', 'sgld_wd_noise', 'sgld_nonoise']


@pytest.mark.parametrize('ci,case', list(enumerate(K1_CASES)))
def test_port_step_bitwise_vs_reference(golden_dir, ci, case):
    g = np.load(os.path.join(golden_dir, 'k1_steps.npz'))
    shapes = json.loads(str(g['shapes']))
    sizes = [int(np.prod(s)) for s in shapes]
    split = lambda v: [torch.tensor(c).view(s) for c, s in zip(np.split(v, np.cumsum(sizes)[:-1]), shapes)]
    momentum, wd, N = g[f'{case}/hyper']
    params = [torch.nn.Parameter(t) for t in split(g[f'{case}/theta0'])]
    state = {}
    for k, lr in enumerate(g[f'{case}/lr']):
        for p, gr in zip(params, split(g[f'{case}/grad'][k])):
            p.grad = gr
        torch.manual_seed(7000 + 31 * ci + k)                  # the seeds tools/gen_golden.py used
        port.sgmcmc_step_per_tensor(params, state, lr=float(lr), momentum=float(momentum), weight_decay=float(wd),
                                    num_training_samples=int(N), add_langevin_noise=bool(g[f'{case}/noise'][k]))
        flat = torch.cat([p.detach().reshape(-1) for p in params]).numpy()
        assert np.array_equal(flat, g[f'{case}/theta'][k]), (case, k)


def test_port_prediction_bitwise_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, 'tasks.npz'))
    for tag in ('c10', 'c100'):
        B = int(g[f'{tag}/batch'])
        x = torch.tensor(g[f'{tag}/x'])
        ms = []
        for W, b in zip(g[f'{tag}/W'], g[f'{tag}/b']):
            m = torch.nn.Linear(W.shape[1], W.shape[0])
            with torch.no_grad():
                m.weight.copy_(torch.tensor(W)); m.bias.copy_(torch.tensor(b))
            ms.append(m)
        batches = [(x[i:i + B], None) for i in range(0, len(x), B)]
        p, e, rows, _ = port.prediction_accumulate(ms, batches, W.shape[0], len(x))
        assert rows == len(x)
        assert np.array_equal(p.numpy(), g[f'{tag}/pred_proba']) and np.array_equal(e.numpy(), g[f'{tag}/pred_ent'])
