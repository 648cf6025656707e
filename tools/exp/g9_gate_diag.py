"""G9 (the reference's PreResNet-8 SGHMC run) replayed on the CPU through the oracle kernel set and on the GPU (eager),
ReLU gates of every train-mode BatchNorm call compared, next to the final predictive error. Test infrastructure only."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ursabench_amd.models as M  # noqa: E402
from ursabench_amd import fused_bn, inference, tasks, tuning  # noqa: E402

tuning.use_shipped_miopen_db()
from oracle_kernels import OracleKernels  # noqa: E402
from test_samplers_cpu import _load_preresnet8, _preresnet8_inputs, pad_eps  # noqa: E402


def replay(dev, g, fused):
    hyp = json.loads(str(g['hyper']))
    train, test = _preresnet8_inputs(g)
    gates = []
    orig, orig_add = M.bn_relu, M.add_bn_relu

    def spy(bn, x, relu=True):
        y = orig(bn, x, relu)
        if bn.training:
            gates.append((y.detach() > 0).cpu())
        return y

    def spy_add(bn, x, relu=True):
        z, y = orig_add(bn, x, relu)
        if bn.training:
            gates.append((y.detach() > 0).cpu())
        return z, y
    M.bn_relu, M.add_bn_relu = spy, spy_add
    fused_bn.enabled(fused)
    try:
        if dev == 'cpu':
            K = OracleKernels()
            s = inference.SGHMC(dict(hyp), _load_preresnet8(g), train, kernels=K, use_graph=False)
            s.eps_provider = lambda k: pad_eps(s.arena, g['eps'][k])
            ens = s.sample()
            pred = tasks.Prediction({'in_distribution_test': test}, 10, torch.device('cpu'), 'ALL', kernels=K)
        else:
            s = inference.SGHMC(dict(hyp), _load_preresnet8(g), train, device=torch.device('cuda'), use_graph=False)

            def eps(k):
                e = torch.zeros(s.arena.n, device='cuda')
                e[s.arena.layout.gather_index(torch.device('cuda'))] = torch.tensor(g['eps'][k], device='cuda')
                return e
            s.eps_provider = eps
            ens = s.sample()
            pred = tasks.Prediction({'in_distribution_test': test}, 10, torch.device('cuda'), 'ALL')
        pred.update_statistics(ens, output_performance=False)
    finally:
        M.bn_relu, M.add_bn_relu = orig, orig_add
        fused_bn.enabled(True)
    return gates, pred.ensemble_proba.numpy()


def main():
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'e2e_preresnet8.npz'))
    gc, pc = replay('cpu', g, False)
    print('cpu replay vs golden proba:', float(np.abs(pc / g['proba_sum'] - 1).max()))
    for fused in (True, False):
        gg, pg = replay('cuda', g, fused)
        flips = [int((a != b).sum()) for a, b in zip(gg, gc)]
        print(json.dumps(dict(fused=fused, train_mode_bn_calls=len(gg), calls_cpu=len(gc), gates=int(sum(a.numel() for a in gc)),
                              flips_per_call=flips, total_flips=sum(flips),
                              max_rel_err_proba=float(np.abs(pg / g['proba_sum'] - 1).max()))))


if __name__ == '__main__':
    main()
