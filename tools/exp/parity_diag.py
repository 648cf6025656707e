"""Where does a GPU sample differ from the torch-CPU port's after ONE SGHMC step (bench.py parity_block's setting)?
Per state_dict entry: max |gpu - cpu| / max |cpu|, for fused / stock BatchNorm and graph / eager stepping."""
import copy
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ursabench_amd import fused_bn, inference, models  # noqa: E402
from ursabench_amd.data import DeviceLoader  # noqa: E402


def run(fused, use_graph, rows=128):
    dev = torch.device('cuda')
    port = bench.load_port()
    fused_bn.enabled(fused)
    torch.manual_seed(4242)
    net_cpu = models.PreResNet(10, 20)
    net_gpu = copy.deepcopy(net_cpu)
    g = torch.Generator().manual_seed(4243)
    xtr, ytr = torch.randn(rows, 3, 32, 32, generator=g), torch.randint(0, 10, (rows,), generator=g)
    hyp = dict(bench.HYP, num_samples=1)
    torch.manual_seed(777)
    eps_steps = [[torch.randn_like(p) for p in net_cpu.parameters()]]
    train = DeviceLoader(xtr.to(dev), ytr.to(dev), rows)
    s = inference.SGHMC(dict(hyp), net_gpu, train, device=dev, seed=1, use_graph=use_graph)
    s.optimizer.param_groups[0]['num_training_samples'] = 50000
    idx = s.arena.layout.gather_index(dev)

    def eps(k):
        e = torch.zeros(s.arena.n, device=dev)
        e[idx] = torch.cat([t.reshape(-1) for t in eps_steps[0]]).to(dev)
        return e
    s.eps_provider = eps
    a = s.arena
    if use_graph:
        saved = (a.theta.clone(), None if a.fbuf is None else a.fbuf.clone(), [b.clone() for _, b in a.ibufs])
        s.engine.WARMUP_STEPS = 1
        for _ in range(2):
            s.engine.run_epoch(train, True, eps_per_step=eps)
        with torch.no_grad():
            a.theta.copy_(saved[0])
            a.fbuf.copy_(saved[1])
            for (_, b), v in zip(a.ibufs, saved[2]):
                b.copy_(v)
            a.mom.zero_()
        s.optimizer._step, s.optimizer._has_mom = 0, [False]
        s.optimizer.state.clear()
    lr = s.optimizer.param_groups[0]['lr']
    member = s.sample_iterative()
    torch.manual_seed(777)
    state = {}
    port.sghmc_epoch(net_cpu, [(xtr, ytr)], state, lr=lr, momentum=1 - bench.HYP['alpha'],
                     weight_decay=1 / bench.HYP['prior_std'] ** 2, num_training_samples=50000)
    sd_g, sd_c = member.state_dict(), net_cpu.state_dict()
    worst = []
    for k, vc in sd_c.items():
        vg = sd_g[k].detach().cpu()
        if vc.dtype.is_floating_point:
            worst.append((float((vg - vc).abs().max()) / max(float(vc.abs().max()), 1e-30), k))
        else:
            worst.append((float((vg - vc).abs().max()), k))
    worst.sort(reverse=True)
    fused_bn.enabled(True)
    return dict(fused=fused, graph=use_graph, engine=dict(s.engine.stats), worst=[(f'{e:.3e}', k) for e, k in worst[:8]])


if __name__ == '__main__':
    for fused in (True, False):
        for graph in (False, True):
            print(json.dumps(run(fused, graph)), flush=True)
