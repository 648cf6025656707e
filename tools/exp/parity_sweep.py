"""Margin of bench.py's in-run parity leg (GPU path vs torch-CPU port, PreResNet, injected noise) over
its size parameters: how far from 1e-5 relative is the predictive after a few SGHMC steps?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device('cuda', 0)
res = []
for kw in (dict(steps_per_sample=1, samples=3, rows=64), dict(steps_per_sample=1, samples=3, rows=128),
           dict(steps_per_sample=1, samples=3, rows=128, n_noise=50000), dict(steps_per_sample=2, samples=3, rows=128, n_noise=50000),
           dict(steps_per_sample=1, samples=3, rows=128, depth=8, n_noise=50000), dict(steps_per_sample=1, samples=1, rows=128, n_noise=50000),
           dict(steps_per_sample=1, samples=2, rows=128, n_noise=50000), dict(steps_per_sample=1, samples=3, rows=64, n_noise=50000)):
    for rep in range(2):
        try:
            out = bench.parity_block(dev, **kw)
        except AssertionError as e:
            out = eval(str(e).split(': ', 2)[2])
        r = dict(kw, rep=rep, proba=out['max_rel_err_proba'], ent=out['max_rel_err_entropy'], params=out['max_abs_diff_params_last_member'])
        print(r, flush=True); res.append(r)
json.dump(res, open('gpurun_out/parity_sweep.json', 'w'), indent=1)
