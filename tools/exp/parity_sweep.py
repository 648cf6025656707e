"""Margin of bench.py's in-run parity leg (GPU path vs torch-CPU port, PreResNet, injected noise) over
its size parameters: how far from 1e-5 relative is the predictive after a few SGHMC steps?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
dev = torch.device('cuda', 0)
res = []
for kw in (dict(), dict(rows=64), dict(steps_per_sample=2), dict(depth=8), dict(samples=4)):
    for rep in range(2):
        try:
            out = bench.parity_block(dev, **kw)
        except AssertionError as e:
            out = json.loads(str(e).split(': ', 2)[2])
        r = dict(kw, rep=rep, sampler=out['sampler_first_sample'], bma=out['bma_same_members'], growth=out['trajectory_growth_reported_not_asserted'])
        print(json.dumps(r), flush=True); res.append(r)
json.dump(res, open('gpurun_out/parity_sweep.json', 'w'), indent=1)
