"""relu(bn(x)) forward / backward per layer shape: the K6 launches vs the stock MIOpen / ATen launches, HIP events
around hipGraph replays of 50 back-to-back calls (what a layer costs inside the captured training step).
    python3 tools/exp/bn_fused_bench.py > gpurun_out/bn_fused_bench.json
"""
import json
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import fused_bn, tuning  # noqa: E402

tuning.use_shipped_miopen_db()

SHAPES = [(128, 16, 32, 32), (128, 32, 16, 16), (128, 64, 8, 8),            # PreResNet-20 (C2)
          (128, 160, 32, 32), (128, 320, 16, 16), (128, 640, 8, 8),         # WideResNet-28-10 (C4)
          (1024, 64, 32, 32), (1024, 128, 16, 16), (1024, 256, 8, 8)]       # PreResNet-164 at the HMC batch (C5)
REPS = 50


def timed(fn, reps=REPS, rounds=7):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(rounds):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / reps)
    out.sort()
    return out[len(out) // 2]


def main():
    res = []
    for shape in SHAPES:
        C = shape[1]
        x = torch.randn(shape, device='cuda')
        dy = torch.randn(shape, device='cuda')
        row = dict(shape=list(shape), mbytes=x.numel() * 4 / 1e6)
        for fused in (True, 'two_launch', False):
            bn = nn.BatchNorm2d(C).cuda().train()
            fused_bn.enabled(bool(fused))
            fused_bn._two_launch = fused == 'two_launch'
            xg = x.clone().requires_grad_(True)

            def fwd():
                with torch.no_grad():
                    return fused_bn.bn_relu(bn, x)

            def fwd_bwd():
                y = fused_bn.bn_relu(bn, xg)
                torch.autograd.grad(y, (xg, bn.weight, bn.bias), dy)

            bn.eval()

            def ev():
                with torch.no_grad():
                    return fused_bn.bn_relu(bn, x)
            t_eval = timed(ev)
            bn.train()
            t_f = timed(fwd)
            t_fb = timed(fwd_bwd)
            key = {True: 'fused', 'two_launch': 'fused_two_launch', False: 'stock'}[fused]
            row[key] = dict(fwd_us=round(t_f, 2), bwd_us=round(t_fb - t_f, 2), eval_us=round(t_eval, 2))
        fused_bn.enabled(True)
        fused_bn._two_launch = False
        n = x.numel() * 4
        # algorithmic minimum: forward x in + y out = 8 B/element, backward x, dy in + dx out = 12, evaluation 8
        # (the two-launch form moves 12 / 20: it reads its inputs twice)
        row['fused']['fwd_frac_of_8TBs'] = round(8 * x.numel() / (row['fused']['fwd_us'] * 1e-6) / 8e12, 3)
        row['fused']['bwd_frac_of_8TBs'] = round(12 * x.numel() / (row['fused']['bwd_us'] * 1e-6) / 8e12, 3)
        row['fused']['eval_frac_of_8TBs'] = round(8 * x.numel() / (row['fused']['eval_us'] * 1e-6) / 8e12, 3)
        row['fused']['form'] = 'one-pass' if (C >= 48 and x.numel() // C <= 32768) else 'two-launch' 
        res.append(row)
        print(json.dumps(row), file=sys.stderr, flush=True)
    print(json.dumps(dict(what='relu(bn(x)) per layer, us per call inside a hipGraph of 50 calls (HIP events, median of 7)',
                          device=torch.cuda.get_device_name(0), rows=res), indent=1))


if __name__ == '__main__':
    main()
