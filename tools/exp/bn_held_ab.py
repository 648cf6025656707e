"""K6 at activations that do not fit the Infinity Cache-friendly regime: held form (ONE launch, inputs read once) vs the
two-launch form, us per call inside a hipGraph of 20 calls (HIP events, median of 7), fraction of 8 TB/s on the
ALGORITHMIC minimum (8 B/element forward, 12 backward).
    python tools/exp/bn_held_ab.py > gpurun_out/bn_held_ab.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import _native  # noqa: E402

K = _native.knobs_kernels() if os.environ.get('URSA_BN_HELD_MIN_MIB') else _native.default_kernels()
SHAPES = [(1024, 64, 32, 32), (1024, 16, 32, 32), (1024, 128, 16, 16), (1024, 256, 8, 8), (128, 160, 32, 32), (256, 64, 32, 32)]
if os.environ.get('URSA_BN_HELD_MIN_MIB'):          # experiment: the held form at the workload's own (cache-resident) layers
    SHAPES = [(128, 16, 32, 32), (128, 32, 16, 16), (128, 64, 16, 16)]
REPS = 50 if os.environ.get('URSA_BN_HELD_MIN_MIB') else 20


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay()
    torch.cuda.synchronize()
    out = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / REPS)
    return sorted(out)[3]


rows = []
for shape in SHAPES:
    C = shape[1]
    x, dy, dz, ad = (torch.randn(shape, device='cuda') for _ in range(4))
    y, dx, z = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    w, b = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    sm, si, dg, db = (torch.zeros(C, device='cuda') for _ in range(4))
    ws = torch.zeros(_native.bn_ws_floats(C), device='cuda')
    e = x.numel()
    row = dict(shape=list(shape), mbytes=round(e * 4 / 1e6, 1))
    for held in (True, False):
        ws.zero_()
        kw = dict(held=True) if held else dict(two_launch=True)
        f = timed(lambda: K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, **kw))
        bw = timed(lambda: K.bn_relu_backward(x, dy, dx, w, b, sm, si, dg, db, ws, **kw))
        fr = timed(lambda: K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, addend=ad, z_out=z, **kw))
        br = timed(lambda: K.bn_relu_backward(z, dy, dx, w, b, sm, si, dg, db, ws, dz=dz, **kw))
        row['held' if held else 'two_launch'] = dict(
            fwd_us=round(f, 1), bwd_us=round(bw, 1), fwd_frac=round(8 * e / (f * 1e-6) / 8e12, 3), bwd_frac=round(12 * e / (bw * 1e-6) / 8e12, 3),
            fwd_residual_us=round(fr, 1), bwd_residual_us=round(br, 1),
            fwd_residual_frac=round(16 * e / (fr * 1e-6) / 8e12, 3), bwd_residual_frac=round(16 * e / (br * 1e-6) / 8e12, 3))
    torch.cuda.synchronize()
    row['ws_clean'] = None
    rows.append(row)
    print(json.dumps(row), file=sys.stderr, flush=True)
    del x, dy, dz, ad, y, dx, z
    torch.cuda.empty_cache()
print(json.dumps(dict(what=__doc__.strip().split('\n')[0], rows=rows), indent=1))
