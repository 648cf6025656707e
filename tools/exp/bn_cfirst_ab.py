"""K6 grid kernels (two-launch form, evaluation below 16 MiB): which grid dimension is the channel. URSA_BN_CFIRST_MAX_MIB (knobs
build): -1 = grid (splits, channels) everywhere (rounds 3-4), 100000 = grid (channels, splits) everywhere, n = up to n MiB (shipped: 16).
us per call (HIP events; hipGraph-batched below 64 MB) and fraction of 8 TB/s on the algorithmic minimum (8 / 12 B per element).
    URSA_BN_CFIRST_MAX_MIB=100000 python tools/exp/bn_cfirst_ab.py -> one JSON line     (tools/exp/bn_cfirst_ab.sh: side by side)"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import event_time_ms, HBM_PEAK_GBPS
from ursabench_amd import _native
K = _native.knobs_kernels()
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream()
rows = []
for shape in ((128, 16, 32, 32), (128, 32, 16, 16), (128, 64, 8, 8), (128, 160, 32, 32), (128, 320, 16, 16), (256, 64, 32, 32), (1024, 128, 16, 16),
              (1024, 64, 32, 32), (1024, 256, 8, 8), (512, 16, 32, 32)):
    C = shape[1]
    x, a, dy, dz = (torch.randn(shape, device=dev) for _ in range(4))
    y, z, dx = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    w, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    sm, si, dg, db = (torch.zeros(C, device=dev) for _ in range(4))
    ws = torch.zeros(_native.bn_ws_floats(C), device=dev)
    e = x.numel()
    resident = e * 4 < (64 << 20)
    row = dict(shape=list(shape), mbytes=round(e * 4 / 1e6, 1))
    forms = {'fwd': (8, lambda: K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, two_launch=True)),
             'bwd': (12, lambda: K.bn_relu_backward(x, dy, dx, w, b, sm, si, dg, db, ws, two_launch=True)),
             'fwd_residual': (16, lambda: K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, two_launch=True, addend=a, z_out=z)),
             'bwd_residual': (16, lambda: K.bn_relu_backward(z, dy, dx, w, b, sm, si, dg, db, ws, two_launch=True, dz=dz))}
    if e * 4 < (16 << 20):
        forms['eval'] = (8, lambda: K.bn_relu_eval(x, y, w, b, rm, rv, eps=1e-5))
    for name, (bpe, fn) in forms.items():
        bt = sorted(event_time_ms(fn, 1024, stream, graph_batch=128) if resident else event_time_ms(fn, 10, stream) for _ in range(5))
        row[name + '_us'] = round(bt[2] * 1e3, 2)
        row[name + '_frac'] = round(bpe * e / (bt[2] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 3)
    rows.append(row)
    del x, a, dy, dz, y, z, dx
print(json.dumps(dict(setting=os.environ.get('URSA_BN_CFIRST_MAX_MIB'), rows=rows)))
