"""Debug: the held form's sync words after back-to-back launches, and a quick timing."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import _native  # noqa: E402

K = _native.default_kernels()
for shape in ((1024, 64, 32, 32), (128, 160, 32, 32)):
    C = shape[1]
    x, dy = torch.randn(shape, device='cuda'), torch.randn(shape, device='cuda')
    y, dx = torch.empty_like(x), torch.empty_like(x)
    w, b = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    sm, si, dg, db = (torch.zeros(C, device='cuda') for _ in range(4))
    ws = torch.zeros(_native.bn_ws_floats(C), device='cuda')
    for name, fn in (('fwd', lambda: K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, held=True)),
                     ('bwd', lambda: K.bn_relu_backward(x, dy, dx, w, b, sm, si, dg, db, ws, held=True)),
                     ('fwd two-launch', lambda: K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, two_launch=True)),
                     ('bwd two-launch', lambda: K.bn_relu_backward(x, dy, dx, w, b, sm, si, dg, db, ws, two_launch=True))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        e.record()
        torch.cuda.synchronize()
        if 'two' in name:
            ws.zero_()
        sw = ws[C * 256:].view(torch.int32)
        nz = sw.nonzero().flatten().tolist()
        print(shape, name, 'us/call', round(a.elapsed_time(e) * 1e3 / 20, 1), 'nonzero sync words', [(i, int(sw[i])) for i in nz[:10]])
