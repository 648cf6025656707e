"""us per launch and GB/s of the K12 launches (1x1 / stride 1 convolutions, batch 1,024 = an HMC chunk) against torch's
convolution (MIOpen) for the same call, per layer shape of PreResNet-164.
    python3 tools/k12_bench.py [out.json]"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ursabench_amd import _native  # noqa: E402

DEV = torch.device('cuda', 0)
K = _native.default_kernels()
N = int(os.environ.get('URSA_K12_ROWS', '1024'))
SHAPES = [(64, 16, 32), (16, 64, 32), (128, 32, 16), (32, 128, 16), (256, 64, 8), (64, 256, 8)]


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return round(best, 1)


out = {}
for cin, cout, hw in SHAPES:
    x = torch.randn(N, cin, hw, hw, device=DEV)
    dy = torch.randn(N, cout, hw, hw, device=DEV)
    w = torch.randn(cout, cin, 1, 1, device=DEV) * 0.1
    y, dx, dw = torch.empty_like(dy), torch.empty_like(x), torch.empty_like(w)
    ws = torch.empty(K.conv_wgrad_ws_floats(x.shape, cout, 1, 1), device=DEV)
    nbytes = 4 * (x.numel() + dy.numel())
    r = {'fwd': timed(lambda: K.conv1x1(x, w, y)), 'dgrad': timed(lambda: K.conv1x1(dy, w, dx, flip=True)),
         'wgrad': timed(lambda: K.conv_wgrad(x, dy, dw, ws, 1)),
         'miopen_fwd': timed(lambda: F.conv2d(x, w)),
         'miopen_bwd_both': timed(lambda: torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, True, False]))}
    r['fwd_gbps'] = round(nbytes / (r['fwd'] * 1e-6) / 1e9)
    r['dgrad_gbps'] = round(nbytes / (r['dgrad'] * 1e-6) / 1e9)
    r['wgrad_gbps'] = round(nbytes / (r['wgrad'] * 1e-6) / 1e9)
    out[f'{cin}->{cout}@{hw}'] = r
    print(f'{cin}->{cout}@{hw}', json.dumps(r), flush=True)
if len(sys.argv) > 1:
    json.dump(dict(what=f'us per launch, batch {N} (tools/k12_bench.py); GB/s = 4 B x (elements of x + elements of y) / time', shapes=out), open(sys.argv[1], 'w'), indent=1)
