"""us per launch of the evaluation-mode fused unit (ursa_preact_conv3x3_f32 with URSA_PREACT_EVAL) at the BMA predictive's batch
(4,096 rows) against K6's evaluation launch + K8 / MIOpen's convolution, per unit shape; with the knobs build URSA_K8_EVAL_IPW sets
the images one workgroup walks (weights staged once, the next image's rows loaded under this one's matrix work).
    python3 tools/k10_eval_bench.py [out.json]"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ursabench_amd import _native  # noqa: E402

DEV = torch.device('cuda', 0)
K = _native.knobs_kernels() if os.environ.get('URSA_K10_KNOBS') == '1' else _native.default_kernels()
N = int(os.environ.get('URSA_EVAL_ROWS', '4096'))
UNITS = [(16, 16, 32, 1), (32, 32, 16, 1), (64, 64, 8, 1), (16, 32, 32, 2), (32, 64, 16, 2)]


def timed(fn, reps=10):
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
for cin, cout, hw, st in UNITS:
    x = torch.randn(N, cin, hw, hw, device=DEV)
    w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.1
    gamma, beta = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.1
    rm, rv = torch.randn(cin, device=DEV) * 0.2, torch.rand(cin, device=DEV) + 0.5
    y = torch.empty(N, cout, hw // st, hw // st, device=DEV)
    h = torch.empty_like(x)
    add = torch.randn_like(y)
    flops = 2 * N * (hw // st) ** 2 * cin * cout * 9
    r = {'k6_eval': timed(lambda: K.bn_relu_eval(x, h, gamma, beta, rm, rv, eps=1e-5)),
         'k8': timed(lambda: K.conv3x3(h, w, y, stride=st)),
         'miopen': timed(lambda: F.conv2d(h, w, None, st, 1)),
         'fused': timed(lambda: K.preact_eval(x, w, y, gamma, beta, rm, rv, eps=1e-5, stride=st))}
    if st == 1:
        r['fused_add'] = timed(lambda: K.preact_eval(x, w, y, gamma, beta, rm, rv, eps=1e-5, add=add))
    r['fused_tflops'] = round(flops / (r['fused'] * 1e-6) / 1e12, 1)
    out[f'{cin}x{cout}x{hw}s{st}'] = r
    print(f'{cin}x{cout}x{hw}s{st}', json.dumps(r), flush=True)
if len(sys.argv) > 1:
    json.dump(dict(what=f'us per launch, batch {N} (tools/k10_eval_bench.py)', ipw=os.environ.get('URSA_K8_EVAL_IPW'), units=out), open(sys.argv[1], 'w'), indent=1)
