"""K7 (ursa_conv_wgrad_f32) against the CPU float64 weight gradient and against MIOpen's own launch sequence, per layer
shape of the CIFAR pre-activation ResNets: error, run-to-run bit equality, microseconds per call (HIP events, 200 calls
back to back on one stream; MIOpen's figure includes its transposes / zero fills - they are part of what a call costs).

    python tools/exp/conv_wgrad_probe.py -> gpurun_out/conv_wgrad_probe.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ursabench_amd import _native, tuning  # noqa: E402

tuning.use_shipped_miopen_db()
dev = torch.device('cuda', 0)
K = _native.knobs_kernels() if os.environ.get('URSA_PROBE_KNOBS') == '1' else _native.default_kernels()
REPS = 200


def stock(x, dy, w, stride):
    pad = w.shape[2] // 2
    return torch.ops.aten.convolution_backward(dy, x, w, None, [stride, stride], [pad, pad], [1, 1], False, [0, 0], 1,
                                               [False, True, False])[1]


def timed(fn):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / REPS


out = []
shapes = [(3, 16, 32, 3, 1), (16, 16, 32, 3, 1), (32, 32, 16, 3, 1), (64, 64, 8, 3, 1), (16, 32, 32, 3, 2), (32, 64, 16, 3, 2),
          (16, 32, 32, 1, 2), (32, 64, 16, 1, 2)]
for cin, cout, hw, ks, stride in shapes:
    for n in (128, 3):
        torch.manual_seed(n + cin)
        x = torch.randn(n, cin, hw, hw, device=dev)
        dy = torch.randn(n, cout, hw // stride, hw // stride, device=dev)
        w = torch.randn(cout, cin, ks, ks, device=dev)
        wsn = K.conv_wgrad_ws_floats(x.shape, cout, ks, stride)
        rec = dict(cin=cin, cout=cout, hw=hw, ksize=ks, stride=stride, n=n, ws_floats=wsn)
        if not wsn:
            rec['covered'] = False
            out.append(rec)
            continue
        ws = torch.empty(wsn, device=dev)
        dw = torch.full((cout, cin, ks, ks), float('nan'), device=dev)
        K.conv_wgrad(x, dy, dw, ws, stride)
        dw2 = torch.empty_like(dw)
        ws.fill_(float('nan'))
        K.conv_wgrad(x, dy, dw2, ws, stride)
        ref = torch.nn.grad.conv2d_weight(x.double().cpu(), w.shape, dy.double().cpu(), stride, ks // 2)
        mi = stock(x, dy, w, stride)
        scale = float(ref.abs().max())
        rec.update(err_k7=float((dw.double().cpu() - ref).abs().max()) / scale, err_miopen=float((mi.double().cpu() - ref).abs().max()) / scale,
                   bit_equal_runs=bool(torch.equal(dw, dw2)), finite=bool(torch.isfinite(dw).all()))
        if n == 128:
            rec['us_k7'] = timed(lambda: K.conv_wgrad(x, dy, dw, ws, stride))
            rec['us_miopen'] = timed(lambda: stock(x, dy, w, stride))
        out.append(rec)
        print(rec, flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', os.environ.get('URSA_PROBE_OUT', 'conv_wgrad_probe.json')), 'w'), indent=1)
