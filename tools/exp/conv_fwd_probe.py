"""K8 (ursa_conv3x3_f32) and K9 (ursa_conv1x1s2_f32) against the CPU float64 convolution and against MIOpen's launches for the
same call, forward and input gradient, per layer shape of the CIFAR pre-activation ResNets: error, bit reproducibility, and (run it under
`rocprofv3 --kernel-trace --stats`) the kernels' durations. 200 calls each, alternating over 8 input buffers so that the
inputs are not L2-resident from the previous call.

    python tools/exp/conv_fwd_probe.py -> gpurun_out/conv_fwd_probe.json
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ursabench_amd import _native, tuning  # noqa: E402

tuning.use_shipped_miopen_db()
dev = torch.device('cuda', 0)
K = _native.knobs_kernels() if os.environ.get('URSA_PROBE_KNOBS') == '1' else _native.default_kernels()
REPS = 200


def timed(fn):
    for i in range(10):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(REPS):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / REPS


out = []
for cin, cout, hw, ks, st in ((16, 16, 32, 3, 1), (3, 16, 32, 3, 1), (32, 32, 16, 3, 1), (64, 64, 8, 3, 1), (16, 32, 32, 3, 2), (32, 64, 16, 3, 2),
                              (16, 32, 32, 1, 2), (32, 64, 16, 1, 2)):
    pad = ks // 2
    k_fwd = (lambda x, w, y=None: K.conv3x3(x, w, y, stride=st)) if ks == 3 else (lambda x, w, y=None: K.conv1x1s2(x, w, y))
    k_bwd = (lambda dy, w, y=None: K.conv3x3(dy, w, y, flip=True, stride=st)) if ks == 3 else (lambda dy, w, y=None: K.conv1x1s2(dy, w, y, flip=True))
    for n in (128, 3):
        torch.manual_seed(n + cin)
        xs = [torch.randn(n, cin, hw, hw, device=dev) for _ in range(8 if n == 128 else 1)]
        dys = [torch.randn(n, cout, hw // st, hw // st, device=dev) for _ in range(8 if n == 128 else 1)]
        w = torch.randn(cout, cin, ks, ks, device=dev) * 0.1
        x, dy = xs[0], dys[0]
        rec = dict(cin=cin, cout=cout, hw=hw, ksize=ks, stride=st, n=n)
        y = k_fwd(x, w)
        ref = F.conv2d(x.double().cpu(), w.double().cpu(), None, st, pad)
        mi = F.conv2d(x, w, None, st, pad)
        sc = float(ref.abs().max())
        rec.update(fwd_err=float((y.double().cpu() - ref).abs().max()) / sc, fwd_err_miopen=float((mi.double().cpu() - ref).abs().max()) / sc,
                   fwd_bit_equal_runs=bool(torch.equal(y, k_fwd(x, w))))
        has_bwd = cin != 3
        if has_bwd:
            dx = k_bwd(dy, w)
            refd = torch.nn.grad.conv2d_input(x.shape, w.double().cpu(), dy.double().cpu(), st, pad)
            mid = torch.ops.aten.convolution_backward(dy, x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])[0]
            sc = float(refd.abs().max())
            rec.update(dgrad_err=float((dx.double().cpu() - refd).abs().max()) / sc,
                       dgrad_err_miopen=float((mid.double().cpu() - refd).abs().max()) / sc)
        if n == 128:
            yb = torch.empty_like(y)
            rec['us_fwd_host_loop'] = timed(lambda i: k_fwd(xs[i % 8], w, yb))
            rec['us_miopen_fwd_host_loop'] = timed(lambda i: F.conv2d(xs[i % 8], w, None, st, pad))
            if has_bwd:
                db = torch.empty_like(x)
                rec['us_dgrad_host_loop'] = timed(lambda i: k_bwd(dys[i % 8], w, db))
                rec['us_miopen_dgrad_host_loop'] = timed(lambda i: torch.ops.aten.convolution_backward(
                    dys[i % 8], x, w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False]))
        out.append(rec)
        print(rec, flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'conv_fwd_probe.json'), 'w'), indent=1)
