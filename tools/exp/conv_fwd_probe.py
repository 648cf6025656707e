"""K8 (ursa_conv3x3_f32) against the CPU float64 convolution and against MIOpen's launch for the same call, forward and
input gradient, per stride-1 3x3 layer shape of the CIFAR pre-activation ResNets: error, bit reproducibility, and (run it under
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
for cin, cout, hw in ((16, 16, 32), (3, 16, 32), (32, 32, 16), (64, 64, 8)):
    for n in (128, 3):
        torch.manual_seed(n + cin)
        xs = [torch.randn(n, cin, hw, hw, device=dev) for _ in range(8 if n == 128 else 1)]
        dys = [torch.randn(n, cout, hw, hw, device=dev) for _ in range(8 if n == 128 else 1)]
        w = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
        x, dy = xs[0], dys[0]
        rec = dict(cin=cin, cout=cout, hw=hw, n=n)
        y = K.conv3x3(x, w)
        ref = F.conv2d(x.double().cpu(), w.double().cpu(), None, 1, 1)
        mi = F.conv2d(x, w, None, 1, 1)
        sc = float(ref.abs().max())
        rec.update(fwd_err_k8=float((y.double().cpu() - ref).abs().max()) / sc, fwd_err_miopen=float((mi.double().cpu() - ref).abs().max()) / sc,
                   fwd_bit_equal_runs=bool(torch.equal(y, K.conv3x3(x, w))))
        if K.conv3x3_supported(dy.shape, cin):
            dx = K.conv3x3(dy, w, flip=True)
            refd = torch.nn.grad.conv2d_input(x.shape, w.double().cpu(), dy.double().cpu(), 1, 1)
            mid = torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
            sc = float(refd.abs().max())
            rec.update(dgrad_err_k8=float((dx.double().cpu() - refd).abs().max()) / sc,
                       dgrad_err_miopen=float((mid.double().cpu() - refd).abs().max()) / sc)
        if n == 128:
            yb = torch.empty_like(y)
            rec['us_k8_fwd'] = timed(lambda i: K.conv3x3(xs[i % 8], w, yb))
            rec['us_miopen_fwd'] = timed(lambda i: F.conv2d(xs[i % 8], w, None, 1, 1))
            if 'dgrad_err_k8' in rec:
                db = torch.empty_like(x)
                rec['us_k8_dgrad'] = timed(lambda i: K.conv3x3(dys[i % 8], w, db, flip=True))
                rec['us_miopen_dgrad'] = timed(lambda i: torch.ops.aten.convolution_backward(
                    dys[i % 8], x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False]))
        out.append(rec)
        print(rec, flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'conv_fwd_probe.json'), 'w'), indent=1)
