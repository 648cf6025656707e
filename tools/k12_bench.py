"""us per launch and GB/s of the K12 launches (1x1 / stride 1 convolutions, batch 1,024 = an HMC chunk) against torch's
convolution (MIOpen) for the same call, per layer shape of PreResNet-164; K13 (statistics + merge, forward / weight gradient with
relu(bn(x)) rebuilt while staged) and K14 (the narrowing layers' backward: sums pass + merge + dx pass) against the K6 launches they
replace.
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
    # K13: statistics launch + merge, then the forward / weight gradient with relu(bn(x)) rebuilt while staged; the K6 launches they replace
    gamma, beta = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.1
    save, bws = torch.empty(4, cin, device=DEV), torch.empty(_native.bn_ws_floats(cin), device=DEV)
    h, addend, z = torch.empty_like(x), torch.randn_like(x), torch.empty_like(x)
    r['k13_stats'] = timed(lambda: K.bn_stats(x, gamma, beta, None, None, save, bws, eps=1e-5, momentum=0.0))
    r['k13_stats_add'] = timed(lambda: K.bn_stats(x, gamma, beta, None, None, save, bws, eps=1e-5, momentum=0.0, addend=addend, z_out=z))
    r['k6_fwd_two_launch'] = timed(lambda: K.bn_relu_forward(x, h, gamma, beta, None, None, save[0], save[1], bws, eps=1e-5, momentum=0.0, relu=True,
                                                              two_launch=True, save_gate=save[2:]))
    r['k13_fwd'] = timed(lambda: K.preact_conv1x1(x, save, w, y))
    r['k13_wgrad'] = timed(lambda: K.conv_wgrad_reduce([(K.preact_wgrad1x1_partial(x, save, dy, w.shape, ws), dw)]))
    r['k13_fwd_gbps'] = round(nbytes / (r['k13_fwd'] * 1e-6) / 1e9)
    r['k13_stats_gbps'] = round(4 * x.numel() / (r['k13_stats'] * 1e-6) / 1e9)
    r['k13_stats_add_gbps'] = round(12 * x.numel() / (r['k13_stats_add'] * 1e-6) / 1e9)
    # K14 (narrowing layers): the flipped GEMM twice against K12's flipped launch + K6's two backward launches
    if K.preact_conv1x1_bwd_nl(dy.shape, cin) > 0:
        dgb, dz, dh = torch.empty(2, cin, device=DEV), torch.randn_like(x), torch.empty_like(x)
        r['k14_all'] = timed(lambda: K.preact_conv1x1_bwd(dy, w, x, save, gamma, dx, dgb[0], dgb[1], dz=dz))
        r['k12_flip_plus_k6_bwd'] = timed(lambda: (K.conv1x1(dy, w, dh, flip=True),
                                                   K.bn_relu_backward(x, dh, dx, gamma, beta, save[0], save[1], dgb[0], dgb[1], bws, relu=True, dz=dz,
                                                                      two_launch=True, gate=save[2:])))
        k14_bytes = 4 * ((dy.numel() + x.numel()) + (dy.numel() + 3 * x.numel()))          # sums pass + dx pass (x, dz read, dx written)
        r['k14_gbps'] = round(k14_bytes / (r['k14_all'] * 1e-6) / 1e9)
    r['fwd_gbps'] = round(nbytes / (r['fwd'] * 1e-6) / 1e9)
    r['dgrad_gbps'] = round(nbytes / (r['dgrad'] * 1e-6) / 1e9)
    r['wgrad_gbps'] = round(nbytes / (r['wgrad'] * 1e-6) / 1e9)
    out[f'{cin}->{cout}@{hw}'] = r
    print(f'{cin}->{cout}@{hw}', json.dumps(r), flush=True)
if len(sys.argv) > 1:
    json.dump(dict(what=f'us per launch, batch {N} (tools/k12_bench.py); GB/s = 4 B x (elements of x + elements of y) / time; k13_stats: 4 B x elements of x (with addend: 12 B); k14: 4 B x (dy + x) + 4 B x (dy + 3 x) over the three launches', shapes=out), open(sys.argv[1], 'w'), indent=1)
