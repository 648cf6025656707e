"""us per launch of the K10 launches against the K6 / K8 / K7 launches they replace, inside hipGraph replays at the workload
batch (128): the fused forward / input-gradient / weight-gradient forms of every unit shape, and the `dx` launch.
    python3 tools/k10_bench.py [out.json]          (URSA_K10_KNOBS=1: bind csrc/libursa_hip_knobs.so so URSA_K10_DBG applies)
Each timed graph holds REP launches on rotating operand sets (so consecutive launches do not hit the same lines); HIP events
around 20 replays."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from ursabench_amd import _native  # noqa: E402

DEV = torch.device('cuda', 0)
K = _native.knobs_kernels() if os.environ.get('URSA_K10_KNOBS') == '1' else _native.default_kernels()
N, REP, SETS = 128, 24, 6
UNITS = [(16, 16, 32, 1), (32, 32, 16, 1), (64, 64, 8, 1), (16, 32, 32, 2), (32, 64, 16, 2)]
if os.environ.get('URSA_K10_DBG'):
    UNITS = UNITS[:1]                      # the knob's variants exist for the first shape only
if os.environ.get('URSA_K10_UNITS'):
    UNITS = UNITS[:int(os.environ['URSA_K10_UNITS'])]


def timed(fn):
    """fn(i) enqueues launch i; returns us per launch inside a replayed graph."""
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(3):
            fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        for i in range(REP):
            fn(i)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / (20 * REP))
    return round(best, 2)


def sums(x):
    xd = x.double()
    return torch.stack([xd.sum((0, 2, 3)), (xd * xd).sum((0, 2, 3))], -1)[:, None, :].contiguous()


out = {}
for cin, cout, hw, st in UNITS:
    ho = hw // st
    xs = [torch.randn(N, cin, hw, hw, device=DEV) for _ in range(SETS)]
    ys = [torch.empty(N, cout, ho, ho, device=DEV) for _ in range(SETS)]
    adds = [torch.randn(N, cout, ho, ho, device=DEV) for _ in range(SETS)]
    dys = [torch.randn(N, cout, ho, ho, device=DEV) for _ in range(SETS)]
    gs = [torch.empty(N, cin, hw, hw, device=DEV) for _ in range(SETS)]
    w = torch.randn(cout, cin, 3, 3, device=DEV) * 0.1
    gamma, beta = torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV) * 0.1
    ip = sums(xs[0])
    save = torch.empty(4, cin, device=DEV)
    r = {}
    # forward
    r['k8_fwd'] = timed(lambda i: K.conv3x3(xs[i % SETS], w, ys[i % SETS], stride=st))
    ws6 = torch.empty(_native.bn_ws_floats(cin), device=DEV)
    hs = [torch.empty_like(xs[0]) for _ in range(SETS)]
    stt = torch.empty(4, cin, device=DEV)
    r['k6_fwd'] = timed(lambda i: K.bn_relu_forward(xs[i % SETS], hs[i % SETS], gamma, beta, None, None, stt[0], stt[1], ws6, eps=1e-5, momentum=0.0,
                                                    relu=True, save_gate=stt[2:]))
    geo = K.preact_geometry(xs[0].shape, cout, stride=st, bn=True)
    sc = torch.zeros(geo[1], dtype=torch.uint8, device=DEV)
    part = torch.empty(cout, geo[0], 2, dtype=torch.float64, device=DEV)
    bn = (ip, gamma, beta, None, None, save, 1e-5, 0.0)
    r['k10_fwd'] = timed(lambda i: K.preact_conv3x3(xs[i % SETS], w, ys[i % SETS], part, sc, stride=st, bn=bn))
    if st == 1:
        r['k10_fwd_add'] = timed(lambda i: K.preact_conv3x3(xs[i % SETS], w, ys[i % SETS], part, sc, stride=st, bn=bn, add=adds[i % SETS]))
    # backward
    r['k8_dgrad'] = timed(lambda i: K.conv3x3(dys[i % SETS], w, gs[i % SETS], flip=True, stride=st))
    geob = K.preact_geometry(dys[0].shape, cin, flip=True, stride=st)
    scb = None
    pb = torch.empty(cin, geob[0], 2, dtype=torch.float64, device=DEV)
    K.preact_conv3x3(xs[0], w, ys[0], part, sc, stride=st, bn=bn)          # fills save
    r['k10_dgrad'] = timed(lambda i: K.preact_conv3x3(dys[i % SETS], w, gs[i % SETS], pb, scb, stride=st, flip=True, bwd=(xs[i % SETS], save)))
    dxs = [torch.empty_like(xs[0]) for _ in range(SETS)]
    dgb = torch.empty(2, cin, device=DEV)
    r['k6_bwd_2launch'] = timed(lambda i: K.bn_relu_backward(xs[i % SETS], gs[i % SETS], dxs[i % SETS], gamma, beta, save[0], save[1], dgb[0], dgb[1], ws6,
                                                             relu=True, gate=save[2:]))
    r['dx'] = timed(lambda i: K.bn_bwd_dx(xs[i % SETS], gs[i % SETS], dxs[i % SETS], gamma, save, pb, dgb[0], dgb[1]))
    r['dx_res'] = timed(lambda i: K.bn_bwd_dx(xs[i % SETS], gs[i % SETS], dxs[i % SETS], gamma, save, pb, dgb[0], dgb[1], dz=hs[i % SETS]))
    wsf = K.conv_wgrad_ws_floats(xs[0].shape, cout, 3, st)
    wss = [torch.empty(wsf, device=DEV) for _ in range(SETS)]
    r['k7_partial'] = timed(lambda i: K.conv_wgrad_partial(xs[i % SETS], dys[i % SETS], w.shape, wss[i % SETS], st))
    r['k7_partial_xbn'] = timed(lambda i: K.preact_wgrad_partial(xs[i % SETS], save, dys[i % SETS], w.shape, wss[i % SETS], st))
    r['bwd_pair'] = timed(lambda i: K.preact_bwd_pair(dys[i % SETS], w, gs[i % SETS], xs[i % SETS], save, pb, wss[i % SETS], st))
    out[f'{cin}x{cout}x{hw}s{st}'] = r
    print(f'{cin}x{cout}x{hw}s{st}', json.dumps(r), flush=True)
# K7's second launch for the 21 convolutions of a PreResNet-20 step at once (what the engine's flush issues)
LAYERS = [(3, 16, 32, 3, 1)] + [(16, 16, 32, 3, 1)] * 6 + [(16, 32, 32, 3, 2), (16, 32, 32, 1, 2)] + [(32, 32, 16, 3, 1)] * 5 + \
         [(32, 64, 16, 3, 2), (32, 64, 16, 1, 2)] + [(64, 64, 8, 3, 1)] * 5
pend, tot = [], 0
for cin, cout, hw, ks, st in LAYERS:
    x, dy = torch.randn(N, cin, hw, hw, device=DEV), torch.randn(N, cout, hw // st, hw // st, device=DEV)
    ws = torch.empty(K.conv_wgrad_ws_floats(x.shape, cout, ks, st), device=DEV)
    tot += ws.numel() * 4
    pend.append((K.conv_wgrad_partial(x, dy, (cout, cin, ks, ks), ws, st), torch.empty(cout, cin, ks, ks, device=DEV)))
r = {'reduce_21_layers': timed(lambda i: K.conv_wgrad_reduce(pend)), 'partial_sum_bytes': tot}
out['reduce_step'] = r
print('reduce_step', json.dumps(r), flush=True)
if len(sys.argv) > 1:
    json.dump(dict(what='us per launch inside hipGraph replays, batch 128 (tools/k10_bench.py)', dbg=os.environ.get('URSA_K10_DBG'), units=out),
              open(sys.argv[1], 'w'), indent=1)
