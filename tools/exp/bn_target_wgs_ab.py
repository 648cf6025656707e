"""K6 two-launch form at the workload's own layers: workgroups per launch (URSA_BN_TARGET_WGS in the knobs build; the
library's constant is 1024). us per call inside a hipGraph of 50 calls.   python tools/exp/bn_target_wgs_ab.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    out = []
    for t in ('256', '512', '768', '1024', '2048'):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=dict(os.environ, URSA_BN_TARGET_WGS=t),
                           capture_output=True, text=True, check=True)
        out += [dict(json.loads(l), target_wgs=int(t)) for l in r.stdout.splitlines() if l.startswith('{')]
    for o in out:
        print(json.dumps(o))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/bn_target_wgs_ab.json', 'w'), indent=1)
    sys.exit(0)
import torch
from ursabench_amd import _native
K = _native.knobs_kernels()
REPS = 50


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
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / REPS)
    return sorted(out)[4]


for shape in [(128, 16, 32, 32), (128, 32, 16, 16), (128, 64, 8, 8), (128, 160, 32, 32), (512, 16, 32, 32)]:
    C = shape[1]
    x, dy, ad, dz = (torch.randn(shape, device='cuda') for _ in range(4))
    y, dx, z = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    w, b = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    sm, si, dg, db = (torch.zeros(C, device='cuda') for _ in range(4))
    ws = torch.zeros(_native.bn_ws_floats(C), device='cuda')
    f = timed(lambda: K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, two_launch=True))
    bw = timed(lambda: K.bn_relu_backward(x, dy, dx, w, b, sm, si, dg, db, ws, two_launch=True))
    fr = timed(lambda: K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, addend=ad, z_out=z, two_launch=True))
    br = timed(lambda: K.bn_relu_backward(z, dy, dx, w, b, sm, si, dg, db, ws, dz=dz, two_launch=True))
    print(json.dumps(dict(shape=list(shape), fwd_us=round(f, 2), bwd_us=round(bw, 2), fwd_residual_us=round(fr, 2), bwd_residual_us=round(br, 2))), flush=True)
