"""K6 held forms: compile-time variants of the workgroup shape (tools/exp/bn_held_fwd_ab.sh -> tools/exp/_libs/libursa_hip_heldfwd_*.so;
F = forward, B = backward: threads per workgroup, e = float4 per thread in registers, l = in LDS) against the shipped library
(forward 512 / e32 / l9, backward 512 / e8 / l4) and the knobs build with the held form taken from URSA_BN_HELD_MIN_MIB on: us per call inside a hipGraph of 20 calls, and the same floats as the two-launch form.
    bash tools/exp/bn_held_fwd_ab.sh && python tools/exp/bn_held_fwd_ab.py"""
import glob, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ursabench_amd import _native

SHAPES = [(1024, 64, 32, 32), (1024, 128, 16, 16), (128, 160, 32, 32), (1024, 16, 32, 32), (1024, 256, 8, 8), (256, 64, 32, 32), (512, 64, 32, 32),
          (512, 16, 32, 32), (1024, 32, 16, 16), (384, 16, 32, 32), (128, 96, 32, 32), (256, 16, 32, 32), (128, 64, 32, 32)]
os.environ.setdefault('URSA_BN_HELD_MIN_MIB', '8')      # (knobs builds only: the forward takes the held form from 24 MiB too, to see where it pays)
REPS = 20


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
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        out.append(a.elapsed_time(b) * 1e3 / REPS)
    return sorted(out)[3]


libs = [('shipped', _native.LIB_PATH), ('knobs_held_from_%s_MiB' % os.environ.get('URSA_BN_HELD_MIN_MIB'), _native.KNOBS_LIB_PATH)] + [(os.path.basename(p)[len('libursa_hip_heldfwd_'):-3], p)
                                          for p in sorted(glob.glob(os.path.join(ROOT, 'tools/exp/_libs/libursa_hip_heldfwd_*.so')))]
res = []
for tag, path in libs:
    K = _native.HipKernels(_native.load_library(path))
    for shape in SHAPES:
        C = shape[1]
        x, dy = torch.randn(shape, device='cuda'), torch.randn(shape, device='cuda')
        y, y2, dx, dx2 = (torch.empty_like(x) for _ in range(4))
        w, b = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
        sm, si, dg, db = (torch.zeros(C, device='cuda') for _ in range(4))
        ws = torch.zeros(_native.bn_ws_floats(C), device='cuda')
        f = timed(lambda: K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, held=True))
        bw = timed(lambda: K.bn_relu_backward(x, dy, dx, w, b, sm, si, dg, db, ws, held=True))
        ad, dz, z = torch.randn(shape, device='cuda'), torch.randn(shape, device='cuda'), torch.empty(shape, device='cuda')
        fr = timed(lambda: K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, addend=ad, z_out=z, held=True))
        br = timed(lambda: K.bn_relu_backward(z, dy, dx, w, b, sm, si, dg, db, ws, dz=dz, held=True))
        f2 = timed(lambda: K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, two_launch=True))
        fr2 = timed(lambda: K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, addend=ad, z_out=z, two_launch=True))
        del ad, dz, z
        K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, held=True)
        K.bn_relu_backward(x, dy, dx, w, b, sm, si, dg, db, ws, held=True)
        clean = bool((ws[C * 256:] == 0).all())
        K.bn_relu_forward(x, y2, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, two_launch=True)
        K.bn_relu_backward(x, dy, dx2, w, b, sm, si, dg, db, ws, two_launch=True)
        e = x.numel()
        r = dict(lib=tag, shape=list(shape), mb=round(x.numel() * 4 / 1e6), fwd_us=round(f, 1), fwd_two_launch_us=round(f2, 1), fwd_residual_us=round(fr, 1), fwd_residual_two_launch_us=round(fr2, 1), bwd_us=round(bw, 1), bwd_residual_us=round(br, 1), fwd_frac=round(8 * e / (f * 1e-6) / 8e12, 3),
                 bwd_frac=round(12 * e / (bw * 1e-6) / 8e12, 3), same_floats=bool(torch.equal(y, y2) and torch.equal(dx, dx2)), ws_clean=clean)
        print(json.dumps(r), flush=True)
        res.append(r)
        del x, dy, y, y2, dx, dx2
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/bn_held_fwd_ab.json', 'w'), indent=1)
