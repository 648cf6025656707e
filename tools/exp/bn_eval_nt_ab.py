"""K6 evaluation launch (k_bn_eval): nontemporal (streaming) accesses beyond the Infinity Cache vs temporal ones
(URSA_BN_EVAL_NO_NT=1 in the knobs build), plain and residual forms; us per call inside a hipGraph of 20 calls.
    python tools/exp/bn_eval_nt_ab.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:                       # the knob is read once per process: one child per setting
    out = []
    for no_nt in ('0', '1'):
        env = dict(os.environ, URSA_BN_EVAL_NO_NT=no_nt)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=env, capture_output=True, text=True, check=True)
        out += [dict(json.loads(l), nontemporal=no_nt == '0') for l in r.stdout.splitlines() if l.startswith('{')]
    for o in out:
        print(json.dumps(o))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(out, open('gpurun_out/bn_eval_nt_ab.json', 'w'), indent=1)
    sys.exit(0)
import torch
from ursabench_amd import _native
K = _native.knobs_kernels()
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


for shape in [(4096, 16, 32, 32), (1024, 64, 32, 32), (4096, 32, 16, 16), (4096, 64, 8, 8), (1024, 16, 32, 32), (128, 160, 32, 32), (512, 16, 32, 32)]:
    C = shape[1]
    x, ad = torch.randn(shape, device='cuda'), torch.randn(shape, device='cuda')
    y, z = torch.empty_like(x), torch.empty_like(x)
    w, b, rm, rv = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda'), torch.randn(C, device='cuda'), torch.rand(C, device='cuda') + 0.5
    e = x.numel()
    t = timed(lambda: K.bn_relu_eval(x, y, w, b, rm, rv, eps=1e-5))
    tr = timed(lambda: K.bn_relu_eval(x, y, w, b, rm, rv, eps=1e-5, addend=ad, z_out=z))
    print(json.dumps(dict(shape=list(shape), mb=round(e * 4 / 1e6), eval_us=round(t, 1), frac=round(8 * e / t / 8e6, 3),
                          eval_residual_us=round(tr, 1), frac_residual=round(16 * e / tr / 8e6, 3))), flush=True)
    del x, ad, y, z
