"""K5 lane-group kernel: members in flight per lane (U) A/B — the shipped library against builds with
-DURSA_BMA_U_EPL8=3/4 (C = 100: 8 classes per lane) and -DURSA_BMA_U_EPL16=2 (C = 256)."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import torch
    from ursabench_amd import _native
    _native.LIB_PATH = sys.argv[1]
    from tools.kbench import timeit
    K = _native.default_kernels()
    for (S, B, C) in ((30, 10000, 100), (30, 10000, 128), (30, 10000, 256), (8, 10000, 100)):
        z = torch.randn(S, B, C, device='cuda') * 3
        p, e = torch.zeros(B, C, device='cuda'), torch.zeros(B, device='cuda')
        med, best = timeit(lambda: K.bma_accumulate(z, p, e, one_minus_gamma=0.9999, gamma_over_c=1e-4 / C, smoothed=False), 20)
        byt = 4 * S * B * C + 8 * B * (C + 1)
        print(json.dumps(dict(lib=os.path.basename(sys.argv[1]), S=S, B=B, C=C, median_us=round(med * 1e6, 2), frac=round(byt / med / 8e12, 4))), flush=True)
else:
    libs = [os.path.join(ROOT, 'ursabench_amd', 'csrc', 'libursa_hip.so')] + sorted(glob.glob(os.path.join(ROOT, 'tools', 'exp', 'build', 'libursa_u*.so')))
    for lib in libs:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), lib], capture_output=True, text=True)
        print(p.stdout, end='')
        sys.stderr.write(p.stderr[-500:])
