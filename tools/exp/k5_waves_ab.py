"""K5 at C > 64 (k_bma_accumulate<16, 8, V4>): waves per block (= member ranges a row group is split into) x software
pipelining x members in flight per lane (U, a compile-time constant: one knobs library per value under tools/exp/_libs/,
built by tools/exp/k5_waves_ab.sh). HIP-event timing of graph-batched launches, as tools/k5_bench.py.
    python tools/exp/k5_waves_ab.py [S B C ...]"""
import glob, json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ursabench_amd import _native
from tools.kbench import timeit

args = [int(a) for a in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(30, 10000, 100), (30, 10000, 128), (30, 10000, 64), (50, 10000, 100), (8, 10000, 100), (30, 10000, 256), (30, 10000, 32), (30, 128, 100)]
libs = [('U2', _native.KNOBS_LIB_PATH)] + [(os.path.basename(p)[len('libursa_hip_knobs_'):-3], p)
                                             for p in sorted(glob.glob(os.path.join(ROOT, 'tools/exp/_libs/libursa_hip_knobs_*.so')))]
res = []
# the practical ceiling of a READ-ONLY launch over the same bytes: sum of squares (K4's reduction) of the logits slab
K0 = _native.HipKernels(_native.load_library(_native.KNOBS_LIB_PATH))
for (S, B, C) in shapes:
    z = torch.randn(S * B * C, device='cuda')
    out, ws = torch.zeros(1, device='cuda'), torch.zeros(_native.REDUCE_WS_FLOATS, device='cuda')
    med, best = timeit(lambda: K0.sumsq(z, out, ws), 20)
    r = dict(lib='read_only_sumsq', S=S, B=B, C=C, median_us=round(med * 1e6, 2), best_us=round(best * 1e6, 2),
             frac_of_8TBps=round(4 * S * B * C / med / 8e12, 4))
    print(json.dumps(r), flush=True)
    res.append(r)
for tag, path in libs:
    K = _native.HipKernels(_native.load_library(path))
    if tag == 'U2':                       # what the library picks by itself (bma_form)
        for k in ('URSA_BMA_WAVES', 'URSA_BMA_PREFETCH', 'URSA_BMA_EARLY'):
            os.environ.pop(k, None)
        for (S, B, C) in shapes:
            z = torch.randn(S, B, C, device='cuda') * 3
            p, e = torch.zeros(B, C, device='cuda'), torch.zeros(B, device='cuda')
            med, best = timeit(lambda: K.bma_accumulate(z, p, e, one_minus_gamma=0.9999, gamma_over_c=1e-4 / C, smoothed=False), 20)
            r = dict(lib='as_shipped', S=S, B=B, C=C, median_us=round(med * 1e6, 2), best_us=round(best * 1e6, 2),
                     frac_of_8TBps=round((4 * S * B * C + 8 * B * (C + 1)) / med / 8e12, 4))
            print(json.dumps(r), flush=True)
            res.append(r)
    for W in (1, 2, 4, 8):
        for pf, early in ((0, 0), (1, 0), (0, 1), (1, 1)):
            os.environ['URSA_BMA_WAVES'], os.environ['URSA_BMA_PREFETCH'], os.environ['URSA_BMA_EARLY'] = str(W), str(pf), str(early)
            for (S, B, C) in shapes:
                z = torch.randn(S, B, C, device='cuda') * 3
                p, e = torch.zeros(B, C, device='cuda'), torch.zeros(B, device='cuda')
                fn = lambda: K.bma_accumulate(z, p, e, one_minus_gamma=0.9999, gamma_over_c=1e-4 / C, smoothed=False)
                med, best = timeit(fn, 20)
                byt = 4 * S * B * C + 8 * B * (C + 1)
                r = dict(lib=tag, waves=W, prefetch=pf, early=early, S=S, B=B, C=C, median_us=round(med * 1e6, 2), best_us=round(best * 1e6, 2),
                         frac_of_8TBps=round(byt / med / 8e12, 4))
                print(json.dumps(r), flush=True)
                res.append(r)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/k5_waves_ab.json', 'w'), indent=1)
