"""Kernel micro-benchmark on the GPU box: achieved algorithmic GB/s of each kernel at a
roofline-sized working set (> 256 MiB Infinity Cache) and at the PreResNet-20 size.
    python tools/kbench.py [--n 67108864] [--iters 20]
"""
import argparse
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ursabench_amd import _native  # noqa: E402


def timeit(fn, iters, warmup=3, batch=16):
    """Median / best duration of one launch. `batch` launches are captured into a hipGraph and the
    replay is timed with HIP events, so the host's ~10 us of Python/ctypes per call is not in the
    figure (it includes the ~1.5 us kernel-to-kernel boundary instead)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(batch):
            fn()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3 / batch)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=int, default=1 << 26)
    ap.add_argument('--iters', type=int, default=20)
    a = ap.parse_args()
    K = _native.default_kernels()
    res = []
    for n in (a.n, 272282):
        th, g, m, snap = (torch.randn(n, device='cuda') for _ in range(4))
        eps = torch.randn(n, device='cuda')
        ws, acc = torch.zeros(2048, device='cuda'), torch.zeros(1, device='cuda')
        sc = dict(lr=1e-3, c_wd=8e-5, c_noise=0.03, n_train=50000.0, seed=1)
        cases = {
            'k1_sghmc_philox': (20, lambda: K.sgmcmc_step(th, g, m, mu=0.5, flags=0x1 | 0x8, step=3, **sc)),
            'k1_sghmc_philox_zero_grad': (24, lambda: K.sgmcmc_step(th, g, m, mu=0.5, flags=0x1 | 0x8 | 0x4, step=3, **sc)),
            'k1_sghmc_philox_zg_snapshot': (28, lambda: K.sgmcmc_step(th, g, m, mu=0.5, flags=0x1 | 0x8 | 0x4, step=3, snapshot=snap, **sc)),
            'k1_sghmc_nonoise': (20, lambda: K.sgmcmc_step(th, g, m, mu=0.5, flags=0x8, step=3, **sc)),
            'k1_sghmc_eps_ptr': (24, lambda: K.sgmcmc_step(th, g, m, mu=0.5, flags=0x1 | 0x8, eps=eps, step=3, **sc)),
            'k1_sgld_philox': (12, lambda: K.sgmcmc_step(th, g, None, mu=0.0, flags=0x1 | 0x8, step=3, **sc)),
            'k2_swag_collect': (20, lambda: K.swag_collect(th, m, g, decay=0.75, denom=4.0)),
            'k3_swag_draw_philox': (12, lambda: K.swag_draw(snap, th, m, var_clamp=1e-30, seed=1, draw=2)),
            'k4_leapfrog_kick_drift': (20, lambda: K.leapfrog(th, m, g, kick_coef=1e-4, step_size=2e-4, inv_mass=1.0, flags=0x3)),
            'k4_leapfrog_kick_drift_kinetic': (20, lambda: K.leapfrog(th, m, g, kick_coef=1e-4, step_size=2e-4, inv_mass=1.0,
                                                                   flags=0x3, kinetic_out=acc, ws=ws)),
            'k4_sumsq': (4, lambda: K.sumsq(th, acc, ws)),
            'philox_normal_fill': (4, lambda: K.philox_normal(snap, seed=1, step=2)),
            'torch_copy_(ref)': (8, lambda: snap.copy_(th)),
        }
        for name, (bpe, fn) in cases.items():
            med, best = timeit(fn, a.iters)
            res.append(dict(kernel=name, n=n, bytes_per_elem=bpe, median_us=med * 1e6, best_us=best * 1e6,
                            GBps_median=bpe * n / med / 1e9, GBps_best=bpe * n / best / 1e9))
            print(json.dumps(res[-1]))
    for (S, B, C) in ((30, 10000, 100), (50, 10000, 10)):
        z = torch.randn(S, B, C, device='cuda')
        p, e = torch.zeros(B, C, device='cuda'), torch.zeros(B, device='cuda')
        fn = lambda: K.bma_accumulate(z, p, e, one_minus_gamma=0.9999, gamma_over_c=1e-4 / C, smoothed=False)
        med, best = timeit(fn, a.iters)
        byt = 4 * S * B * C + 8 * B * (C + 1)
        res.append(dict(kernel='k5_bma', S=S, B=B, C=C, median_us=med * 1e6, best_us=best * 1e6,
                        GBps_median=byt / med / 1e9, GBps_best=byt / best / 1e9))
        print(json.dumps(res[-1]))
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(res, open('gpurun_out/kbench.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
