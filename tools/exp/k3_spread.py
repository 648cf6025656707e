"""Why does K3 (SWAG draw) read 0.56-0.74 of peak at the WideResNet-28-10 arena (36,546,980 elements) when the same
kernel reads 0.78 at 2^26? (VERDICT r2 #2.) One process, HIP events, median of 7 batches of 10 launches per point:
  * sizes: 2^24 ... 2^27 and the arena size, Philox noise vs noise from memory (eps pointer: 16 B/elem, no generator
    arithmetic) — separates "bytes" from "generator VALU work";
  * cold (first thing the process does) vs hot (right after 20 s of back-to-back fp32 GEMMs), with `rocm-smi -c`
    sclk/mclk and power snapshots taken while the kernels run;
  * output into a fresh buffer vs into a member-bank row.
Writes gpurun_out/k3_spread.json."""
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import _native  # noqa: E402

K = _native.default_kernels()


def smi():
    """Current clocks / power as rocm-smi prints them (an ordinary user may read, not set)."""
    out = {}
    try:
        txt = subprocess.run(['rocm-smi', '-c', '-P', '--json'], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(txt)
        card = d[sorted(d)[0]]
        for k, v in card.items():
            if any(s in k.lower() for s in ('sclk', 'mclk', 'fclk', 'socclk', 'power')):
                out[k] = v
    except Exception as e:       # noqa: BLE001
        out['error'] = repr(e)
    return out


def timed(fn, launches=10, batches=7):
    ts = []
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    for _ in range(batches):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(launches):
            fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / launches)
    ts.sort()
    return ts[len(ts) // 2], ts[0], ts[-1]


def smi_while(fn, seconds=1.5):
    """Run fn() back to back for `seconds` while a thread takes an rocm-smi snapshot in the middle."""
    snap = {}
    th = threading.Thread(target=lambda: snap.update(smi()))
    t0 = time.time()
    started = False
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        if not started and time.time() - t0 > seconds / 3:
            th.start()
            started = True
    torch.cuda.synchronize()
    if started:
        th.join()
    return snap


def point(n, state):
    mean, sq, eps, out = (torch.randn(n, device='cuda') for _ in range(4))
    sq.abs_().add_(mean * mean)
    res = {'elements': n, 'state': state}
    sd = torch.empty_like(mean)
    K.swag_std(sd, mean, sq, var_clamp=1e-30)
    for name, bpe, fn in (('philox', 12, lambda: K.swag_draw(out, mean, sq, var_clamp=1e-30, scale=1.0, seed=3, draw=1)),
                          ('philox_std_hoisted', 12, lambda: K.swag_draw_std(out, mean, sd, seed=3, draw=1)),
                          ('std_once', 12, lambda: K.swag_std(sd, mean, sq, var_clamp=1e-30)),
                          ('eps_ptr', 16, lambda: K.swag_draw(out, mean, sq, var_clamp=1e-30, scale=1.0, eps=eps)),
                          ('k2_collect', 20, lambda: K.swag_collect(mean, sq, eps, decay=0.75, denom=4.0)),
                          ('torch_copy', 8, lambda: out.copy_(mean))):
        med, best, worst = timed(fn)
        res[name] = {'us_median': round(med, 2), 'us_best': round(best, 2), 'us_worst': round(worst, 2),
                     'TBps_median': round(bpe * n / med / 1e6, 3), 'frac_of_8TBps': round(bpe * n / med / 1e6 / 8, 4),
                     'Gelem_per_s': round(n / med / 1e3, 1)}
    res['smi_during_philox'] = smi_while(lambda: K.swag_draw(out, mean, sq, var_clamp=1e-30, scale=1.0, seed=3, draw=1))
    res['smi_during_philox_std_hoisted'] = smi_while(lambda: K.swag_draw_std(out, mean, sd, seed=3, draw=1))
    return res


def main():
    arena = 36546980 + (-36546980) % 64
    results = {'smi_idle': smi(), 'points': []}
    for n in (arena, 1 << 26):
        results['points'].append(point(n, 'cold'))
        print(json.dumps(results['points'][-1]), flush=True)
    # heat: 20 s of fp32 GEMMs
    a, b = torch.randn(8192, 8192, device='cuda'), torch.randn(8192, 8192, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 20:
        for _ in range(20):
            a @ b
        torch.cuda.synchronize()
    results['smi_after_heat'] = smi()
    for n in (arena, 1 << 26, 1 << 24):
        results['points'].append(point(n, 'hot'))
        print(json.dumps(results['points'][-1]), flush=True)
    # output into a member-bank row layout ([theta_pad | fbuf_pad]) vs a fresh buffer: same kernel, different write target
    row = torch.empty(arena + 4096, device='cuda')
    mean, sq = torch.randn(arena, device='cuda'), torch.rand(arena, device='cuda') + 2
    med, best, worst = timed(lambda: K.swag_draw(row[:arena], mean, sq, var_clamp=1e-30, scale=1.0, seed=3, draw=1))
    results['bank_row_target'] = {'us_median': round(med, 2), 'us_best': round(best, 2), 'frac_of_8TBps': round(12 * arena / med / 1e6 / 8, 4)}
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(results, open('gpurun_out/k3_spread.json', 'w'), indent=1)
    print(json.dumps({k: v for k, v in results.items() if k != 'points'}))


if __name__ == '__main__':
    main()
