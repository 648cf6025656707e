"""K5 (ursa_bma_accumulate_f32) micro-benchmark: graph-batched HIP-event timing at the shapes the tasks feed
it, with the debug switches for A/B (URSA_BMA_NO_ROWLANE / URSA_BMA_NO_V4 select the generic lane-group kernel).
    python tools/k5_bench.py [S B C ...]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ursabench_amd import _native
from tools.kbench import timeit
K = _native.knobs_kernels()        # the -DURSA_DEBUG_KNOBS build: the shipped library reads no environment
args = [int(a) for a in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(50, 10000, 10), (20, 10000, 10), (3, 10000, 10), (3, 128, 10),
                                                                  (30, 10000, 100), (30, 128, 100), (30, 10000, 16), (30, 10000, 64)]
res = []
VARIANTS = [('fast', {}), ('lane_group_scalar', {'URSA_BMA_NO_ROWLANE': '1', 'URSA_BMA_NO_V4': '1'})]
VARIANTS += [(f'rowlane_waves{w}', {'URSA_BMA_RL_WAVES': str(w)}) for w in (1, 2, 4, 8)]
VARIANTS += [('prefetch', {'URSA_BMA_PREFETCH': '1'}), ('no_prefetch', {'URSA_BMA_PREFETCH': '0'})]
if os.environ.get('K5_ONLY'):
    VARIANTS = [v for v in VARIANTS if v[0] in os.environ['K5_ONLY'].split(',')]
for variant, env in VARIANTS:
    for k in ('URSA_BMA_NO_ROWLANE', 'URSA_BMA_NO_V4', 'URSA_BMA_RL_WAVES', 'URSA_BMA_NO_G4', 'URSA_BMA_PREFETCH'):
        os.environ.pop(k, None)
    os.environ.update(env)
    for (S, B, C) in shapes:
        z = torch.randn(S, B, C, device='cuda') * 3
        p, e = torch.zeros(B, C, device='cuda'), torch.zeros(B, device='cuda')
        fn = lambda: K.bma_accumulate(z, p, e, one_minus_gamma=0.9999, gamma_over_c=1e-4 / C, smoothed=False)
        med, best = timeit(fn, 20)
        byt = 4 * S * B * C + 8 * B * (C + 1)
        r = dict(variant=variant, S=S, B=B, C=C, median_us=round(med * 1e6, 2), best_us=round(best * 1e6, 2),
                 GBps_median=round(byt / med / 1e9, 1), frac_of_8TBps=round(byt / med / 8e12, 4))
        print(json.dumps(r), flush=True)
        res.append(r)
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/k5_bench.json', 'w'), indent=1)
