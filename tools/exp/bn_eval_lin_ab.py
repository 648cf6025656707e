"""K6 evaluation launch: the linear form (round 5: contiguous 16 KB span per workgroup, channel looked up per float4) against the
channel-grid form (rounds 3-4: grid = (splits, channels), channel c walked as N runs of H*W floats), plain and residual (addend)
forms, us per call (HIP events; hipGraph-batched for cache-resident sizes) and fraction of 8 TB/s on the bytes moved (8 B/element,
residual form 16). Same floats (asserted).
    python tools/exp/bn_eval_lin_ab.py  -> one JSON line"""
import json
import os
import sys
os.environ['URSA_BN_EVAL_GRID'] = '1'                      # read by the KNOBS build only: its evaluation launch keeps the channel grid
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import event_time_ms, HBM_PEAK_GBPS
from ursabench_amd import _native
K_lin, K_grid = _native.default_kernels(), _native.knobs_kernels()
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream()
out = []
for shape in ((128, 16, 32, 32), (128, 32, 16, 16), (128, 64, 8, 8), (4096, 16, 32, 32), (4096, 32, 16, 16), (4096, 64, 8, 8), (1808, 16, 32, 32),
              (128, 160, 32, 32), (128, 320, 16, 16), (128, 640, 8, 8), (1024, 64, 32, 32), (1024, 256, 8, 8), (30, 50, 12, 12)):
    C = shape[1]
    x, a = torch.randn(shape, device=dev), torch.randn(shape, device=dev)
    w, b, rm, rv = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev), torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
    e = x.numel()
    resident = e * 4 * 2 < (128 << 20)
    row = dict(shape=list(shape), mbytes=round(e * 4 / 1e6, 1))
    for form, kw, bpe in (('plain', {}, 8), ('residual', dict(addend=a), 16)):
        res = {}
        for tag, K in (('linear', K_lin), ('channel_grid', K_grid)):
            y, z = torch.empty_like(x), torch.empty_like(x)
            kk = dict(kw, z_out=z) if kw else {}
            fn = lambda: K.bn_relu_eval(x, y, w, b, rm, rv, eps=1e-5, **kk)
            fn()
            res[tag] = (y.clone(), z.clone() if kw else None)
            bt = sorted(event_time_ms(fn, 1024, stream, graph_batch=128) if resident else event_time_ms(fn, 10, stream) for _ in range(5))
            row[f'{form}_{tag}_us'] = round(bt[2] * 1e3, 2)
            row[f'{form}_{tag}_frac'] = round(bpe * e / (bt[2] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 3)
        assert torch.equal(res['linear'][0], res['channel_grid'][0]), (shape, form)
        if kw:
            assert torch.equal(res['linear'][1], res['channel_grid'][1]), (shape, form)
    out.append(row)
    del x, a
print(json.dumps(dict(what='K6 evaluation launch, linear vs channel-grid form; same floats asserted', rows=out)))
