"""In-network ReLU gates after every BatchNorm: GPU (fused / stock launches) vs the CPU run from the same weights."""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ursabench_amd import fused_bn, models, tuning  # noqa: E402
import ursabench_amd.models as M  # noqa: E402

tuning.use_shipped_miopen_db()


def main():
    torch.manual_seed(4242)
    m0 = models.PreResNet(10, 20)
    g = torch.Generator().manual_seed(4243)
    x = torch.randn(128, 3, 32, 32, generator=g)
    runs = {}
    orig, orig_add = M.bn_relu, M.add_bn_relu
    for name in ('cpu', 'fused', 'stock'):
        rec = []

        def spy(bn, xx, relu=True, _rec=rec):
            y = orig(bn, xx, relu)
            _rec.append((xx.detach().cpu().clone(), y.detach().cpu().clone(), bn.weight.detach().cpu().clone(),
                         bn.bias.detach().cpu().clone()))
            return y

        def spy_add(bn, xx, relu=True, _rec=rec):
            z, y = orig_add(bn, xx, relu)
            _rec.append((z.detach().cpu().clone(), y.detach().cpu().clone(), bn.weight.detach().cpu().clone(),
                         bn.bias.detach().cpu().clone()))
            return z, y
        M.bn_relu, M.add_bn_relu = spy, spy_add
        mm = models.PreResNet(10, 20)
        mm.load_state_dict(m0.state_dict())
        dev = 'cpu' if name == 'cpu' else 'cuda'
        mm = mm.to(dev).train()
        fused_bn.enabled(name == 'fused')
        with torch.no_grad():
            mm(x.to(dev))
        runs[name] = rec
    M.bn_relu, M.add_bn_relu = orig, orig_add
    fused_bn.enabled(True)
    for li, (xc, yc, w, b) in enumerate(runs['cpu']):
        line = f'layer {li:2d} {tuple(xc.shape)}'
        for name in ('fused', 'stock'):
            xg, yg, _, _ = runs[name][li]
            flips = (yg > 0) != (yc > 0)
            line += (f' | {name}: dx_max {float((xg - xc).abs().max()):.2e} x_differing {int((xg != xc).sum()):8d} '
                     f'y_differing {int((yg != yc).sum()):8d} gate_flips {int(flips.sum())}')
            if flips.any():
                idx = flips.nonzero()[0]
                i = tuple(int(v) for v in idx)
                line += f' [first flip at {i}: y_gpu {float(yg[i]):.3e} y_cpu {float(yc[i]):.3e} x_gpu-x_cpu {float(xg[i] - xc[i]):.3e}]'
        print(line)


if __name__ == '__main__':
    main()
