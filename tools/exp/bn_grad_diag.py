"""Per-tensor gradient differences fused / stock / CPU on PreResNet-20 (B=128, same weights), and for every fused BN
layer the backward's own outputs against a float64 recomputation from the x / dy it was given."""
import json
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ursabench_amd import _native, fused_bn, models, tuning  # noqa: E402

tuning.use_shipped_miopen_db()


def main():
    torch.manual_seed(4242)
    m0 = models.PreResNet(10, 20)
    g = torch.Generator().manual_seed(4243)
    x, t = torch.randn(128, 3, 32, 32, generator=g), torch.randint(0, 10, (128,), generator=g)
    K = _native.default_kernels()
    rec = []
    orig = K.bn_relu_backward

    def spy(x_, dy, dx, gamma, beta, sm, si, dg, db, ws, relu=True, dz=None, **kw):
        orig(x_, dy, dx, gamma, beta, sm, si, dg, db, ws, relu=relu, dz=dz, **kw)
        plain = dx if dz is None else dx - dz           # the BatchNorm's own dx (the residual form adds dz)
        rec.append([v.detach().clone() for v in (x_, dy, plain, gamma, beta, sm, si, dg, db)])
    K.bn_relu_backward = spy
    grads = {}
    for name in ('fused', 'stock', 'cpu'):
        mm = models.PreResNet(10, 20)
        mm.load_state_dict(m0.state_dict())
        dev = 'cpu' if name == 'cpu' else 'cuda'
        mm = mm.to(dev).train()
        fused_bn.enabled(name == 'fused')
        F.cross_entropy(mm(x.to(dev)), t.to(dev)).backward()
        grads[name] = {k: p.grad.detach().cpu() for k, p in mm.named_parameters()}
    fused_bn.enabled(True)
    K.bn_relu_backward = orig
    rows = []
    for k in grads['cpu']:
        c = grads['cpu'][k]
        sc = max(float(c.abs().max()), 1e-30)
        rows.append((float((grads['fused'][k] - c).abs().max()) / sc, float((grads['stock'][k] - c).abs().max()) / sc, k))
    rows.sort(reverse=True)
    print('worst tensors (fused vs cpu, stock vs cpu, relative to the tensor max):')
    for r in rows[:12]:
        print(f'  {r[0]:.3e} {r[1]:.3e} {r[2]}')
    print('fused BN layers, backward outputs vs float64 from the same x / dy (in backward order):')
    for x_, dy, dx, gamma, beta, sm, si, dg, db in rec:
        xd, dyd = x_.double().cpu().requires_grad_(True), dy.double().cpu()
        w, b = gamma.double().cpu().requires_grad_(True), beta.double().cpu().requires_grad_(True)
        pre = F.batch_norm(xd, None, None, w, b, True, 0.0, 1e-5)
        alpha = si * gamma
        shift = torch.addcmul(beta, -sm, alpha)          # not bit-identical to the kernel's fma, close enough to count
        gate_gpu = (torch.addcmul(shift.view(1, -1, 1, 1), x_, alpha.view(1, -1, 1, 1)) > 0).cpu()
        gate_64 = pre.detach() > 0
        (pre * gate_64).backward(dyd)
        rows = dict(shape=list(x_.shape), gates_differing_from_float64=int((gate_gpu != gate_64).sum()),
                    dbeta=float((db.cpu() - b.grad.float()).abs().max() / b.grad.abs().max()),
                    dgamma=float((dg.cpu() - w.grad.float()).abs().max() / w.grad.abs().max()),
                    dx=float((dx.cpu() - xd.grad.float()).abs().max() / xd.grad.abs().max()))
        print(' ', json.dumps(rows))


if __name__ == '__main__':
    main()
