"""How close are relu(bn(x)) implementations to torch's CPU BatchNorm — the reference's arithmetic — where it matters:
the ReLU gate. PreResNet-20, 128 rows, train mode: every BatchNorm layer's input is captured from one stock forward on
the GPU; each implementation then normalises the SAME x, and its output is compared with the CPU kernel's: elements
whose gate differs (open on one side, closed on the other), elements that differ at all, channels whose mean / invstd
differ. One flipped gate moves that element's gradient by O(dy): ~1e-3 relative on the weights it touches.
    python3 tools/exp/bn_mask_flips.py > gpurun_out/bn_mask_flips.json
"""
import json
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import _native, fused_bn, models, tuning  # noqa: E402

tuning.use_shipped_miopen_db()


def main():
    torch.manual_seed(0)
    m = models.PreResNet(num_classes=10, depth=20).cuda().train()
    x = torch.randn(128, 3, 32, 32, device='cuda')
    caught = []
    hooks = [mod.register_forward_pre_hook(lambda mod, inp: caught.append((mod, inp[0].detach().clone())))
             for mod in m.modules() if isinstance(mod, nn.BatchNorm2d)]
    fused_bn.enabled(False)
    m(x)
    fused_bn.enabled(True)
    for h in hooks:
        h.remove()
    K = _native.default_kernels()
    rows, tot = [], dict(elements=0, fused_gate=0, stock_gate=0, fused_any=0, stock_any=0)
    for mod, xi in caught:
        C = xi.shape[1]
        w, b = mod.weight.detach(), mod.bias.detach()
        y_f = torch.empty_like(xi)
        sm, si = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
        ws = torch.empty(_native.bn_ws_floats(C), device='cuda')
        K.bn_relu_forward(xi, y_f, w, b, None, None, sm, si, ws, eps=mod.eps, momentum=0.0, relu=True)
        y_s = F.relu(F.batch_norm(xi, None, None, w, b, True, 0.0, mod.eps))
        out_c, m_c, i_c = torch.native_batch_norm(xi.cpu(), w.cpu(), b.cpu(), None, None, True, 0.0, mod.eps)
        y_c = F.relu(out_c)
        _, m_s, i_s = torch.native_batch_norm(xi, w, b, None, None, True, 0.0, mod.eps)
        yf, ys = y_f.cpu(), y_s.cpu()
        row = dict(shape=list(xi.shape),
                   fused_gate_flips=int(((yf > 0) != (y_c > 0)).sum()), stock_gate_flips=int(((ys > 0) != (y_c > 0)).sum()),
                   fused_differing=int((yf != y_c).sum()), stock_differing=int((ys != y_c).sum()),
                   fused_mean_differs=int((sm.cpu() != m_c).sum()), fused_invstd_differs=int((si.cpu() != i_c).sum()),
                   stock_mean_differs=int((m_s.cpu() != m_c).sum()), stock_invstd_differs=int((i_s.cpu() != i_c).sum()))
        rows.append(row)
        tot['elements'] += xi.numel()
        tot['fused_gate'] += row['fused_gate_flips']
        tot['stock_gate'] += row['stock_gate_flips']
        tot['fused_any'] += row['fused_differing']
        tot['stock_any'] += row['stock_differing']
        print(json.dumps(row), file=sys.stderr, flush=True)
    # end to end: gradients fused vs stock vs CPU from the same weights
    t = torch.randint(0, 10, (128,), device='cuda')
    grads = {}
    for name in ('fused', 'stock', 'cpu'):
        mm = models.PreResNet(num_classes=10, depth=20)
        mm.load_state_dict(m.state_dict())
        dev = 'cpu' if name == 'cpu' else 'cuda'
        mm = mm.to(dev).train()
        fused_bn.enabled(name == 'fused')
        F.cross_entropy(mm(x.to(dev)), t.to(dev)).backward()
        grads[name] = torch.cat([p.grad.flatten().cpu() for p in mm.parameters()])
    fused_bn.enabled(True)
    sc = float(grads['cpu'].abs().max())
    e2e = {f'{a}_vs_{b}_max_abs_over_max_grad': float((grads[a] - grads[b]).abs().max()) / sc
           for a, b in (('fused', 'cpu'), ('stock', 'cpu'), ('fused', 'stock'))}
    print(json.dumps(dict(what=__doc__.split('\n')[0], totals=tot, end_to_end_gradient=e2e, layers=rows), indent=1))


if __name__ == '__main__':
    main()
