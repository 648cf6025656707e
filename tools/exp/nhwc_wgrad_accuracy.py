import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from ursabench_amd.tuning import use_shipped_miopen_db
use_shipped_miopen_db('ursa_diag_miopen_')
import torch, torch.nn as nn
from ursabench_amd import fused_bn, fused_conv, models
torch.manual_seed(0)
net = models.WideResNet(100, 28, 10).cuda().train()
x, y = torch.randn(8, 3, 32, 32, device='cuda'), torch.randint(0, 100, (8,), device='cuda')
def run(mode):
    old = fused_bn.nhwc_twins(mode)
    try:
        loss = torch.nn.functional.cross_entropy(net(x), y)
        return torch.autograd.grad(loss, list(net.parameters()))
    finally:
        fused_bn.nhwc_twins(old)
a, b, c = run(True), run(False), run(False)
net64 = models.WideResNet(100, 28, 10).double()
net64.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
net64.train()
loss = torch.nn.functional.cross_entropy(net64(x.double().cpu()), y.cpu())
r = torch.autograd.grad(loss, list(net64.parameters()))
worst = []
for (k, p), g1, g2, g3, g64 in zip(net.named_parameters(), a, b, c, r):
    s = float(g64.abs().max())
    worst.append((float((g1.cpu().double() - g64).abs().max()) / s, float((g2.cpu().double() - g64).abs().max()) / s, float((g2 - g3).abs().max()) / s, k, tuple(p.shape)))
worst.sort(reverse=True)
print('err(nhwc twins) vs f64 | err(stock) vs f64 | stock run-to-run | param')
for w in worst[:12]:
    print('%.2e %.2e %.2e %s %s' % w)
import numpy as np
print('median over params: twins %.2e stock %.2e' % (np.median([w[0] for w in worst]), np.median([w[1] for w in worst])))
