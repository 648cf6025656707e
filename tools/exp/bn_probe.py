import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import models
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda')
_orig = F.batch_norm
def native_bn(input, running_mean, running_var, weight=None, bias=None, training=False, momentum=0.1, eps=1e-5):
    torch._C._set_cudnn_enabled(False)      # ATen picks the BN backend from the GLOBAL flag
    try:
        return _orig(input, running_mean, running_var, weight, bias, training, momentum, eps)
    finally:
        torch._C._set_cudnn_enabled(True)
_orig = F.batch_norm
net = models.PreResNet(10, 20).to(dev)
x = torch.randn(128, 3, 32, 32, device=dev); y = torch.randint(0, 10, (128,), device=dev)
for name, bn in (('MIOpen', _orig), ('native', native_bn)):
    F.batch_norm = bn
    for mode in ('eval', 'train'):
        net.train(mode == 'train')
        def run():
            if mode == 'eval':
                with torch.no_grad(): net(x)
            else:
                net.zero_grad(); F.cross_entropy(net(x), y).backward()
        for _ in range(3): run()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(5): run()
            torch.cuda.synchronize()
        rows = sorted(prof.key_averages(), key=lambda e: -e.device_time_total)[:7]
        print(f'--- {name} {mode}: total device ms/iter', sum(e.device_time_total for e in prof.key_averages()) / 5e3)
        for e in rows:
            print(f'   {e.key[:70]:70s} calls {e.count:4d}  avg {e.device_time_total / e.count:8.1f} us')
