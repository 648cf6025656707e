"""PreResNet-20 B=128: MIOpen BatchNorm vs ATen's native BatchNorm kernels (torch.batch_norm with
cudnn_enabled=False), training step and eval forward, under hipGraph replay."""
import os, sys, time
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import models

dev = torch.device('cuda')
crit = torch.nn.CrossEntropyLoss()
_orig = F.batch_norm


def native_bn(input, running_mean, running_var, weight=None, bias=None, training=False, momentum=0.1, eps=1e-5):
    torch._C._set_cudnn_enabled(False)      # ATen picks the BN backend from the GLOBAL flag
    try:
        return _orig(input, running_mean, running_var, weight, bias, training, momentum, eps)
    finally:
        torch._C._set_cudnn_enabled(True)


def graph_time(fn, steps=200):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for name, bn in (('MIOpen BN', _orig), ('native BN', native_bn), ('MIOpen BN', _orig), ('native BN', native_bn)):
    F.batch_norm = bn
    torch.manual_seed(0)
    net = models.PreResNet(10, 20).to(dev)
    x = torch.randn(128, 3, 32, 32, device=dev); y = torch.randint(0, 10, (128,), device=dev)
    params = list(net.parameters())

    def train_step():
        loss = crit(net(x), y)
        loss.backward()
        for p in params:
            p.grad = None
    net.train()
    t_train = graph_time(train_step)
    net.eval()
    with torch.no_grad():
        t_eval = graph_time(lambda: net(x))
    print(f'{name}: train step {t_train:.3f} ms, eval forward {t_eval:.3f} ms', flush=True)
