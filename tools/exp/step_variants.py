"""What does a PreResNet-20 B=128 training step cost under hipGraph replay, and which host-side
choices move it? (fp32, MI355X). Prints ms/step for several variants."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import models

dev = torch.device('cuda')
crit = torch.nn.CrossEntropyLoss()


def bench(name, channels_last=False, benchmark=False, grads_none=False, no_nbt=False, flat_grad_views=False, pack=None, steps=200):
    torch.backends.cudnn.benchmark = benchmark
    torch.manual_seed(0)
    net = models.PreResNet(10, 20).to(dev)
    x = torch.randn(128, 3, 32, 32, device=dev)
    y = torch.randint(0, 10, (128,), device=dev)
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    if no_nbt:
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.num_batches_tracked = None
    params = list(net.parameters())
    if flat_grad_views:
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n, device=dev)
        off = 0
        for p in params:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
    if pack:
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n, device=dev)
        views, off = [], 0
        for p in params:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
    net.train()

    def step():
        loss = crit(net(x), y)
        loss.backward()
        if pack == 'foreach_copy':
            torch._foreach_copy_(views, [p.grad for p in params])
        elif pack == 'foreach_add':
            torch._foreach_add_(views, [p.grad for p in params])
            flat.mul_(0.0)       # stands in for the fused re-zeroing in K1
        elif pack == 'cat':
            torch.cat([p.grad.reshape(-1) for p in params], out=flat)
        if grads_none:
            for p in params:
                p.grad = None
        elif flat_grad_views:
            flat.zero_()

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    print(f'{name:55s} {(time.perf_counter() - t0) / steps * 1e3:7.3f} ms/step', flush=True)


bench('grads accumulate into flat views (+memset)   [old]', flat_grad_views=True)
bench('grads=None (no packing at all)', grads_none=True)
bench('grads=None + _foreach_copy_ into flat', grads_none=True, pack='foreach_copy')
bench('grads=None + _foreach_add_ into zeroed flat', grads_none=True, pack='foreach_add')
bench('grads=None + torch.cat(out=flat)', grads_none=True, pack='cat')
