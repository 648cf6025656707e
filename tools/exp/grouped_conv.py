"""Would K chains batched as ONE grouped convolution beat K separate convolutions? (fp32, B=128)"""
import os, tempfile, time
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='ursa_gc_'))
import torch
dev = torch.device('cuda')


def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


for (c, hw) in ((16, 32), (32, 16), (64, 8)):
    for K in (1, 4, 8):
        x = torch.randn(128, c * K, hw, hw, device=dev, requires_grad=True)
        conv = torch.nn.Conv2d(c * K, c * K, 3, padding=1, groups=K, bias=False).to(dev)
        bn = torch.nn.BatchNorm2d(c * K).to(dev)

        def fb():
            y = bn(conv(x)).relu()
            y.sum().backward()
        ms = t(fb)
        print(f'C={c:3d} HW={hw:2d} K={K}: conv+bn+relu fwd+bwd {ms:7.3f} ms  -> per chain {ms / K:7.3f} ms', flush=True)
