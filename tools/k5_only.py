import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ursabench_amd import _native
K = _native.default_kernels()
S, B, C = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (30, 10000, 100)
z = torch.randn(S, B, C, device='cuda') * 3
p, e = torch.zeros(B, C, device='cuda'), torch.zeros(B, device='cuda')
for _ in range(10):
    K.bma_accumulate(z, p, e, one_minus_gamma=0.9999, gamma_over_c=1e-4 / C, smoothed=False)
torch.cuda.synchronize()
