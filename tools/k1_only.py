"""Launch only the fused update kernel (for rocprofv3 --pmc passes): 20 launches at the PreResNet-20
arena size through the control-block entry point (the bench's launch) and 10 at 2^26 elements."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ursabench_amd import _native
K = _native.default_kernels()
which = sys.argv[1] if len(sys.argv) > 1 else 'both'
if which in ('small', 'both'):
    n = 273408
    th, g, m = (torch.randn(n, device='cuda') for _ in range(3))
    c = _native.StepCtl(lr=0.1, mu=0.5, c_wd=8e-5, c_noise=0.3, n_train=50000.0, flags=0x1 | 0x8 | 0x20, seed=1, step=0)     # NOISE | WD | ADVANCE
    ctl = torch.frombuffer(bytearray(bytes(c)), dtype=torch.uint8).cuda()
    for _ in range(20):
        K.sgmcmc_step_ctl(th, g, m, ctl)           # advances its own control block
if which in ('large', 'both'):
    n = 1 << 26
    th, g, m = (torch.randn(n, device='cuda') for _ in range(3))
    for k in range(10):
        K.sgmcmc_step(th, g, m, lr=0.1, mu=0.5, c_wd=8e-5, c_noise=0.3, n_train=50000.0, flags=0x1 | 0x8, seed=1, step=k)
torch.cuda.synchronize()
