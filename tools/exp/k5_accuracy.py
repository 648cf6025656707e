import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as O
from ursabench_amd import _native
K = _native.default_kernels()
rng = np.random.default_rng(0)
for C, scale in ((10, 4), (100, 4), (100, 10), (10, 12), (17, 6), (200, 6)):
    z = (rng.standard_normal((8, 4096, C)) * scale).astype(np.float32)
    po, eo = np.zeros((4096, C), np.float32), np.zeros(4096, np.float32)
    O.bma_accumulate(z, po, eo, one_minus_gamma=1 - 1e-4, gamma_over_c=1e-4 / C, smoothed=False)
    pd, ed = torch.zeros(4096, C, device="cuda"), torch.zeros(4096, device="cuda")
    K.bma_accumulate(torch.from_numpy(z).cuda(), pd, ed, one_minus_gamma=1 - 1e-4, gamma_over_c=1e-4 / C, smoothed=False)
    p, e = pd.cpu().numpy(), ed.cpu().numpy()
    big = po > 1e-6
    print("C", C, "scale", scale, "max rel err proba (p>1e-6)", float(np.max(np.abs(p - po)[big] / po[big])), "entropy", float(np.max(np.abs(e - eo) / np.abs(eo))))
