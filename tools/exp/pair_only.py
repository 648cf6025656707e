"""The 16-channel paired backward launch (batch 128) a few times, through the knobs library: for rocprofv3 --pmc passes
(tools/exp/r06_pair_xcd_ab.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from ursabench_amd import _native  # noqa: E402

K = _native.knobs_kernels()
dev = torch.device('cuda', 0)
N, C, H = 128, 16, 32
sets = 6
xs = [torch.randn(N, C, H, H, device=dev) for _ in range(sets)]
dys = [torch.randn(N, C, H, H, device=dev) for _ in range(sets)]
gs = [torch.empty(N, C, H, H, device=dev) for _ in range(sets)]
w = torch.randn(C, C, 3, 3, device=dev) * 0.1
save = torch.stack([torch.zeros(C), torch.ones(C), torch.ones(C), torch.zeros(C)]).to(dev)
geo = K.preact_geometry(dys[0].shape, C, flip=True)
pb = torch.empty(C, geo[0], 2, dtype=torch.float64, device=dev)
wsf = K.conv_wgrad_ws_floats(xs[0].shape, C, 3, 1)
wss = [torch.empty(wsf, device=dev) for _ in range(sets)]
for i in range(24):
    K.preact_bwd_pair(dys[i % sets], w, gs[i % sets], xs[i % sets], save, pb, wss[i % sets], 1)
torch.cuda.synchronize()
