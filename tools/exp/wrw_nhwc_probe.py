"""Weight gradient alone of PreResNet-20's 3x3 convolutions (batch 128): NCHW operands (what the networks hand MIOpen; its
NHWC implicit-GEMM kernel then transposes x and dy itself) vs channels-last operands under PYTORCH_MIOPEN_SUGGEST_NHWC=1.
HIP events around hipGraph replays of 20 calls.  Run:  PYTORCH_MIOPEN_SUGGEST_NHWC=1 python3 tools/exp/wrw_nhwc_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import tuning  # noqa: E402

tuning.use_shipped_miopen_db()
res = {}
for cin, cout, hw in ((16, 16, 32), (32, 32, 16), (64, 64, 8)):
    row = {}
    for layout in ('nchw', 'nhwc'):
        fmt = torch.channels_last if layout == 'nhwc' else torch.contiguous_format
        w = torch.randn(cout, cin, 3, 3, device='cuda').contiguous(memory_format=fmt)
        x = torch.randn(128, cin, hw, hw, device='cuda').contiguous(memory_format=fmt)
        dy = torch.randn(128, cout, hw, hw, device='cuda').contiguous(memory_format=fmt)

        def step():
            return torch.ops.aten.convolution_backward(dy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                       [False, True, False])[1]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                gw = step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                step()
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / 20)
        ts.sort()
        row[layout] = dict(us=round(ts[3], 2), grad_weight_is_channels_last=bool(gw.is_contiguous(memory_format=torch.channels_last) and not gw.is_contiguous()))
    res[f'{cin}x{hw}x{hw}'] = row
print(json.dumps({'suggest_nhwc': os.environ.get('PYTORCH_MIOPEN_SUGGEST_NHWC'), 'weight_gradient_us': res}))
