"""3x3 convolution forward + backward (data and weight gradient) of PreResNet-20's three stages at batch 128: NCHW (what the
networks run) vs channels_last with PYTORCH_MIOPEN_SUGGEST_NHWC=1 (MIOpen's NHWC kernels without its own transposes).
HIP events around hipGraph replays of 20 forward+backward passes. Run once per layout (the switch is read at import):
    python3 tools/exp/conv_layout_ab.py nchw ; PYTORCH_MIOPEN_SUGGEST_NHWC=1 python3 tools/exp/conv_layout_ab.py nhwc
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import tuning  # noqa: E402

if os.environ.get('URSA_CONV_AB_TUNED', '1') == '1':
    tuning.use_shipped_miopen_db()
layout = sys.argv[1]
fmt = torch.channels_last if layout == 'nhwc' else torch.contiguous_format
res = {}
for cin, cout, hw in ((16, 16, 32), (32, 32, 16), (64, 64, 8)):
    conv = torch.nn.Conv2d(cin, cout, 3, 1, 1, bias=False).cuda().to(memory_format=fmt)
    x = torch.randn(128, cin, hw, hw, device='cuda').contiguous(memory_format=fmt).requires_grad_(True)
    dy = torch.randn(128, cout, hw, hw, device='cuda').contiguous(memory_format=fmt)

    def step():
        y = conv(x)
        return torch.autograd.grad(y, (x, conv.weight), dy)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
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
    res[f'{cin}x{hw}x{hw}'] = round(ts[3], 2)
print(json.dumps({'layout': layout, 'suggest_nhwc': os.environ.get('PYTORCH_MIOPEN_SUGGEST_NHWC'), 'us_fwd_bwd_per_conv': res}))
