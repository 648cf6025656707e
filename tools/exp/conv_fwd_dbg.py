"""What bounds a K8 launch: the product kernel beside two knobs-build variants that are WRONG on purpose - no stores
(URSA_CONV_FWD_DBG=1), no matrix work (=2) - timed by HIP events over 128-launch graph replays, 8 rotating inputs.
    URSA_CONV_FWD_DBG=<0|1|2> python tools/exp/conv_fwd_dbg.py"""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ursabench_amd import _native
K = _native.knobs_kernels()
dev = torch.device('cuda', 0)
out = {}
for cin, hw in ((16, 32), (32, 16)):
    xs = [torch.randn(128, cin, hw, hw, device=dev) for _ in range(8)]
    w = torch.randn(cin, cin, 3, 3, device=dev) * 0.1
    y = torch.empty_like(xs[0])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(16):
            K.conv3x3(xs[i % 8], w, y)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for i in range(128):
                K.conv3x3(xs[i % 8], w, y)
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); g.replay(); b.record(s); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / 128)
    out[f'{cin}x{hw}'] = sorted(ts)[2]
print(os.environ.get('URSA_CONV_FWD_DBG', '0'), json.dumps(out))
