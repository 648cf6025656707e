"""SURVEY.md 8f-1's two ways to run K chains per GPU, at the convolution level (the 3x3 convolutions are 48 % of a
PreResNet-20 step's kernel time): (a) K independent convolutions on K forked streams inside one hipGraph - what
inference.ChainGroup captures - against (b) ONE grouped convolution over [B, K*C, H, W] with groups=K (a chain per
group). Forward + backward-data + backward-weight per layer, 6 layers deep per graph, us per (layer, chain).
    python tools/exp/grouped_conv_ab.py > gpurun_out/grouped_conv_ab.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ursabench_amd.tuning import use_shipped_miopen_db  # noqa: E402
use_shipped_miopen_db('ursa_gconv_miopen_')
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from ursabench_amd._capture import capture, side_streams  # noqa: E402

dev = torch.device('cuda')
DEPTH, B = 6, 128


def timed_graph(body, reps=30):
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with capture(g):
        body()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) * 1e3 / reps          # us per replay


def layer_fwd_bwd(x, w, groups):
    x = x.requires_grad_(True)
    y = F.conv2d(x, w, None, 1, 1, 1, groups)
    gx, gw = torch.autograd.grad(y, (x, w), torch.ones_like(y))
    return gx, gw


def main():
    out = []
    for C, HW in ((16, 32), (32, 16), (64, 8)):
        for K in (1, 2, 4, 8, 16):
            # (a) K branches
            xs = [torch.randn(B, C, HW, HW, device=dev) for _ in range(K)]
            ws = [torch.randn(C, C, 3, 3, device=dev, requires_grad=True) for _ in range(K)]
            side = side_streams(dev, K)

            def branches():
                cur = torch.cuda.current_stream()
                for k in range(K):
                    side[k].wait_stream(cur)
                    with torch.cuda.stream(side[k]):
                        for _ in range(DEPTH):
                            layer_fwd_bwd(xs[k].detach(), ws[k], 1)
                for k in range(K):
                    cur.wait_stream(side[k])
            us_a = timed_graph(branches)
            # (b) one grouped convolution
            xg = torch.randn(B, K * C, HW, HW, device=dev)
            wg = torch.randn(K * C, C, 3, 3, device=dev, requires_grad=True)

            def grouped():
                for _ in range(DEPTH):
                    layer_fwd_bwd(xg.detach(), wg, K)
            try:
                us_b = timed_graph(grouped)
            except Exception as e:      # noqa: BLE001
                us_b = None
                print('grouped failed', C, HW, K, repr(e)[:200], file=sys.stderr)
            rec = dict(channels=C, hw=HW, chains=K, branches_us_per_layer_chain=round(us_a / DEPTH / K, 2),
                       grouped_us_per_layer_chain=None if us_b is None else round(us_b / DEPTH / K, 2))
            out.append(rec)
            print(json.dumps(rec), file=sys.stderr)
            del xs, ws, xg, wg
            torch.cuda.empty_cache()
    print(json.dumps(dict(what='fwd + bwd-data + bwd-weight of one 3x3 convolution at batch 128, us per (layer, chain): K stream '
                               'branches in one hipGraph vs one grouped convolution (groups = K)', rows=out), indent=1))


if __name__ == '__main__':
    main()
