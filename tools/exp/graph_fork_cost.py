"""What a cross-stream edge costs inside a hipGraph replay: a chain of N small kernels on the capture stream, with every
k-th of them also forking a small kernel onto a second stream (side.wait_stream(main) per fork, ONE join at the end),
against the same kernels all on one stream. us per replay, HIP events.
    python tools/exp/graph_fork_cost.py > gpurun_out/graph_fork_cost.json"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd._capture import capture, side_streams  # noqa: E402

dev = torch.device('cuda')
N = 60
a = [torch.randn(1 << 18, device=dev) for _ in range(N)]         # 1 MB each: ~2 us kernels
b = [torch.randn(1 << 18, device=dev) for _ in range(N)]


def timed(body, reps=50):
    for _ in range(3):
        body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with capture(g):
        body()
    g.replay()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        g.replay()
    t1.record()
    t1.synchronize()
    return t0.elapsed_time(t1) * 1e3 / reps


def make(forks):
    side = side_streams(dev, 2)[1]

    def body():
        cur = torch.cuda.current_stream()
        every = N // forks if forks else 0
        for k in range(N):
            a[k].mul_(1.0001)                                     # the chain
            if forks and k % every == 0:
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    b[k].mul_(1.0001)                             # a leaf beside it
            elif not forks and k % max(1, N // 20) == 0:
                b[k].mul_(1.0001)                                 # the same leaves on the chain's own stream
        if forks:
            cur.wait_stream(side)
    return body


res = {'kernels_on_chain': N}
res['one_stream_20_leaves_us'] = round(timed(make(0)), 1)
for f in (1, 3, 6, 20):
    res[f'{f}_forks_us'] = round(timed(make(f)), 1)
print(json.dumps(res))
