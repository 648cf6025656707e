"""Is it torch's own reduction inside a hipGraph? A pure-PyTorch reproduction attempt of the GateProbe.n_open corruption
(tools/exp/gate_probe_overwrite2.py: the count `(y > 0).sum()` of a 2 M-element activation, taken inside the captured training
step through a graph-pool temporary, came back as float bit patterns on the third replay; MIOpen's BatchNorm in K6's place
shows it too, eager launches and a persistent output tensor do not). No ursabench_amd code runs here.

The captured body imitates what surrounds the count in the step: small float tensors are written and freed (so that the
512-byte blocks the reduction's output / staging buffer / semaphores land in have held float data), then a global reduction
of a large bool tensor goes to a graph-pool temporary and is copied to a persistent counter; afterwards more small float
tensors are written (the backward's per-channel gradients). Between replays: the small-pool allocations of the epoch end.

    python tools/exp/graph_reduce_repro.py      -> gpurun_out/graph_reduce_repro.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device('cuda', 0)
torch.manual_seed(0)
N_CALLS = 7
SHAPES = [(128, 16, 32, 32)] * 3 + [(128, 32, 16, 16)] * 2 + [(128, 64, 8, 8)] * 2


def body(xs, n_open, mode, held):
    for k, x in enumerate(xs):
        C = x.shape[1]
        y = torch.relu(x * 1.0001)                                  # a fresh activation in the graph pool
        stats = x.new_empty(2, C).fill_(3.1e-5 * (k + 1))           # small float tensors before the count ...
        flat = y.reshape(-1)
        if mode == 'tmp':
            t = (flat > 0).sum()
            if held is not None:
                held.append(t)
            n_open[k].copy_(t)
        elif mode == 'out':
            torch.sum(flat > 0, dim=0, keepdim=True, out=n_open[k:k + 1])
        del stats
    for k, x in enumerate(xs):                                      # ... and after it (the backward's per-channel gradients)
        C = x.shape[1]
        dwb = x.new_empty(2, C).fill_(4.2e-5 * (k + 1))
        dwb.mul_(1.5)
        del dwb


def run(mode, hold, side_warmup, between):
    xs = [torch.randn(s, device=dev) for s in SHAPES]
    want = torch.tensor([int((x * 1.0001 > 0).sum()) for x in xs])
    n_open = torch.zeros(N_CALLS, dtype=torch.int64, device=dev)
    held = [] if hold else None
    if side_warmup:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            body(xs, n_open, mode, None)
        torch.cuda.current_stream().wait_stream(s)
    else:
        body(xs, n_open, mode, None)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        body(xs, n_open, mode, held)
    bad = []
    for rep in range(6):
        for x in xs:
            x.normal_()
        want = torch.tensor([int((x * 1.0001 > 0).sum()) for x in xs])
        n_open.zero_()
        g.replay()
        got = n_open.cpu()
        if not torch.equal(got, want):
            bad.append(dict(replay=rep, got=got.tolist(), want=want.tolist()))
        if between == 'small_alloc':
            torch.zeros(8193, dtype=torch.int32, device=dev)
            row = torch.empty(79000, device=dev).normal_()          # a bank-row-sized small-pool allocation (316 KB)
            del row
    return bad


out = {}
for mode in ('tmp', 'out'):
    for hold in (False, True):
        for side in (False, True):
            for between in ('none', 'small_alloc'):
                if mode == 'out' and hold:
                    continue
                key = f'{mode}/hold={hold}/side_warmup={side}/{between}'
                try:
                    bad = run(mode, hold, side, between)
                except Exception as e:       # noqa: BLE001
                    bad = repr(e)
                out[key] = bad
                print(key, 'MISMATCH' if bad else 'ok', (bad if isinstance(bad, str) else bad[:2]), flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'graph_reduce_repro.json'), 'w'), indent=1)
