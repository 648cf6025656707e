"""Second pure-PyTorch reproduction attempt of the corrupted `(y > 0).sum()` inside replayed hipGraphs (profiles/
r05_gate_probe_root_cause.txt). NOTHING of ursabench_amd is imported: a small pre-activation ResNet written here with stock
torch ops (MIOpen convolutions and BatchNorm, ATen ReLU), a per-tensor torch SGD update, the training step captured with
torch.cuda.graph after eager warm-up on a side stream - and, at every relu(bn(x)) of the forward, the open-gate count taken
the way round 4's GateProbe took it: n_open[k].copy_((y > 0).sum()). Between replays: host-to-device copies and small-pool
allocations, as the G16 replays have them. Before every replay the expected counts are computed by an eager no-grad forward on
the same weights and batch (the forward is deterministic), then compared.

    python tools/exp/graph_reduce_repro2.py [replays] -> gpurun_out/graph_reduce_repro2.json
"""
import json
import os
import sys

import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = torch.device('cuda', 0)
REPLAYS = int(sys.argv[1]) if len(sys.argv) > 1 else 8


class Block(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.bn1, self.conv1 = nn.BatchNorm2d(cin), nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn2, self.conv2 = nn.BatchNorm2d(cout), nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.down = None if stride == 1 and cin == cout else nn.Conv2d(cin, cout, 1, stride, bias=False)

    def forward(self, x, obs):
        h = obs(F.relu(self.bn1(x)))
        y = self.conv2(obs(F.relu(self.bn2(self.conv1(h)))))
        return y + (x if self.down is None else self.down(x))


class Net(nn.Module):                         # PreResNet-8-like: 7 relu(bn(.)) sites
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 16, 3, 1, 1, bias=False)
        self.b1, self.b2, self.b3 = Block(16, 16, 1), Block(16, 32, 2), Block(32, 64, 2)
        self.bn, self.fc = nn.BatchNorm2d(64), nn.Linear(64, 10)

    def forward(self, x, obs):
        x = self.b3(self.b2(self.b1(self.conv(x), obs), obs), obs)
        return self.fc(obs(F.relu(self.bn(x))).mean((2, 3)))


def run(mode, between, graph=True):
    torch.manual_seed(0)
    net = Net().to(dev).train()
    params = list(net.parameters())
    x, y = torch.randn(128, 3, 32, 32, device=dev), torch.randint(0, 10, (128,), device=dev)
    n_open = torch.zeros(7, dtype=torch.int64, device=dev)
    state = {'k': 0}

    def obs(t):
        k = state['k']
        state['k'] += 1
        flat = t.detach().reshape(-1)
        if mode == 'tmp_copy':
            n_open[k].copy_((flat > 0).sum())
        else:                                                   # rows of 4096 per workgroup, then the row sums: no multi-workgroup reduction
            torch.sum((flat > 0).view(-1, 4096).sum(1), dim=0, keepdim=True, out=n_open[k:k + 1])
        return t

    def step():
        state['k'] = 0
        loss = F.cross_entropy(net(x, obs), y)
        for p in params:
            p.grad = None
        loss.backward()
        with torch.no_grad():
            for p in params:
                p.add_(p.grad, alpha=-0.05)

    def expected():
        cnt = []
        with torch.no_grad():
            net(x, lambda t: (cnt.append(int((t > 0).sum())), t)[1])
        for m in net.modules():                                 # (the eager forward above moved the running statistics: irrelevant in train mode)
            pass
        return cnt

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = None
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            step()
    bad = []
    keep = []
    for rep in range(REPLAYS):
        x.copy_(torch.randn(128, 3, 32, 32).to(dev))           # a new batch (host-to-device copy, like the probe's gate lists)
        want = expected()
        n_open.zero_()
        if g is not None:
            g.replay()
        else:
            step()                                              # the same step through eager launches
        got = n_open.cpu().tolist()
        if got != want:
            bad.append(dict(replay=rep, got=got, want=want))
        if between == 'small_alloc':
            torch.zeros(8193, dtype=torch.int32, device=dev)
            keep.append(torch.empty(79000, device=dev).normal_())   # bank-row sized small-pool allocations that stay alive
    return bad


out = {}
for mode, graph in (('tmp_copy', True), ('two_stage', True), ('tmp_copy', False)):
    for between in ('none', 'small_alloc'):
        for trial in range(3):
            key = f'{mode}/{"hipGraph replay" if graph else "eager launches"}/{between}/trial{trial}'
            try:
                bad = run(mode, between, graph)
            except Exception as e:          # noqa: BLE001
                bad = repr(e)
            out[key] = bad
            print(key, 'MISMATCH' if bad else 'ok', bad if isinstance(bad, str) else [(b['replay'], [(i, g_, w_) for i, (g_, w_) in enumerate(zip(b['got'], b['want'])) if g_ != w_]) for b in bad][:3], flush=True)
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'graph_reduce_repro2.json'), 'w'), indent=1)
