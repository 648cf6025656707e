"""Pure-PyTorch reproduction attempts of the hipGraphLaunch segfault (hip::Graph::UpdateStreams) seen when graph
objects of samplers and tasks are created and destroyed in one process. No ursabench_amd code involved.
    python tools/exp/hipgraph_lifetime_repro.py            # runs every pattern in a child process"""
import os, subprocess, sys
PATTERNS = ['E_T_T', 'T_T', 'E_T', 'E_T_T_sync_del', 'E_T_T_lanes1', 'E_T_T_noE_destroy', 'E_T_T_linear_E_only_ops']
if len(sys.argv) == 1:
    for pat in PATTERNS:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), pat], capture_output=True, text=True, timeout=600)
        last = [ln for ln in p.stdout.splitlines() if ln.startswith('it ')]
        print(f'{pat:28s} rc={p.returncode:4d} last: {last[-1] if last else None}', flush=True)
    sys.exit(0)
import torch
pat = sys.argv[1]
dev = torch.device('cuda', 0)
x = torch.randn(64, 16, 32, 32, device=dev)


def net():
    return torch.nn.Sequential(torch.nn.Conv2d(16, 16, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(16, 16, 3, padding=1)).to(dev)


def capture_single(train=True):
    m = net()
    sx = x.clone()

    def step():
        if train:
            m.zero_grad(set_to_none=True)
            m(sx).square().mean().backward()
        else:
            m(sx)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode='thread_local'):
        step()
    return g, m, sx


def capture_multi(lanes=4):
    ms = [net().eval() for _ in range(lanes)]
    sx = x.clone()
    side = [torch.cuda.Stream() for _ in ms]
    with torch.no_grad():
        for m in ms:
            m(sx)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        outs = []
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            cap = torch.cuda.current_stream()
            for m, st in zip(ms, side):
                st.wait_stream(cap)
                with torch.cuda.stream(st):
                    outs.append(m(sx))
            for st in side:
                cap.wait_stream(st)
    return g, ms, sx, outs


lanes = 1 if 'lanes1' in pat else 4
keepE = []
for it in range(8):
    if pat.startswith('E'):
        if pat == 'E_T_T_sync_del' and it:
            torch.cuda.synchronize()
            del E, T
        E = capture_single(train='linear' not in pat)
        for _ in range(5):
            E[0].replay()
        if 'noE_destroy' in pat:
            keepE.append(E)
    for k in range(2 if pat != 'E_T' else 1):
        T = capture_multi(lanes)          # rebinding destroys the previous multi-branch graph first... after this one is built
        for _ in range(5):
            T[0].replay()
    torch.cuda.synchronize()
    print('it', it, 'ok', flush=True)
