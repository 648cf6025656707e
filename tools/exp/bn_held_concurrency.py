"""K6 held forms: how many launches may be in flight at once before one starves (bounded wait -> err word). Scenarios of
forward (32 pieces per channel, one 512-thread workgroup per CU) and backward (64 pieces, two per CU) launches on separate
streams, repeated; prints per scenario how many launches raised their err word and the wall time.
    python tools/exp/bn_held_concurrency.py [fwd_streams bwd_streams reps] ..."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from ursabench_amd import _native
K = _native.default_kernels()
C = 8
FS = tuple(int(v) for v in os.environ.get('FSHAPE', '2624,8,32,32').split(','))
BS = tuple(int(v) for v in os.environ.get('BSHAPE', '1536,8,32,32').split(','))
args = [int(a) for a in sys.argv[1:]]
scen = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(4, 0, 3), (8, 0, 3), (0, 4, 3), (0, 8, 3), (4, 4, 3), (2, 2, 3), (1, 7, 3), (7, 1, 3)]
g = torch.Generator().manual_seed(1)
w, bb = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
new = lambda: torch.zeros(C, device='cuda')
res = []
for nf, nb, reps in scen:
    xs = [torch.randn(FS, generator=g).cuda() for _ in range(nf)]
    ys = [torch.empty_like(x) for x in xs]
    bx = [(torch.randn(BS, generator=g).cuda(), torch.randn(BS, generator=g).cuda()) for _ in range(nb)]
    dxs = [torch.empty_like(x) for x, _ in bx]
    st = [(new(), new() + 1) for _ in range(nb)]
    wss = [torch.zeros(_native.bn_ws_floats(C), device='cuda') for _ in range(nf + nb)]
    streams = [torch.cuda.Stream() for _ in range(nf + nb)]
    torch.cuda.synchronize()
    bad_runs = 0
    t0 = time.perf_counter()
    TRIALS = int(os.environ.get('TRIALS', '5'))
    for trial in range(TRIALS):
        for rep in range(reps):
            for k in range(max(nf, nb)):
                if k < nf:
                    with torch.cuda.stream(streams[k]):
                        K.bn_relu_forward(xs[k], ys[k], w, bb, None, None, new(), new(), wss[k], eps=1e-5, momentum=0.0, held=True)
                if k < nb:
                    with torch.cuda.stream(streams[nf + k]):
                        K.bn_relu_backward(bx[k][0], bx[k][1], dxs[k], w, bb, st[k][0], st[k][1], new(), new(), wss[nf + k], held=True)
        torch.cuda.synchronize()
        errs = [int(ws[C * 256:].view(torch.int32)[C * 256 + 33]) for ws in wss]
        if any(errs):
            bad_runs += 1
            for ws in wss:
                ws.zero_()
    r = dict(fwd_streams=nf, bwd_streams=nb, reps=reps, trials=TRIALS, trials_with_a_starved_launch=bad_runs, seconds=round(time.perf_counter() - t0, 2),
             hw_queues=os.environ.get('GPU_MAX_HW_QUEUES', 'default'))
    print(json.dumps(r), flush=True)
    res.append(r)
    del xs, ys, bx, dxs
os.makedirs('gpurun_out', exist_ok=True)
json.dump(res, open('gpurun_out/bn_held_concurrency.json', 'w'), indent=1)
