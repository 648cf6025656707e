"""K independent PreResNet-20 SGHMC steps captured as parallel branches of ONE hipGraph (fork/join on K
side streams inside the capture): does the runtime overlap the branches?"""
import os, sys, tempfile, time
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='ursa_mc1_'))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import inference, models, util
from ursabench_amd.data import synthetic

dev = torch.device('cuda', 0)
train = synthetic(128 * 8, (3, 32, 32), 10, seed=0, device=dev, batch_size=128)
hyp = {'lr': 0.1, 'prior_std': 0.5, 'num_samples': 1, 'alpha': 0.5, 'burn_in_epochs': 0}
x, y = next(iter(train))
for K in (1, 2, 4, 8):
    chains = []
    for k in range(K):
        util.set_random_seed(k)
        s = inference.SGHMC(dict(hyp), models.PreResNet(10, 20).to(dev), train, device=dev, use_graph=False)
        s.sample_iterative()                      # eager warm-up epoch (MIOpen search, allocator)
        s.optimizer.ctl_begin(True)
        chains.append(s)
    torch.cuda.synchronize()
    side = [torch.cuda.Stream(dev) for _ in range(K)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        cap = torch.cuda.current_stream()
        for s, st in zip(chains, side):
            st.wait_stream(cap)
            with torch.cuda.stream(st):
                s.engine._train_step(x, y)
        for st in side:
            cap.wait_stream(st)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'K={K}: one graph with {K} parallel branches: {dt / n * 1e3:7.3f} ms per replay -> {K * n / dt:8.1f} aggregate steps/s', flush=True)
    del chains, g
