"""K independent PreResNet-20 SGHMC chains on ONE GPU, each a captured hipGraph replayed on its own
stream: does kernel-level concurrency raise aggregate minibatch-steps/s?"""
import os, sys, tempfile, time
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='ursa_mc_'))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import inference, models, util
from ursabench_amd.data import synthetic

dev = torch.device('cuda', 0)
train = synthetic(128 * 40, (3, 32, 32), 10, seed=0, device=dev, batch_size=128)
hyp = {'lr': 0.1, 'prior_std': 0.5, 'num_samples': 1, 'alpha': 0.5, 'burn_in_epochs': 0}
for K in (1, 2, 4, 8):
    chains, streams = [], []
    for k in range(K):
        util.set_random_seed(k)
        s = inference.SGHMC(dict(hyp), models.PreResNet(10, 20).to(dev), train, device=dev)
        st = torch.cuda.Stream(dev)
        with torch.cuda.stream(st):
            s.sample_iterative()          # warm-up epoch: captures the graph on this stream's context
        chains.append(s); streams.append(st)
    torch.cuda.synchronize()
    # drive the captured graphs directly, round-robin over chains
    xs = [b for b in train]
    t0 = time.perf_counter()
    steps = 0
    for rep in range(3):
        for s, st in zip(chains, streams):
            s.optimizer.ctl_begin(True)
        for x, y in xs:
            for s, st in zip(chains, streams):
                with torch.cuda.stream(st):
                    s.engine._static[0].copy_(x); s.engine._static[1].copy_(y)
                    s.engine._graph.replay()
            steps += K
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'K={K}: {steps / dt:8.1f} aggregate minibatch steps/s   ({dt / (steps / K) * 1e3:.3f} ms per round of K steps)', flush=True)
    del chains, streams
