"""Headline benchmark: posterior-samples/sec (+ BMA-predictions/sec) for PreResNet-20 / CIFAR-10-
shaped synthetic data, SGHMC, one chain per GPU (BASELINE.json configs[1]; configs[2] at N > 1).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one posterior sample: one epoch of ceil(50000/128) = 391 minibatch updates
(forward + backward in stock PyTorch-ROCm, then ONE fused gfx950 update launch) followed by a
device-to-device snapshot into the member bank (URSABench/inference/sghmc.py:65-101; protocol of
URSABench/time_script.py:89-114: burn-in 0). W untimed samples, then exactly K timed samples
bracketed by barrier + synchronize; rank 0 prints ONE JSON line. value = samples of ALL ranks /
max-over-ranks time. After the timed region the K-member ensemble of every rank is evaluated on
the 10,000-row test set (Prediction.update_statistics + one all-reduce): `bma_preds_per_s`.

Extra objects: `roofline` (the update kernel, HIP events on its own stream, same launch as the
workload's) and `cpu_baseline` (the torch-CPU port of the reference path, oracle/torch_cpu_path.py,
timed on this box's host cores on a bounded sample; rank 0, N = 1 only).
"""
import argparse
import importlib.util
import json
import os
import sys
import tempfile
import time

# A private MIOpen user find-db for this process: MIOpen caches its per-layer solver search under
# $HOME and reuses it across processes, whatever switches the recording process ran with (a search
# recorded in deterministic mode made the same benchmark 8x slower on the same box). Every bench
# run therefore does its own (sub-second per layer, inside the warm-up sample) search.
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='ursa_bench_miopen_'))

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured float4 copy ceiling: 6290
HBM_COPY_GBPS = 6290.0

# C2 hyper-parameters: URSABench/hyperparams/ResNet50CIFAR10/sghmc_hyperparams.json (no PreResNet-20
# file exists in the reference), burn-in forced to 0 as time_script.py:89-90 does.
HYP = {'lr': 0.1, 'prior_std': 0.5, 'alpha': 0.5, 'burn_in_epochs': 0}
N_TRAIN, N_TEST, BATCH, CLASSES = 50000, 10000, 128, 10


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3, help='timed posterior samples per chain (time_script S=3)')
    ap.add_argument('--warmup', type=int, default=1, help='untimed posterior samples per chain')
    ap.add_argument('--no-graph', action='store_true', help='eager launches instead of hipGraph replay')
    ap.add_argument('--chains-per-gpu', type=int, default=1, help='>1: ChainGroup (parallel graph branches); the headline config is 1')
    ap.add_argument('--multi-chain-probe', type=int, default=4, help='chains of the informational multi-chain run at N=1 (0: skip)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--force-dist', action='store_true', help='join an RCCL process group even when WORLD_SIZE is 1')
    ap.add_argument('--cpu-steps', type=int, default=200, help='minibatch steps of the CPU port to time')
    ap.add_argument('--ref-style-steps', type=int, default=150, help='eager reference-style GPU steps to time (0: skip)')
    ap.add_argument('--large-n', type=int, default=1 << 26, help='elements of the roofline-sized K1 launch')
    return ap.parse_args()


def event_time_ms(fn, iters, stream, graph_batch=0):
    """Average duration of one fn() launch, HIP events recorded on `stream` (the stream the kernel is
    launched on). With graph_batch > 0 the launches are captured `graph_batch` at a time into a
    hipGraph and the replay is timed, so the host's per-launch Python/ctypes cost (~10 us) is not
    what is measured; the figure then is kernel duration + the ~1.5 us dependent-kernel boundary."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if graph_batch:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):     # RCCL's watchdog thread may be alive
            for _ in range(graph_batch):
                fn()
        g.replay()
        torch.cuda.synchronize()
        reps = max(1, iters // graph_batch)
        s = torch.cuda.current_stream()
        a.record(s)
        for _ in range(reps):
            g.replay()
        b.record(s)
        b.synchronize()
        return a.elapsed_time(b) / (reps * graph_batch)
    with torch.cuda.stream(stream):
        for _ in range(5):
            fn()
        a.record(stream)
        for _ in range(iters):
            fn()
        b.record(stream)
    b.synchronize()
    return a.elapsed_time(b) / iters


def pmc_traffic(elements):
    """HBM bytes per launch of the update kernel from PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate
    rocprofv3 --pmc passes, tools/pmc_collect.sh): PMC needs the profiler, so the figure is the committed
    measurement in profiles/ for a launch of exactly this size, or None."""
    path = os.path.join(ROOT, 'profiles', 'r01_k1_pmc.json')
    if not os.path.exists(path):
        return None
    for v in json.load(open(path))['kernels'].values():
        if v['elements'] == elements:
            return v['hbm_bytes_per_launch_corrected']
    return None


def roofline_block(sampler, large_n):
    """Dominant kernel = the fused update (k_sgmcmc_step_ctl): identical launch to the one inside
    the timed region (same arena, same control block), timed with HIP events on the stream it is
    launched on. Algorithmic bytes: SGHMC 20 B/param (theta, grad, mom read; theta, mom written) x
    arena elements per launch."""
    opt, arena = sampler.optimizer, sampler.arena
    K = opt.kernels
    stream = torch.cuda.current_stream()
    opt.ctl_begin(True)
    ms = event_time_ms(lambda: K.sgmcmc_step_ctl(arena.theta, arena.grad, arena.mom, opt._ctl), 2048, stream,
                       graph_batch=256)
    bytes_per_launch = 20 * arena.n
    achieved = bytes_per_launch / (ms * 1e-3) / 1e9
    traffic = pmc_traffic(arena.n)
    out = {'bound': 'hbm', 'kernel': 'k_sgmcmc_step_ctl', 'achieved': round(achieved, 1), 'peak': HBM_PEAK_GBPS,
           'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBPS, 4), 'traffic': traffic,
           'bytes_per_launch': bytes_per_launch, 'us_per_launch': round(ms * 1e3, 3),
           'note': 'workload-sized launch (5.5 MB of state, L2/Infinity-Cache resident, one float4 per lane): '
                   'latency-bound; us_per_launch is a 256-launch hipGraph replay / 256 and includes the '
                   '~1.5 us kernel boundary; see roofline_large for the HBM-sized launch of the same arithmetic'}
    # the same arithmetic at a working set beyond the 256 MiB Infinity Cache
    n = large_n
    th, g, m = (torch.randn(n, device='cuda') for _ in range(3))
    sc = dict(lr=HYP['lr'], mu=1 - HYP['alpha'], c_wd=(1 / HYP['prior_std'] ** 2) / N_TRAIN, c_noise=0.3,
              n_train=float(N_TRAIN), seed=1, step=1)
    ms_l = event_time_ms(lambda: K.sgmcmc_step(th, g, m, flags=0x1 | 0x8, **sc), 30, stream)
    ach_l = 20 * n / (ms_l * 1e-3) / 1e9
    large = {'kernel': 'k_sgmcmc_step<mom,philox>', 'elements': n, 'achieved': round(ach_l, 1), 'peak': HBM_PEAK_GBPS,
             'unit': 'GB/s', 'frac': round(ach_l / HBM_PEAK_GBPS, 4), 'frac_of_measured_copy_ceiling':
             round(ach_l / HBM_COPY_GBPS, 4), 'us_per_launch': round(ms_l * 1e3, 2), 'bytes_per_launch': 20 * n,
             'traffic': pmc_traffic(n)}
    del th, g, m
    return out, large


def cpu_baseline_block(steps):
    """Reference CPU path (port): PreResNet-20 forward/backward + per-tensor torch update loop on
    this box's host cores, `steps` minibatch steps of the same workload; extrapolated to
    posterior-samples/s = 1 / (391 x seconds-per-step)."""
    spec = importlib.util.spec_from_file_location('torch_cpu_path', os.path.join(ROOT, 'oracle', 'torch_cpu_path.py'))
    port = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(port)
    from ursabench_amd import models
    torch.manual_seed(0)
    # probed on the GPU box (256 logical CPUs, tools/cpu_threads_probe.py): 8/16/32/64/128 threads ->
    # 76/50/76/222/1119 ms per forward+backward; 16 is the fastest, more threads only add sync cost
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    net = models.PreResNet(CLASSES, 20)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((steps + 3) * BATCH, 3, 32, 32, generator=g)
    y = torch.randint(0, CLASSES, (len(x),), generator=g)
    batches = [(x[i:i + BATCH], y[i:i + BATCH]) for i in range(0, len(x), BATCH)]
    kw = dict(lr=HYP['lr'], momentum=1 - HYP['alpha'], weight_decay=1 / HYP['prior_std'] ** 2,
              num_training_samples=N_TRAIN)
    state = {}
    port.sghmc_epoch(net, batches[:3], state, **kw)                       # warm up
    n, secs = port.sghmc_epoch(net, batches[3:], state, **kw)
    steps_per_sample = (N_TRAIN + BATCH - 1) // BATCH
    threads = torch.get_num_threads()
    torch.set_num_threads(1)                                              # the scalar figure, on a few steps
    n1, secs1 = port.sghmc_epoch(net, batches[3:3 + max(4, steps // 20)], state, **kw)
    torch.set_num_threads(threads)
    cpu_model = 'unknown'
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                cpu_model = ln.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return {'value': round(1.0 / (steps_per_sample * secs / n), 5), 'unit': 'posterior-samples/s',
            'cores': threads, 'kind': 'port', 'cpu_model': cpu_model, 'logical_cpus': os.cpu_count(),
            'value_1_thread': round(1.0 / (steps_per_sample * secs1 / n1), 5),
            'sample': f'{n} minibatch steps (B={BATCH}) of PreResNet-20 SGHMC on torch-CPU '
                      f'({secs:.1f} s, {1e3 * secs / n:.1f} ms/step), extrapolated to {steps_per_sample} steps/sample',
            'ms_per_minibatch_step': round(1e3 * secs / n, 2)}


def multi_chain_block(k, make_chain, inference):
    """Informational (NOT the headline config, which is one chain per GPU): K independent chains on this one
    GPU stepped as K parallel branches of one hipGraph (inference/chain_group.py). One untimed sample per
    chain, then one timed."""
    group = inference.ChainGroup([make_chain(100 + c) for c in range(k)])
    group.sample_iterative()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    group.sample_iterative()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {'chains_per_gpu': k, 'value': round(k / dt, 4), 'unit': 'posterior-samples/s (aggregate over the chains)',
            'ms_per_lockstep_round': round(1e3 * dt / len(group.loader), 3), 'engine': group.stats}


def reference_style_gpu_block(steps, dev):
    """Informational: the REFERENCE'S execution pattern on this same GPU (stock torch ops only, written out
    here — not the product, not the oracle): eager forward/backward, per-parameter-tensor update with 8 small
    ops each (optim_sghmc.py:43-67), `loss.item()` every step (sghmc.py:82) and a `deepcopy(model.cpu())`
    per sample (sghmc.py:99). Extrapolated to posterior-samples/s like the CPU baseline."""
    import copy
    import math
    from ursabench_amd import models
    torch.manual_seed(0)
    net = models.PreResNet(CLASSES, 20).to(dev)
    crit = torch.nn.CrossEntropyLoss()
    params = list(net.parameters())
    g = torch.Generator().manual_seed(0)
    x = torch.randn(8 * BATCH, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, CLASSES, (8 * BATCH,), generator=g).to(dev)
    lr, mu, wd = HYP['lr'], 1 - HYP['alpha'], 1 / HYP['prior_std'] ** 2
    state = {}

    def step(k):
        i = (k % 8) * BATCH
        logits = net(x[i:i + BATCH])
        for p in params:
            p.grad = None
        loss = crit(logits, y[i:i + BATCH])
        loss.backward()
        total = loss.item() * BATCH
        with torch.no_grad():
            for p in params:
                d_p = p.grad.add(p, alpha=wd / N_TRAIN)
                buf = state.get(p)
                if buf is None:
                    buf = torch.clone(d_p).detach()
                buf.mul_(mu).add_(d_p, alpha=-lr)
                d_p = buf.add(torch.randn_like(buf) * math.sqrt(2 * (1 - mu) * lr) / N_TRAIN)
                p.add_(d_p)
                state[p] = d_p
        return total

    net.train()
    for k in range(10):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / steps
    t0 = time.perf_counter()
    snap = copy.deepcopy(net.cpu())
    net.to(dev)
    torch.cuda.synchronize()
    t_snap = time.perf_counter() - t0
    del snap
    per_sample = ((N_TRAIN + BATCH - 1) // BATCH) * t_step + t_snap
    return {'value': round(1.0 / per_sample, 4), 'unit': 'posterior-samples/s', 'kind': 'reference-style eager loop, same GPU',
            'ms_per_minibatch_step': round(1e3 * t_step, 3), 'ms_snapshot_via_cpu': round(1e3 * t_snap, 2),
            'sample': f'{steps} eager minibatch steps, extrapolated'}


def main():
    a = parse()
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != a.gpus:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (no CPU fallback)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 or a.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)      # RCCL over xGMI

    from ursabench_amd import inference, models, tasks, util
    from ursabench_amd.data import synthetic

    train = synthetic(N_TRAIN, (3, 32, 32), CLASSES, seed=0, device=dev, batch_size=BATCH)
    test = synthetic(N_TEST, (3, 32, 32), CLASSES, seed=1, device=dev, batch_size=BATCH)
    hyp = dict(HYP, num_samples=a.steps + a.warmup)
    kpg = max(1, a.chains_per_gpu)

    def make_chain(c):
        util.set_random_seed(c)                                # chain c uses seed c (experiment.py:170)
        return inference.SGHMC(dict(hyp), models.PreResNet(CLASSES, 20).to(dev), train, device=dev,
                               use_graph=not a.no_graph)

    chains = [make_chain(rank * kpg + k) for k in range(kpg)]
    sampler = chains[0]
    group = inference.ChainGroup(chains, use_graph=not a.no_graph) if kpg > 1 else None

    def one_sample():                                          # one posterior sample from every local chain
        return group.sample_iterative() if group is not None else [sampler.sample_iterative()]

    use_dist = dist.is_initialized()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        one_sample()
    barrier()
    t0 = time.perf_counter()
    ensemble = [m for _ in range(a.steps) for m in one_sample()]          # EXACTLY K timed steps (per chain)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    # ---- BMA predictive over the test set: members sharded over ranks, one all-reduce -------------
    pred = tasks.Prediction({'in_distribution_test': test}, CLASSES, dev, 'ALL')
    pred.update_statistics(ensemble[:1], output_performance=False)        # warm up MIOpen eval-mode kernels
    pred.reset()
    barrier()
    t1 = time.perf_counter()
    pred.update_statistics(ensemble, output_performance=False)
    barrier()
    dt_bma = time.perf_counter() - t1
    if use_dist:
        t = torch.tensor([dt_bma], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_bma = t.item()
    metrics = pred.get_performance_metrics()
    members = pred.num_samples_collected

    if rank == 0:
        roof, roof_large = roofline_block(sampler, a.large_n)
        steps_per_sample = len(train)
        line = {
            'metric': f'posterior-samples/sec (PreResNet-20 SGHMC, {kpg} chain{"s" if kpg > 1 else ""} per GPU); '
                      'bma_preds_per_s beside it',
            'value': round(world * kpg * a.steps / dt, 4), 'unit': 'posterior-samples/s', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': round(1e3 * dt / a.steps, 2), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'PreResNet-20 / CIFAR-10-shaped synthetic, SGHMC {kpg} chain(s) per GPU '
                                   '(BASELINE configs[1] at 1 chain, 1 GPU; configs[2] when n_gpus > 1)', 'n_train': N_TRAIN,
                       'n_test': N_TEST, 'batch': BATCH, 'minibatch_steps_per_sample': steps_per_sample,
                       'params': sampler.arena.num_parameters, 'hyper': HYP, 'chains': world * kpg, 'chains_per_gpu': kpg,
                       'hip_graph': not a.no_graph, 'sharding': 'one independent chain per rank; members stay on '
                       'their rank; one RCCL all-reduce of [N*C + N] fp32 for the predictive'},
            'minibatch_steps_per_s': round(world * kpg * a.steps * steps_per_sample / dt, 1),
            'bma_preds_per_s': round(N_TEST / dt_bma, 1), 'bma_members': members,
            'bma_member_forwards_per_s': round(members * N_TEST / dt_bma, 1),
            'bma_nll': round(float(metrics['nll']), 5),
            'engine': group.stats if group is not None else sampler.engine.stats,
            'roofline': roof, 'roofline_large': roof_large,
        }
        if world == 1 and kpg == 1 and a.multi_chain_probe > 1:
            line['multi_chain_per_gpu'] = multi_chain_block(a.multi_chain_probe, make_chain, inference)
        if world == 1 and a.ref_style_steps > 0:
            line['reference_style_gpu'] = reference_style_gpu_block(a.ref_style_steps, dev)
        if world == 1 and not a.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline_block(a.cpu_steps)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
