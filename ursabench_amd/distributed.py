"""One process per GPU. Chains and ensemble members never interact, so sampling needs no
communication at all; the only exchange step on the path is the final predictive average:
one all-reduce(sum) of the concatenated [N*C + N (+ N*C)] fp32 accumulators (RCCL over xGMI on
GPUs — backend "nccl" IS RCCL on ROCm — gloo on CPU in the tests). See tasks/task_base.py
EnsembleAccumulator.reduced()."""
import os

import torch
import torch.distributed as dist


def init_from_env(device_type=None):
    """Read RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run sets them) and join the
    job. Returns (rank, world, device). World size 1 does not initialise a process group."""
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if device_type is None:
        device_type = 'cuda' if torch.cuda.is_available() else 'cpu'
    if device_type == 'cuda':
        torch.cuda.set_device(local)
        device = torch.device('cuda', local)
    else:
        device = torch.device('cpu')
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if device_type == 'cuda':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group('gloo')
    return rank, world, device


def shard(items, rank, world):
    """Members / chains owned by `rank` (round robin: uneven counts are fine, the predictive is a sum)."""
    return list(items)[rank::world]


def chain_seed(base_seed, rank):
    """Chain c uses seed base + c, mirroring URSABench/experiment.py:169-170 (set_random_seed(s) per trial)."""
    return int(base_seed) + int(rank)


def share_swag_moments(sampler, src=0, group=None):
    """BASELINE configs[3] on N GPUs: rank `src` ran the SWAG trajectory (`sampler.run_trajectory()`); every other
    rank receives the two moment vectors (2 x 4 B x n_pad: 2 x 146 MB for WideResNet-28-10) and the collected
    count with RCCL broadcasts over xGMI and marks them final, so each rank can draw its own share of members.
    A no-op without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    dist.broadcast(sampler._mean, src, group=group)
    dist.broadcast(sampler._sq, src, group=group)
    count = sampler.num_models_collected.to(sampler._mean.device)
    dist.broadcast(count, src, group=group)
    sampler.num_models_collected = count.cpu()
    sampler.adopt_moments()


def describe_group(device, all_reduce_bytes, repeats=20, group=None):
    """What the process group actually is, measured from inside it (every rank must call this: it runs collectives):
    backend, world size, the ranks that answered an all-reduce, the device each rank computes on (index, name, PCI bus
    id where the runtime gives one - N ranks must name N different devices), and the duration of the predictive
    all-reduce at its real size (`all_reduce_bytes` of fp32, HIP events on the collective's stream; host clock on CPU
    tensors). Without a process group: world 1, no collective. bench.py puts the result in its line as `rccl`, so the
    first multi-GPU run of the job proves by itself that RCCL saw N ranks on N GPUs."""
    import time
    device = torch.device(device)
    me = {'rank': int(os.environ.get('RANK', 0)), 'local_rank': int(os.environ.get('LOCAL_RANK', 0)), 'device': str(device), 'pid': os.getpid()}
    if device.type == 'cuda':
        p = torch.cuda.get_device_properties(device)
        me.update(name=p.name, pci_bus_id=getattr(p, 'pci_bus_id', None), pci_device_id=getattr(p, 'pci_device_id', None),
                  uuid=str(getattr(p, 'uuid', '')) or None)
    if not (dist.is_available() and dist.is_initialized()):
        return {'backend': None, 'world': 1, 'ranks_seen': [0], 'devices': [me], 'all_reduce_bytes': 0, 'all_reduce_us': None,
                'note': 'single process: no process group, no collective on the path'}
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    seen = torch.zeros(world, device=device)
    seen[rank] = rank + 1.0
    dist.all_reduce(seen, group=group)
    every = [None] * world
    dist.all_gather_object(every, me, group=group)
    buf = torch.ones(max(1, int(all_reduce_bytes) // 4), device=device)
    for _ in range(3):
        dist.all_reduce(buf, group=group)
        buf.fill_(1.0)
    if device.type == 'cuda':
        torch.cuda.synchronize(device)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(repeats):
            dist.all_reduce(buf, group=group)
        b.record()
        b.synchronize()
        us = a.elapsed_time(b) * 1e3 / repeats
    else:
        t0 = time.perf_counter()
        for _ in range(repeats):
            dist.all_reduce(buf, group=group)
        us = (time.perf_counter() - t0) * 1e6 / repeats
    t = torch.tensor([us], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return {'backend': dist.get_backend(group), 'world': world,
            'ranks_seen': [r for r in range(world) if float(seen[r]) == r + 1.0], 'devices': every,
            'distinct_devices': len({(d.get('pci_bus_id'), d.get('uuid'), d.get('device')) for d in every}),
            'all_reduce_bytes': int(buf.numel() * 4), 'all_reduce_us': round(float(t.item()), 2),
            'all_reduce_what': f'sum over {world} ranks of the predictive buffer [N*C + N + count] fp32, max over ranks of the mean of {repeats}'}
