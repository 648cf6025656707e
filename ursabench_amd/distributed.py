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
