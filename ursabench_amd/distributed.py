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
