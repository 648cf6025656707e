"""BASELINE configs[2] in miniature, as a job of N processes (one per GPU, RCCL; `--cpu`: gloo + the tests' oracle kernel
set, for the CPU suite): rank c runs chain c (seed c, experiment.py:170) of PreResNet-8 SGHMC, keeps its members, and the
predictive comes out of ONE all-reduce. Self-check, on every rank: the all-reduced accumulators equal the sum - taken in
one process, in rank order - of every rank's LOCAL accumulators (gathered separately), the member count is chains x
samples, every row's probabilities sum to the member count, the chains differ, and the process group really spans N
ranks on N distinct devices (`distributed.describe_group`). Rank 0 prints one JSON line.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/c3_partition_check.py [--cpu]
Started by tests/test_samplers_gpu.py::test_c3_partition_over_rccl_on_every_visible_gpu as a fresh child process (it
must start before anything in it touches a GPU) and by tests/test_distributed_cpu.py at world size 2 on CPU."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cpu', action='store_true')
    ap.add_argument('--samples', type=int, default=2)
    a = ap.parse_args()
    if not a.cpu:
        from ursabench_amd.tuning import use_shipped_miopen_db
        use_shipped_miopen_db('ursa_c3check_miopen_')
    import numpy as np
    import torch
    import torch.distributed as dist
    from ursabench_amd import inference, models, tasks, util
    from ursabench_amd.data import synthetic
    from ursabench_amd.distributed import describe_group, init_from_env
    rank, world, dev = init_from_env('cpu' if a.cpu else 'cuda')
    kw = {}
    if a.cpu:
        from oracle_kernels import OracleKernels
        kw = dict(kernels=OracleKernels())
    n_test, C = 256, 10
    train = synthetic(384, (3, 32, 32), C, seed=0, device=dev, batch_size=128)
    test = synthetic(n_test, (3, 32, 32), C, seed=1, device=dev, batch_size=128)
    util.set_random_seed(rank)
    hyp = {'lr': 0.05, 'prior_std': 0.5, 'num_samples': a.samples, 'alpha': 0.5, 'burn_in_epochs': 0}
    s = inference.SGHMC(hyp, models.PreResNet(C, 8).to(dev), train, device=dev, seed=rank, **({'use_graph': False, **kw} if a.cpu else {}))
    ens = s.sample()
    local = tasks.Prediction({'in_distribution_test': test}, C, dev, 'ALL', process_group=False, **kw)   # this rank's members only
    local.update_statistics(ens, output_performance=False)
    pred = tasks.Prediction({'in_distribution_test': test}, C, dev, 'ALL', **kw)
    pred.update_statistics(ens, output_performance=False)                                                   # the one all-reduce
    info = describe_group(dev, 4 * (n_test * C + n_test + 1))
    mine = torch.cat([local.ensemble_proba.reshape(-1), local.expected_data_uncertainty]).to(dev)
    every = [torch.empty_like(mine) for _ in range(world)] if world > 1 else [mine]
    if world > 1:
        dist.all_gather(every, mine)
    total = torch.zeros_like(mine, device='cpu')
    for t in every:                                        # one process, rank order
        total += t.cpu()
    got = torch.cat([pred.ensemble_proba.reshape(-1), pred.expected_data_uncertainty])
    ok = {'count': pred.num_samples_collected == world * a.samples and local.num_samples_collected == a.samples,
          'sum_equals_all_reduce': bool(np.allclose(got.numpy(), total.numpy(), rtol=2e-6, atol=1e-6)),
          'rows_sum_to_members': bool(np.allclose(pred.ensemble_proba.sum(1).numpy(), world * a.samples, rtol=2e-6)),
          'chains_differ': world == 1 or not any(torch.equal(every[0].cpu(), t.cpu()) for t in every[1:]),
          'ranks_seen': info['ranks_seen'] == list(range(world)),
          'distinct_devices': a.cpu or world == 1 or info['distinct_devices'] == world}
    line = {'check': 'c3_partition', 'world': world, 'backend': info['backend'], 'ok': ok, 'pass': all(ok.values()), 'rccl': info,
            'max_abs_diff': float((got - total).abs().max()), 'engine': dict(s.engine.stats)}
    verdicts = [None] * world
    if world > 1:
        dist.all_gather_object(verdicts, line['pass'])
        dist.barrier()
        dist.destroy_process_group()
    else:
        verdicts = [line['pass']]
    line['pass_on_every_rank'] = all(verdicts)
    if rank == 0:
        print(json.dumps(line))
    sys.exit(0 if all(verdicts) else 1)


if __name__ == '__main__':
    main()
