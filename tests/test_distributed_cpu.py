"""CPU, world_size 2, gloo: the N > 1 path. Members are sharded over ranks, every rank reduces its
own members over the full test set, ONE all-reduce produces the predictive on every rank; it must
equal the single-process result over all members."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _members(C=10, d=12, S=5):
    out = []
    for s in range(S):
        g = torch.Generator().manual_seed(100 + s)
        m = torch.nn.Linear(d, C)
        with torch.no_grad():
            m.weight.copy_(torch.randn(C, d, generator=g) * (0.5 + s))
            m.bias.copy_(torch.randn(C, generator=g))
        out.append(m)
    return out


def _loaders():
    from torch.utils.data import DataLoader, TensorDataset
    g = torch.Generator().manual_seed(7)
    x, y = torch.randn(45, 12, generator=g), torch.randint(0, 10, (45,), generator=g)
    xo = torch.randn(23, 12, generator=g) * 3
    mk = lambda a, b: DataLoader(TensorDataset(a, b), batch_size=16, shuffle=False)
    return mk(x, y), mk(xo, torch.zeros(23, dtype=torch.long))


def _evaluate(members, group_ok):
    from ursabench_amd import tasks
    from ursabench_amd.tasks.decision_making import CIFAR10_cost
    from oracle_kernels import OracleKernels
    l_in, l_out = _loaders()
    dev = torch.device('cpu')
    pred = tasks.Prediction({'in_distribution_test': l_in}, 10, dev, 'ALL', kernels=OracleKernels())
    pred.update_statistics(members, output_performance=False)
    ood = tasks.OODDetection({'in_distribution_test': l_in, 'out_distribution_test': l_out}, 10, dev, kernels=OracleKernels())
    om = ood.update_statistics(members, output_performance=True)
    dec = tasks.Decision({'decision_data_test': l_in}, 10, dev, cost_mat=CIFAR10_cost(10), kernels=OracleKernels())
    dm = dec.update_statistics(members, output_performance=True)
    return dict(proba=pred.ensemble_proba.numpy(), ent=pred.expected_data_uncertainty.numpy(),
                count=pred.num_samples_collected, metrics=pred.get_performance_metrics(), ood=om,
                risk=dec.risk.numpy(), decision=dm['Decision'].numpy(), ood_count=ood.num_samples_collected)


def _worker(rank, world, port, outdir, n_members=5):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from ursabench_amd.distributed import init_from_env, shard
    r, w, dev = init_from_env('cpu')
    assert (r, w) == (rank, world) and dist.is_initialized() and dist.get_backend() == 'gloo'
    mine = shard(_members(S=n_members), rank, world)   # round robin: 5 over 2 ranks = 3 + 2; 30 over 4 = 8, 8, 7, 7
    res = _evaluate(mine, True)
    np.savez(os.path.join(outdir, f'rank{rank}.npz'), proba=res['proba'], ent=res['ent'], risk=res['risk'],
             decision=res['decision'], count=res['count'], nll=res['metrics']['nll'],
             auroc=res['ood']['model_uncertainty_auroc'], ood_count=res['ood_count'], n_local=len(mine))
    dist.barrier()
    dist.destroy_process_group()


def test_member_sharded_bma_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ref = _evaluate(_members(), False)              # this process: no process group, all 5 members
    r0, r1 = (np.load(tmp_path / f'rank{r}.npz') for r in (0, 1))
    assert int(r0['n_local']) == 3 and int(r1['n_local']) == 2
    for r in (r0, r1):                              # every rank holds the full predictive
        assert int(r['count']) == 5 and int(r['ood_count']) == 5
        np.testing.assert_allclose(r['proba'], ref['proba'], rtol=2e-6, atol=1e-7)   # summation order differs
        np.testing.assert_allclose(r['ent'], ref['ent'], rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(r['risk'], ref['risk'], rtol=2e-6, atol=1e-6)
        assert np.array_equal(r['decision'], ref['decision'])
        assert float(r['nll']) == pytest.approx(ref['metrics']['nll'], rel=1e-5)
        assert float(r['auroc']) == pytest.approx(ref['ood']['model_uncertainty_auroc'], abs=1e-9)
    assert np.array_equal(r0['proba'], r1['proba'])


@pytest.mark.parametrize('n_members,counts', [(30, [8, 8, 7, 7]), (3, [1, 1, 1, 0])])
def test_world_size_4_uneven_shards_and_an_empty_rank(tmp_path, n_members, counts):
    """BASELINE configs[3]'s split (30 members over the ranks, uneven) at world size 4, and a rank that holds NO
    member: it still calls update_statistics (with an empty list) and so takes part in the one all-reduce; no
    task constructor or reset() communicates, so ranks may build their tasks in any order."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(4, port, str(tmp_path), n_members), nprocs=4, join=True)
    ref = _evaluate(_members(S=n_members), False)
    ranks = [np.load(tmp_path / f'rank{r}.npz') for r in range(4)]
    assert [int(r['n_local']) for r in ranks] == counts
    for r in ranks:
        assert int(r['count']) == n_members and int(r['ood_count']) == n_members
        np.testing.assert_allclose(r['proba'], ref['proba'], rtol=3e-6, atol=1e-7)
        np.testing.assert_allclose(r['ent'], ref['ent'], rtol=3e-6, atol=1e-6)
        np.testing.assert_allclose(r['risk'], ref['risk'], rtol=3e-6, atol=1e-6)
        assert np.array_equal(r['decision'], ref['decision'])
        assert float(r['nll']) == pytest.approx(ref['metrics']['nll'], rel=1e-5)
        assert np.array_equal(r['proba'], ranks[0]['proba'])


def test_task_construction_and_reset_do_not_communicate():
    """collectives hidden in constructors deadlock a job in which one rank builds a task and another
    does not. With a process group initialised (world size 1 here: the collective would still be issued) the
    constructors and reset() must not call all_reduce."""
    import torch.distributed as dist
    from ursabench_amd import tasks
    from ursabench_amd.tasks.decision_making import CIFAR10_cost
    from oracle_kernels import OracleKernels
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1)
    calls = []
    orig = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        l_in, l_out = _loaders()
        dev = torch.device('cpu')
        pred = tasks.Prediction({'in_distribution_test': l_in}, 10, dev, 'ALL', kernels=OracleKernels())
        ood = tasks.OODDetection({'in_distribution_test': l_in, 'out_distribution_test': l_out}, 10, dev, kernels=OracleKernels())
        dec = tasks.Decision({'decision_data_test': l_in}, 10, dev, cost_mat=CIFAR10_cost(10), kernels=OracleKernels())
        for t in (pred, ood, dec):
            t.reset()
        assert calls == []
        pred.update_statistics(_members(S=2), output_performance=False)
        assert len(calls) == 1                          # exactly one all-reduce per update (one accumulator)
        ood.update_statistics(_members(S=2), output_performance=False)
        assert len(calls) == 3                          # in- and out-of-distribution accumulators
    finally:
        dist.all_reduce = orig
        dist.destroy_process_group()


def _swag_worker(rank, world, port, outdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    from ursabench_amd import inference, util
    from ursabench_amd.distributed import init_from_env, share_swag_moments
    from oracle_kernels import OracleKernels
    from test_samplers_cpu import tiny_loader, tiny_net
    init_from_env('cpu')
    hyp = {'swag_lr': 0.01, 'swag_wd': 1e-3, 'lr_init': 0.05, 'num_samples': 2, 'momentum': 0.9, 'burn_in_epochs': 1,
           'num_iterates': 2}
    util.set_random_seed(0)
    s = inference.SWAG(dict(hyp), tiny_net(), tiny_loader(), kernels=OracleKernels(), use_graph=False,
                       reference_quirks=False, seed=100 + rank)
    if rank == 0:
        s.run_trajectory()
    share_swag_moments(s, src=0)
    assert s.burnt_in and int(s.num_models_collected) == 2
    m = s.sample_iterative()                       # a draw from the shared moments with this rank's own noise key
    np.savez(os.path.join(outdir, f'swag{rank}.npz'), mean=s._mean.numpy(), sq=s._sq.numpy(),
             member=torch.cat([p.detach().reshape(-1) for p in m.parameters()]).numpy(), steps=len(s._kernels.step_log))
    dist.barrier()
    dist.destroy_process_group()


def test_swag_moments_are_shared_and_members_differ_per_rank(tmp_path):
    """BASELINE configs[3]'s multi-GPU split: rank 0 runs the trajectory, the moments are broadcast, every rank
    draws its own members (own Philox key) without training."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    mp.spawn(_swag_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (np.load(tmp_path / f'swag{r}.npz') for r in (0, 1))
    assert np.array_equal(r0['mean'], r1['mean']) and np.array_equal(r0['sq'], r1['sq']) and r0['mean'].any()
    assert int(r0['steps']) > 0 and int(r1['steps']) == 0          # only rank 0 trained
    assert not np.array_equal(r0['member'], r1['member'])


def test_shard_and_seed_helpers():
    from ursabench_amd.distributed import chain_seed, shard
    assert shard(range(30), 7, 8) == [7, 15, 23] and sum(len(shard(range(30), r, 8)) for r in range(8)) == 30
    assert [chain_seed(0, r) for r in range(3)] == [0, 1, 2]


def test_c3_partition_check_job_world_size_2_gloo():
    """tools/c3_partition_check.py (the job the GPU suite starts over RCCL on every visible GPU) at world size 2 on CPU:
    the all-reduced predictive equals the one-process sum of the ranks' local accumulators, the count is chains x
    samples, the chains differ, both ranks answered."""
    import json
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    root = os.path.dirname(HERE)
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), os.path.join(root, 'tools', 'c3_partition_check.py'), '--cpu'],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, PYTHONPATH=root))
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert p.returncode == 0 and len(lines) == 1, (p.stdout[-1500:], p.stderr[-1500:])
    line = json.loads(lines[0])
    assert line['pass'] and line['pass_on_every_rank'] and line['world'] == 2 and all(line['ok'].values()), line
    assert line['rccl']['ranks_seen'] == [0, 1] and line['rccl']['backend'] == 'gloo'
