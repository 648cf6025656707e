"""GPU: the harness rows of SURVEY.md §8(f) on the HIP path — the experiment driver's CSV/npy row against the
column golden derived from the reference (G11, URSABench/experiment.py:249-266), time_script's JSON
(time_script.py:114-125), the member-bank checkpoint round trip on the device, MCdropout, and update_hyp on
the members of a ChainGroup."""
import csv
import json
import os

import numpy as np
import pytest
import torch

import ursabench_amd.inference as inference
from ursabench_amd import checkpoint, experiment, models, tasks, time_script, util
from ursabench_amd.data import synthetic

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def flat_params(m):
    return torch.cat([p.detach().reshape(-1) for p in m.parameters()])


def test_experiment_row_matches_the_reference_format(tmp_path, golden_dir):
    gold = json.load(open(os.path.join(golden_dir, 'experiment_columns.json')))
    hyp = {'lr': 0.05, 'prior_std': 0.5, 'num_samples': 2, 'alpha': 0.5, 'burn_in_epochs': 0}
    args = experiment.build_parser().parse_args([
        '--dataset', 'CIFAR10', '--model', 'PreResNet8', '--inference_method', 'SGHMC', '--hyperparams', json.dumps(hyp),
        '--save_path', str(tmp_path) + '/', '--num_trials', '2', '--batch_size', '128', '--train_size', '1024',
        '--test_size', '256'])
    res = experiment.run(args)                                    # default device cuda:0, default (HIP) kernels
    assert sorted(res) == gold['datasets']['CIFAR10']             # the reference's result columns, in its order
    row = next(csv.reader(open(str(tmp_path) + '/results.csv')))
    assert row[:6] == ['CIFAR10', 'PreResNet8', '1', 'SGHMC', 'Prediction', '128'] and len(gold['fixed_columns']) == 6
    assert len(row) == 6 + len(hyp) + len(gold['datasets']['CIFAR10'])
    assert [float(v) for v in row[6:6 + len(hyp)]] == [hyp[k] for k in sorted(hyp)]
    saved = torch.load(str(tmp_path) + '/_tests.npy')
    assert sorted(saved) == gold['datasets']['CIFAR10']
    assert all(np.isfinite(float(v)) for v in saved.values())
    assert 0 <= float(saved['error_rate_mean']) <= 1 and float(saved['nll_mean']) > 0


def test_time_script_json_on_gpu(tmp_path, golden_dir):
    gold = json.load(open(os.path.join(golden_dir, 'experiment_columns.json')))
    p = str(tmp_path / 'timing')
    methods = ['SGLD', 'SGHMC', 'cSGLD', 'cSGHMC', 'SWAG', 'MCdropout', 'SGD']
    assert set(methods) <= set(gold['time_script_methods'])
    class A:  # noqa: E701
        dataset, model, seed, hyperparams_path, batch_size, save_path = 'CIFAR10', 'PreResNet8', 1, None, 128, p
        device_num, samples, trials, train_size, test_size = 0, 2, 2, 512, 128
    A.methods = methods
    out = time_script.run(A)
    assert sorted(out) == sorted(m + s for m in methods for s in ('_mean', '_std'))     # time_script.py:114-115
    assert json.load(open(p + '.json')) == out and all(v >= 0 for v in out.values())


def test_checkpoint_round_trip_on_device(tmp_path):
    util.set_random_seed(2)
    train = synthetic(512, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128)
    test = synthetic(300, (3, 32, 32), 10, seed=1, device=DEV, batch_size=128)
    net = models.PreResNet(10, 8).to(DEV)
    s = inference.SGHMC({'lr': 0.05, 'prior_std': 0.5, 'num_samples': 3, 'alpha': 0.5, 'burn_in_epochs': 0}, net, train, device=DEV)
    ens = s.sample()
    path = str(tmp_path / 'ens.pt')
    checkpoint.save_ensemble(ens, path)
    back = checkpoint.load_ensemble(path, models.PreResNet(10, 8), device=DEV)
    assert len(back) == 3 and all(b._ursa_row.is_cuda for b in back)
    for a, b in zip(ens, back):
        assert torch.equal(a._ursa_row, b._ursa_row)
        for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert ka == kb and torch.equal(va, vb)
    pa = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    pb = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    pa.update_statistics(ens, output_performance=False)
    pb.update_statistics(back, output_performance=False)          # loaded members are bank-resident: twin + graph path
    assert pb._acc.stats['twin_forwards'] == 3 * 1 and pb._acc.stats['eager_forwards'] == 0      # 300 rows: one evaluation batch
    assert torch.equal(pa.ensemble_proba, pb.ensemble_proba)
    assert torch.equal(pa.expected_data_uncertainty, pb.expected_data_uncertainty)
    sd = checkpoint.to_state_dicts(back)
    ref = models.PreResNet(10, 8)
    ref.load_state_dict(sd[0])
    assert torch.equal(flat_params(ref), flat_params(ens[0]).cpu())


def test_mcdropout_on_gpu():
    """FlatSGD trajectory in hipGraph replay with the per-minibatch (lr, momentum) table walked on the device;
    T stochastic forwards of the one live model through Prediction."""
    util.set_random_seed(4)
    train = synthetic(1024, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128)
    test = synthetic(256, (3, 32, 32), 10, seed=1, device=DEV, batch_size=128)
    hyp = {'lr': 0.01, 'epochs': 1, 'dropout': 0.2, 'lengthscale': 0.01, 'num_samples': 3, 'momentum': 0.9, 'weight_decay': 0}
    s = inference.MCdropout(dict(hyp), models.PreResNet(10, 8), train, device=DEV)
    assert type(s.model).__name__ == 'PreResNet_dropout'
    th0 = flat_params(s.model).clone()
    ens = s.sample()
    assert len(ens) == 3 and all(m is s.model for m in ens)
    assert s.engine.stats['graph_replays'] > 0 and s.optimizer._step == 4 * 8
    assert torch.isfinite(flat_params(s.model)).all() and not torch.equal(th0, flat_params(s.model))
    # the control block ended on the schedule's last (lr, momentum): OneCycleLR anneals lr down and momentum back up
    from ursabench_amd._native import StepCtl
    c = StepCtl.from_buffer_copy(bytes(s.optimizer._ctl.cpu().numpy()))
    assert c.step == 32 and 0.85 <= c.mu <= 0.95 + 1e-6
    pred = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    pred.update_statistics(ens, output_performance=False)
    assert pred.num_samples_collected == 3 and pred._acc.stats['eager_forwards'] == 3 * 1
    np.testing.assert_allclose(pred.ensemble_proba.sum(1).numpy(), np.full(256, 3.0, np.float32), rtol=1e-5)
    one = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    one.update_statistics(ens[:1], output_performance=False)
    assert not np.allclose(pred.ensemble_proba.numpy() / 3, one.ensemble_proba.numpy(), rtol=1e-4)    # fresh masks per forward


def test_update_hyp_on_group_members_recaptures_on_gpu():
    """update_hyp rebuilds every chain's optimizer; the group's captured round must be dropped (not
    replayed with the old control block) and the chains must land where the same chains land when run alone."""
    hyp = {'lr': 0.05, 'prior_std': 0.5, 'num_samples': 1, 'alpha': 0.5, 'burn_in_epochs': 0}
    hyp2 = dict(hyp, lr=0.02, alpha=0.3)
    train = synthetic(1024, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128)

    def make(k):
        util.set_random_seed(k)
        return inference.SGHMC(dict(hyp), models.PreResNet(10, 8).to(DEV), train, device=DEV, seed=k)

    def alone(k):
        s = make(k)
        s.sample_iterative()
        util.set_random_seed(10 + k)
        s.update_hyp(dict(hyp2))
        return s.sample_iterative(), s
    ref = [alone(k) for k in range(2)]
    chains = [make(k) for k in range(2)]
    group = inference.ChainGroup(chains)
    group.sample_iterative()
    assert group.stats['captures'] == 1
    ctl = [c.optimizer._ctl.data_ptr() for c in chains]
    for k, c in enumerate(chains):
        util.set_random_seed(10 + k)
        c.update_hyp(dict(hyp2))
    assert [c.optimizer._ctl.data_ptr() for c in chains] == ctl            # no freed address inside any graph
    together = group.sample_iterative()
    assert group.stats['captures'] == 2                                     # stale capture dropped, round re-captured
    for (ma, sa), mb, c in zip(ref, together, chains):
        assert sa.optimizer._step == c.optimizer._step == 16
        np.testing.assert_allclose(flat_params(ma).cpu().numpy(), flat_params(mb).cpu().numpy(), rtol=2e-3, atol=3e-4)
    assert abs(float(chains[0].optimizer.param_groups[0]['momentum']) - 0.7) < 1e-12


def test_chain_checkpoint_resume_on_device(tmp_path):
    """save_chain / load_chain on the HIP path: a PreResNet-8 SGHMC chain saved after two samples and resumed in a fresh
    sampler has the same update counter (= Philox call index), learning rate and BatchNorm counters, and lands where the
    uninterrupted chain lands up to MIOpen's run-to-run rounding (weight-gradient atomics), through graph replay."""
    from ursabench_amd import inference
    hyp = {'lr': 0.05, 'prior_std': 0.5, 'num_samples': 4, 'alpha': 0.5, 'burn_in_epochs': 0}
    train = synthetic(640, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128)

    def make():
        util.set_random_seed(2)
        return inference.SGHMC(dict(hyp), models.PreResNet(10, 8).to(DEV), train, device=DEV, seed=31)
    straight = make()
    want = [flat_params(straight.sample_iterative()).clone() for _ in range(4)]
    first = make()
    for _ in range(2):
        first.sample_iterative()
    p = str(tmp_path / 'chain.pt')
    checkpoint.save_chain(first, p)
    resumed = checkpoint.load_chain(make(), p)
    assert resumed.optimizer._step == first.optimizer._step == 10
    assert resumed.optimizer.param_groups[0]['lr'] == first.optimizer.param_groups[0]['lr']
    assert torch.equal(resumed.arena.theta, first.arena.theta) and torch.equal(resumed.arena.mom, first.arena.mom)
    got = [flat_params(resumed.sample_iterative()) for _ in range(2)]
    assert resumed.engine.stats['graph_replays'] > 0 and resumed.optimizer._step == straight.optimizer._step == 20
    for a, b in zip(want[2:], got):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-3, atol=3e-4)
    assert int(dict(resumed.model.named_buffers())['bn.num_batches_tracked']) == 20
