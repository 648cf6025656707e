"""GPU: samplers and tasks end to end on the HIP path (C-ABI kernels + hipGraph chain engine)."""
import json
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

import oracle_lib as O
import ursabench_amd.inference as inference
from ursabench_amd import models, tasks
from ursabench_amd.data import synthetic
from test_tasks_cpu import check_tasks

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def flat_params(m):
    return torch.cat([p.detach().reshape(-1) for p in m.parameters()])


@pytest.mark.parametrize('name', ['SGLD', 'SGHMC'])
def test_end_to_end_lenet5_vs_reference_on_gpu(golden_dir, name):
    """The reference's CPU run (LeNet-5, 6 minibatch steps, 2 samples) replayed on the GPU with the
    captured noise. Forward/backward are MIOpen/rocBLAS instead of oneDNN, so parameters agree to
    rounding amplified over 6 steps (1e-4 relative, 2e-5 absolute here); the predictive probabilities — north_star's
    criterion — agree within 1e-5 relative."""
    g = np.load(os.path.join(golden_dir, 'e2e_lenet5.npz'))
    hyp = json.loads(str(g[f'{name}/hyper']))
    train = DataLoader(TensorDataset(torch.tensor(g['x_train']), torch.tensor(g['y_train'])), batch_size=32)
    test = DataLoader(TensorDataset(torch.tensor(g['x_test']), torch.tensor(g['y_test'])), batch_size=16)
    net = models.LeNet5(10)
    with torch.no_grad():
        off = 0
        for p in net.parameters():
            p.copy_(torch.tensor(g[f'{name}/theta0'][off:off + p.numel()]).view_as(p))
            off += p.numel()
    s = getattr(inference, name)(dict(hyp), net, train, device=DEV)

    def eps(k):
        e = torch.zeros(s.arena.n, device=DEV)
        e[s.arena.layout.gather_index(DEV)] = torch.tensor(g[f'{name}/eps'][k], device=DEV)
        return e
    s.eps_provider = eps
    ens = s.sample()
    for m, ref in zip(ens, g[f'{name}/samples']):
        np.testing.assert_allclose(flat_params(m).cpu().numpy(), ref, rtol=1e-4, atol=2e-5)
    pred = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    pred.update_statistics(ens, output_performance=False)
    np.testing.assert_allclose(pred.ensemble_proba.numpy(), g[f'{name}/proba_sum'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(pred.expected_data_uncertainty.numpy(), g[f'{name}/ent_sum'], rtol=1e-5, atol=1e-6)
    gold = json.loads(str(g[f'{name}/metrics']))
    got = pred.get_performance_metrics()
    for k in ('error_rate', 'nll', 'brier_score', 'ece'):
        assert got[k] == pytest.approx(gold[k], rel=1e-4, abs=1e-6), k


def _run_chain(use_graph, cls=None, hyp=None, seed=5, depth=8, n=1024 + 40):
    torch.manual_seed(seed)
    train = synthetic(n, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128)      # ragged tail: 40
    net = models.PreResNet(10, depth).to(DEV)
    hyp = hyp or {'lr': 0.05, 'prior_std': 0.5, 'num_samples': 2, 'alpha': 0.5, 'burn_in_epochs': 1}
    s = (cls or inference.SGHMC)(dict(hyp), net, train, device=DEV, use_graph=use_graph)
    return s, s.sample()


def test_graph_replay_equals_eager_launches():
    """hipGraph replay is the same computation as eager launches: same seeds -> same posterior samples
    up to the chaotic amplification of rounding differences over 27 noisy SGHMC steps (MIOpen's solver
    choice during the eager warm-up steps vs the captured graph is not bit-reproducible)."""
    torch.backends.cudnn.deterministic = True        # MIOpen: no atomics-based weight-gradient solvers
    try:
        sg, eg = _run_chain(True)
        se, ee = _run_chain(False)
    finally:
        torch.backends.cudnn.deterministic = False
    assert sg.engine.stats['graph_replays'] > 0 and sg.engine.stats['captures'] == 1
    assert se.engine.stats['graph_replays'] == 0
    assert sg.optimizer._step == se.optimizer._step == 3 * 9
    for a, b in zip(eg, ee):
        np.testing.assert_allclose(flat_params(a).cpu().numpy(), flat_params(b).cpu().numpy(), rtol=2e-3, atol=3e-4)
        for (ka, ba), (kb, bb) in zip(a.named_buffers(), b.named_buffers()):
            assert ka == kb
            np.testing.assert_allclose(ba.float().cpu().numpy(), bb.float().cpu().numpy(), rtol=2e-3, atol=3e-4)
    # members carry their own BN statistics and are distinct snapshots
    assert any(k.endswith('running_mean') for k, _ in eg[0].named_buffers())
    assert int(dict(eg[1].named_buffers())['bn.num_batches_tracked']) == 27
    assert int(dict(eg[0].named_buffers())['bn.num_batches_tracked']) == 18
    assert not torch.equal(flat_params(eg[0]), flat_params(eg[1]))


def test_update_matches_oracle_inside_the_training_loop():
    """Take the gradients a real backward pass produced on the GPU and check the fused update that
    the engine applied against the oracle, bit for bit (Philox mode, fused zero-grad)."""
    torch.manual_seed(1)
    train = synthetic(256, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128)
    net = models.PreResNet(10, 8).to(DEV)
    s = inference.SGHMC({'lr': 0.1, 'prior_std': 0.5, 'num_samples': 1, 'alpha': 0.5, 'burn_in_epochs': 0},
                        net, train, device=DEV, use_graph=False)
    a = s.arena
    th0 = a.theta.cpu().numpy().copy()
    x, y = next(iter(train))
    s.model.train()
    s.loss_criterion(s.model(x), y).backward()
    g = a.grad.cpu().numpy().copy()
    s.optimizer.ctl_begin(True)
    s.optimizer.ctl_step()
    s.optimizer.ctl_end(1)
    mom = np.zeros_like(th0)
    O.sgmcmc_step(th0, g, mom, flags=O.STEP_NOISE | O.STEP_WD | O.STEP_FIRST | O.STEP_ZERO_GRAD,
                  seed=s.optimizer.seed, step=0, **O.step_scalars(0.1, 0.5, 1 / 0.5 ** 2, 256))
    assert np.array_equal(a.theta.cpu().numpy(), th0) and np.array_equal(a.mom.cpu().numpy(), mom)
    assert not a.grad.any()
    assert torch.equal(s.optimizer.state[next(net.parameters())]['momentum_buffer'].reshape(-1),
                       a.mom[:next(net.parameters()).numel()])


def test_csghmc_on_gpu_walks_the_device_schedule():
    hyp = {'lr_0': 0.05, 'prior_std': 1.0, 'num_samples_per_cycle': 1, 'cycle_length': 3, 'burn_in_epochs': 1,
           'num_cycles': 1, 'alpha': 0.3}
    s, ens = _run_chain(True, inference.cSGHMC, hyp)
    assert len(ens) == 1 and s.epochs_run == 3 and s.engine.stats['graph_replays'] > 0
    from ursabench_amd._native import StepCtl
    c = StepCtl.from_buffer_copy(bytes(s.optimizer._ctl.cpu().numpy()))
    assert c.step == 27 and np.float32(s.lr) == np.float32(s._epoch_table()[-1, 0])
    assert np.isfinite(flat_params(ens[0]).cpu().numpy()).all()


@pytest.mark.parametrize('tag', ['c10', 'c100', 'mnist'])
def test_tasks_vs_reference_on_gpu(golden_dir, tag):
    check_tasks(golden_dir, tag, DEV, None)           # None -> the product HIP kernel set


def test_bma_full_size_properties():
    """10,000-row test set, PreResNet-20 members: an ensemble of S copies of one member has the
    member's own softmax as its mean (idempotence), every row of proba_sum sums to S, and the result
    does not depend on how the members are split over update_statistics calls."""
    torch.manual_seed(0)
    test = synthetic(10000, (3, 32, 32), 10, seed=1, device=DEV, batch_size=128)
    m1, m2 = models.PreResNet(10, 20).to(DEV), models.PreResNet(10, 20).to(DEV)
    pred = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    pred.update_statistics([m1, m1, m1], output_performance=False)
    with torch.no_grad():
        p1 = torch.cat([torch.softmax(m1.eval()(x), -1) for x, _ in test]).cpu()
    np.testing.assert_allclose((pred.ensemble_proba / 3).numpy(), p1.numpy(), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(pred.ensemble_proba.sum(1).numpy(), np.full(10000, 3.0, np.float32), rtol=1e-6)
    a = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    a.update_statistics([m1, m2], output_performance=False)
    b = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    b.update_statistics(m1, output_performance=False)
    b.update_statistics([m2], output_performance=False)
    assert torch.equal(a.ensemble_proba, b.ensemble_proba) and a.num_samples_collected == b.num_samples_collected == 2
    assert torch.equal(a.expected_data_uncertainty, b.expected_data_uncertainty)
