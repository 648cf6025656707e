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

# G9 = the reference's PreResNet-8 (BatchNorm + ReLU) SGHMC run, 4 noisy minibatch steps. MIOpen's and oneDNN's
# convolutions differ by ~1e-6, so a pre-activation within that of zero opens its ReLU gate on one device and not on the
# other - for any BatchNorm arithmetic - and every such gate moves the gradients it feeds. The fixture lists the ~750
# pre-activations per step the reference computed within 1e-4 of zero and the gate it took there; the replays below hand
# them to the backward launch (fused_bn.GateProbe -> ursa_bn_relu_bwd_gated_f32), after which GPU and CPU evaluate the
# same piecewise-linear function and north_star's 1e-5 holds outright (measured 6e-7). The natural runs (no gates
# given), eight seeds, K6 vs MIOpen's BatchNorm launches paired: tests/test_gate_parity_gpu.py.
def _give_reference_gates(samplers, g):
    """Install a forcing GateProbe on every sampler's engine, fed with the G9 fixture's lists."""
    from gate_lists import unpack
    from ursabench_amd import fused_bn
    lists = unpack(g)
    cap = int(g['gate_counts'].max())
    for s in samplers:
        s.engine.gate_probe = fused_bn.GateProbe(len(lists[0]), cap, DEV, force=True)
        s.gate_provider = lambda k, _l=lists: _l[k]


def _assert_reference_gates_were_the_only_difference(s, g):
    """Nothing outside the listed band changed sides: open-gate counts equal the reference's once the listed ones are."""
    hist = s.engine.gate_probe.history
    assert len(hist) == len(g['n_open'])
    for h, ref in zip(hist, g['n_open']):
        assert h['n_open_as_reference'] == ref.tolist()


def flat_params(m):
    return torch.cat([p.detach().reshape(-1) for p in m.parameters()])


def _replay_mode(s, use_graph):
    """use_graph=True: ONE eager warm-up step (MIOpen's solver search cannot run inside a capture), then every
    full-size batch is a hipGraph replay that reads the injected noise from the engine's persistent buffer — the
    execution path bench.py times."""
    if use_graph:
        s.engine.WARMUP_STEPS = 1


@pytest.mark.parametrize('use_graph', [False, True])
@pytest.mark.parametrize('name', ['SGLD', 'SGHMC'])
def test_end_to_end_lenet5_vs_reference_on_gpu(golden_dir, name, use_graph):
    """The reference's CPU run (LeNet-5, 6 minibatch steps, 2 samples) replayed on the GPU with the
    captured noise — with eager launches and through hipGraph replay (the timed path). Forward/backward are
    MIOpen/rocBLAS instead of oneDNN, so parameters agree to
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
    s = getattr(inference, name)(dict(hyp), net, train, device=DEV, use_graph=use_graph)
    _replay_mode(s, use_graph)

    def eps(k):
        e = torch.zeros(s.arena.n, device=DEV)
        e[s.arena.layout.gather_index(DEV)] = torch.tensor(g[f'{name}/eps'][k], device=DEV)
        return e
    s.eps_provider = eps
    ens = s.sample()
    assert (s.engine.stats['graph_replays'] >= 4) if use_graph else (s.engine.stats['graph_replays'] == 0), s.engine.stats
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
    sg, eg = _run_chain(True)
    se, ee = _run_chain(False)
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
    O.sgmcmc_step(th0, g.copy(), mom, flags=O.STEP_NOISE | O.STEP_WD | O.STEP_FIRST,
                  seed=s.optimizer.seed, step=0, **O.step_scalars(0.1, 0.5, 1 / 0.5 ** 2, 256))
    assert np.array_equal(a.theta.cpu().numpy(), th0) and np.array_equal(a.mom.cpu().numpy(), mom)
    assert np.array_equal(a.grad.cpu().numpy(), g)      # the engine packs fresh gradients each step: no re-zeroing
    assert torch.equal(s.optimizer.state[next(net.parameters())]['momentum_buffer'].reshape(-1),
                       a.mom[:next(net.parameters()).numel()])


class _SpyLoader:
    """Wraps a loader and calls `on_next()` before every batch is handed out and once after the last one: between
    two calls exactly one minibatch step of the engine (eager or one hipGraph replay) has been enqueued."""

    def __init__(self, loader, on_next):
        self.loader, self.on_next = loader, on_next
        self.dataset, self.batch_size = loader.dataset, loader.batch_size

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for b in self.loader:
            self.on_next()
            yield b
        self.on_next()


def _check_steps_against_oracle(recs, seed, lr, mu, wd, n_train, first_step=0):
    """recs[k] = (theta, mom, grad) on the host BEFORE step k (grad = what step k-1 consumed: the engine packs fresh
    gradients every step and does not re-zero them). Step k, recomputed by the oracle from the GPU's own gradients,
    must reproduce the GPU's theta and momentum bit for bit. Returns the number of steps checked."""
    sc = O.step_scalars(lr, mu, wd, n_train)
    for k in range(len(recs) - 1):
        th, mo = recs[k][0].copy(), recs[k][1].copy()
        flags = O.STEP_NOISE | O.STEP_WD | (O.STEP_FIRST if first_step + k == 0 else 0)
        O.sgmcmc_step(th, recs[k + 1][2].copy(), mo if mu else None, flags=flags, seed=seed, step=first_step + k, **sc)
        assert np.array_equal(th, recs[k + 1][0]), f'theta differs from the oracle at step {first_step + k}'
        if mu:
            assert np.array_equal(mo, recs[k + 1][1]), f'momentum differs from the oracle at step {first_step + k}'
    return len(recs) - 1


@pytest.mark.parametrize('size', ['small', 'c2_full_sample'])
def test_replayed_steps_equal_oracle_given_the_gpus_gradients(size):
    """The path bench.py times — hipGraph replays of {forward, backward, pack, k_sgmcmc_step_ctl with Philox noise and
    the folded control-block advance} — checked against the oracle step by step: before every step theta / momentum
    are copied out, after it the gradients the step consumed; the oracle's update from those inputs (same Philox key
    and call index) must equal the GPU's bit for bit. `c2_full_sample` is BASELINE configs[1] at full size: all 391
    minibatch steps of one PreResNet-20 posterior sample (387 of them replays) — the implementation adds nothing to the
    MIOpen-vs-oneDNN gradient differences (URSABench/inference/sghmc.py:72-87 + optim_sghmc.py:43-67)."""
    from ursabench_amd import util
    n_rows, depth = (50000, 20) if size == 'c2_full_sample' else (1152 + 40, 8)
    util.set_random_seed(3)
    recs = []
    holder = {}

    def on_next():
        a = holder['s'].arena
        recs.append(tuple(t.cpu().numpy().copy() for t in (a.theta, a.mom if a.mom is not None else a.theta, a.grad)))
    train = _SpyLoader(synthetic(n_rows, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128), on_next)
    hyp = {'lr': 0.1, 'prior_std': 0.5, 'num_samples': 3, 'alpha': 0.5, 'burn_in_epochs': 0}
    s = holder['s'] = inference.SGHMC(dict(hyp), models.PreResNet(10, depth).to(DEV), train, device=DEV, seed=77)
    s.arena.ensure_mom()
    s.sample_iterative()
    steps = _check_steps_against_oracle(recs, 77, 0.1, 0.5, 1 / 0.5 ** 2, n_rows)
    st = s.engine.stats
    assert steps == len(train) and st['graph_replays'] + st['eager_steps'] == steps
    assert st['graph_replays'] == (n_rows // 128) - 3 and st['captures'] == 1, st     # all full-size batches after 3 warm-up steps
    # second sample: CosineAnnealingLR moved lr (0.1 -> 0.075), the Philox call index continues, FIRST is gone
    recs.clear()
    lr2 = s.optimizer.param_groups[0]['lr']
    assert 0 < lr2 < 0.1
    s.sample_iterative()
    _check_steps_against_oracle(recs, 77, lr2, 0.5, 1 / 0.5 ** 2, n_rows, first_step=steps)
    assert s.engine.stats['captures'] == 1


def test_chain_group_replayed_steps_equal_oracle_for_every_chain():
    """Same check for every branch of a ChainGroup of 3: ONE ursa_sgmcmc_step_multi_f32 launch per lock-step round
    over the [3, n] slabs (asserted: update launches == rounds), each chain bit-identical to the oracle run on that
    chain's own gradients, Philox key and call index."""
    from ursabench_amd import util
    recs = [[], [], []]
    holder = {}

    def on_next():
        for k, s in enumerate(holder['g'].samplers):
            a = s.arena
            recs[k].append(tuple(t.cpu().numpy().copy() for t in (a.theta, a.mom, a.grad)))
    train = _SpyLoader(synthetic(1152 + 40, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128), on_next)
    hyp = {'lr': 0.05, 'prior_std': 0.5, 'num_samples': 1, 'alpha': 0.5, 'burn_in_epochs': 0}

    def make(k):
        util.set_random_seed(k)
        return inference.SGHMC(dict(hyp), models.PreResNet(10, 8).to(DEV), train, device=DEV, seed=40 + k)
    group = holder['g'] = inference.ChainGroup([make(k) for k in range(3)])
    assert all(s.arena.theta.data_ptr() == group.theta[k].data_ptr() for k, s in enumerate(group.samplers))
    group.sample_iterative()
    for k in range(3):
        assert _check_steps_against_oracle(recs[k], 40 + k, 0.05, 0.5, 1 / 0.5 ** 2, 1152 + 40) == 10
    st = group.stats
    assert st['graph_replays'] == 6 and st['eager_rounds'] == 4 and st['captures'] == 1, st      # 3 warm-up + the ragged tail
    assert st['update_launches'] == 10                                                           # one per round, not K
    assert not np.array_equal(recs[0][-1][0], recs[1][-1][0])


def test_chain_group_replays_reference_run_with_injected_noise(golden_dir):
    """G9 through the group path: two chains of a group, both fed the reference's captured noise from the same
    initial weights, each reproduce the reference's PreResNet-8 SGHMC run (predictive within 1e-5) through graph
    replays whose multi-chain update reads the [K, n] injected-noise slab."""
    from test_samplers_cpu import _load_preresnet8, _preresnet8_inputs
    g = np.load(os.path.join(golden_dir, 'e2e_preresnet8.npz'))
    hyp = json.loads(str(g['hyper']))
    train, test = _preresnet8_inputs(g)
    chains = [inference.SGHMC(dict(hyp), _load_preresnet8(g), train, device=DEV, seed=k) for k in range(2)]
    group = inference.ChainGroup(chains)
    group.WARMUP_STEPS = 1

    def provider(s):
        def eps(k):
            e = torch.zeros(s.arena.n, device=DEV)
            e[s.arena.layout.gather_index(DEV)] = torch.tensor(g['eps'][k], device=DEV)
            return e
        return eps
    for s in chains:
        s.eps_provider = provider(s)
    _give_reference_gates(chains, g)
    per_chain = group.sample()
    assert group.stats['graph_replays'] >= 2
    for s, ens in zip(chains, per_chain):
        _assert_reference_gates_were_the_only_difference(s, g)
        for m, ref in zip(ens, g['samples']):
            np.testing.assert_allclose(flat_params(m).cpu().numpy(), ref, rtol=1e-5, atol=2e-6)
        pred = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
        pred.update_statistics(ens, output_performance=False)
        np.testing.assert_allclose(pred.ensemble_proba.numpy(), g['proba_sum'], rtol=1e-5, atol=1e-7)


def test_csghmc_on_gpu_walks_the_device_schedule():
    hyp = {'lr_0': 0.05, 'prior_std': 1.0, 'num_samples_per_cycle': 1, 'cycle_length': 3, 'burn_in_epochs': 1,
           'num_cycles': 1, 'alpha': 0.3}
    s, ens = _run_chain(True, inference.cSGHMC, hyp)
    assert len(ens) == 1 and s.epochs_run == 3 and s.engine.stats['graph_replays'] > 0
    from ursabench_amd._native import StepCtl
    c = StepCtl.from_buffer_copy(bytes(s.optimizer._ctl.cpu().numpy()))
    assert c.step == 27 and c.tickets_clear()
    want = s._adjust_learning_rate(s.optimizer, s.epochs_run - 1, len(s.train_loader) - 1)   # last iteration run
    assert s.lr == pytest.approx(want) and s.optimizer.param_groups[0]['lr'] == pytest.approx(want)
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


def test_swag_on_gpu_both_modes():
    hyp = {'swag_lr': 0.01, 'swag_wd': 1e-4, 'lr_init': 0.05, 'num_samples': 3, 'momentum': 0.9,
           'burn_in_epochs': 1, 'num_iterates': 2}
    torch.manual_seed(0)
    train = synthetic(512 + 40, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128)
    # bug-compatible (reference) mode: every "sample" is the last SGD iterate + a BN refresh
    s = inference.SWAG(dict(hyp), models.PreResNet(10, 8).to(DEV), train, device=DEV)
    ens = s.sample()
    live = s.arena.flatten()
    assert s.num_models_collected.item() == 0 and s.epochs_run == 3 and s.engine.stats['graph_replays'] > 0
    assert all(torch.equal(flat_params(m), live) for m in ens)
    assert torch.equal(s.weight_mean, live)
    np.testing.assert_allclose(s.sq_mean.cpu().numpy(), (live * live).cpu().numpy(), rtol=1e-6)
    bn = dict(ens[0].named_buffers())
    assert float(bn['bn.running_var'].min()) > 0 and not torch.equal(bn['bn.running_mean'], torch.zeros_like(bn['bn.running_mean']))
    # corrected mode: real moments, distinct draws that follow our Philox stream through K3
    s2 = inference.SWAG(dict(hyp), models.PreResNet(10, 8).to(DEV), train, device=DEV, reference_quirks=False, seed=9)
    ens2 = s2.sample()
    assert s2.num_models_collected.item() == 2
    assert not torch.equal(flat_params(ens2[0]), flat_params(ens2[1]))
    mean, var = s2._get_mean_and_variance()
    idx = s2.arena.layout.gather_index('cpu').numpy()
    for d, m in enumerate(ens2):
        eps = O.philox_normal(s2.arena.n, 9, d)[idx]
        np.testing.assert_allclose(flat_params(m).cpu().numpy(), eps * np.sqrt(var.cpu().numpy()) + mean.cpu().numpy(),
                                   rtol=1e-6, atol=1e-7)


def test_flat_sgd_equals_torch_sgd_on_gpu():
    torch.manual_seed(0)
    a, b = models.PreResNet(10, 8).to(DEV), models.PreResNet(10, 8).to(DEV)
    b.load_state_dict(a.state_dict())
    oa = inference.FlatSGD(a.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    ob = torch.optim.SGD(b.parameters(), lr=0.05, momentum=0.9, weight_decay=5e-4)
    x = torch.randn(64, 3, 32, 32, device=DEV)
    y = torch.randint(0, 10, (64,), device=DEV)
    for k in range(3):
        for net, opt in ((a, oa), (b, ob)):
            opt.zero_grad()
            torch.nn.functional.cross_entropy(net(x), y).backward()
            opt.step()
    # MIOpen's weight-gradient kernels use float atomics (run-to-run rounding differences), hence not bitwise
    np.testing.assert_allclose(flat_params(a).cpu().numpy(), flat_params(b).cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_hmc_on_gpu_small():
    torch.manual_seed(0)
    train = synthetic(256, (1, 28, 28), 10, seed=0, device=DEV, batch_size=128)
    s = inference.HMC({'step_size': 1e-4, 'num_samples': 3, 'L': 2, 'tau': 1.0, 'burn': 0, 'mass': 1.0},
                      models.LeNet5(10).to(DEV), train, device=DEV, seed=4)
    out = s.sample()
    assert len(out) == 4 and s.accepted == 3 and s.x.shape[0] == 256
    th = [flat_params(m) for m in out]
    assert all(torch.isfinite(t).all() for t in th) and not torch.equal(th[0], th[-1])


def test_compute_val_loss_matches_torch():
    torch.manual_seed(0)
    train = synthetic(256, (1, 28, 28), 10, seed=0, device=DEV, batch_size=128)
    val = synthetic(200, (1, 28, 28), 10, seed=2, device=DEV, batch_size=64)
    s = inference.SGLD({'lr': 0.01, 'prior_std': 1.0, 'num_samples': 1, 'alpha': 1.0, 'burn_in_epochs': 0},
                       models.LeNet5(10).to(DEV), train, device=DEV)
    got = s.compute_val_loss(val)
    with torch.no_grad():
        ref = torch.nn.functional.cross_entropy(s.model(val.dataset.x), val.dataset.y).item()
    assert got == pytest.approx(ref, rel=1e-5)


def test_bma_graph_replay_equals_eager_member_forwards(monkeypatch):
    """Bank-resident members are evaluated through one hipGraph-captured twin; foreign modules (here:
    deep copies, which lose the bank handle) eagerly. Same accumulators either way."""
    import copy
    from ursabench_amd.tasks.task_base import EnsembleAccumulator
    monkeypatch.setattr(EnsembleAccumulator, 'EVAL_ROWS_SMALL', 1024)   # (small networks merge up to 4,096 rows: two batch shapes wanted here)
    torch.manual_seed(0)
    train = synthetic(512, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128)
    test = synthetic(1500, (3, 32, 32), 10, seed=1, device=DEV, batch_size=128)     # evaluation batches: 1024 + 476 rows
    s = inference.SGHMC({'lr': 0.05, 'prior_std': 0.5, 'num_samples': 3, 'alpha': 0.5, 'burn_in_epochs': 0},
                        models.PreResNet(10, 8).to(DEV), train, device=DEV)
    ens = s.sample()
    a = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    a.update_statistics(ens, output_performance=False)
    assert len(a._acc._twins) == 1 and len(next(iter(a._acc._twins.values()))['runners']) == 2
    foreign = []
    for m in ens:
        c = copy.deepcopy(m)
        del c._ursa_bank
        foreign.append(c)
    b = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    b.update_statistics(foreign, output_performance=False)
    assert len(b._acc._twins) == 0
    np.testing.assert_allclose(a.ensemble_proba.numpy(), b.ensemble_proba.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(a.expected_data_uncertainty.numpy(), b.expected_data_uncertainty.numpy(), rtol=1e-5, atol=1e-6)
    # the twin must not alias a member: evaluating leaves every member's weights untouched
    w = [m._ursa_row.clone() for m in ens]
    a.update_statistics(ens[:2], output_performance=False)
    assert all(torch.equal(x, m._ursa_row) for x, m in zip(w, ens))
    # 3 members on 4 lanes, then 6 members (two lane groups, second one partial)
    c = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    c.update_statistics(ens + ens, output_performance=False)
    np.testing.assert_allclose(c.ensemble_proba.numpy(), 2 * b.ensemble_proba.numpy(), rtol=1e-5, atol=1e-7)


def _rccl_worker(rank, world, port, q):
    try:
        _rccl_worker_body(rank, world, port, q)
    except BaseException as exc:          # surface the failure instead of letting the parent time out
        import traceback
        q.put(('error', traceback.format_exc()))
        raise


def _rccl_worker_body(rank, world, port, q):
    import os
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0', MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from ursabench_amd import inference, models, tasks
    from ursabench_amd.data import synthetic
    from ursabench_amd.distributed import init_from_env
    r, w, dev = init_from_env('cuda')                 # world size 1: leaves the process group alone
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    assert dist.is_initialized() and dist.get_backend() == 'nccl'
    torch.manual_seed(0)
    train = synthetic(640, (3, 32, 32), 10, seed=0, device=dev, batch_size=128)
    test = synthetic(300, (3, 32, 32), 10, seed=1, device=dev, batch_size=128)
    s = inference.SGHMC({'lr': 0.05, 'prior_std': 0.5, 'num_samples': 2, 'alpha': 0.5, 'burn_in_epochs': 0},
                        models.PreResNet(10, 8).to(dev), train, device=dev)      # hipGraph capture with RCCL alive
    ens = s.sample()
    pred = tasks.Prediction({'in_distribution_test': test}, 10, dev, 'ALL')
    pred.update_statistics(ens, output_performance=False)                        # all-reduce over RCCL
    q.put((pred.num_samples_collected, float(pred.ensemble_proba.sum()), s.engine.stats['graph_replays']))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_process_group_with_graph_capture_world_size_1():
    """One process per GPU over RCCL (backend 'nccl'): on a 1-GPU box the world is 1, which still exercises
    process-group init, hipGraph capture beside RCCL's watchdog thread, and the predictive all-reduce."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(0, 1, port, q))
    p.start()
    res = q.get(timeout=180)
    p.join(timeout=60)
    assert res[0] != 'error', res[1]
    count, total, replays = res
    assert p.exitcode == 0 and count == 2 and replays > 0
    assert total == pytest.approx(2 * 300, rel=1e-5)         # every member's probabilities sum to 1 per row


def test_c3_partition_over_rccl_on_every_visible_gpu():
    """BASELINE configs[2] in miniature over REAL RCCL: tools/c3_partition_check.py as a torch.distributed.run job with
    one rank per visible GPU (at most 4: the box admits 6 GPU processes), started as a fresh child process so that
    nothing in it has touched a GPU before its own ranks do. Each rank checks that the RCCL all-reduce of the predictive
    equals the one-process sum of every rank's local accumulators, that N ranks answered on N distinct devices, and
    that the chains differ. Skips on a 1-GPU box (world size 1 over RCCL is test_rccl_process_group_..._world_size_1)."""
    import socket
    import subprocess
    import sys
    n = min(torch.cuda.device_count(), 4)
    if n < 2:
        pytest.skip(f'{torch.cuda.device_count()} GPU visible: the N > 1 RCCL job needs at least 2')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, PYTHONPATH=root, HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr',
                        '127.0.0.1', '--master-port', str(port), os.path.join(root, 'tools', 'c3_partition_check.py')],
                       capture_output=True, text=True, timeout=900, env=env)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert p.returncode == 0 and len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    line = json.loads(lines[0])
    assert line['pass_on_every_rank'] and line['world'] == n and line['backend'] == 'nccl', line
    assert line['rccl']['ranks_seen'] == list(range(n)) and line['rccl']['distinct_devices'] == n, line['rccl']
    assert line['engine']['graph_replays'] > 0


@pytest.mark.parametrize('use_graph', [False, True])
def test_end_to_end_preresnet8_vs_reference_on_gpu(golden_dir, use_graph):
    """BASELINE configs[1]'s network family (PreResNet, BatchNorm) and sampler (SGHMC): the reference's CPU run
    replayed on the GPU with its captured noise. north_star's criterion — fp32 predictive probabilities
    within 1e-5 relative of the reference CPU path — on 64 test rows after 4 noisy SGHMC steps, with the ReLU gates
    the reference took at its near-zero pre-activations given (top of this file)."""
    from test_samplers_cpu import _load_preresnet8, _preresnet8_inputs
    g = np.load(os.path.join(golden_dir, 'e2e_preresnet8.npz'))
    hyp = json.loads(str(g['hyper']))
    train, test = _preresnet8_inputs(g)
    s = inference.SGHMC(dict(hyp), _load_preresnet8(g), train, device=DEV, use_graph=use_graph)
    _replay_mode(s, use_graph)

    def eps(k):
        e = torch.zeros(s.arena.n, device=DEV)
        e[s.arena.layout.gather_index(DEV)] = torch.tensor(g['eps'][k], device=DEV)
        return e
    s.eps_provider = eps
    _give_reference_gates([s], g)
    ens = s.sample()
    assert (s.engine.stats['graph_replays'] >= 2) if use_graph else (s.engine.stats['graph_replays'] == 0), s.engine.stats
    _assert_reference_gates_were_the_only_difference(s, g)
    for m, ref in zip(ens, g['samples']):
        np.testing.assert_allclose(flat_params(m).cpu().numpy(), ref, rtol=1e-5, atol=2e-6)
    pred = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    pred.update_statistics(ens, output_performance=False)
    np.testing.assert_allclose(pred.ensemble_proba.numpy(), g['proba_sum'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(pred.expected_data_uncertainty.numpy(), g['ent_sum'], rtol=1e-5, atol=1e-6)


def test_chain_group_parallel_branches_match_single_chains():
    """K chains as K parallel branches of one hipGraph: each chain ends where it ends when run alone (up to
    MIOpen's atomics-level nondeterminism), the branches really are captured together, chains differ."""
    from ursabench_amd import util
    hyp = {'lr': 0.05, 'prior_std': 0.5, 'num_samples': 1, 'alpha': 0.5, 'burn_in_epochs': 1}
    train = synthetic(1024 + 40, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128)

    def make(k):
        util.set_random_seed(k)
        return inference.SGHMC(dict(hyp), models.PreResNet(10, 8).to(DEV), train, device=DEV)
    alone = [make(k).sample()[0] for k in range(3)]
    group = inference.ChainGroup([make(k) for k in range(3)])
    together = [c[0] for c in group.sample()]
    assert group.stats['captures'] == 1 and group.stats['graph_replays'] > 0
    for a, b in zip(alone, together):
        np.testing.assert_allclose(flat_params(a).cpu().numpy(), flat_params(b).cpu().numpy(), rtol=2e-3, atol=3e-4)
    assert not torch.equal(flat_params(together[0]), flat_params(together[1]))
    assert all(s.optimizer._step == 18 for s in group.samplers)


def test_many_samplers_and_tasks_in_one_process(monkeypatch):
    """What a hyper-optimisation loop does: a new sampler and new tasks per trial, the old ones dropped whenever
    Python gets to it. With side streams taken from PyTorch's round-robin pool per capture this segfaulted in
    hipGraphLaunch (hip::Graph::UpdateStreams, ROCm 7.2) on the third trial — the pool wraps after 32 streams and a
    new multi-branch capture forks onto streams a live graph was captured on; the package now forks every capture
    onto ONE fixed set of side streams (ursabench_amd/_capture.py; tools/exp/graph_stress.py)."""
    from ursabench_amd import util
    from ursabench_amd.tasks.task_base import EnsembleAccumulator
    monkeypatch.setattr(EnsembleAccumulator, 'EVAL_ROWS', 0)      # a captured graph per loader-batch shape: the pattern that crashed
    train = synthetic(512, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128)
    test = synthetic(300, (3, 32, 32), 10, seed=1, device=DEV, batch_size=128)
    hyp = {'lr': 0.05, 'prior_std': 0.5, 'num_samples': 3, 'alpha': 0.5, 'burn_in_epochs': 0}
    sums = []
    for trial in range(8):
        util.set_random_seed(trial)
        s = inference.SGHMC(dict(hyp), models.PreResNet(10, 8).to(DEV), train, device=DEV)
        ens = s.sample()
        for _ in range(2):
            p = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
            p.update_statistics(ens, output_performance=False)
        sums.append(float(p.ensemble_proba.sum()))
    assert all(abs(v - 3 * 300) < 1e-2 for v in sums)


@pytest.mark.parametrize('use_graph', [False, True])
@pytest.mark.parametrize('name', ['cSGHMC', 'cSGLD'])
def test_cyclic_samplers_end_to_end_vs_reference_on_gpu(golden_dir, name, use_graph):
    """G12 on the HIP path: the reference's cSGHMC / cSGLD trajectories (tiny MLP, 16 steps, 4 samples) with its
    captured noise; the per-iteration lr walks the device schedule table. rocBLAS vs oneDNN gradients differ in the
    last bits, amplified over 16 steps: parameters agree to 1e-4 relative / 1e-5 absolute."""
    from test_samplers_cpu import _cyclic_replay
    s, ens, g = _cyclic_replay(golden_dir, name, DEV, use_graph=use_graph, warmup_steps=1 if use_graph else None)
    assert (s.engine.stats['graph_replays'] >= 8) if use_graph else (s.engine.stats['graph_replays'] == 0), s.engine.stats
    for m, ref in zip(ens, g[f'{name}/samples']):
        np.testing.assert_allclose(flat_params(m).cpu().numpy(), ref, rtol=1e-4, atol=1e-5)


def test_swag_grouped_sampling_on_gpu():
    """SWAG.sample() refreshes the BatchNorm statistics of LANES members in one pass over the training set, their
    forwards running on LANES side streams: from the same moments and the same draw indices the members equal the
    ones formed one at a time."""
    from ursabench_amd import util
    hyp = {'swag_lr': 0.01, 'swag_wd': 1e-4, 'lr_init': 0.05, 'num_samples': 5, 'momentum': 0.9, 'burn_in_epochs': 1,
           'num_iterates': 2}
    train = synthetic(512, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128)
    util.set_random_seed(3)
    s = inference.SWAG(dict(hyp), models.PreResNet(10, 8).to(DEV), train, device=DEV, reference_quirks=False, seed=5)
    s.run_trajectory()
    s.LANES = 1
    one = s.sample()
    s._draws = 0                                           # same Philox draw indices again
    s.LANES, s.GROUP_MIN_PARAMS = 4, 0
    grp = s.sample()                                       # a group of 4 and a group of 1
    assert len(one) == len(grp) == 5
    for a, b in zip(one, grp):
        assert torch.equal(flat_params(a), flat_params(b))                     # same draw, bit for bit
        for (ka, va), (kb, vb) in zip(a.named_buffers(), b.named_buffers()):
            assert ka == kb
            if va.dtype == torch.float32:                                      # BatchNorm statistics of the refresh pass
                np.testing.assert_allclose(va.cpu().numpy(), vb.cpu().numpy(), rtol=1e-5, atol=1e-6)
    steps = len(train)
    cnt = lambda m: int(dict(m.named_buffers())['bn.num_batches_tracked'])
    assert [cnt(m) - cnt(one[0]) for m in one] == [k * steps for k in range(5)]       # the reference's cumulative counter
    assert [cnt(m) - cnt(grp[0]) for m in grp] == [k * steps for k in range(5)]
    assert not torch.equal(flat_params(grp[0]), flat_params(grp[1]))


# ---- GPU replays of the reference's own runs that round 2 only replayed on CPU (G8, G10, G13) ---------------------
@pytest.mark.parametrize('tag', ['mlp', 'bn'])
@pytest.mark.parametrize('cls_name', ['SWAG', 'SWA'])
def test_swag_swa_reference_run_on_gpu(golden_dir, tag, cls_name):
    """G8 on the HIP path: the reference's SWA / SWAG runs (URSABench/inference/swag.py:51-129, swa.py) — FlatSGD
    trajectory through hipGraph replays, K2 moments, the (bug-compatible) draw, and `bn_update`
    (URSABench/util.py:212-247): BatchNorm running statistics of every member and the members' predictive through
    Prediction within 1e-5 relative of the reference's. rocBLAS / MIOpen gradients differ from oneDNN's in the last
    bits: parameters and moments agree to 1e-4 relative / 1e-6 absolute over the 8-step trajectory."""
    from test_swag_cpu import bn_loader, bn_net
    from test_samplers_cpu import tiny_loader, tiny_net
    g = np.load(os.path.join(golden_dir, 'swag_e2e.npz'))
    hyp = json.loads(str(g['hyper']))
    torch.manual_seed(0)
    net = (tiny_net if tag == 'mlp' else bn_net)()
    assert np.array_equal(flat_params(net).numpy(), g[f'{tag}/{cls_name}/theta0'])
    mk = tiny_loader if tag == 'mlp' else bn_loader
    s = getattr(inference, cls_name)(dict(hyp), net, mk(), device=DEV)
    s.engine.WARMUP_STEPS = 1
    ens = s.sample(num_samples=2)
    assert s.engine.stats['graph_replays'] >= 4, s.engine.stats
    assert s.epochs_run == int(g[f'{tag}/{cls_name}/epochs_run'])
    assert np.array_equal(s.num_models_collected.cpu().numpy(), g[f'{tag}/{cls_name}/n_collected'])
    tol = dict(rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(flat_params(s.model).cpu().numpy(), g[f'{tag}/{cls_name}/live_theta'], **tol)
    np.testing.assert_allclose(s.weight_mean.cpu().numpy(), g[f'{tag}/{cls_name}/weight_mean'], **tol)
    np.testing.assert_allclose(s.sq_mean.cpu().numpy(), g[f'{tag}/{cls_name}/sq_mean'], **tol)
    assert (ens[0] is ens[1]) == bool(g[f'{tag}/{cls_name}/same_object'])
    for m, ref, refb in zip(ens, g[f'{tag}/{cls_name}/samples'], g[f'{tag}/{cls_name}/sample_buffers']):
        np.testing.assert_allclose(flat_params(m).cpu().numpy(), ref, **tol)
        if refb.size:                                    # BatchNorm statistics after bn_update (row a8)
            got = torch.cat([b.detach().float().reshape(-1) for b in m.buffers()]).cpu().numpy()
            np.testing.assert_allclose(got, refb, rtol=1e-5, atol=1e-6)
    pred = tasks.Prediction({'in_distribution_test': mk(seed=1)}, 4, DEV, 'ALL')
    pred.update_statistics(ens, output_performance=False)
    np.testing.assert_allclose(pred.ensemble_proba.numpy(), g[f'{tag}/{cls_name}/proba_sum'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(pred.expected_data_uncertainty.numpy(), g[f'{tag}/{cls_name}/ent_sum'], rtol=1e-5, atol=1e-6)


def test_sgd_sampler_reference_run_on_gpu(golden_dir):
    """G13 on the HIP path: the reference's SGD baseline sampler (URSABench/inference/sgd.py:69-113), constructor run
    and the run after update_hyp, FlatSGD (K1 SGD mode) under hipGraph replay."""
    from test_samplers_cpu import tiny_loader, tiny_net
    from ursabench_amd import util
    g = np.load(os.path.join(golden_dir, 'sgd_sampler.npz'))
    hyp, hyp2 = json.loads(str(g['hyper'])), json.loads(str(g['hyper2']))
    util.set_random_seed(5)
    net = tiny_net()
    assert np.array_equal(flat_params(net).numpy(), g['theta0'])
    s = inference.SGD(dict(hyp), net, tiny_loader(), device=DEV)
    s.engine.WARMUP_STEPS = 1
    m = s.sample(num_samples=2)
    assert s.engine.stats['graph_replays'] >= 4
    np.testing.assert_allclose(flat_params(m[0]).cpu().numpy(), g['sample'], rtol=1e-4, atol=1e-6)
    assert s.optimizer.param_groups[0]['lr'] == pytest.approx(float(g['lr_after']), rel=1e-12)
    util.set_random_seed(6)
    s.update_hyp(dict(hyp2))
    s.engine.WARMUP_STEPS = 1
    # update_hyp re-initialises on the device (HIP generator, not the reference's CPU stream): start the second run from
    # the reference's re-initialised weights
    s.arena.load_flat(torch.tensor(g['theta1']))
    m = s.sample()
    np.testing.assert_allclose(flat_params(m[0]).cpu().numpy(), g['sample2'], rtol=1e-4, atol=1e-6)
    assert s.optimizer.param_groups[0]['lr'] == pytest.approx(float(g['lr_after2']), rel=1e-12)


def cpu_stream_dropout(x, p=0.5, training=True, inplace=False):
    """F.dropout with the mask drawn on the HOST from torch's global CPU generator, exactly as ATen's CPU dropout
    draws it (empty_like(input).bernoulli_(1 - p).div_(1 - p); aten/src/ATen/native/Dropout.cpp): lets a device run
    consume the very masks the reference's CPU run consumed. TEST ONLY (checked bit-exact on CPU in test_mcdropout_cpu)."""
    if not training or p == 0:
        return x
    noise = torch.empty(x.shape, dtype=x.dtype).bernoulli_(1 - p).div_(1 - p)
    return x * noise.to(x.device)


def test_mcdropout_reference_run_on_gpu(golden_dir, monkeypatch):
    """G10 on the HIP path: the reference's MCdropout run (URSABench/inference/vi_dropout.py:87-131) — model swap,
    per-minibatch OneCycleLR (lr, momentum) walked from the device schedule table by the self-advancing K1 launch
    (SGD mode), always-on dropout. The dropout masks are the reference's (host-drawn, see cpu_stream_dropout), so the
    steps run eagerly (a captured graph cannot take a host-drawn mask)."""
    from test_samplers_cpu import tiny_loader
    import torch.nn.functional as F
    monkeypatch.setattr(F, 'dropout', cpu_stream_dropout)
    g = np.load(os.path.join(golden_dir, 'mcdropout.npz'))
    hyp, hyp2 = json.loads(str(g['hyper'])), json.loads(str(g['hyper2']))
    torch.manual_seed(21)
    s = inference.MCdropout(dict(hyp), models.MLP(16, 12, 4), tiny_loader(), device=DEV, use_graph=False)
    np.testing.assert_array_equal(flat_params(s.model).cpu().numpy(), g['theta0'])
    for k in range(2):
        m = s.sample_iterative()
        np.testing.assert_allclose(flat_params(m).cpu().numpy(), g['samples'][k], rtol=1e-4, atol=1e-6)
    from ursabench_amd._native import StepCtl
    c = StepCtl.from_buffer_copy(bytes(s.optimizer._ctl.cpu().numpy()))
    assert c.step == len(g['lr_mom']) and c.tickets_clear()
    s.model.eval()
    with torch.no_grad():
        mc = np.stack([s.model(torch.tensor(g['x_test']).to(DEV)).cpu().numpy() for _ in range(3)])
    np.testing.assert_allclose(mc, g['mc_logits'], rtol=1e-4, atol=1e-5)
    # update_hyp re-initialises the model: the reference drew those weights from the CPU generator (and so moved the
    # stream its later dropout masks come from); the device run draws them from the HIP generator. Re-initialise a host
    # copy the reference's way — same draws, same stream position — and start the second run from those weights.
    import copy
    from ursabench_amd import util
    host_twin = copy.deepcopy(s.model).cpu()
    s.update_hyp(dict(hyp2))
    util.reset_model(host_twin)
    np.testing.assert_array_equal(flat_params(host_twin).numpy(), g['theta1'])
    s.arena.load_flat(flat_params(host_twin))
    np.testing.assert_allclose(flat_params(s.sample_iterative()).cpu().numpy(), g['sample2'], rtol=1e-4, atol=1e-6)


def _hmc_proposal_vs_oracle(model, train, L, eps, tau, mass, seed, use_graph, expect_params=None):
    """Run ONE HMC proposal of `model` on the device, tapping the gradients of every potential evaluation and the K4
    launches, then replay hamiltorch's sub-step sequence (half kick; L x (drift, gradient, kick); half-kick correction;
    call site URSABench/inference/hmc.py:71-75) in the oracle on those gradients: theta and momentum bit for bit, both
    kinetic energies within 2e-6 relative; the host issues L + 3 K4 launches."""
    from ursabench_amd import _native
    s = inference.HMC({'step_size': eps, 'num_samples': 1, 'L': L, 'tau': tau, 'burn': 0, 'mass': mass},
                      model, train, device=DEV, seed=seed, use_graph=use_graph)
    s._bind()
    if expect_params is not None:
        assert s.arena.num_parameters == expect_params
    th0 = s.arena.theta.cpu().numpy().copy()
    grads, launches = [], []
    orig_eval, orig_leap = s._neg_logp_and_grad, s.kernels.leapfrog

    def tap_eval():
        u = orig_eval()
        grads.append(s._glogp.cpu().numpy().copy())
        return u

    def tap_leap(*a, **k):
        launches.append(k['flags'])
        return orig_leap(*a, **k)
    s._neg_logp_and_grad = tap_eval
    import builtins
    log = []
    real_print = builtins.print
    s.kernels.leapfrog = tap_leap
    try:
        builtins.print = lambda d, *a, **k: log.append(d)
        s.sample(debug=True)
    finally:
        builtins.print = real_print
        s.kernels.leapfrog = orig_leap
    KD = _native.LEAP_KICK | _native.LEAP_DRIFT
    assert launches == [0] + [KD] * L + [_native.LEAP_KICK, _native.LEAP_KICK] and len(launches) == L + 3
    assert len(grads) == L + 1 and s.accepted == 1
    n = s.arena.n
    mask = s._mask.cpu().numpy()
    p = (O.philox_normal(n, seed, 0) * mask) * np.float32(np.sqrt(mass))
    th = th0.copy()
    inv_mass = 1.0 / mass
    ke0 = O.leapfrog(None, p, None, kick_coef=0.0, step_size=0.0, inv_mass=inv_mass, flags=0, want_kinetic=True)
    O.leapfrog(None, p, grads[0], kick_coef=0.5 * eps, step_size=eps, inv_mass=inv_mass, flags=O.LEAP_KICK)
    for l in range(L):
        O.leapfrog(th, p, None, kick_coef=0.0, step_size=eps, inv_mass=inv_mass, flags=O.LEAP_DRIFT)
        O.leapfrog(None, p, grads[l + 1], kick_coef=eps, step_size=eps, inv_mass=inv_mass, flags=O.LEAP_KICK)
    ke1 = O.leapfrog(None, p, grads[L], kick_coef=-0.5 * eps, step_size=eps, inv_mass=inv_mass, flags=O.LEAP_KICK,
                     want_kinetic=True)
    assert np.array_equal(s.arena.theta.cpu().numpy(), th)            # accepted: the chain sits at the proposal
    assert np.array_equal(s._p.cpu().numpy(), p)
    assert float(s._acc[0]) == pytest.approx(ke1, rel=2e-6)           # the closing launch's fused kinetic-energy reduction
    assert np.isfinite(log[0]['H0']) and np.isfinite(log[0]['H1'])
    s._p.copy_(torch.from_numpy((O.philox_normal(n, seed, 0) * mask) * np.float32(np.sqrt(mass))).to(DEV))
    assert float(s._kinetic()) == pytest.approx(ke0, rel=2e-6)
    return s, log[0]


def test_hmc_fused_proposal_equals_oracle_substeps():
    """One full HMC proposal (L = 3) at PreResNet-164's parameter count, P = 1,726,388, as the host issues it —
    kinetic energy, fused half-kick + drift, two fused kick + drift launches, full kick, half-kick correction with the
    kinetic-energy reduction: L + 3 K4 launches — against the oracle's sub-step sequence on the GPU's gradients."""
    class Wide(torch.nn.Module):                         # 3072 x 561 + 561 + 2435 = 1,726,388 parameters, cheap to evaluate
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(3072, 561)
            self.extra = torch.nn.Parameter(torch.randn(2435) * 0.01)

        def forward(self, x):
            return self.lin(x.flatten(1))[:, :100] + self.extra[:100]
    torch.manual_seed(0)
    train = synthetic(256, (3, 32, 32), 100, seed=0, device=DEV, batch_size=128)
    _, rec = _hmc_proposal_vs_oracle(Wide().to(DEV), train, L=3, eps=2e-4, tau=1.0, mass=2.0, seed=13, use_graph=False,
                                     expect_params=1726388)
    assert abs(rec['H0'] - rec['H1']) < 1.0


def test_hmc_sampler_on_preresnet164_equals_oracle_substeps():
    """BASELINE configs[4]'s own network through the sampler on the GPU (no GPU test ran a PreResNet-164
    sampler): PreResNet-164 / CIFAR-100-shaped, full-batch potential over 64 images (BatchNorm in training mode, as
    hamiltorch runs the model), one proposal of L = 2 with the potential replayed from a hipGraph after two eager
    evaluations — theta / momentum bit-identical to the oracle's sub-step sequence on the GPU's gradients."""
    from ursabench_amd import util
    util.set_random_seed(0)
    train = synthetic(64, (3, 32, 32), 100, seed=0, device=DEV, batch_size=64)
    s, rec = _hmc_proposal_vs_oracle(models.PreResNet(100, 164).to(DEV), train, L=2, eps=1e-4, tau=1.0, mass=1.0, seed=3,
                                     use_graph=True, expect_params=1726388)
    assert s._graph is not None                                       # the third evaluation was a graph replay


def test_swag_sampler_on_wideresnet28_10_follows_the_oracle_draw():
    """BASELINE configs[3]'s own network through the sampler on the GPU (no GPU test ran a WideResNet-28-10
    sampler): WideResNet-28-10 (36,546,980 parameters), SWAG as published on a 256-image set — the SGD trajectory
    through hipGraph replays with the roofline-sized, self-advancing K1 launch (non-temporal; 17,846 workgroups on the
    ticket tree), K2
    moments, then two members: each equals, bit for bit, the oracle's draw from the moments the device holds (same
    Philox key and draw index) through the ensemble form (std once + square-root-free draw); BatchNorm statistics
    refreshed (grouped path: 2 members in one pass)."""
    from ursabench_amd import util
    from ursabench_amd._native import StepCtl
    util.set_random_seed(0)
    train = synthetic(256, (3, 32, 32), 100, seed=0, device=DEV, batch_size=64)
    hyp = {'swag_lr': 0.01, 'swag_wd': 3e-4, 'lr_init': 0.05, 'num_samples': 2, 'momentum': 0.9, 'burn_in_epochs': 1,
           'num_iterates': 2}
    s = inference.SWAG(dict(hyp), models.WideResNet(100, 28, 10).to(DEV), train, device=DEV, reference_quirks=False, seed=21)
    assert s.num_parameters == 36546980
    ens = s.sample()
    assert len(ens) == 2 and s.num_models_collected.item() == 2 and s.engine.stats['graph_replays'] > 0
    assert s.optimizer.self_advance is True                           # also at 36.5 M elements (17,846 workgroups on the ticket tree)
    c = StepCtl.from_buffer_copy(bytes(s.optimizer._ctl.cpu().numpy()))
    assert c.step == 3 * 4 and c.tickets_clear()
    mean, sq = s._mean.cpu().numpy(), s._sq.cpu().numpy()
    for d, m in enumerate(ens):
        want = np.empty_like(mean)
        O.swag_draw(want, mean, sq, var_clamp=1e-30, scale=1.0, seed=21, draw=d)
        got = m._ursa_row[:s.arena.n].cpu().numpy()
        assert np.array_equal(got, want), d
        bn = dict(m.named_buffers())
        assert float(bn['bn1.running_var'].min()) > 0 and torch.isfinite(bn['bn1.running_mean']).all()
    assert not torch.equal(ens[0]._ursa_row, ens[1]._ursa_row)


def test_swag_grouped_sampling_with_a_host_resident_loader():
    """in the grouped BatchNorm refresh every batch is allocated on the current stream (here: the H2D copy of a
    host-resident loader's batch) and read by member forwards on side streams; without `record_stream` the caching
    allocator could hand the block to the next batch's copy while a forward still reads it. Grouped members must equal
    the ones formed one at a time, on a loader whose batches are fresh device allocations."""
    from ursabench_amd import util
    hyp = {'swag_lr': 0.01, 'swag_wd': 1e-4, 'lr_init': 0.05, 'num_samples': 4, 'momentum': 0.9, 'burn_in_epochs': 1,
           'num_iterates': 1}
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(1024, 3, 32, 32, generator=g), torch.randint(0, 10, (1024,), generator=g)
    train = DataLoader(TensorDataset(x, y), batch_size=64, shuffle=False)            # host tensors: x.to(device) per batch
    util.set_random_seed(3)
    s = inference.SWAG(dict(hyp), models.PreResNet(10, 8).to(DEV), train, device=DEV, reference_quirks=False, seed=5)
    s.run_trajectory()
    s.LANES = 1
    one = s.sample()
    s._draws = 0
    s.LANES, s.GROUP_MIN_PARAMS = 4, 0
    for _ in range(3):                                        # a race is a matter of timing: several passes
        s._draws = 0
        grp = s.sample()
        for a, b in zip(one, grp):
            assert torch.equal(flat_params(a), flat_params(b))
            for (ka, va), (kb, vb) in zip(a.named_buffers(), b.named_buffers()):
                if va.dtype == torch.float32:
                    np.testing.assert_allclose(va.cpu().numpy(), vb.cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=ka)


def test_chain_group_of_cyclic_samplers_replays_the_reference_run(golden_dir):
    """G12 through the group path: two cSGHMC chains of one ChainGroup, both fed the reference's captured noise from its
    initial weights, each reproduce the reference's cSGHMC trajectory — every chain's per-iteration (lr, noise scale)
    table is walked by the ONE multi-chain update launch (its own `sched` pointer and call index in ctl[k]) under graph
    replay, noise masks and collect epochs included."""
    from test_samplers_cpu import tiny_loader, tiny_net
    g = np.load(os.path.join(golden_dir, 'e2e_cyclic.npz'))
    name = 'cSGHMC'
    hyp = json.loads(str(g[f'{name}/hyper']))
    loader = tiny_loader()
    chains = []
    for k in range(2):
        net = tiny_net()
        with torch.no_grad():
            off = 0
            for p in net.parameters():
                p.copy_(torch.tensor(g[f'{name}/theta0'][off:off + p.numel()]).view_as(p))
                off += p.numel()
        chains.append(inference.cSGHMC(dict(hyp), net, loader, device=DEV, seed=k))
    group = inference.ChainGroup(chains)
    group.WARMUP_STEPS = 1

    def provider(s):
        def eps(k):
            e = torch.zeros(s.arena.n, device=DEV)
            e[s.arena.layout.gather_index(DEV)] = torch.tensor(g[f'{name}/eps'][k], device=DEV)
            return e
        return eps
    for s in chains:
        s.eps_provider = provider(s)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        per_chain = group.sample()
    assert group.stats['graph_replays'] >= 8 and group.stats['update_launches'] == 16
    for ens in per_chain:
        assert len(ens) == len(g[f'{name}/samples'])
        for m, ref in zip(ens, g[f'{name}/samples']):
            np.testing.assert_allclose(flat_params(m).cpu().numpy(), ref, rtol=1e-4, atol=1e-5)


def test_c3_partition_eight_chains_equals_one_ensemble():
    """BASELINE configs[2] (8 independent SGHMC chains, members stay with their chain, ONE sum for the predictive) on
    one GPU: 8 chains stepped as a ChainGroup (one multi-chain update launch per round), every chain's members
    accumulated by its own Prediction task — what rank c of the 8-GPU job holds locally — and the sum of the 8 local
    accumulators (the all-reduce is a sum) equals one Prediction over all 16 members to fp32 summation
    order; the chains are distinct (seed = chain id, experiment.py:170)."""
    from ursabench_amd import util
    hyp = {'lr': 0.05, 'prior_std': 0.5, 'num_samples': 2, 'alpha': 0.5, 'burn_in_epochs': 0}
    train = synthetic(640, (3, 32, 32), 10, seed=0, device=DEV, batch_size=128)
    test = synthetic(512, (3, 32, 32), 10, seed=1, device=DEV, batch_size=128)
    chains = []
    for c in range(8):
        util.set_random_seed(c)
        chains.append(inference.SGHMC(dict(hyp), models.PreResNet(10, 8).to(DEV), train, device=DEV, seed=c))
    group = inference.ChainGroup(chains)
    per_chain = group.sample()
    assert group.stats['update_launches'] == 2 * len(train) and len(per_chain) == 8 and all(len(e) == 2 for e in per_chain)
    firsts = [flat_params(e[0]) for e in per_chain]
    assert all(not torch.equal(firsts[0], f) for f in firsts[1:])
    p_sum, e_sum = torch.zeros(512, 10), torch.zeros(512)
    for ens in per_chain:                                   # "rank c": its own members only, no process group
        t = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
        t.update_statistics(ens, output_performance=False)
        assert t.num_samples_collected == 2
        p_sum += t.ensemble_proba
        e_sum += t.expected_data_uncertainty
    whole = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    whole.update_statistics([m for ens in per_chain for m in ens], output_performance=False)
    assert whole.num_samples_collected == 16
    np.testing.assert_allclose(p_sum.numpy(), whole.ensemble_proba.numpy(), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(e_sum.numpy(), whole.expected_data_uncertainty.numpy(), rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(whole.ensemble_proba.sum(1).numpy(), np.full(512, 16.0, np.float32), rtol=1e-6)


@pytest.mark.parametrize('use_graph', [False, True])
def test_csghmc_across_update_hyp_vs_reference_on_gpu(golden_dir, use_graph):
    """G15 on the HIP path: the reference's cSGHMC run, `update_hyp` (new optimizer adopting the control block and the
    schedule table a captured graph holds; the Philox / update counter keeps counting), second run whose cyclical
    schedule keeps the constructor's `total_iterations` (csghmc.py:48-62) — eager and through graph replay."""
    from test_samplers_cpu import _cyclic_update_hyp_replay
    s, ens, g, hyp2 = _cyclic_update_hyp_replay(golden_dir, DEV, use_graph=use_graph, warmup_steps=1 if use_graph else None)
    for m, ref in zip(ens, g['samples']):
        np.testing.assert_allclose(flat_params(m).cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
    # the device re-initialises from the HIP generator: continue from the reference's re-initialised weights
    s.arena.load_flat(torch.tensor(g['theta1']))
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        ens2 = s.sample()
    assert (s.engine.stats['graph_replays'] >= 15) if use_graph else (s.engine.stats['graph_replays'] == 0), s.engine.stats
    for m, ref in zip(ens2, g['samples2']):
        np.testing.assert_allclose(flat_params(m).cpu().numpy(), ref, rtol=1e-4, atol=1e-5)
