"""CPU: call sequences of the reference's callers against our classes (oracle kernel set):
hyper-optimisation (`_HypOpt.inference_step`, URSABench/hyperopt/hyper_optimization.py:51-73), the SGD
baseline sampler, member-bank checkpoints."""
import numpy as np
import pytest
import torch

import ursabench_amd.inference as inference
from ursabench_amd import checkpoint, tasks
from oracle_kernels import OracleKernels
from test_samplers_cpu import tiny_loader, tiny_net


def test_hyperopt_call_sequence():
    """update_hyp -> reset -> sample -> update_statistics(output_performance=True) returns a float, twice,
    with the sampler restarting from a re-initialised model each time."""
    K = OracleKernels()
    hyp = {'lr': 0.05, 'prior_std': 1.0, 'num_samples': 2, 'alpha': 0.5, 'burn_in_epochs': 1}
    sampler = inference.SGHMC(dict(hyp), tiny_net(), tiny_loader(), kernels=K, use_graph=False)
    obj = tasks.Prediction({'in_distribution_test': tiny_loader(seed=3)}, 4, torch.device('cpu'), ['nll'], kernels=K)
    vals = []
    for lr in (0.05, 0.01):
        sampler.update_hyp(dict(hyp, lr=lr))
        obj.reset()
        samples = sampler.sample()
        v = obj.update_statistics(samples, output_performance=True)
        assert isinstance(v, float) and np.isfinite(v) and obj.num_samples_collected == 2
        vals.append(v)
    assert vals[0] != vals[1]
    for name in ('SGLD', 'SGHMC', 'cSGLD', 'cSGHMC', 'SWA', 'SWAG', 'HMC', 'SGD', 'MCdropout', 'optimSGHMC'):
        assert hasattr(inference, name)
    with pytest.raises(AttributeError):          # model=None has no class to swap, exactly as in the reference (vi_dropout.py:14)
        inference.MCdropout(None)


def test_sgd_baseline_sampler():
    K = OracleKernels()
    s = inference.SGD({'lr': 0.1, 'epochs': 2, 'momentum': 0.9, 'weight_decay': 1e-3}, tiny_net(), tiny_loader(),
                      kernels=K, use_graph=False)
    out = s.sample(num_samples=2)
    assert out[0] is out[1] is s.model and len(K.step_log) == 3 * 2      # epochs+1 epochs once, then nothing
    assert all(f & 16 for _, _, f, _ in K.step_log)                       # SGD mode, no Langevin noise
    assert s.optimizer.param_groups[0]['lr'] == pytest.approx(0.1 / 100)  # cosine to eta_min = lr/100


def test_member_bank_checkpoint_roundtrip(tmp_path):
    K = OracleKernels()
    net = torch.nn.Sequential(torch.nn.Linear(12, 8), torch.nn.BatchNorm1d(8), torch.nn.ReLU(), torch.nn.Linear(8, 4))
    s = inference.SGHMC({'lr': 0.05, 'prior_std': 1.0, 'num_samples': 3, 'alpha': 0.5, 'burn_in_epochs': 0}, net,
                        tiny_loader(), kernels=K, use_graph=False)
    ens = s.sample()
    p = str(tmp_path / 'ens.pt')
    checkpoint.save_ensemble(ens, p)
    back = checkpoint.load_ensemble(p, net)
    assert len(back) == 3
    x = torch.randn(5, 12)
    for a, b in zip(ens, back):
        a.eval(); b.eval()
        assert torch.equal(a(x), b(x))
        assert all(torch.equal(u, v) for u, v in zip(a.state_dict().values(), b.state_dict().values()))
    sds = checkpoint.to_state_dicts(ens)
    fresh = torch.nn.Sequential(torch.nn.Linear(12, 8), torch.nn.BatchNorm1d(8), torch.nn.ReLU(), torch.nn.Linear(8, 4))
    fresh.load_state_dict(sds[1])
    fresh.eval()
    assert torch.equal(fresh(x), ens[1](x))
    with pytest.raises(ValueError):
        checkpoint.save_ensemble([net], p)


def test_sgd_sampler_equals_the_reference_run(golden_dir):
    """G13: the reference's SGD baseline (inference/sgd.py) — constructor run, update_hyp (re-init in place, new
    optimizer and cosine floor, same epoch count: the quirk), second run — bit-identical parameters on CPU."""
    import json
    import os
    from ursabench_amd import util
    g = np.load(os.path.join(golden_dir, 'sgd_sampler.npz'))
    hyp, hyp2 = json.loads(str(g['hyper'])), json.loads(str(g['hyper2']))
    flat = lambda m: torch.cat([p.detach().reshape(-1) for p in m.parameters()]).numpy()
    util.set_random_seed(5)
    net = tiny_net()
    assert np.array_equal(flat(net), g['theta0'])
    s = inference.SGD(dict(hyp), net, tiny_loader(), kernels=OracleKernels(), use_graph=False)
    m = s.sample(num_samples=2)
    assert np.array_equal(flat(m[0]), g['sample'])
    assert s.optimizer.param_groups[0]['lr'] == pytest.approx(float(g['lr_after']), rel=1e-12)
    util.set_random_seed(6)
    s.update_hyp(dict(hyp2))
    assert np.array_equal(flat(s.model), g['theta1'])
    m = s.sample()
    assert np.array_equal(flat(m[0]), g['sample2'])
    assert s.optimizer.param_groups[0]['lr'] == pytest.approx(float(g['lr_after2']), rel=1e-12)


@pytest.mark.parametrize('name', ['SGHMC', 'SGLD', 'cSGHMC'])
def test_chain_checkpoint_resume_is_bit_identical(tmp_path, name):
    """A chain saved between two samples and resumed in a fresh sampler continues bit for bit: the update counter is the
    Philox call index (counter-based noise), momentum / BatchNorm buffers / scheduler / epoch bookkeeping travel along.
    (The reference has no resume for samplers: SURVEY.md §5.)"""
    import ursabench_amd.inference as inference
    from oracle_kernels import OracleKernels
    from ursabench_amd import util
    from test_swag_cpu import bn_loader, bn_net
    hyp = {'SGHMC': {'lr': 0.05, 'prior_std': 1.0, 'num_samples': 4, 'alpha': 0.5, 'burn_in_epochs': 1},
           'SGLD': {'lr': 0.05, 'prior_std': 1.0, 'num_samples': 4, 'alpha': 1.0, 'burn_in_epochs': 1},
           'cSGHMC': {'lr_0': 0.05, 'prior_std': 1.0, 'num_samples_per_cycle': 2, 'cycle_length': 4, 'burn_in_epochs': 1,
                      'num_cycles': 2, 'alpha': 0.5}}[name]
    flat = lambda m: torch.cat([p.detach().reshape(-1) for p in m.parameters()] + [b.detach().float().reshape(-1) for b in m.buffers()])

    def make():
        util.set_random_seed(4)
        return getattr(inference, name)(dict(hyp), bn_net(), bn_loader(), kernels=OracleKernels(), use_graph=False, seed=17)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        straight = make()
        want = [flat(straight.sample_iterative()) for _ in range(4)]
        first = make()
        got = [flat(first.sample_iterative()) for _ in range(2)]
        p = str(tmp_path / 'chain.pt')
        checkpoint.save_chain(first, p)
        resumed = make()                                   # fresh weights, fresh optimizer, counter at 0 ...
        checkpoint.load_chain(resumed, p)                  # ... put where `first` stopped
        assert resumed.optimizer._step == first.optimizer._step > 0
        got += [flat(resumed.sample_iterative()) for _ in range(2)]
    for a, b in zip(want, got):
        assert torch.equal(a, b)
    other = inference.SGHMC({'lr': 0.05, 'prior_std': 1.0, 'num_samples': 1, 'alpha': 0.5, 'burn_in_epochs': 0},
                            torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(36, 4)), bn_loader(), kernels=OracleKernels(), use_graph=False)
    with pytest.raises(ValueError):
        checkpoint.load_chain(other, p)


def test_swag_checkpoint_resume_is_bit_identical(tmp_path):
    """SWAG (as published) saved after its trajectory and first member, resumed in a fresh sampler: the moments, the
    collected count and the Philox draw index travel along, so the next members are the uninterrupted run's, bit for bit
    (parameters and refreshed BatchNorm statistics)."""
    import ursabench_amd.inference as inference
    from oracle_kernels import OracleKernels
    from ursabench_amd import util
    from test_swag_cpu import bn_loader, bn_net
    hyp = {'swag_lr': 0.01, 'swag_wd': 1e-4, 'lr_init': 0.05, 'num_samples': 3, 'momentum': 0.9, 'burn_in_epochs': 1,
           'num_iterates': 2}
    flat = lambda m: torch.cat([p.detach().reshape(-1) for p in m.parameters()] + [b.detach().float().reshape(-1) for b in m.buffers()])

    def make():
        util.set_random_seed(4)
        return inference.SWAG(dict(hyp), bn_net(), bn_loader(), kernels=OracleKernels(), use_graph=False, reference_quirks=False, seed=23)
    straight = make()
    want = [flat(straight.sample_iterative()) for _ in range(3)]
    first = make()
    got = [flat(first.sample_iterative())]
    p = str(tmp_path / 'swag.pt')
    checkpoint.save_chain(first, p)
    resumed = checkpoint.load_chain(make(), p)
    assert resumed.burnt_in and resumed._draws == 1 and resumed.num_models_collected.item() == 2
    got += [flat(resumed.sample_iterative()) for _ in range(2)]
    for a, b in zip(want, got):
        assert torch.equal(a, b)
