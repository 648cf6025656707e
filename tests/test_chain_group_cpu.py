"""CPU: K chains stepped in lock-step by ChainGroup compute exactly what each chain computes alone."""
import numpy as np
import pytest
import torch

import ursabench_amd.inference as inference
from ursabench_amd import util
from oracle_kernels import OracleKernels
from test_samplers_cpu import tiny_loader, tiny_net


def flat(m):
    return torch.cat([p.detach().reshape(-1) for p in m.parameters()]).numpy()


@pytest.mark.parametrize('cls,hyp', [
    (inference.SGHMC, {'lr': 0.05, 'prior_std': 1.0, 'num_samples': 2, 'alpha': 0.5, 'burn_in_epochs': 1}),
    (inference.cSGLD, {'lr_0': 0.05, 'prior_std': 1.0, 'num_samples_per_cycle': 1, 'cycle_length': 3,
                       'burn_in_epochs': 1, 'num_cycles': 2, 'alpha': 1.0}),
])
def test_group_equals_individual_chains(cls, hyp):
    loader = tiny_loader()

    def make(k):
        util.set_random_seed(k)
        return cls(dict(hyp), tiny_net(), loader, kernels=OracleKernels(), use_graph=False)
    alone = [make(k).sample() for k in range(3)]
    group = inference.ChainGroup([make(k) for k in range(3)], use_graph=False)
    together = group.sample()
    assert len(together) == 3 and all(len(c) == len(alone[0]) for c in together)
    for a, b in zip(alone, together):
        for ma, mb in zip(a, b):
            assert np.array_equal(flat(ma), flat(mb))
    assert not np.array_equal(flat(together[0][0]), flat(together[1][0]))       # chains differ (seed = chain id)


def test_group_argument_checks():
    loader = tiny_loader()
    hyp = {'lr': 0.05, 'prior_std': 1.0, 'num_samples': 1, 'alpha': 0.5, 'burn_in_epochs': 0}
    a = inference.SGHMC(dict(hyp), tiny_net(), loader, kernels=OracleKernels(), use_graph=False)
    b = inference.SGLD(dict(hyp), tiny_net(), loader, kernels=OracleKernels(), use_graph=False)
    c = inference.SGHMC(dict(hyp), tiny_net(), tiny_loader(), kernels=OracleKernels(), use_graph=False)
    with pytest.raises(ValueError):
        inference.ChainGroup([a, b])
    with pytest.raises(ValueError):
        inference.ChainGroup([a, c])
    with pytest.raises(TypeError):
        inference.ChainGroup([])
    d = inference.SGHMC(dict(hyp, burn_in_epochs=2), tiny_net(), loader, kernels=OracleKernels(), use_graph=False)
    with pytest.raises(ValueError, match='Philox key'):          # same torch.initial_seed(): identical noise streams
        inference.ChainGroup([a, d], use_graph=False)
    d = inference.SGHMC(dict(hyp, burn_in_epochs=2), tiny_net(), loader, kernels=OracleKernels(), use_graph=False,
                        seed=a.seed + 1)
    with pytest.raises(RuntimeError, match='out of step'):
        inference.ChainGroup([a, d], use_graph=False).sample_iterative()


def test_update_hyp_on_group_members_invalidates_and_matches_single_chains():
    """update_hyp rebuilds each chain's optimizer; the group must drop its captured round (here: its
    record of the device state) and the chains must continue exactly like chains run alone — with the
    control block re-used (no freed address inside a graph) and the Philox counter carried over."""
    loader = tiny_loader()
    hyp = {'lr': 0.05, 'prior_std': 1.0, 'num_samples': 1, 'alpha': 0.5, 'burn_in_epochs': 0}
    hyp2 = dict(hyp, lr=0.02, alpha=0.3)

    def make(k):
        util.set_random_seed(k)
        return inference.SGHMC(dict(hyp), tiny_net(), loader, kernels=OracleKernels(), use_graph=False, seed=k)

    def run_alone(k):
        s = make(k)
        s.sample_iterative()
        util.set_random_seed(10 + k)                       # reset_model re-initialises from the global generator
        s.update_hyp(dict(hyp2))
        return s.sample_iterative(), s
    alone = [run_alone(k) for k in range(2)]
    chains = [make(k) for k in range(2)]
    group = inference.ChainGroup(chains, use_graph=False)
    group.sample_iterative()
    group._graph, group._captured_with = object(), group._device_state()       # stand-in for a captured round
    ctl_before = [c.optimizer._ctl.data_ptr() for c in chains]
    steps_before = [c.optimizer._step for c in chains]
    for k, c in enumerate(chains):
        util.set_random_seed(10 + k)
        c.update_hyp(dict(hyp2))
    assert [c.optimizer._ctl.data_ptr() for c in chains] == ctl_before         # control block survives the rebuild
    assert [c.optimizer._step for c in chains] == steps_before and steps_before[0] > 0
    group.use_graph = True                                                     # only so that _run_epoch checks ...
    group._run_epoch = (lambda orig: (lambda plans: (setattr(group, 'use_graph', False), orig(plans))[1]))(group._run_epoch)
    together = group.sample_iterative()
    assert group._captured_with is None                                        # ... and found the stale capture
    for (ma, sa), mb, c in zip(alone, together, chains):
        assert np.array_equal(flat(ma), flat(mb))
        assert sa.optimizer._step == c.optimizer._step


def test_regrouping_a_chain_is_refused_by_the_old_group():
    """A sampler lives in ONE group's slabs: putting it into a second ChainGroup re-homes its vectors, and the first
    group must refuse to step (its one launch would update stale slab rows) instead of doing so silently."""
    import pytest
    K = OracleKernels()
    hyp = {'lr': 0.05, 'prior_std': 1.0, 'num_samples': 1, 'alpha': 0.5, 'burn_in_epochs': 0}
    loader = tiny_loader()
    chains = [inference.SGHMC(dict(hyp), tiny_net(), loader, kernels=K, use_graph=False, seed=k) for k in range(3)]
    g1 = inference.ChainGroup(chains[:2], use_graph=False)
    g1.sample_iterative()
    g2 = inference.ChainGroup(chains[1:], use_graph=False)
    g2.sample_iterative()
    with pytest.raises(RuntimeError):
        g1.sample_iterative()
