"""CPU: HMC host logic through the oracle kernel set. hamiltorch (the reference's HMC arithmetic) is
absent: PARITY UNPINNED — validated by invariants (acceptance -> 1 as step_size -> 0, energy error
O(eps^2), exactness of the thinning layout, and the sampler targeting the right Gaussian)."""
import math

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

import ursabench_amd.inference as inference
from oracle_kernels import OracleKernels
from test_samplers_cpu import tiny_loader, tiny_net


def test_hmc_layout_and_acceptance():
    torch.manual_seed(0)
    hyp = {'step_size': 1e-3, 'num_samples': 4, 'L': 3, 'tau': 1.0, 'burn': 0, 'mass': 1.0}
    s = inference.HMC(dict(hyp), tiny_net(), tiny_loader(), kernels=OracleKernels())
    assert s.x.shape == (64, 12) and s.y.shape == (64,)
    out = s.sample()
    assert len(out) == len(range(3 * 4 + 1)[0::3]) == 5      # initial + one per proposal (hmc.py:80)
    assert s.accepted == 4                                    # tiny step: energy conserved, always accept
    th = [torch.cat([p.detach().reshape(-1) for p in m.parameters()]) for m in out]
    assert all(not torch.equal(th[i], th[i + 1]) for i in range(4))
    # burn = 2 drops the first two proposals' positions; burn = -1 keeps one position (hmc.py:80 quirk)
    for burn, n in ((2, 3), (-1, 1)):
        torch.manual_seed(0)
        s2 = inference.HMC(dict(hyp, burn=burn), tiny_net(), tiny_loader(), kernels=OracleKernels())
        assert len(s2.sample()) == n
    with pytest.raises(NotImplementedError):
        inference.HMC(dict(hyp), 'nope', tiny_loader(), kernels=OracleKernels()).sample()


class Const(torch.nn.Module):
    """A model whose likelihood does not depend on its weights: the posterior is the N(0, 1/tau) prior."""

    def __init__(self, init=None):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(512) if init is None else init)

    def forward(self, x):
        return torch.zeros(x.shape[0], 2) + 0.0 * self.w.sum()


def const_loader():
    return DataLoader(TensorDataset(torch.zeros(4, 1), torch.zeros(4, dtype=torch.long)), batch_size=4)


def test_hmc_energy_error_scales_quadratically():
    """Smooth (quadratic) target: leapfrog's energy error is O(step_size^2)."""
    import builtins
    errs = []
    for eps in (0.2, 0.1, 0.05):
        torch.manual_seed(3)
        s = inference.HMC({'step_size': eps, 'num_samples': 1, 'L': int(round(0.7 / eps)), 'tau': 4.0, 'burn': 0,
                           'mass': 1.0}, Const(torch.randn(512)), const_loader(), kernels=OracleKernels(), seed=11)
        log = []
        orig = builtins.print
        builtins.print = lambda d, *a, **k: log.append(d)
        try:
            s.sample(debug=True)
        finally:
            builtins.print = orig
        errs.append(abs(log[0]['H0'] - log[0]['H1']))
    assert errs[0] / errs[1] > 3 and errs[1] / errs[2] > 3, errs


def test_hmc_samples_a_gaussian_posterior():
    """No data term (zero-size likelihood via a constant model): the posterior is the N(0, 1/tau) prior;
    HMC with exact-ish leapfrog must reproduce its variance."""
    loader = const_loader()
    torch.manual_seed(3)
    s = inference.HMC({'step_size': 0.03, 'num_samples': 60, 'L': 25, 'tau': 4.0, 'burn': 10, 'mass': 1.0},
                      Const(), loader, kernels=OracleKernels(), seed=11)
    out = s.sample()
    w = torch.stack([m.w.detach() for m in out])
    assert s.accepted >= 50
    assert abs(float(w.var()) - 0.25) < 0.03 and abs(float(w.mean())) < 0.02


def test_wrapper_semantics_vs_reference(golden_dir):
    """G14: the reference's HMC wrapper run around a stand-in hamiltorch (tools/gen_golden.py gen_hmc_wrapper): which
    trajectory positions become members for each (num_samples, L, burn) — including burn = -1 (the last L positions
    only) and burn = 0 (the initial position IS returned) —, how many, that they are independent modules, and the
    arguments the wrapper derives (one tau per tensor, inv_mass = 1/mass, tau_out = 1, burn = -1 passed down,
    the whole training set as one batch). hamiltorch's arithmetic itself stays unpinned."""
    import json
    import os
    g = json.load(open(os.path.join(golden_dir, 'hmc_wrapper.json')))
    call = g['call']
    assert call['burn'] == -1 and call['tau_out'] == 1.0 and call['model_loss'] == 'multi_class_linear_output'
    for case in g['cases']:
        torch.manual_seed(1)
        hyp = {'step_size': 1e-4, 'num_samples': case['num_samples'], 'L': case['L'], 'tau': 2.5, 'burn': case['burn'],
               'mass': 4.0}
        h = inference.HMC(dict(hyp), tiny_net(), tiny_loader(), kernels=OracleKernels(), use_graph=False, seed=3)
        assert h._wanted_indices() == case['selected'], case
        assert len(h.x) == call['x_rows'] and 1.0 / h.mass == call['inv_mass_unique'][0]
        assert all(t == h.tau for t in call['tau_list'])
        ens = h.sample()
        assert len(ens) == case['n_members'] and h.accepted == case['num_samples']      # tiny step: every proposal accepted
        assert case['independent'] and len({id(m) for m in ens}) == len(ens) and all(m is not h.model for m in ens)
        ptrs = {m._ursa_row.data_ptr() for m in ens}
        assert len(ptrs) == len(ens)
        if 0 in case['selected']:                     # burn = 0: the first member is the initial position itself
            assert torch.equal(torch.cat([p.detach().reshape(-1) for p in ens[0].parameters()]), theta0_of(hyp))


def theta0_of(hyp):
    torch.manual_seed(1)
    return torch.cat([p.detach().reshape(-1) for p in tiny_net().parameters()])


def test_host_philox_matches_oracle_and_known_answers():
    """The Metropolis uniform's generator: the host Philox4x32-10 of inference/hmc.py == the oracle's (itself pinned
    by Random123's known-answer vectors, tests/test_oracle_golden.py) on those vectors and on random counters."""
    import oracle_lib as O
    from ursabench_amd.inference.hmc import mh_uniform, philox4x32_10
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        assert philox4x32_10(ctr, key) == want == tuple(O.philox4x32_10(ctr, key))
    rng = np.random.default_rng(0)
    for _ in range(50):
        ctr, key = tuple(int(v) for v in rng.integers(0, 2 ** 32, 4)), tuple(int(v) for v in rng.integers(0, 2 ** 32, 2))
        assert list(philox4x32_10(ctr, key)) == O.philox4x32_10(ctr, key)
    us = [mh_uniform(7, k) for k in range(2000)]
    assert all(0.0 < u < 1.0 for u in us) and abs(np.mean(us) - 0.5) < 0.03 and len(set(us)) == 2000
    assert mh_uniform(7, 3) == mh_uniform(7, 3) != mh_uniform(8, 3)


def test_accept_sequence_is_the_chains_own():
    """the MH uniform came from the process-global torch.rand, so a chain's accept sequence
    depended on how many chains shared the process. Now it is Philox under the chain's key: a chain run alone and the
    same chain run interleaved with another (and with foreign torch.rand calls) accept the same proposals."""
    hyp = {'step_size': 0.5, 'num_samples': 12, 'L': 2, 'tau': 4.0, 'burn': 0, 'mass': 1.0}       # coarse step: some rejects

    def chain(seed):
        torch.manual_seed(5)
        return inference.HMC(dict(hyp, num_samples=1), Const(torch.randn(512)), const_loader(), kernels=OracleKernels(), seed=seed)
    alone = chain(21)
    seq_alone = []
    for _ in range(12):
        before = alone.accepted
        alone.sample()
        seq_alone.append(alone.accepted - before)
    a, b = chain(21), chain(22)
    seq_mixed = []
    for _ in range(12):
        before = a.accepted
        a.sample()
        torch.rand(3)
        b.sample()
        seq_mixed.append(a.accepted - before)
    assert seq_alone == seq_mixed and 0 < sum(seq_alone) < 12
