"""CPU: G16 (tests/golden/e2e_preresnet8_seeds.npz - the reference's PreResNet-8 SGHMC run for 8 seeds with its
near-zero ReLU gates, tools/gen_golden.py gen_e2e_preresnet_seeds) regenerates from torch's CPU generator, replays
through the oracle kernel set with the same pre-activations and gates, and the gate instruments do what they say."""
import json
import os

import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, TensorDataset

import oracle_lib as O
import ursabench_amd.inference as inference
from gate_lists import NearZeroGates, pack, unpack
from oracle_kernels import OracleKernels
from test_samplers_cpu import pad_eps
from ursabench_amd import fused_bn, models, tasks, util

SEEDS = list(range(8))


def g16_case(g, sd):
    """Inputs, initial network, per-step Langevin noise (flat, parameters() order) and gate lists of seed `sd`,
    regenerated the way tools/gen_golden.py made them; checksums verified."""
    gen = torch.Generator().manual_seed(9000 + sd)
    xtr, ytr = torch.randn(128, 3, 32, 32, generator=gen), torch.randint(0, 10, (128,), generator=gen)
    xte, yte = torch.randn(64, 3, 32, 32, generator=gen), torch.randint(0, 10, (64,), generator=gen)
    util.set_random_seed(sd)
    net = models.PreResNet(10, 8)
    theta0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    eps, eps_sums = [], []
    for k in range(4):
        torch.manual_seed(5000 + 100 * sd + k)
        e = [torch.randn_like(p) for p in net.parameters()]
        eps_sums.append(float(sum(t.double().sum() for t in e)))
        eps.append(torch.cat([t.reshape(-1) for t in e]))
    chk = [float(xtr.double().sum()), float(ytr.sum()), float(xte.double().sum()), float(yte.sum()),
           float(theta0.double().sum())] + eps_sums
    # float64 sums of 393,216 values: torch's reduction order follows the host's thread count, so compare to 1e-12
    assert np.allclose(chk, g[f's{sd}/checksums'], rtol=1e-12, atol=1e-9), 'torch CPU generator stream differs from the one the fixture was made with'
    train = DataLoader(TensorDataset(xtr, ytr), batch_size=128)
    test = DataLoader(TensorDataset(xte, yte), batch_size=64)
    return net, train, test, eps, unpack(g, f's{sd}/')


@pytest.mark.parametrize('sd', [0, 5])
def test_g16_replays_on_cpu_with_the_references_gates(golden_dir, sd):
    g = np.load(os.path.join(golden_dir, 'e2e_preresnet8_seeds.npz'))
    net, train, test, eps, gates = g16_case(g, sd)
    K = OracleKernels()
    log = NearZeroGates(net)
    s = inference.SGHMC(json.loads(str(g['hyper'])), net, train, kernels=K, use_graph=False)
    s.eps_provider = lambda k: pad_eps(s.arena, eps[k])
    steps = []
    orig = s.engine.forward_backward

    def fb(x, y):
        r = orig(x, y)
        steps.append(log.take())
        return r
    s.engine.forward_backward = fb
    ens = s.sample()
    got = pack(steps)
    for k in ('gate_idx', 'gate_open', 'gate_counts', 'n_open', 'numel'):
        assert np.array_equal(got[k], g[f's{sd}/{k}']), k
    for k, m in enumerate(ens):
        pred = tasks.Prediction({'in_distribution_test': test}, 10, torch.device('cpu'), 'ALL', kernels=K)
        pred.update_statistics([m], output_performance=False)
        np.testing.assert_allclose(pred.ensemble_proba.numpy(), g[f's{sd}/proba_step'][k], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(pred.expected_data_uncertainty.numpy(), g[f's{sd}/ent_step'][k], rtol=1e-5, atol=1e-6)


def test_oracle_gated_backward():
    """oracle_bn_relu_bwd_gated_f32: no list / padding only / every element's own gate == the plain backward; a listed
    gate overrides the recomputed one and moves dx, dgamma and dbeta of its channel only."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal((4, 3, 5, 5)).astype(np.float32)
    dy = rng.standard_normal(x.shape).astype(np.float32)
    w, b = (rng.random(3) + .5).astype(np.float32), (rng.standard_normal(3) * .3).astype(np.float32)
    y, sm, si = O.bn_relu_fwd(x, w, b, momentum=0.0)
    plain = O.bn_relu_bwd(x, dy, w, b, sm, si)
    own = (np.arange(x.size, dtype=np.int32), (y.reshape(-1) > 0).astype(np.uint8))
    for gates in (([], []), ([2 ** 31 - 1] * 3, [1, 0, 1]), own):
        for a, c in zip(plain, O.bn_relu_bwd(x, dy, w, b, sm, si, gates=gates)):
            assert np.array_equal(a, c)
    e = 1 * 25 + 7                                           # sample 0, channel 1
    flipped = O.bn_relu_bwd(x, dy, w, b, sm, si, gates=([e], [1 - own[1][e]]))
    assert not np.array_equal(plain[0][:, 1], flipped[0][:, 1])
    assert np.array_equal(plain[0][:, [0, 2]], flipped[0][:, [0, 2]])
    assert plain[2][1] != flipped[2][1] and plain[2][0] == flipped[2][0]
    # by hand: the flipped element's dy enters (or leaves) the sums
    d = dy.reshape(-1)[e] * (1 if own[1][e] == 0 else -1)
    assert flipped[2][1] == pytest.approx(plain[2][1] + d, rel=1e-6)


def test_gate_probe_bookkeeping():
    """GateProbe on host tensors: load() pads and validates, observe() reads this run's gates at the listed elements,
    collect() counts differing gates among the listed ones and corrects the open-gate count for them."""
    p = fused_bn.GateProbe(2, 4, 'cpu', force=True)
    with pytest.raises(ValueError):
        p.load([([3, 1], [0, 1]), ([], [])])                                      # not ascending
    with pytest.raises(ValueError):
        p.load([(list(range(5)), [0] * 5), ([], [])])                             # over capacity
    p.load([(np.array([1, 4], np.int32), np.array([1, 0], np.uint8)), (np.array([], np.int32), np.array([], np.uint8))])
    assert p.idx[0].tolist() == [1, 4, p.PAD, p.PAD] and p.open[0].tolist() == [1, 0, 0, 0]
    with fused_bn.probing(p):
        k0 = p.slot()
        p.observe(k0, torch.tensor([[0., 0., 1., 0., 2., 0.]]))                  # own gates at 1, 4: closed, open
        k1 = p.slot()
        p.observe(k1, torch.tensor([1., 1., 0.]))
        with pytest.raises(RuntimeError):
            p.slot()
    assert fused_bn._probe is None
    rec = p.collect()
    assert rec['flips'] == [2, 0] and rec['listed'] == [2, 0]
    assert rec['n_open_as_reference'] == [2 - (0 - 1) - (1 - 0), 2]
    assert p.gates(0)[0].data_ptr() == p.idx[0].data_ptr()
    assert fused_bn.GateProbe(1, 1, 'cpu', force=False).gates(0) is None
