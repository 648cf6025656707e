"""Pin the CPU oracle (oracle/ursa_oracle.c) against golden vectors captured from the imported
reference (tools/gen_golden.py). CPU only."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert O.philox4x32_10([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert O.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert O.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_philox_normal_moments_and_independence():
    from scipy import stats
    z = O.philox_normal(1 << 20, seed=42, step=3).astype(np.float64)
    assert abs(z.mean()) < 4e-3 and abs(z.std() - 1) < 3e-3
    assert abs((z ** 3).mean()) < 2e-2 and abs((z ** 4).mean() - 3) < 5e-2
    assert stats.kstest(z[:200000], 'norm').pvalue > 1e-3
    z2 = O.philox_normal(1 << 20, seed=42, step=4).astype(np.float64)      # next step: independent stream
    z3 = O.philox_normal(1 << 20, seed=43, step=3).astype(np.float64)      # other chain
    assert abs(np.corrcoef(z, z2)[0, 1]) < 5e-3 and abs(np.corrcoef(z, z3)[0, 1]) < 5e-3
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 5e-3
    # prefix property: element i does not depend on n
    assert np.array_equal(O.philox_normal(37, 42, 3), O.philox_normal(1 << 20, 42, 3)[:37])


K1_CASES = ['sghmc_wd_noise', 'sghmc_nowd_mixed', 'sghmc_sched', 'sgld_wd_noise', 'sgld_nonoise']


@pytest.mark.parametrize('case', K1_CASES)
def test_k1_bitwise_vs_reference(golden_dir, case):
    g = np.load(os.path.join(golden_dir, 'k1_steps.npz'))
    momentum, wd, N = g[f'{case}/hyper']
    theta = g[f'{case}/theta0'].copy()
    mom = np.zeros_like(theta) if momentum != 0 else None
    for k, lr in enumerate(g[f'{case}/lr']):
        flags = (O.STEP_NOISE if g[f'{case}/noise'][k] else 0) | (O.STEP_WD if wd != 0 else 0)
        if k == 0 and momentum != 0:
            flags |= O.STEP_FIRST
        grad = g[f'{case}/grad'][k].copy()
        O.sgmcmc_step(theta, grad, mom, eps=g[f'{case}/eps'][k].copy(), flags=flags,
                      **O.step_scalars(float(lr), float(momentum), float(wd), N))
        assert np.array_equal(theta, g[f'{case}/theta'][k]), (case, k)
        if momentum != 0:
            assert np.array_equal(mom, g[f'{case}/mom'][k]), (case, k)


def test_k1_fusions_do_not_change_the_update():
    rng = np.random.default_rng(0)
    n = 1001
    th, gr, mo = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
    kw = dict(flags=O.STEP_NOISE | O.STEP_WD, seed=5, step=9, **O.step_scalars(0.01, 0.9, 4.0, 777))
    a = (th.copy(), gr.copy(), mo.copy())
    O.sgmcmc_step(*a, **kw)
    b = (th.copy(), gr.copy(), mo.copy())
    snap = np.empty_like(th)
    kw['flags'] |= O.STEP_ZERO_GRAD
    O.sgmcmc_step(*b, snapshot=snap, **kw)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    assert np.array_equal(snap, b[0]) and not b[1].any() and np.array_equal(a[1], gr)
    # philox mode == eps mode fed with the same stream
    c = (th.copy(), gr.copy(), mo.copy())
    kw['flags'] &= ~O.STEP_ZERO_GRAD
    O.sgmcmc_step(*c, eps=O.philox_normal(n, 5, 9), **kw)
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[2], c[2])


@pytest.mark.parametrize('mode', ['degenerate', 'counting'])
def test_k2_k3_bitwise_vs_reference(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, 'swag_moments.npz'))
    assert bool(g[f'{mode}/eps_replay_bitwise'])
    w = g['w']
    mean, sq = np.zeros(w.shape[1], np.float32), np.zeros(w.shape[1], np.float32)
    for k in range(w.shape[0]):
        n = k if mode == 'counting' else 0                       # SWAG never increments n
        O.swag_collect(mean, sq, w[k].copy(), decay=n / (n + 1.0), denom=n + 1.0)
        assert np.array_equal(mean, g[f'{mode}/mean'][k]) and np.array_equal(sq, g[f'{mode}/sq'][k]), k
    out = np.empty_like(mean)
    O.swag_draw(out, mean, sq, var_clamp=1e-30, eps=g[f'{mode}/eps'].copy())
    assert np.array_equal(out, g[f'{mode}/draw'])


@pytest.mark.parametrize('tag', ['c10', 'c100', 'mnist'])
def test_k5_vs_reference_tasks(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, 'tasks.npz'))
    z, zo = g[f'{tag}/logits'], g[f'{tag}/logits_out']
    S, N, C = z.shape
    gam = dict(one_minus_gamma=1 - 1e-4, gamma_over_c=1e-4 * 1 / C)
    rtol = 1e-5                                                   # north_star tolerance on fp32 probabilities
    # Prediction: raw-p sum + smoothed entropy sum (prediction.py:60-63); two calls like the fixture
    p, e = np.zeros((N, C), np.float32), np.zeros(N, np.float32)
    O.bma_accumulate(z[:1].copy(), p, e, smoothed=False, **gam)
    O.bma_accumulate(z[1:].copy(), p, e, smoothed=False, **gam)
    np.testing.assert_allclose(p, g[f'{tag}/pred_proba'], rtol=rtol, atol=1e-9)
    np.testing.assert_allclose(e, g[f'{tag}/pred_ent'], rtol=rtol, atol=1e-7)
    # OOD: smoothed sums on both loaders (ood_detection.py:62-65)
    for zz, kp, ke in ((z, 'ood_in_proba', 'ood_in_ent'), (zo, 'ood_out_proba', 'ood_out_ent')):
        p, e = np.zeros(zz.shape[1:], np.float32), np.zeros(zz.shape[1], np.float32)
        O.bma_accumulate(zz.copy(), p, e, smoothed=True, **gam)
        np.testing.assert_allclose(p, g[f'{tag}/{kp}'], rtol=rtol, atol=1e-9)
        np.testing.assert_allclose(e, g[f'{tag}/{ke}'], rtol=rtol, atol=1e-7)
    # Decision: smoothed sum + risk (decision_making.py:127-129)
    p, r = np.zeros((N, C), np.float32), np.zeros((N, C), np.float32)
    O.bma_accumulate(z.copy(), p, None, smoothed=True, risk_sum=r, cost=g[f'{tag}/dec_cost_mat'].copy(), **gam)
    np.testing.assert_allclose(p, g[f'{tag}/dec_proba'], rtol=rtol, atol=1e-9)
    np.testing.assert_allclose(r, g[f'{tag}/dec_risk'], rtol=rtol, atol=1e-7)
    assert np.array_equal((r / S).argmin(1), g[f'{tag}/dec_decision'])


def test_k4_leapfrog_physics():
    """hamiltorch is absent (parity unpinned): validate the restatement by invariants instead."""
    rng = np.random.default_rng(1)
    n = 513
    th0, p0 = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)

    def grad_logp(th):                      # standard normal target: grad log p = -theta
        return (-th).astype(np.float32)

    def run(th, p, eps, L):
        th, p = th.copy(), p.copy()
        O.leapfrog(th, p, grad_logp(th), kick_coef=0.5 * eps, step_size=eps, inv_mass=1.0, flags=O.LEAP_KICK)
        for _ in range(L):
            O.leapfrog(th, p, None, kick_coef=0.0, step_size=eps, inv_mass=1.0, flags=O.LEAP_DRIFT)
            O.leapfrog(th, p, grad_logp(th), kick_coef=eps, step_size=eps, inv_mass=1.0, flags=O.LEAP_KICK)
        O.leapfrog(th, p, grad_logp(th), kick_coef=-0.5 * eps, step_size=eps, inv_mass=1.0, flags=O.LEAP_KICK)
        return th, p

    def H(th, p):
        return 0.5 * O.sumsq(th) + 0.5 * O.sumsq(p)

    h0 = H(th0, p0)
    errs = []
    for eps in (0.1, 0.05):
        th, p = run(th0, p0, eps, int(round(1.0 / eps)))
        errs.append(abs(H(th, p) - h0))
    assert errs[1] < errs[0] / 3                                      # O(eps^2) energy error
    th, p = run(th0, p0, 0.05, 20)
    thb, pb = run(th, -p, 0.05, 20)                                   # reversibility
    np.testing.assert_allclose(thb, th0, atol=2e-5)
    np.testing.assert_allclose(-pb, p0, atol=2e-5)
    ke = O.leapfrog(th0.copy(), p0.copy(), None, kick_coef=0.0, step_size=0.0, inv_mass=2.0, flags=0, want_kinetic=True)
    np.testing.assert_allclose(ke, 0.5 * 2.0 * float((p0.astype(np.float64) ** 2).sum()), rtol=1e-12)


@pytest.mark.parametrize('case', ['sgd_mom_wd', 'sgd_mom_nowd', 'sgd_plain_wd'])
def test_sgd_mode_bitwise_vs_torch_sgd(golden_dir, case):
    """The SWA/SWAG trajectory optimizer (torch.optim.SGD, swa.py:41-42) as a K1 mode."""
    g = np.load(os.path.join(golden_dir, 'sgd_steps.npz'))
    momentum, wd = g[f'{case}/hyper']
    theta = g[f'{case}/theta0'].copy()
    mom = np.zeros_like(theta) if momentum != 0 else None
    for k, lr in enumerate(g[f'{case}/lr']):
        flags = O.STEP_SGD | (O.STEP_WD if wd != 0 else 0) | (O.STEP_FIRST if (k == 0 and momentum != 0) else 0)
        O.sgmcmc_step(theta, g[f'{case}/grad'][k].copy(), mom, lr=float(lr), mu=float(momentum), c_wd=float(wd),
                      c_noise=0.0, n_train=1.0, flags=flags)
        assert np.array_equal(theta, g[f'{case}/theta'][k]), (case, k)
        if momentum != 0:
            assert np.array_equal(mom, g[f'{case}/mom'][k]), (case, k)


@pytest.mark.parametrize('shape', [(32, 16, 16, 16), (16, 8, 8, 8), (7, 5, 3, 3), (64, 4, 1, 6)])
def test_bn_relu_restatement_vs_torch_cpu_batchnorm(shape):
    """K6's restatement against the ops the reference's networks execute on the CPU path (models/preresnet.py:40-41:
    torch's BatchNorm + ReLU and their autograd backward), on seeded inputs: the batch mean is the same float in every
    channel; invstd in most (torch's own variance is not exactly rounded) and within one unit in the last place in the
    rest; wherever both agree every output has the same bits; evaluation mode has the same bits everywhere."""
    import torch
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(sum(shape))
    C = shape[1]
    x = torch.randn(shape, generator=g) * 1.5 + 0.4
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    dy = torch.randn(shape, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    rm_t, rv_t = rm.clone(), rv.clone()
    xt, wt, bt = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out_t, m_t, i_t = torch.native_batch_norm(xt, wt, bt, rm_t, rv_t, True, 0.1, 1e-5)
    y_t = F.relu(out_t)
    y_t.backward(dy)
    rm_o, rv_o = rm.numpy().copy(), rv.numpy().copy()
    y, sm, si = O.bn_relu_fwd(x.numpy(), w.numpy(), b.numpy(), rm_o, rv_o, eps=1e-5, momentum=0.1, relu=True)
    assert np.array_equal(sm, m_t.detach().numpy())
    same = si == i_t.detach().numpy()
    assert same.sum() >= 0.6 * C, f'invstd equal in only {same.sum()} of {C} channels'
    np.testing.assert_allclose(si, i_t.detach().numpy(), rtol=1.3e-7)
    assert np.array_equal(y[:, same], y_t.detach().numpy()[:, same])
    np.testing.assert_allclose(rm_o, rm_t.numpy(), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rv_o, rv_t.numpy(), rtol=1e-6, atol=1e-7)
    # backward from torch's own saved statistics: same gates, sums in double on both sides
    dx, dg, db = O.bn_relu_bwd(x.numpy(), dy.numpy(), w.numpy(), b.numpy(), m_t.detach().numpy(), i_t.detach().numpy(), relu=True)
    # (torch's channel sums carry float accumulation error, ours are exact doubles: compare at the tensor's scale)
    np.testing.assert_allclose(db, bt.grad.numpy(), rtol=0, atol=1e-6 * float(bt.grad.abs().max()))
    np.testing.assert_allclose(dg, wt.grad.numpy(), rtol=0, atol=1e-6 * float(wt.grad.abs().max()))
    np.testing.assert_allclose(dx, xt.grad.numpy(), rtol=0, atol=1e-6 * float(xt.grad.abs().max()))
    y_e = O.bn_relu_eval(x.numpy(), w.numpy(), b.numpy(), rm.numpy(), rv.numpy(), eps=1e-5, relu=True)
    assert np.array_equal(y_e, F.relu(F.batch_norm(x, rm, rv, w, b, False, 0.0, 1e-5)).numpy())
