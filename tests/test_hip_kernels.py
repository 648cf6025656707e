"""GPU parity tests proper: every kernel, called through the C ABI (ctypes -> libursa_hip.so),
against the CPU oracle on the same seeded inputs and against the committed golden vectors
captured from the reference. Bit-exact for the elementwise kernels (K1, K2, K3, Philox noise,
leapfrog); 1e-5 relative (north_star's fp32 tolerance) for the softmax/entropy reductions."""
import json
import os

import numpy as np
import pytest
import torch

import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def K():
    from ursabench_amd import _native
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    return _native.default_kernels()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def test_philox_normal_bitwise_vs_oracle(K):
    for n, seed, step in ((1, 1, 0), (3, 2, 1), (4, 3, 2), (1027, 0xdeadbeefcafe, 2 ** 40 + 5), (1 << 20, 42, 3)):
        out = torch.empty(n, device='cuda')
        K.philox_normal(out, seed=seed, step=step)
        assert np.array_equal(host(out), O.philox_normal(n, seed, step)), (n, seed, step)
    # unaligned output pointer (scalar store path)
    buf = torch.empty(1030, device='cuda')
    K.philox_normal(buf[1:1028], seed=9, step=9)
    assert np.array_equal(host(buf[1:1028]), O.philox_normal(1027, 9, 9))


def test_generator_fast_math_is_ieee_exact_on_every_input(K):
    """The in-register Box-Muller radius uses a 6-instruction division and a 9-instruction square root (csrc/ursa_rng.h).
    On this device, for ALL 2^32 Philox words, radius and logarithm equal the ones computed with the compiler's IEEE
    division / square root — the forms the scalar C oracle uses — bit for bit."""
    assert K.selftest_rng() == (0, 0)


K1_CASES = ['sghmc_wd_noise', 'sghmc_nowd_mixed', 'sghmc_sched', 'sgld_wd_noise', 'sgld_nonoise']


@pytest.mark.parametrize('case', K1_CASES)
@pytest.mark.parametrize('offset', [0, 1])          # 0: float4 path; 1: misaligned -> 4-byte path
def test_k1_bitwise_vs_reference_golden(K, golden_dir, case, offset):
    g = np.load(os.path.join(golden_dir, 'k1_steps.npz'))
    momentum, wd, N = g[f'{case}/hyper']
    n = int(g['n'])

    def slot(a=None):
        buf = torch.zeros(n + 8, device='cuda')
        v = buf[offset:offset + n]
        if a is not None:
            v.copy_(dev(a))
        return v

    theta = slot(g[f'{case}/theta0'])
    mom = slot() if momentum != 0 else None
    for k, lr in enumerate(g[f'{case}/lr']):
        flags = (O.STEP_NOISE if g[f'{case}/noise'][k] else 0) | (O.STEP_WD if wd != 0 else 0)
        if k == 0 and momentum != 0:
            flags |= O.STEP_FIRST
        K.sgmcmc_step(theta, slot(g[f'{case}/grad'][k]), mom, eps=slot(g[f'{case}/eps'][k]), flags=flags,
                      **O.step_scalars(float(lr), float(momentum), float(wd), N))
        assert np.array_equal(host(theta), g[f'{case}/theta'][k]), (case, k)
        if momentum != 0:
            assert np.array_equal(host(mom), g[f'{case}/mom'][k]), (case, k)


@pytest.mark.parametrize('n', [1, 2, 3, 4, 5, 63, 64, 255, 1024, 272282, 1 << 22])
@pytest.mark.parametrize('mu', [0.0, 0.9])
def test_k1_bitwise_vs_oracle_all_modes(K, n, mu):
    rng = np.random.default_rng(n * 7 + int(mu * 10))
    th0, gr0, mo0 = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
    eps = rng.standard_normal(n).astype(np.float32)
    sc = O.step_scalars(0.05, mu, 4.0, 50000)
    for flags, use_eps, fuse in ((0, False, False), (O.STEP_NOISE | O.STEP_WD, True, False),
                                 (O.STEP_NOISE | O.STEP_WD, False, True), (O.STEP_NOISE | O.STEP_FIRST, False, False),
                                 (O.STEP_WD | O.STEP_ZERO_GRAD, False, True)):
        a = [th0.copy(), gr0.copy(), mo0.copy() if mu else None]
        snap_o = np.empty_like(th0) if fuse else None
        if fuse:
            flags |= O.STEP_ZERO_GRAD
        O.sgmcmc_step(a[0], a[1], a[2], eps=eps.copy() if use_eps else None, snapshot=snap_o, flags=flags,
                      seed=1234567, step=77, **sc)
        b = [dev(th0), dev(gr0), dev(mo0) if mu else None]
        snap_d = torch.empty(n, device='cuda') if fuse else None
        K.sgmcmc_step(b[0], b[1], b[2], eps=dev(eps) if use_eps else None, snapshot=snap_d, flags=flags,
                      seed=1234567, step=77, **sc)
        assert np.array_equal(host(b[0]), a[0]), (n, mu, flags)
        assert np.array_equal(host(b[1]), a[1]), (n, mu, flags)
        if mu:
            assert np.array_equal(host(b[2]), a[2]), (n, mu, flags)
        if fuse:
            assert np.array_equal(host(snap_d), snap_o)


def test_k1_trajectory_bitwise_philox(K):
    """20 steps in production (Philox) mode: the GPU trajectory equals the oracle's bit for bit."""
    n = 61706
    rng = np.random.default_rng(5)
    th, mo = rng.standard_normal(n).astype(np.float32), np.zeros(n, np.float32)
    dth, dmo = dev(th), dev(mo)
    for k in range(20):
        gr = rng.standard_normal(n).astype(np.float32)
        sc = O.step_scalars(0.1 * (1 - k / 40), 0.5, 4.0, 50000)
        flags = O.STEP_NOISE | O.STEP_WD | (O.STEP_FIRST if k == 0 else 0)
        O.sgmcmc_step(th, gr.copy(), mo, flags=flags, seed=3, step=k, **sc)
        K.sgmcmc_step(dth, dev(gr), dmo, flags=flags, seed=3, step=k, **sc)
    assert np.array_equal(host(dth), th) and np.array_equal(host(dmo), mo)


def _ctl_tensor(*blocks):
    """Device array of control blocks from keyword dicts."""
    from ursabench_amd._native import StepCtl
    raw = b''.join(bytes(StepCtl(**kw)) for kw in blocks)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).cuda()


def _ctl_back(ctl, k=0):
    from ursabench_amd._native import CTL_BYTES, StepCtl
    return StepCtl.from_buffer_copy(bytes(ctl.cpu().numpy())[k * CTL_BYTES:(k + 1) * CTL_BYTES])


@pytest.mark.parametrize('self_advance', [True, False])
def test_k1_ctl_variant_and_advance(K, self_advance):
    """Device-control-block launch (graph-replayable) == the scalar-argument launch == the oracle, bit for bit; the
    advance (folded into the update launch: last retiring workgroup, or the explicit 1-thread launch) walks the
    (lr, c_noise) schedule table whose address rides in the block, clears FIRST and re-arms the ticket."""
    from ursabench_amd._native import STEP_ADVANCE
    n = 3 * 2048 + 4 + 2                                   # several workgroups at every block size + a scalar tail
    rng = np.random.default_rng(11)
    th0, mo0 = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    lrs = [0.1, 0.07, 0.03]
    sched = np.array([[lr, np.sqrt(2 * (1 - 0.5) * lr)] for lr in lrs], np.float32)
    buf = [torch.zeros(n + 2, device='cuda') for _ in range(6)]        # n is not a multiple of 4: keep bases aligned
    a = [buf[0][:n], buf[1][:n]]
    b = [buf[2][:n], buf[3][:n]]
    for t, v in ((a[0], th0), (a[1], mo0), (b[0], th0), (b[1], mo0)):
        t.copy_(dev(v))
    oth, omo = th0.copy(), mo0.copy()
    sc0 = O.step_scalars(lrs[0], 0.5, 4.0, 1000)
    dsched = dev(sched)
    base = O.STEP_NOISE | O.STEP_WD | O.STEP_ZERO_GRAD
    ctl = _ctl_tensor(dict(lr=sc0['lr'], mu=0.5, c_wd=sc0['c_wd'], c_noise=float(sched[0, 1]), n_train=1000.0,
                           flags=base | O.STEP_FIRST | (STEP_ADVANCE if self_advance else 0), seed=99, step=0,
                           sched=dsched.data_ptr(), sched_len=3))
    for k in range(5):
        gr = rng.standard_normal(n).astype(np.float32)
        ga, gb = buf[4][:n], buf[5][:n]
        ga.copy_(dev(gr)); gb.copy_(dev(gr))
        flags = base | (O.STEP_FIRST if k == 0 else 0)
        lr, cn = sched[k % 3]
        kw = dict(lr=float(lr), mu=0.5, c_wd=sc0['c_wd'], c_noise=float(cn), n_train=1000.0, flags=flags, seed=99, step=k)
        K.sgmcmc_step(a[0], ga, a[1], **kw)
        O.sgmcmc_step(oth, gr.copy(), omo, **kw)
        K.sgmcmc_step_ctl(b[0], gb, b[1], ctl)
        if not self_advance:
            K.step_ctl_advance(ctl)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and not gb.any(), k
        assert np.array_equal(host(b[0]), oth) and np.array_equal(host(b[1]), omo), k
    back = _ctl_back(ctl)
    assert back.step == 5 and not (back.flags & O.STEP_FIRST) and back.lr == sched[5 % 3, 0] and back.tickets_clear()
    assert back.c_noise == sched[5 % 3, 1] and back.sched == dsched.data_ptr()


@pytest.mark.parametrize('n,stride', [(273408, 273408), (61706, 61760), (5, 8), (1 << 23, 1 << 23)])
@pytest.mark.parametrize('inject', [False, True])
def test_k1_multi_chain_launch_bitwise(K, n, stride, inject):
    """ONE launch over a [K, stride] slab (SURVEY.md 8b n_chains): chain k == the oracle's single-chain update with
    ctl[k]'s own scalars, Philox key and call index, for SGHMC / SGLD / noise-off / scheduled chains side by side; every
    chain's block advances itself. (1 << 23) x 3 chains x 12 B is past the 256 MiB non-temporal threshold."""
    from ursabench_amd._native import STEP_ADVANCE
    if inject and n >= (1 << 23):
        pytest.skip('injected noise is covered at the smaller sizes')
    Kc, steps = 4 if n < (1 << 23) else 3, 3
    rng = np.random.default_rng(n % 1000 + 7)
    th = rng.standard_normal((Kc, stride)).astype(np.float32)
    mo = rng.standard_normal((Kc, stride)).astype(np.float32)
    sched = np.array([[0.05, 0.21], [0.04, 0.2], [0.03, 0.17]], np.float32)
    dsched = dev(sched)
    chains = [dict(lr=0.1, mu=0.5, c_wd=8e-5, c_noise=0.3162, n_train=50000.0, flags=O.STEP_NOISE | O.STEP_WD | O.STEP_FIRST, seed=11),
              dict(lr=0.02, mu=0.0, c_wd=4e-5, c_noise=0.2, n_train=1000.0, flags=O.STEP_NOISE | O.STEP_WD, seed=12),
              dict(lr=0.05, mu=0.9, c_wd=0.0, c_noise=0.1, n_train=10.0, flags=0, seed=13),
              dict(lr=float(sched[0, 0]), mu=0.3, c_wd=1e-4, c_noise=float(sched[0, 1]), n_train=64.0,
                   flags=O.STEP_NOISE | O.STEP_WD, seed=14, sched=dsched.data_ptr(), sched_len=3)][:Kc]
    ctl = _ctl_tensor(*[dict(c, flags=c['flags'] | STEP_ADVANCE, step=100 * k, sched_base=100 * k) for k, c in enumerate(chains)])
    dth, dmo = dev(th), dev(mo)
    dgr = torch.empty_like(dth)
    deps = torch.empty_like(dth) if inject else None
    for it in range(steps):
        gr = rng.standard_normal((Kc, stride)).astype(np.float32)
        ep = rng.standard_normal((Kc, stride)).astype(np.float32) if inject else None
        dgr.copy_(dev(gr))
        if inject:
            deps.copy_(dev(ep))
        K.sgmcmc_step_multi(dth, dgr, dmo, ctl, n_per_chain=n, eps=deps)
        for k, c in enumerate(chains):
            kw = {f: c[f] for f in ('lr', 'mu', 'c_wd', 'c_noise', 'n_train', 'seed')}
            flags = c['flags'] & ~(O.STEP_FIRST if it else 0)
            if 'sched' in c:
                kw['lr'], kw['c_noise'] = float(sched[it % 3, 0]), float(sched[it % 3, 1])
            O.sgmcmc_step(th[k, :n], gr[k, :n].copy(), mo[k, :n] if c['mu'] else None, flags=flags, step=100 * k + it,
                          eps=None if not inject else ep[k, :n], **kw)
    got_th, got_mo = host(dth), host(dmo)
    for k, c in enumerate(chains):
        assert np.array_equal(got_th[k, :n], th[k, :n]), k
        if c['mu']:
            assert np.array_equal(got_mo[k, :n], mo[k, :n]), k
        assert np.array_equal(got_th[k, n:], th[k, n:]) and np.array_equal(got_mo[k, n:], mo[k, n:])     # row pads untouched
        b = _ctl_back(ctl, k)
        assert b.step == 100 * k + steps and b.tickets_clear() and not (b.flags & O.STEP_FIRST)
    # argument checks happen on the host, before any launch
    with pytest.raises(ValueError):
        K.sgmcmc_step_multi(dth, dgr[:, :-4].contiguous(), dmo, ctl)
    with pytest.raises(ValueError):
        K.sgmcmc_step_multi(dth, dgr, dmo, ctl[:-8])


@pytest.mark.parametrize('mode', ['degenerate', 'counting'])
def test_k2_k3_bitwise_vs_reference_golden(K, golden_dir, mode):
    g = np.load(os.path.join(golden_dir, 'swag_moments.npz'))
    w = g['w']
    mean, sq = torch.zeros(w.shape[1], device='cuda'), torch.zeros(w.shape[1], device='cuda')
    for k in range(w.shape[0]):
        n = k if mode == 'counting' else 0
        K.swag_collect(mean, sq, dev(w[k]), decay=n / (n + 1.0), denom=n + 1.0)
        assert np.array_equal(host(mean), g[f'{mode}/mean'][k]) and np.array_equal(host(sq), g[f'{mode}/sq'][k])
    out = torch.empty_like(mean)
    K.swag_draw(out, mean, sq, var_clamp=1e-30, eps=dev(g[f'{mode}/eps']))
    assert np.array_equal(host(out), g[f'{mode}/draw'])
    # the ensemble form (standard deviation stored once, square-root-free per-member draw) gives the same bits
    sd, out2 = torch.empty_like(mean), torch.empty_like(mean)
    K.swag_std(sd, mean, sq, var_clamp=1e-30)
    K.swag_draw_std(out2, mean, sd, eps=dev(g[f'{mode}/eps']))
    assert np.array_equal(host(out2), g[f'{mode}/draw'])


@pytest.mark.parametrize('n', [1, 7, 1000, 36546980 // 16 + 3])
def test_k2_k3_bitwise_vs_oracle(K, n):
    rng = np.random.default_rng(n)
    mean, w = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    sq = (mean ** 2 + rng.random(n).astype(np.float32) * 0.1).astype(np.float32)
    sq[::5] = mean[::5] ** 2 - 0.01                                  # negative variance -> clamp
    dm, ds = dev(mean), dev(sq)
    O.swag_collect(mean, sq, w, decay=3 / 4.0, denom=4.0)
    K.swag_collect(dm, ds, dev(w), decay=3 / 4.0, denom=4.0)
    assert np.array_equal(host(dm), mean) and np.array_equal(host(ds), sq)
    for eps in (None, rng.standard_normal(n).astype(np.float32)):
        o, d = np.empty(n, np.float32), torch.empty(n, device='cuda')
        O.swag_draw(o, mean, sq, var_clamp=1e-30, scale=0.5, seed=8, draw=2, eps=eps)
        K.swag_draw(d, dm, ds, var_clamp=1e-30, scale=0.5, seed=8, draw=2, eps=None if eps is None else dev(eps))
        assert np.array_equal(host(d), o)
        for off in (0, 1):                                            # std hoisted out of the draw; float4 and 4-byte paths
            buf = [torch.zeros(n + 8, device='cuda') for _ in range(4)]
            m_, s_, sd, d2 = (b[off:off + n] for b in buf)
            m_.copy_(dm); s_.copy_(ds)
            K.swag_std(sd, m_, s_, var_clamp=1e-30, scale=0.5)
            K.swag_draw_std(d2, m_, sd, seed=8, draw=2, eps=None if eps is None else dev(eps))
            assert np.array_equal(host(d2), o), off


@pytest.mark.parametrize('tag', ['c10', 'c100', 'mnist'])
def test_k5_vs_reference_golden(K, golden_dir, tag):
    g = np.load(os.path.join(golden_dir, 'tasks.npz'))
    z, zo = g[f'{tag}/logits'], g[f'{tag}/logits_out']
    S, N, C = z.shape
    gam = dict(one_minus_gamma=1 - 1e-4, gamma_over_c=1e-4 * 1 / C)
    rtol = 1e-5
    p, e = torch.zeros(N, C, device='cuda'), torch.zeros(N, device='cuda')
    K.bma_accumulate(dev(z[:1]), p, e, smoothed=False, **gam)
    K.bma_accumulate(dev(z[1:]), p, e, smoothed=False, **gam)
    np.testing.assert_allclose(host(p), g[f'{tag}/pred_proba'], rtol=rtol, atol=1e-9)
    np.testing.assert_allclose(host(e), g[f'{tag}/pred_ent'], rtol=rtol, atol=1e-7)
    for zz, kp, ke in ((z, 'ood_in_proba', 'ood_in_ent'), (zo, 'ood_out_proba', 'ood_out_ent')):
        p, e = torch.zeros(zz.shape[1], C, device='cuda'), torch.zeros(zz.shape[1], device='cuda')
        K.bma_accumulate(dev(zz), p, e, smoothed=True, **gam)
        np.testing.assert_allclose(host(p), g[f'{tag}/{kp}'], rtol=rtol, atol=1e-9)
        np.testing.assert_allclose(host(e), g[f'{tag}/{ke}'], rtol=rtol, atol=1e-7)
    p, r = torch.zeros(N, C, device='cuda'), torch.zeros(N, C, device='cuda')
    K.bma_accumulate(dev(z), p, None, smoothed=True, risk_sum=r, cost=dev(g[f'{tag}/dec_cost_mat']), **gam)
    np.testing.assert_allclose(host(p), g[f'{tag}/dec_proba'], rtol=rtol, atol=1e-9)
    np.testing.assert_allclose(host(r), g[f'{tag}/dec_risk'], rtol=rtol, atol=1e-7)
    assert np.array_equal(host(r / S).argmin(1), g[f'{tag}/dec_decision'])


@pytest.mark.parametrize('S,B,C', [(1, 1, 1), (2, 5, 2), (3, 100, 3), (4, 1000, 10), (2, 333, 17), (3, 257, 64),
                                   (30, 512, 100), (2, 65, 129), (2, 40, 1000), (1, 3, 1024), (5, 10000, 10),
                                   # the member-range forms of the float4 lane-group kernel (bma_form in ursa_kernels.hip):
                                   # 2 ranges (many row groups, 8 classes per lane), 1 range (16 per lane), 8 ranges (few
                                   # rows, ragged ranges), 4 ranges with fewer members than ranges
                                   (5, 8200, 100), (4, 8200, 132), (17, 100, 100), (3, 9000, 100), (30, 1001, 36)])
def test_k5_vs_oracle_shapes(K, S, B, C):
    rng = np.random.default_rng(S * 1000 + C)
    z = (rng.standard_normal((S, B, C)) * 4).astype(np.float32)
    z[0, 0, :] = 50.0 * rng.standard_normal(C)                       # a very peaked row
    cost = rng.random((C, C)).astype(np.float32)
    gam = dict(one_minus_gamma=1 - 1e-4, gamma_over_c=1e-4 * 1 / C)
    for smoothed, risk in ((False, False), (True, True)):
        p0 = rng.random((B, C)).astype(np.float32)                    # accumulate on top of existing sums
        e0 = rng.random(B).astype(np.float32)
        r0 = rng.random((B, C)).astype(np.float32)
        po, eo, ro = p0.copy(), e0.copy(), r0.copy()
        O.bma_accumulate(z, po, eo, smoothed=smoothed, risk_sum=ro if risk else None, cost=cost if risk else None, **gam)
        pd, ed, rd = dev(p0), dev(e0), dev(r0)
        K.bma_accumulate(dev(z), pd, ed, smoothed=smoothed, risk_sum=rd if risk else None,
                         cost=dev(cost) if risk else None, **gam)
        np.testing.assert_allclose(host(pd), po, rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(host(ed), eo, rtol=1e-5, atol=1e-6)
        if risk:
            np.testing.assert_allclose(host(rd), ro, rtol=1e-5, atol=1e-6)
        else:
            assert np.array_equal(host(rd), r0)
    # property at full size: probabilities of each member sum to one => row sums grow by exactly S
    p = torch.zeros(B, C, device='cuda')
    K.bma_accumulate(dev(z), p, None, smoothed=False, **gam)
    np.testing.assert_allclose(host(p.sum(1)), np.full(B, S, np.float32), rtol=2e-6)


def _k5_check(K, z, smoothed, risk, rng, rtol=1e-5):
    S, B, C = z.shape
    cost = rng.random((C, C)).astype(np.float32)
    gam = dict(one_minus_gamma=1 - 1e-4, gamma_over_c=1e-4 * 1 / C)
    p0, e0, r0 = rng.random((B, C)).astype(np.float32), rng.random(B).astype(np.float32), rng.random((B, C)).astype(np.float32)
    po, eo, ro = p0.copy(), e0.copy(), r0.copy()
    O.bma_accumulate(z, po, eo, smoothed=smoothed, risk_sum=ro if risk else None, cost=cost if risk else None, **gam)
    pd, ed, rd = dev(p0), dev(e0), dev(r0)
    K.bma_accumulate(dev(z), pd, ed, smoothed=smoothed, risk_sum=rd if risk else None, cost=dev(cost) if risk else None, **gam)
    np.testing.assert_allclose(host(pd), po, rtol=rtol, atol=1e-8)
    np.testing.assert_allclose(host(ed), eo, rtol=rtol, atol=1e-6)
    if risk:
        np.testing.assert_allclose(host(rd), ro, rtol=rtol, atol=1e-6)
    else:
        assert np.array_equal(host(rd), r0)
    return host(pd), host(ed), host(rd)


@pytest.mark.parametrize('C', [1, 2, 3, 4, 5, 7, 8, 9, 10, 12, 16])
@pytest.mark.parametrize('S,B', [(1, 16), (3, 128), (16, 48), (17, 20), (50, 1000), (64, 16), (130, 36), (5, 4), (33, 10000)])
def test_k5_rowlane_few_classes(K, S, B, C):
    """C <= 16 with 16-byte aligned member tiles (B*C % 4 == 0) runs k_bma_rowlane (one lane per row and member
    slot, LDS-staged tiles): every member-slot split (S < slots, S = slots, several rounds, several chunks),
    partial last tile, risk on/off — against the oracle; and bit-for-bit against nothing: the fold order is
    slot order, so only the 1e-5 bar applies."""
    if (B * C) % 4:
        B = B + (4 - B % 4) % 4 if C % 2 else B + B % 2       # make B*C a multiple of 4 so the row-lane kernel is the one tested
    assert (B * C) % 4 == 0
    rng = np.random.default_rng(S * 131 + B * 7 + C)
    z = (rng.standard_normal((S, B, C)) * 4).astype(np.float32)
    z[0, 0, :] = 50.0 * rng.standard_normal(C)
    _k5_check(K, z, False, False, rng)
    _k5_check(K, z, True, True, rng)
    p = torch.zeros(B, C, device='cuda')
    K.bma_accumulate(dev(z), p, None, smoothed=False, one_minus_gamma=1 - 1e-4, gamma_over_c=1e-4 / C)
    np.testing.assert_allclose(host(p.sum(1)), np.full(B, S, np.float32), rtol=2e-6)


@pytest.mark.parametrize('S,B,C', [(3, 7, 20), (2, 129, 32), (4, 65, 64), (30, 257, 100), (5, 31, 104), (9, 16, 112), (1, 1, 100),
                                   (2, 33, 128), (3, 17, 200), (2, 9, 256)])
def test_k5_float4_rows(K, S, B, C):
    """C % 4 == 0, 16 < C <= 256: the lane-group kernel with four consecutive classes per float4 load."""
    rng = np.random.default_rng(S * 17 + B + C)
    z = (rng.standard_normal((S, B, C)) * 4).astype(np.float32)
    _k5_check(K, z, False, False, rng)
    _k5_check(K, z, True, True, rng)


@pytest.mark.parametrize('S,B,C', [(50, 1000, 10), (7, 64, 16), (30, 256, 100), (3, 64, 64)])
def test_k5_fast_paths_agree_with_the_lane_group_kernel(K, S, B, C, monkeypatch):
    """The knobs build (csrc/libursa_hip_knobs.so; the shipped library reads no environment) selects the generic
    lane-group kernel (scalar loads); both must satisfy the same bar and agree with each other to rounding. The shipped
    library must ignore the same variables."""
    from ursabench_amd import _native
    rng = np.random.default_rng(5)
    z = (rng.standard_normal((S, B, C)) * 3).astype(np.float32)
    fast = _k5_check(K, z, True, True, np.random.default_rng(9))
    monkeypatch.setenv('URSA_BMA_NO_ROWLANE', '1')
    monkeypatch.setenv('URSA_BMA_NO_V4', '1')
    same = _k5_check(K, z, True, True, np.random.default_rng(9))
    for a, b in zip(fast, same):
        assert np.array_equal(a, b)                       # shipped library: the variables change nothing
    slow = _k5_check(_native.knobs_kernels(), z, True, True, np.random.default_rng(9))
    for a, b in zip(fast, slow):
        np.testing.assert_allclose(a, b, rtol=2e-6, atol=1e-7)


@pytest.mark.parametrize('C,B', [(10, 64), (100, 16), (7, 5)])
def test_k5_masked_classes_and_zero_smoothing(K, C, B):
    """a -inf logit (masked class) contributes probability 0 like the reference's log_softmax().exp()
    — not NaN — and with gamma_over_c = 0 (allowed by the header) an underflowing probability contributes
    0 ln 0 = 0 to the entropy."""
    rng = np.random.default_rng(C)
    z = (rng.standard_normal((3, B, C)) * 2).astype(np.float32)
    z[:, :, 1] = -np.inf
    z[1, 0, :] = -300.0
    z[1, 0, 0] = 0.0                                        # 300-logit gap: p underflows to exactly 0
    for omg, goc in ((1 - 1e-4, 1e-4 / C), (1.0, 0.0)):
        po, eo = np.zeros((B, C), np.float32), np.zeros(B, np.float32)
        O.bma_accumulate(z, po, eo, one_minus_gamma=omg, gamma_over_c=goc, smoothed=False)
        pd, ed = torch.zeros(B, C, device='cuda'), torch.zeros(B, device='cuda')
        K.bma_accumulate(dev(z), pd, ed, one_minus_gamma=omg, gamma_over_c=goc, smoothed=False)
        assert np.isfinite(host(pd)).all() and np.isfinite(host(ed)).all() and np.isfinite(po).all() and np.isfinite(eo).all()
        assert (host(pd)[:, 1] == 0).all()
        np.testing.assert_allclose(host(pd), po, rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(host(ed), eo, rtol=1e-5, atol=1e-6)


def test_k5_empty_and_errors(K):
    p = torch.zeros(4, 10, device='cuda')
    K.bma_accumulate(torch.zeros(0, 4, 10, device='cuda'), p, None, one_minus_gamma=0.9999, gamma_over_c=1e-5, smoothed=False)
    assert not p.any()
    with pytest.raises(ValueError):
        K.bma_accumulate(torch.zeros(1, 4, 10, device='cuda'), p, None, one_minus_gamma=0.9999, gamma_over_c=1e-5,
                         smoothed=True, risk_sum=torch.zeros(4, 10, device='cuda'))        # risk without cost
    with pytest.raises(ValueError):
        K.bma_accumulate(torch.zeros(1, 4, 2000, device='cuda'), torch.zeros(4, 2000, device='cuda'), None,
                         one_minus_gamma=0.9999, gamma_over_c=1e-5, smoothed=True)          # C too large
    with pytest.raises(ValueError):
        K.bma_accumulate(torch.zeros(1, 5, 10, device='cuda'), p, None, one_minus_gamma=0.9999, gamma_over_c=1e-5,
                         smoothed=True)                                                      # shape mismatch


@pytest.mark.parametrize('n', [1, 5, 1023, 1726388])
def test_k4_leapfrog_vs_oracle(K, n):
    rng = np.random.default_rng(n)
    th, p, g = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
    dth, dp, dg = dev(th), dev(p), dev(g)
    ws = torch.empty(2048, device='cuda')
    ke = torch.zeros(1, device='cuda')
    k_o = O.leapfrog(th, p, g, kick_coef=0.5e-3, step_size=1e-3, inv_mass=2.0, flags=O.LEAP_KICK | O.LEAP_DRIFT,
                     want_kinetic=True)
    K.leapfrog(dth, dp, dg, kick_coef=0.5e-3, step_size=1e-3, inv_mass=2.0, flags=O.LEAP_KICK | O.LEAP_DRIFT,
               kinetic_out=ke, ws=ws)
    assert np.array_equal(host(dth), th) and np.array_equal(host(dp), p)
    np.testing.assert_allclose(float(ke), k_o, rtol=2e-6)
    out = torch.zeros(1, device='cuda')
    K.sumsq(dth, out, ws)
    np.testing.assert_allclose(float(out), O.sumsq(th), rtol=2e-6)
    # deterministic reduction: same bits on a second launch
    out2 = torch.zeros(1, device='cuda')
    K.sumsq(dth, out2, ws)
    assert torch.equal(out, out2)


def test_k1_roofline_sized_properties(K):
    """Full-size arena (2^26 elements, > Infinity Cache): size-independent properties.
    (a) noise off, wd off, mu=0: theta' == theta - lr*g exactly (linearity);
    (b) Philox noise statistics over the whole arena; (c) ZERO_GRAD leaves grad all-zero."""
    n = 1 << 26
    th = torch.randn(n, device='cuda')
    g = torch.randn(n, device='cuda')
    ref = th + g * (-0.125)
    K.sgmcmc_step(th, g, None, lr=0.125, mu=0.0, c_wd=0.0, c_noise=0.0, n_train=1.0, flags=0)
    assert torch.equal(th, ref)
    th.zero_(); g.zero_()
    K.sgmcmc_step(th, g, None, lr=0.5, mu=0.0, c_wd=0.0, c_noise=1.0, n_train=1.0, flags=O.STEP_NOISE | O.STEP_ZERO_GRAD,
                  seed=7, step=1)
    assert abs(float(th.mean())) < 1e-3 and abs(float(th.std()) - 1) < 1e-3 and not g.any()
    assert float(th.abs().max()) < 6.8
    assert np.array_equal(host(th[:4096]), O.philox_normal(4096, 7, 1))


@pytest.mark.parametrize('case', ['sgd_mom_wd', 'sgd_mom_nowd', 'sgd_plain_wd'])
def test_sgd_mode_bitwise_vs_torch_sgd_golden(K, golden_dir, case):
    g = np.load(os.path.join(golden_dir, 'sgd_steps.npz'))
    momentum, wd = g[f'{case}/hyper']
    theta = dev(g[f'{case}/theta0'])
    mom = torch.zeros_like(theta) if momentum != 0 else None
    for k, lr in enumerate(g[f'{case}/lr']):
        flags = O.STEP_SGD | (O.STEP_WD if wd != 0 else 0) | (O.STEP_FIRST if (k == 0 and momentum != 0) else 0)
        K.sgmcmc_step(theta, dev(g[f'{case}/grad'][k]), mom, lr=float(lr), mu=float(momentum), c_wd=float(wd),
                      c_noise=0.0, n_train=1.0, flags=flags)
        assert np.array_equal(host(theta), g[f'{case}/theta'][k]), (case, k)
        if momentum != 0:
            assert np.array_equal(host(mom), g[f'{case}/mom'][k]), (case, k)
    with pytest.raises(ValueError):
        K.sgmcmc_step(theta, theta.clone(), mom, lr=0.1, mu=float(momentum), c_wd=0.0, c_noise=0.0, n_train=1.0,
                      flags=O.STEP_SGD | O.STEP_NOISE)


def test_k1_beyond_2_31_elements(K):
    """Maximum-size edge: an arena of 2^31 + 4100 elements (17 GB of theta + grad): 64-bit indexing in the
    grid/element arithmetic and in the Philox counter. Noise-only SGLD update on zeros => theta is the noise
    stream itself; both ends are compared with the oracle bit for bit, the bulk statistically."""
    n = (1 << 31) + 4100
    free, _ = torch.cuda.mem_get_info()
    if free < 3 * n * 4:
        pytest.skip('needs 26 GB of free HBM')
    th = torch.zeros(n, device='cuda')
    g = torch.zeros(n, device='cuda')
    K.sgmcmc_step(th, g, None, lr=0.5, mu=0.0, c_wd=0.0, c_noise=1.0, n_train=1.0, flags=O.STEP_NOISE, seed=21, step=5)
    assert np.array_equal(host(th[:4096]), O.philox_normal_range(0, 4096, 21, 5))
    tail0 = ((1 << 31) - 8)
    assert np.array_equal(host(th[tail0:]), O.philox_normal_range(tail0, n - tail0, 21, 5))      # crosses 2^31, scalar tail
    m, sd = float(th.mean()), float(th.std())
    assert abs(m) < 2e-4 and abs(sd - 1) < 2e-4 and float(th.abs().max()) < 6.8
    # the two halves of the arena are different streams (no 32-bit wrap of the counter)
    assert not torch.equal(th[:4096], th[(1 << 31):(1 << 31) + 4096])


def test_k5_many_rows_grid_loop(K):
    """More row tiles than the grid cap (2^20 blocks of 4 rows): the kernel's outer loop over row tiles.
    Property at full size: every member's probabilities sum to one, so row sums grow by exactly S; the first
    and last rows are compared with the oracle."""
    S, B, C = 2, (1 << 22) + 5, 10
    z = torch.randn(S, B, C, device='cuda') * 3
    p, e = torch.zeros(B, C, device='cuda'), torch.zeros(B, device='cuda')
    gam = dict(one_minus_gamma=1 - 1e-4, gamma_over_c=1e-4 * 1 / C)
    K.bma_accumulate(z, p, e, smoothed=False, **gam)
    rs = p.sum(1)
    assert float((rs - S).abs().max()) < 1e-5 and float(e.min()) > 0
    for sl in (slice(0, 64), slice(B - 64, B)):
        zo = np.ascontiguousarray(host(z[:, sl]))
        po, eo = np.zeros((64, C), np.float32), np.zeros(64, np.float32)
        O.bma_accumulate(zo, po, eo, smoothed=False, **gam)
        np.testing.assert_allclose(host(p[sl]), po, rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(host(e[sl]), eo, rtol=1e-5, atol=1e-6)


def test_k2_k3_full_size_properties(K):
    """WideResNet-28-10 size (36,546,980 parameters, BASELINE configs[3]). Properties: collecting the SAME
    iterate twice (n = 0, then n = 1) leaves mean = w and sq = w*w exactly (halves add back exactly in fp32);
    with zero variance the draw is the mean up to the 1e-30 clamp; a unit-variance draw follows our Philox
    stream at both ends of the vector."""
    n = 36546980
    w = torch.randn(n, device='cuda')
    mean, sq = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    K.swag_collect(mean, sq, w, decay=0.0, denom=1.0)
    K.swag_collect(mean, sq, w, decay=1 / 2.0, denom=2.0)
    assert torch.equal(mean, w) and torch.equal(sq, w * w)
    out = torch.empty(n, device='cuda')
    K.swag_draw(out, mean, sq, var_clamp=1e-30, seed=3, draw=0)
    assert float((out - mean).abs().max()) < 1e-13
    K.swag_draw(out, torch.zeros_like(mean), torch.ones_like(sq), var_clamp=1e-30, seed=3, draw=1)
    assert np.array_equal(host(out[:1024]), O.philox_normal_range(0, 1024, 3, 1))
    t0 = (n // 4) * 4 - 1024
    assert np.array_equal(host(out[t0:]), O.philox_normal_range(t0, n - t0, 3, 1))
    assert abs(float(out.std()) - 1) < 1e-3
    # the ensemble form at full size: same bits as the fused draw for a random pair of moments
    mean, sq = torch.randn(n, device='cuda'), torch.rand(n, device='cuda') * 3
    K.swag_draw(out, mean, sq, var_clamp=1e-30, seed=3, draw=7)
    sd, out2 = torch.empty_like(mean), torch.empty_like(mean)
    K.swag_std(sd, mean, sq, var_clamp=1e-30)
    K.swag_draw_std(out2, mean, sd, seed=3, draw=7)
    assert torch.equal(out, out2)


def test_abi_v2_argument_errors_and_empty_inputs(K):
    """The new entry points return URSA_E* codes (never crash, never launch) on bad arguments, and treat empty inputs
    (n = 0, no chains, no control blocks) as a no-op that touches no pointer."""
    lib = K.lib
    E_NULL, E_SIZE, E_ALIGN = -1, -2, -3
    th = torch.zeros(3, 64, device='cuda')
    ctl = torch.zeros(3 * 2304 + 128, dtype=torch.uint8, device='cuda')
    p, c, st = th.data_ptr(), ctl.data_ptr(), torch.cuda.current_stream().cuda_stream
    f = lib.ursa_sgmcmc_step_multi_f32
    assert f(p, p, p, None, None, 0, 3, 64, c, st) == 0                      # n == 0
    assert f(None, None, None, None, None, 64, 0, 64, None, st) == 0         # no chains
    assert f(p, p, p, None, None, 64, 3, 60, c, st) == E_SIZE                # stride < n
    assert f(p, p, p, None, None, 62, 3, 62, c, st) == E_SIZE                # stride not a multiple of 4
    assert f(p, p, p, None, None, 64, 70000, 64, c, st) == E_SIZE            # more chains than a grid's y extent
    assert f(p, p, None, None, None, 64, 3, 64, c, st) == E_NULL             # momentum is required (mu lives on the device)
    assert f(p, p, p, None, None, 64, 3, 64, None, st) == E_NULL
    assert f(p + 4, p, p, None, None, 60, 1, 0, c, st) == E_ALIGN            # float4-only launch
    assert f(p, p, p, None, None, 64, 3, 64, c + 64, st) == E_ALIGN          # ticket counters sit on separate 128-byte lines
    assert lib.ursa_step_ctl_advance(None, 1, st) == E_NULL
    assert lib.ursa_step_ctl_advance(c, -1, st) == E_SIZE and lib.ursa_step_ctl_advance(c, 0, st) == 0
    assert lib.ursa_swag_std_f32(None, p, p, 64, 1e-30, 1.0, st) == E_NULL
    assert lib.ursa_swag_std_f32(p, p, p, 0, 1e-30, 1.0, st) == 0 and lib.ursa_swag_std_f32(p, p, p, -1, 1e-30, 1.0, st) == E_SIZE
    assert lib.ursa_swag_draw_std_f32(p, p, None, None, 64, 1, 0, st) == E_NULL
    assert lib.ursa_swag_draw_std_f32(p + 2, p, p, None, 64, 1, 0, st) == E_ALIGN
    assert lib.ursa_selftest_rng_f32(None, st) == E_NULL
    torch.cuda.synchronize()
    assert not th.any() and not ctl.any()                                    # nothing was launched
    assert lib.ursa_strerror(E_SIZE).decode() == 'invalid size'
