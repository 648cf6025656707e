"""K6 (relu(bn(x)) as gfx950 launches) against torch's own BatchNorm + ReLU — the ops the reference's networks run
(URSABench/models/preresnet.py:40-41,76-85,146; wideresnet.py:47,49,117) — evaluated in float64 on the host.
Floating-point reductions: the bar is a few fp32 ulp of the activation scale (2e-6 relative to max |y|, written below),
an order of magnitude inside north_star's 1e-5; the stock fp32 GPU path's own distance to float64 is measured beside it."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

RTOL = 2e-6          # of max |reference| per tensor


@pytest.fixture(scope='module')
def K():
    from ursabench_amd import _native
    assert torch.cuda.is_available(), 'gpu tests need a HIP device'
    return _native.default_kernels()


def _ref64(x, w, b, eps, relu, dy=None, mask=None):
    """float64 BatchNorm(+ReLU) forward / backward on the host. `mask`: which ReLUs count as open in the backward
    (the device's own decision: an element whose pre-activation sits within fp32 rounding of zero may legitimately
    fall on either side, and a flipped element moves dx by O(dy), not by rounding)."""
    x = x.double().cpu().requires_grad_(True)
    w = w.double().cpu().requires_grad_(True)
    b = b.double().cpu().requires_grad_(True)
    y = F.batch_norm(x, None, None, w, b, True, 0.0, eps)
    if relu:
        y = F.relu(y) if mask is None else y * mask.double().cpu()
    out = dict(y=y.detach())
    dims = [0] + list(range(2, x.dim()))
    out['mean'] = x.detach().mean(dims)
    out['var'] = x.detach().var(dims, unbiased=False)
    if dy is not None:
        y.backward(dy.double().cpu())
        out.update(dx=x.grad, dw=w.grad, db=b.grad)
    return out


def _close(got, ref, what, rtol=RTOL):
    ref = ref.float()
    scale = max(float(ref.abs().max()), 1e-30)
    err = float((got.cpu() - ref).abs().max()) / scale
    assert err <= rtol, f'{what}: {err:.3e} of the tensor scale (bar {rtol:.1e})'
    return err


SHAPES = [(128, 16, 32, 32), (128, 32, 16, 16), (128, 64, 8, 8),      # PreResNet-20's three stages
          (16, 160, 32, 32), (16, 640, 8, 8),                          # WideResNet-28-10 widths
          (7, 5, 3, 3), (2, 3, 1, 2), (33, 2, 5, 4), (1, 4, 2, 2),     # ragged: 4-byte path, tiny, one image
          (3, 1, 257, 4), (2, 70000 // 64, 8, 8)]                      # one channel; more channels than workgroup target


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('relu', [True, False])
def test_training_forward_backward_vs_float64(K, shape, relu):
    g = torch.Generator().manual_seed(hash(shape) % 1000)
    C = shape[1]
    x = (torch.randn(shape, generator=g) * 1.7 + 0.3).cuda()
    w = (torch.rand(C, generator=g) + 0.5).cuda()
    b = (torch.randn(C, generator=g) * 0.2).cuda()
    dy = torch.randn(shape, generator=g).cuda()
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    eps, mom = 1e-5, 0.1
    from ursabench_amd import _native
    y = torch.full_like(x, float('nan'))
    sm, si = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    ws = torch.empty(_native.bn_ws_floats(C), device='cuda')
    K.bn_relu_forward(x, y, w, b, rm, rv, sm, si, ws, eps=eps, momentum=mom, relu=relu)
    ref = _ref64(x, w, b, eps, relu, dy, mask=(y > 0) if relu else None)
    _close(y, ref['y'], 'y')
    _close(sm, ref['mean'], 'save_mean')
    _close(si, 1.0 / torch.sqrt(ref['var'] + eps), 'save_invstd')
    n = x.numel() // C
    _close(rm, mom * ref['mean'], 'running_mean')
    _close(rv, mom * ref['var'] * n / (n - 1) + (1 - mom), 'running_var')
    dx = torch.full_like(x, float('nan'))
    dw, db = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    K.bn_relu_backward(x, dy, dx, w, b, sm, si, dw, db, ws, relu=relu)
    _close(db, ref['db'], 'dbeta', 2e-5)
    _close(dw, ref['dw'], 'dgamma', 2e-5)
    _close(dx, ref['dx'], 'dx', 2e-5)
    # the stock fp32 path on the same device, for scale
    ys = F.batch_norm(x, None, None, w, b, True, 0.0, eps)
    ys = F.relu(ys) if relu else ys
    _close(ys, ref['y'], 'stock y', 1e-5)


def test_large_mean_small_spread(K):
    """Activations with |mean| >> std (mean 300, std 0.01): E[x^2] - E[x]^2 in fp32 cancels to noise; the double
    accumulators do not. y follows torch's CPU association (x * alpha + beta'), whose own cancellation against the
    folded shift is the reference's arithmetic: compared with the CPU kernel, not with float64."""
    from ursabench_amd import _native
    g = torch.Generator().manual_seed(5)
    shape, C = (64, 8, 16, 16), 8
    x = (torch.randn(shape, generator=g) * 0.01 + 300.0).cuda()
    w, b = torch.ones(C, device='cuda'), torch.zeros(C, device='cuda')
    y = torch.empty_like(x)
    sm, si = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    ws = torch.empty(_native.bn_ws_floats(C), device='cuda')
    K.bn_relu_forward(x, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, relu=False)
    ref = _ref64(x, w, b, 1e-5, False)
    assert torch.equal(sm.cpu(), ref['mean'].float())
    assert float((si.cpu().double() * torch.sqrt(ref['var'] + 1e-5) - 1).abs().max()) < 2e-7
    y_c, m_c, i_c = torch.native_batch_norm(x.cpu(), w.cpu(), b.cpu(), None, None, True, 0.0, 1e-5)
    same = (si.cpu() == i_c) & (sm.cpu() == m_c)
    assert torch.equal(y.cpu()[:, same], y_c[:, same])
    assert float((y.cpu() - y_c).abs().max()) < 1e-2       # channels whose invstd differs in the last bit: 30000 * 2^-23


@pytest.mark.parametrize('shape', [(128, 16, 32, 32), (128, 32, 16, 16), (128, 64, 8, 8), (7, 5, 3, 3)])
def test_forward_is_torch_cpu_batchnorm_bit_for_bit(K, shape):
    """The reference's path is torch's CPU BatchNorm. Same x: the batch mean is the same float in every channel, invstd
    in most (torch's own variance is not exact; ours is the correctly rounded one), and wherever both agree every
    output element has the same bits - so the ReLU gates agree."""
    from ursabench_amd import _native
    g = torch.Generator().manual_seed(11)
    C = shape[1]
    x = (torch.randn(shape, generator=g) * 1.3 - 0.2)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    y_c, m_c, i_c = torch.native_batch_norm(x, w, b, None, None, True, 0.0, 1e-5)
    xd = x.cuda()
    y = torch.empty_like(xd)
    sm, si = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    ws = torch.empty(_native.bn_ws_floats(C), device='cuda')
    K.bn_relu_forward(xd, y, w.cuda(), b.cuda(), None, None, sm, si, ws, eps=1e-5, momentum=0.0, relu=True)
    assert torch.equal(sm.cpu(), m_c)
    same = si.cpu() == i_c
    assert int(same.sum()) >= 0.7 * C, f'invstd equal in only {int(same.sum())} of {C} channels'
    assert float((si.cpu() / i_c - 1).abs().max()) < 1.3e-7          # the others: one unit in the last place
    assert torch.equal(y.cpu()[:, same], F.relu(y_c)[:, same])


def test_unaligned_pointers_take_the_scalar_path(K):
    from ursabench_amd import _native
    g = torch.Generator().manual_seed(9)
    shape, C = (8, 6, 4, 4), 6
    buf = torch.randn(8 * 6 * 16 + 1, generator=g).cuda()
    x = buf[1:].view(shape)                       # contiguous, 4-byte aligned only
    assert x.data_ptr() % 16 != 0
    w, b = torch.rand(C, device='cuda') + 0.5, torch.randn(C, device='cuda')
    out = []
    for xx in (x, x.clone()):
        y = torch.empty_like(xx)
        sm, si = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
        ws = torch.empty(_native.bn_ws_floats(C), device='cuda')
        K.bn_relu_forward(xx, y, w, b, None, None, sm, si, ws, eps=1e-5, momentum=0.0, relu=True)
        out.append((y, sm, si))
    ref = _ref64(x, w, b, 1e-5, True)
    _close(out[0][0], ref['y'], 'y (unaligned)')
    _close(out[1][0], ref['y'], 'y (aligned)')


@pytest.mark.parametrize('shape', [(128, 16, 32, 32), (5, 3, 3, 3), (100, 64, 8, 8)])
def test_evaluation_mode_vs_float64(K, shape):
    g = torch.Generator().manual_seed(3)
    C = shape[1]
    x = torch.randn(shape, generator=g).cuda()
    w, b = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    rm, rv = torch.randn(C, generator=g).cuda(), (torch.rand(C, generator=g) + 0.2).cuda()
    y = torch.empty_like(x)
    K.bn_relu_eval(x, y, w, b, rm, rv, eps=1e-5, relu=True)
    ref = F.relu(F.batch_norm(x.double().cpu(), rm.double().cpu(), rv.double().cpu(), w.double().cpu(), b.double().cpu(),
                              False, 0.0, 1e-5))
    _close(y, ref, 'eval y')


def test_module_semantics_follow_nn_batchnorm():
    """bn_relu(bn, x) against relu(bn_stock(x)) on twin modules: running statistics, the batch counter, train / eval,
    the cases that must take the stock path, and the in-place contract (x is never modified)."""
    from ursabench_amd import fused_bn
    torch.manual_seed(0)
    x = torch.randn(32, 8, 8, 8, device='cuda')
    a, s = nn.BatchNorm2d(8).cuda(), nn.BatchNorm2d(8).cuda()
    with torch.no_grad():
        a.weight.uniform_(0.5, 1.5)
        a.bias.normal_()
    s.load_state_dict(a.state_dict())
    for it in range(3):
        xi = x + it
        keep = xi.clone()
        ya = fused_bn.bn_relu(a, xi)
        ys = F.relu(s(xi))
        assert torch.equal(xi, keep)
        assert torch.allclose(ya, ys, rtol=0, atol=2e-6 * float(ys.abs().max()))
        assert torch.allclose(a.running_mean, s.running_mean, rtol=1e-6, atol=1e-7)
        assert torch.allclose(a.running_var, s.running_var, rtol=1e-6, atol=1e-7)
        assert int(a.num_batches_tracked) == int(s.num_batches_tracked) == it + 1
    a.eval(), s.eval()
    with torch.no_grad():
        ya, ys = fused_bn.bn_relu(a, x), F.relu(s(x))
    assert torch.allclose(ya, ys, rtol=0, atol=5e-6 * float(ys.abs().max()))
    assert int(a.num_batches_tracked) == 3
    # gradients through an evaluation-mode layer: stock path (same numbers as torch, by construction)
    xg = x.clone().requires_grad_(True)
    fused_bn.bn_relu(a, xg).sum().backward()
    assert xg.grad is not None
    # momentum=None (cumulative average) and host tensors: stock path
    c = nn.BatchNorm2d(8, momentum=None).cuda()
    c2 = nn.BatchNorm2d(8, momentum=None).cuda()
    assert torch.equal(fused_bn.bn_relu(c, x), F.relu(c2(x)))
    h, h2 = nn.BatchNorm2d(8), nn.BatchNorm2d(8)
    assert torch.equal(fused_bn.bn_relu(h, x.cpu()), F.relu(h2(x.cpu())))
    # the switch
    old = fused_bn.enabled(False)
    try:
        a.train(), s.train()
        assert torch.equal(fused_bn.bn_relu(a, x), F.relu(s(x)))
    finally:
        fused_bn.enabled(old)
    with pytest.raises(ValueError):
        fused_bn.bn_relu(nn.BatchNorm2d(8).cuda(), torch.randn(1, 8, 1, 1, device='cuda'))


def test_autograd_through_the_fused_layer_vs_stock():
    from ursabench_amd import fused_bn
    torch.manual_seed(1)
    x0 = torch.randn(64, 16, 16, 16, device='cuda')
    res = []
    for fused in (True, False):
        bn = nn.BatchNorm2d(16).cuda()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, 16))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, 16))
        x = x0.clone().requires_grad_(True)
        old = fused_bn.enabled(fused)
        try:
            y = fused_bn.bn_relu(bn, x)
            (y * torch.linspace(0, 1, y.numel(), device='cuda').view_as(y)).sum().backward()
        finally:
            fused_bn.enabled(old)
        res.append((y.detach(), x.grad, bn.weight.grad, bn.bias.grad))
    for got, ref, what in zip(res[0], res[1], ('y', 'dx', 'dgamma', 'dbeta')):
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 2e-5 * scale, what


@pytest.mark.parametrize('name', ['PreResNet20', 'PreResNet164', 'WideResNet28x10'])
def test_networks_fused_vs_stock(name):
    """The benchmark networks end to end, train mode, fused vs stock launches from identical weights. The logits are a
    continuous function of every intermediate rounding: tight. The gradients are not: a pre-activation within rounding
    of zero opens its ReLU gate on one side and not on the other (tools/exp/bn_gate_diag.py: ~2 of 24M gates per
    forward between ANY two of {CPU, MIOpen launches, K6}), which moves the gradients it touches by ~1/sqrt(elements per
    channel) - so they are compared at that scale only; the per-layer tests above carry the tight gradient bars."""
    from ursabench_amd import fused_bn, models
    cfg = getattr(models, name)
    classes = 10 if name == 'PreResNet20' else 100
    torch.manual_seed(0)
    ref_model = cfg.base(num_classes=classes, **cfg.kwargs).cuda()
    B = 32 if name == 'PreResNet20' else 8
    x = torch.randn(B, 3, 32, 32, device='cuda')
    t = torch.randint(0, classes, (B,), device='cuda')
    out = []
    for fused in (True, False):
        m = cfg.base(num_classes=classes, **cfg.kwargs).cuda()
        m.load_state_dict(ref_model.state_dict())
        m.train()
        old = fused_bn.enabled(fused)
        try:
            logits = m(x)
            F.cross_entropy(logits, t).backward()
        finally:
            fused_bn.enabled(old)
        out.append((logits.detach(), [p.grad for p in m.parameters()], [b.clone() for b in m.buffers()]))
    (la, ga, ba), (ls, gs, bs) = out
    deep = name != 'PreResNet20'       # 164 / 28 layers: rounding differences compound through the depth
    assert float((la - ls).abs().max()) <= (2e-4 if deep else 2e-5) * float(ls.abs().max())
    for a, s in zip(ga, gs):
        assert float((a - s).abs().max()) <= 0.1 * max(float(s.abs().max()), 1e-6)
    for a, s in zip(ba, bs):
        assert torch.allclose(a.float(), s.float(), rtol=1e-4, atol=1e-5)


def test_fused_layers_replay_inside_a_hipgraph():
    from ursabench_amd import models
    from ursabench_amd._capture import capture
    torch.manual_seed(0)
    m = models.PreResNet(num_classes=10, depth=8).cuda().train()
    x = torch.randn(16, 3, 32, 32, device='cuda')
    t = torch.randint(0, 10, (16,), device='cuda')

    def step():
        for p in m.parameters():
            p.grad = None
        F.cross_entropy(m(x), t).backward()
        return [p.grad for p in m.parameters()]
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    eager = [g.clone() for g in step()]
    m.load_state_dict(state)
    g = torch.cuda.CUDAGraph()
    with capture(g):
        grads = step()
    m.load_state_dict(state)
    g.replay()
    torch.cuda.synchronize()
    for a, b in zip(grads, eager):      # MIOpen's split-K weight gradients add with atomics: not bit-reproducible, and an
        # element that nearly cancels carries the rounding of its largest terms - judged on the tensor's own scale
        # (a 1e-6 bound failed once in ~10 runs of the suite on one element of the first convolution's gradient)
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))


@pytest.mark.parametrize('shape', [(128, 16, 32, 32), (128, 64, 8, 8), (33, 6, 5, 3), (16, 160, 16, 16)])
@pytest.mark.parametrize('relu', [True, False])
def test_k6_vs_oracle(K, shape, relu):
    """K6 through the C ABI against oracle/ursa_oracle.c (scalar C, sequential double sums) on the same seeded inputs:
    both round exact double statistics once, so mean / invstd - and with them every output, every ReLU gate, dbeta and
    dgamma - are the same floats (the double sums differ in association only: 1e-16 relative before the one rounding;
    a tie-straddling channel is allowed for, none observed); dx is elementwise float arithmetic on equal scalars."""
    import oracle_lib as O
    from ursabench_amd import _native
    g = torch.Generator().manual_seed(sum(shape) + relu)
    C = shape[1]
    x = (torch.randn(shape, generator=g) * 1.4 + 0.25)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    dy = torch.randn(shape, generator=g)
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    rm_o, rv_o = rm.numpy().copy(), rv.numpy().copy()
    y_o, sm_o, si_o = O.bn_relu_fwd(x.numpy(), w.numpy(), b.numpy(), rm_o, rv_o, eps=1e-5, momentum=0.1, relu=relu)
    xd, wd, bd, dyd, rmd, rvd = (t.cuda() for t in (x, w, b, dy, rm, rv))
    y = torch.empty_like(xd)
    sm, si = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    ws = torch.empty(_native.bn_ws_floats(C), device='cuda')
    K.bn_relu_forward(xd, y, wd, bd, rmd, rvd, sm, si, ws, eps=1e-5, momentum=0.1, relu=relu)
    same = (sm.cpu().numpy() == sm_o) & (si.cpu().numpy() == si_o)
    assert same.sum() >= C - 1, f'{C - same.sum()} of {C} channels differ in mean / invstd'
    assert np.array_equal(y.cpu().numpy()[:, same], y_o[:, same])
    np.testing.assert_allclose(rmd.cpu().numpy(), rm_o, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rvd.cpu().numpy(), rv_o, rtol=1e-6, atol=1e-7)
    dx_o, dg_o, db_o = O.bn_relu_bwd(x.numpy(), dy.numpy(), w.numpy(), b.numpy(), sm_o, si_o, relu=relu)
    dx = torch.empty_like(xd)
    dw, db = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    K.bn_relu_backward(xd, dyd, dx, wd, bd, torch.from_numpy(sm_o).cuda(), torch.from_numpy(si_o).cuda(), dw, db, ws, relu=relu)
    assert np.count_nonzero(db.cpu().numpy() != db_o) <= 1 and np.count_nonzero(dw.cpu().numpy() != dg_o) <= 1
    np.testing.assert_allclose(db.cpu().numpy(), db_o, rtol=2e-7, atol=0)
    np.testing.assert_allclose(dw.cpu().numpy(), dg_o, rtol=2e-7, atol=0)
    ok = (db.cpu().numpy() == db_o) & (dw.cpu().numpy() == dg_o)
    assert np.array_equal(dx.cpu().numpy()[:, ok], dx_o[:, ok])
    y_e = torch.empty_like(xd)
    K.bn_relu_eval(xd, y_e, wd, bd, rmd, rvd, eps=1e-5, relu=relu)
    assert np.array_equal(y_e.cpu().numpy(), O.bn_relu_eval(x.numpy(), w.numpy(), b.numpy(), rmd.cpu().numpy(), rvd.cpu().numpy(), eps=1e-5, relu=relu))


@pytest.mark.parametrize('shape', [(128, 16, 32, 32), (64, 64, 8, 8), (9, 5, 3, 3)])
def test_residual_form_equals_add_then_k6(K, shape):
    """addend / dz (the residual sum folded into the statistics pass, the gradient accumulation into the backward's
    second pass): z is torch's a + b bit for bit, y / statistics are those of the plain form on z, and the backward is
    the plain form's dx plus dz with one fp32 add - the same bits as running the two add launches separately."""
    from ursabench_amd import _native
    g = torch.Generator().manual_seed(sum(shape))
    C = shape[1]
    a, b, dy, dz = (torch.randn(shape, generator=g).cuda() for _ in range(4))
    w, bb = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    ws = torch.empty(_native.bn_ws_floats(C), device='cuda')

    def stats():
        return torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
    z, y = torch.empty_like(a), torch.empty_like(a)
    sm, si = stats()
    rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    K.bn_relu_forward(a, y, w, bb, rm, rv, sm, si, ws, eps=1e-5, momentum=0.1, addend=b, z_out=z)
    z_ref = a + b
    assert torch.equal(z, z_ref)
    y2 = torch.empty_like(a)
    sm2, si2 = stats()
    rm2, rv2 = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
    K.bn_relu_forward(z_ref, y2, w, bb, rm2, rv2, sm2, si2, ws, eps=1e-5, momentum=0.1)
    assert torch.equal(y, y2) and torch.equal(sm, sm2) and torch.equal(si, si2) and torch.equal(rm, rm2) and torch.equal(rv, rv2)
    dx, dx2 = torch.empty_like(a), torch.empty_like(a)
    dw, db = stats()
    dw2, db2 = stats()
    K.bn_relu_backward(z, dy, dx, w, bb, sm, si, dw, db, ws, dz=dz)
    K.bn_relu_backward(z, dy, dx2, w, bb, sm, si, dw2, db2, ws)
    assert torch.equal(dx, dz + dx2) and torch.equal(dw, dw2) and torch.equal(db, db2)
    ze, ye, ye2 = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
    K.bn_relu_eval(a, ye, w, bb, rm, rv, eps=1e-5, addend=b, z_out=ze)
    K.bn_relu_eval(z_ref, ye2, w, bb, rm, rv, eps=1e-5)
    assert torch.equal(ze, z_ref) and torch.equal(ye, ye2)
    with pytest.raises(ValueError):
        K.bn_relu_forward(a, y, w, bb, rm, rv, sm, si, ws, eps=1e-5, momentum=0.1, addend=b)


def test_add_bn_relu_autograd_vs_stock_ops():
    """(z, y) = add_bn_relu(bn, (a, b)) with z feeding a second consumer, against torch's add / BatchNorm / relu:
    gradients of a and b (equal), of gamma / beta; the unused-output cases (dz None, dy None)."""
    from ursabench_amd import fused_bn
    torch.manual_seed(3)
    a0, b0 = torch.randn(32, 8, 8, 8, device='cuda'), torch.randn(32, 8, 8, 8, device='cuda')
    wz = torch.randn(32, 8, 8, 8, device='cuda')
    out = []
    for fused in (True, False):
        bn = nn.BatchNorm2d(8).cuda()
        with torch.no_grad():
            bn.weight.copy_(torch.linspace(0.5, 1.5, 8))
            bn.bias.copy_(torch.linspace(-0.3, 0.3, 8))
        a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
        old = fused_bn.enabled(fused)
        try:
            z, y = fused_bn.add_bn_relu(bn, (a, b))
            ((z * wz).sum() + (y * y).sum()).backward()
        finally:
            fused_bn.enabled(old)
        out.append((z.detach(), y.detach(), a.grad, b.grad, bn.weight.grad, bn.bias.grad))
    assert torch.equal(out[0][0], out[1][0])
    for got, ref, what in zip(out[0][1:], out[1][1:], ('y', 'da', 'db', 'dgamma', 'dbeta')):
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), what
    assert torch.equal(out[0][2], out[0][3])
    bn = nn.BatchNorm2d(8).cuda()
    a, b = a0.clone().requires_grad_(True), b0.clone().requires_grad_(True)
    z, y = fused_bn.add_bn_relu(bn, (a, b))
    y.sum().backward()                                   # z unused: no dz
    assert a.grad is not None and torch.equal(a.grad, b.grad)
    a.grad = b.grad = None
    z, y = fused_bn.add_bn_relu(bn, (a, b))
    (z * wz).sum().backward()                            # y unused: the sum's own gradient only
    assert torch.equal(a.grad, wz) and torch.equal(b.grad, wz)
    # a plain tensor passes through; host tensors take torch's ops
    z, y = fused_bn.add_bn_relu(bn, a0)
    assert z is a0
    h = nn.BatchNorm2d(8)
    z, y = fused_bn.add_bn_relu(h, (a0.cpu(), b0.cpu()))
    assert torch.equal(z, a0.cpu() + b0.cpu())


@pytest.mark.parametrize('shape', [(128, 64, 8, 8), (128, 64, 16, 16), (16, 640, 8, 8), (2, 48, 2, 2), (30, 50, 12, 12), (64, 96, 8, 8)])
@pytest.mark.parametrize('relu', [True, False])
def test_one_pass_form_equals_two_launch_form(K, shape, relu):
    """A channel that fits one workgroup's registers takes the one-pass kernels (one launch forward, one backward); the
    arithmetic is the two-launch form's: statistics, outputs, gates, parameter gradients and dx are the same floats
    (double sums rounded once on both sides), plain and residual forms."""
    from ursabench_amd import _native
    g = torch.Generator().manual_seed(sum(shape) + 7 * relu)
    C = shape[1]
    a, b, dy, dz = (torch.randn(shape, generator=g).cuda() for _ in range(4))
    w, bb = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    ws = torch.empty(_native.bn_ws_floats(C), device='cuda')
    outs = []
    for two in (False, True):
        for resid in (False, True):
            z, y, dx = torch.full_like(a, float('nan')), torch.full_like(a, float('nan')), torch.full_like(a, float('nan'))
            sm, si, dw, db = (torch.empty(C, device='cuda') for _ in range(4))
            rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
            K.bn_relu_forward(a, y, w, bb, rm, rv, sm, si, ws, eps=1e-5, momentum=0.1, relu=relu, two_launch=two,
                              **(dict(addend=b, z_out=z) if resid else {}))
            K.bn_relu_backward(z if resid else a, dy, dx, w, bb, sm, si, dw, db, ws, relu=relu, two_launch=two,
                               dz=dz if resid else None)
            outs.append((resid, [y, sm, si, rm, rv, dx, dw, db] + ([z] if resid else [])))
    for (r1, o1), (r2, o2) in ((outs[0], outs[2]), (outs[1], outs[3])):
        assert r1 == r2
        for t1, t2, name in zip(o1, o2, ('y', 'mean', 'invstd', 'running_mean', 'running_var', 'dx', 'dgamma', 'dbeta', 'z')):
            assert not torch.isnan(t1).any(), name
            if name in ('dgamma', 'dbeta', 'mean', 'invstd', 'running_mean', 'running_var'):
                assert int((t1 != t2).sum()) <= 1 and torch.allclose(t1, t2, rtol=2e-7, atol=0), name
            elif name in ('dx', 'y'):
                ok = (o1[1] == o2[1]) & (o1[2] == o2[2]) & (o1[6] == o2[6]) & (o1[7] == o2[7])
                assert torch.equal(t1[:, ok], t2[:, ok]), name
            else:
                assert torch.equal(t1, t2), name


@pytest.mark.parametrize('shape', [(128, 16, 32, 32), (64, 64, 8, 8), (7, 5, 3, 3), (3, 1, 257, 4)])
@pytest.mark.parametrize('residual', [False, True])
def test_gated_backward_equals_oracle(K, shape, residual):
    """ursa_bn_relu_bwd_gated_f32 (the parity instrument, include/ursa_hip.h): listed gates are taken as given, unlisted
    ones recomputed; == the oracle's gated backward (dgamma / dbeta: both round exact double sums once; dx: same
    floats), == the plain backward when the list is empty, only padding, or every listed gate is the element's own."""
    import oracle_lib as O
    from ursabench_amd import _native
    g = torch.Generator().manual_seed(11)
    C = shape[1]
    x = torch.randn(shape, generator=g) * 1.3 + 0.2
    dy, dz = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    y, sm, si = O.bn_relu_fwd(x.numpy(), w.numpy(), b.numpy(), momentum=0.0)
    n = x.numel()
    pick = np.sort(np.random.default_rng(5).choice(n, size=min(n, 97), replace=False)).astype(np.int32)
    own = (y.reshape(-1)[pick] > 0).astype(np.uint8)
    flipped = own.copy()
    flipped[::2] ^= 1
    pad = np.full(31, 2 ** 31 - 1, np.int32)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    dX, dDY, dW, dB, dSM, dSI = (t.cuda() for t in (x, dy, w, b, torch.from_numpy(sm), torch.from_numpy(si)))
    dDZ = dz.cuda() if residual else None
    ws = torch.empty(_native.bn_ws_floats(C), device='cuda')

    def run(gates):
        dx = torch.full_like(dX, float('nan'))
        dg, db = torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
        K.bn_relu_backward(dX, dDY, dx, dW, dB, dSM, dSI, dg, db, ws, relu=True, dz=dDZ, two_launch=True, gates=gates)
        return dx.cpu().numpy(), dg.cpu().numpy(), db.cpu().numpy()
    plain = run(None)
    for gates in ((dev(pad[:0]), dev(np.zeros(0, np.uint8))), (dev(pad), dev(np.ones(31, np.uint8))),
                  (dev(np.concatenate([pick, pad])), dev(np.concatenate([own, np.ones(31, np.uint8)])))):
        for a, c in zip(plain, run(gates)):
            assert np.array_equal(a, c)
    got = run((dev(np.concatenate([pick, pad])), dev(np.concatenate([flipped, np.zeros(31, np.uint8)]))))
    odx, odg, odb = O.bn_relu_bwd(x.numpy(), dy.numpy(), w.numpy(), b.numpy(), sm, si, gates=(pick, flipped))
    if residual:
        odx = odx + dz.numpy()
    assert np.array_equal(got[1], odg) and np.array_equal(got[2], odb)
    np.testing.assert_allclose(got[0], odx, rtol=0, atol=2e-7 * float(np.abs(odx).max()))
    assert not np.array_equal(got[0], plain[0])


def test_gated_backward_argument_errors(K):
    from ursabench_amd import _native
    x = torch.randn(4, 3, 2, 2, device='cuda')
    C = 3
    v = torch.ones(C, device='cuda')
    ws = torch.empty(_native.bn_ws_floats(C), device='cuda')
    gi, go = torch.zeros(2, dtype=torch.int32, device='cuda'), torch.zeros(2, dtype=torch.uint8, device='cuda')
    with pytest.raises(ValueError):
        K.bn_relu_backward(x, x, torch.empty_like(x), v, v, v, v, v.clone(), v.clone(), ws, relu=False, gates=(gi, go))   # needs RELU
    with pytest.raises(ValueError):
        K.bn_relu_backward(x, x, torch.empty_like(x), v, v, v, v, v.clone(), v.clone(), ws, relu=True, gates=(gi, go[:1]))
    with pytest.raises(ValueError):
        K.bn_relu_backward(x, x, torch.empty_like(x), v, v, v, v, v.clone(), v.clone(), ws, relu=True, gates=(gi.long(), go))


HELD_SHAPES = [(512, 64, 32, 32),       # 134 MB: forward AND backward take the held form (forward: from 48 MiB on)
               (128, 160, 32, 32),      # WideResNet-28-10 stage 1: 84 MB: both directions held; forward in two pieces of 32 float4 per thread
               (256, 64, 32, 32),       # 67 MB: forward in 4 pieces with the LDS part in use
               (512, 16, 32, 32),       # few channels, 34 MB: backward held (16 x 11 workgroups), forward two-launch
               (1024, 256, 8, 8),       # 67 MB, 256 KB per channel: forward in two pieces of 16 float4 per thread
               (96, 96, 34, 30)]        # ragged: H*W = 1020 (float4 path, chunk tail)


def _sync_words(ws, C):
    """The held form's part of ws (URSA_BN_WS_HELD_OFFSET_FLOATS(C) on) as it must leave it: slots and counters zero.
    (The two-launch form's partials in front of it are scratch; a forward below 48 MiB takes that form.)"""
    return ws[C * 256:].view(torch.int32)


@pytest.mark.parametrize('shape', HELD_SHAPES)
@pytest.mark.parametrize('relu', [True, False])
def test_held_form_equals_two_launch_form(K, shape, relu):
    """URSA_BN_HELD (activations >= 24 MiB whose channels do not fit one workgroup): ONE launch per direction - the
    channel's workgroups hold their chunks in registers, exchange double partial sums through ws and wait for each other -
    must produce the two-launch form's floats (plain and residual forms), leave its sync words zero so that the SAME
    zeroed ws serves call after call, and never run into its bounded wait."""
    from ursabench_amd import _native
    g = torch.Generator().manual_seed(sum(shape) + 3 * relu)
    C = shape[1]
    a, b, dy, dz = (torch.randn(shape, generator=g).cuda() for _ in range(4))
    w, bb = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    assert a.numel() * 4 >= _native.BN_HELD_MIN_BYTES
    ws = torch.zeros(_native.bn_ws_floats(C), device='cuda')              # zeroed ONCE, reused by every held call below
    outs = []
    for held in (True, False):
        for resid in (False, True):
            z, y, dx = torch.full_like(a, float('nan')), torch.full_like(a, float('nan')), torch.full_like(a, float('nan'))
            sm, si, dw, db = (torch.full((C,), float('nan'), device='cuda') for _ in range(4))
            rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
            for _ in range(2 if held else 1):                               # twice: the launch re-arms its own sync words
                rm.zero_(), rv.fill_(1.0)
                K.bn_relu_forward(a, y, w, bb, rm, rv, sm, si, ws, eps=1e-5, momentum=0.1, relu=relu, held=held,
                                  **(dict(addend=b, z_out=z) if resid else {}))
                K.bn_relu_backward(z if resid else a, dy, dx, w, bb, sm, si, dw, db, ws, relu=relu, held=held,
                                   dz=dz if resid else None)
                if held:
                    torch.cuda.synchronize()
                    assert not _sync_words(ws, C).any(), 'sync words not re-armed (or the bounded wait ran out: word 1)'
            outs.append((resid, [y, sm, si, rm, rv, dx, dw, db] + ([z] if resid else [])))
    for (r1, o1), (r2, o2) in ((outs[0], outs[2]), (outs[1], outs[3])):
        assert r1 == r2
        for t1, t2, name in zip(o1, o2, ('y', 'mean', 'invstd', 'running_mean', 'running_var', 'dx', 'dgamma', 'dbeta', 'z')):
            assert not torch.isnan(t1).any(), name
            if name in ('dgamma', 'dbeta', 'mean', 'invstd', 'running_mean', 'running_var'):
                assert int((t1 != t2).sum()) <= 1 and torch.allclose(t1, t2, rtol=2e-7, atol=0), name
            elif name in ('dx', 'y'):
                ok = (o1[1] == o2[1]) & (o1[2] == o2[2]) & (o1[6] == o2[6]) & (o1[7] == o2[7])
                assert torch.equal(t1[:, ok], t2[:, ok]), name
            else:
                assert torch.equal(t1, t2), name


def test_one_held_launch_at_its_largest_split():
    """The contract of csrc/ursa_bn.hip ("held forms"): a held launch has the device to itself, at up to 32 pieces per
    channel forward / 64 backward (what one XCD holds - all pieces of a channel may land on one). Exercised at exactly those
    splits, launch after launch on one stream: never a bounded wait, same floats as the two-launch form. (With other
    launches in flight - held ones, and once in four runs of this suite plain ones too - a launch at these splits DID run
    into its bounded wait: tools/exp/bn_held_concurrency.py, profiles/r04_bn_held_concurrency.json)"""
    from ursabench_amd import _native
    K = _native.default_kernels()
    C = 8
    fshape, bshape = (2624, C, 32, 32), (1536, C, 32, 32)      # 671,744 / 393,216 float4 per channel: 32 / 64 pieces
    g = torch.Generator().manual_seed(11)
    w, bb = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    x = torch.randn(fshape, generator=g).cuda()
    bx, bdy = torch.randn(bshape, generator=g).cuda(), torch.randn(bshape, generator=g).cuda()
    new = lambda: torch.empty(C, device='cuda')
    y_ref, sm, si = torch.empty_like(x), new(), new()
    K.bn_relu_forward(x, y_ref, w, bb, None, None, sm, si, torch.empty(_native.bn_ws_floats(C), device='cuda'), eps=1e-5, momentum=0.0,
                      two_launch=True)
    by, bsm, bsi, dx_ref = torch.empty_like(bx), new(), new(), torch.empty_like(bx)
    wsb = torch.empty(_native.bn_ws_floats(C), device='cuda')
    K.bn_relu_forward(bx, by, w, bb, None, None, bsm, bsi, wsb, eps=1e-5, momentum=0.0, two_launch=True)
    K.bn_relu_backward(bx, bdy, dx_ref, w, bb, bsm, bsi, new(), new(), wsb, two_launch=True)
    del by
    torch.cuda.synchronize()
    ws = torch.zeros(_native.bn_ws_floats(C), device='cuda')
    y, dx = torch.empty_like(x), torch.empty_like(bx)
    fs, dgb = (new(), new()), (new(), new())
    for rep in range(10):
        K.bn_relu_forward(x, y, w, bb, None, None, fs[0], fs[1], ws, eps=1e-5, momentum=0.0, held=True)
        K.bn_relu_backward(bx, bdy, dx, w, bb, bsm, bsi, dgb[0], dgb[1], ws, held=True)
    torch.cuda.synchronize()
    sw = _sync_words(ws, C)
    nz = sw.nonzero().flatten()[:8].tolist()
    assert not nz, (f'bounded wait ran out or sync words not re-armed: int32 words {nz} of the held area = {[int(sw[i]) for i in nz]} '
                    f'(counters start at word {C * 256}: ticket, +32 done, +33 err, +64.. left[c])')
    assert not ws[:C * 256].any(), 'the two-launch partials were written - the held form did not run'
    assert torch.equal(y, y_ref) and torch.equal(dx, dx_ref)


def test_one_held_launch_beside_other_traffic_is_correct_or_loud_never_silently_wrong():
    """round 4 removed the busy side streams from the test above after a single held launch starved beside
    PLAIN launches once in four runs of the suite (not explained: the held form needs co-resident workgroups, and one forward
    workgroup needs a whole CU - foreign waves on every CU can keep it out for longer than the bounded wait). That hazard is why
    the form is opt-in now; what must hold when somebody opts in is that the outcome is never a plausible wrong number: either
    the launch drains and equals the two-launch form bit for bit, or it raises its error word AND its outputs are NaN."""
    from ursabench_amd import _native
    K = _native.default_kernels()
    C = 8
    fshape, bshape = (2624, C, 32, 32), (1536, C, 32, 32)      # 32 / 64 pieces per channel: the largest splits
    g = torch.Generator().manual_seed(11)
    w, bb = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    x = torch.randn(fshape, generator=g).cuda()
    bx, bdy = torch.randn(bshape, generator=g).cuda(), torch.randn(bshape, generator=g).cuda()
    new = lambda: torch.empty(C, device='cuda')
    y_ref, sm, si = torch.empty_like(x), new(), new()
    K.bn_relu_forward(x, y_ref, w, bb, None, None, sm, si, torch.empty(_native.bn_ws_floats(C), device='cuda'), eps=1e-5, momentum=0.0,
                      two_launch=True)
    by, bsm, bsi, dx_ref = torch.empty_like(bx), new(), new(), torch.empty_like(bx)
    wsb = torch.empty(_native.bn_ws_floats(C), device='cuda')
    K.bn_relu_forward(bx, by, w, bb, None, None, bsm, bsi, wsb, eps=1e-5, momentum=0.0, two_launch=True)
    K.bn_relu_backward(bx, bdy, dx_ref, w, bb, bsm, bsi, new(), new(), wsb, two_launch=True)
    del by
    others = [torch.randn(256, 64, 32, 32, generator=g).cuda() for _ in range(3)]
    oys = [torch.empty_like(o) for o in others]
    ows = [torch.empty(_native.bn_ws_floats(64), device='cuda') for _ in range(3)]
    w64, b64 = torch.ones(64, device='cuda'), torch.zeros(64, device='cuda')
    ostats = [(torch.empty(64, device='cuda'), torch.empty(64, device='cuda')) for _ in range(3)]
    side = [torch.cuda.Stream() for _ in range(3)]
    torch.cuda.synchronize()
    outcome = []
    for rep in range(6):
        ws = torch.zeros(_native.bn_ws_floats(C), device='cuda')
        y, dx = torch.zeros_like(x), torch.zeros_like(bx)
        fs, dgb = (new(), new()), (new(), new())
        torch.cuda.synchronize()
        for k, st in enumerate(side):                    # plain launches that wait for nobody, on three other streams
            with torch.cuda.stream(st):
                for _ in range(4):
                    K.bn_relu_forward(others[k], oys[k], w64, b64, None, None, ostats[k][0], ostats[k][1], ows[k], eps=1e-5, momentum=0.0,
                                      two_launch=True)
                    oys[k].mul_(1.0001)
        K.bn_relu_forward(x, y, w, bb, None, None, fs[0], fs[1], ws, eps=1e-5, momentum=0.0, held=True)
        K.bn_relu_backward(bx, bdy, dx, w, bb, bsm, bsi, dgb[0], dgb[1], ws, held=True)
        torch.cuda.synchronize()
        err = int(ws.view(torch.int32)[C * 512 + 33])
        if err == 0:
            assert not _sync_words(ws, C).any()
            assert torch.equal(y, y_ref) and torch.equal(dx, dx_ref), 'drained without an error word but the numbers differ'
            outcome.append('clean')
        else:
            assert torch.isnan(y).any() or torch.isnan(dx).any(), 'a starved launch left plausible numbers behind'
            outcome.append('starved-and-loud')
    print('held launch beside plain traffic:', outcome)


def test_module_path_keeps_one_private_scratch_per_layer_and_check_held_is_loud():
    """fused_bn, held form opted in: every LAYER owns one persistent zeroed scratch (no memset per call; two layers of equal
    width never share), the backward reuses the forward's scratch and decision, and check_held() - called by
    the samplers at their host syncs - raises when a launch left its error word there, then clears it."""
    from ursabench_amd import fused_bn
    C, shape = 16, (512, 16, 32, 32)                    # 34 MB: the backward takes the held form
    bn, bn_b = nn.BatchNorm2d(C).cuda().train(), nn.BatchNorm2d(C).cuda().train()
    x = torch.randn(shape, device='cuda', requires_grad=True)
    assert fused_bn.held() is False                     # opt-in
    y = fused_bn.bn_relu(bn, x)
    y.backward(torch.randn_like(y))
    assert bn not in fused_bn._held_ws and not fused_bn.held_in_use()        # default: two-launch form, nothing registered
    old = fused_bn.held(True)
    try:
        for _ in range(2):
            for m in (bn, bn_b):
                y = fused_bn.bn_relu(m, x)
                y.backward(torch.randn_like(y))
        (ws, c1), (ws_b, _) = fused_bn._held_ws[bn], fused_bn._held_ws[bn_b]
        assert c1 == C and ws.data_ptr() != ws_b.data_ptr() and fused_bn.held_in_use()
        torch.cuda.synchronize()
        assert not _sync_words(ws, C).any() and not _sync_words(ws_b, C).any()
        fused_bn.check_held()                               # quiet
        ws_b.view(torch.int32)[C * 512 + 33] = 1            # what a starved launch leaves behind
        with pytest.raises(RuntimeError, match='starved'):
            fused_bn.check_held(x.device)
        fused_bn.check_held()                               # cleared by the raise: the next check speaks of the next launches
        with fused_bn.several_streams():                    # overlapping callers: private, unzeroed scratch, two-launch form
            s1, h1 = fused_bn._scratch(bn, x, C)
            assert h1 is False and s1.data_ptr() != ws.data_ptr()
    finally:
        fused_bn.held(old)
    del bn, bn_b, m, y
    import gc
    gc.collect()
    assert not fused_bn.held_in_use()                       # the scratch goes with its layer


def test_a_starved_held_launch_poisons_its_outputs_with_nan():
    """round 4's held launch, when its bounded wait ran out, went on with incomplete sums - plausible wrong
    numbers, the err word the only signal. Now the sums are NaN: y / dx of the starved pieces, the channel's saved and running
    statistics, dgamma / dbeta are NaN on the device itself. Starvation is provoked deterministically: the ticket counter
    starts at 1, so piece 0 of channel 0 is never handed out and that channel's other pieces wait out their bound (~3 s)."""
    from ursabench_amd import _native
    K = _native.default_kernels()
    C, shape = 64, (256, 64, 32, 32)                    # 67 MB: forward (>= 48 MiB) and backward take the held form
    g = torch.Generator().manual_seed(3)
    x, dy = torch.randn(shape, generator=g).cuda(), torch.randn(shape, generator=g).cuda()
    w, bb = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    new = lambda: torch.zeros(C, device='cuda')
    for direction in ('forward', 'backward'):
        ws = torch.zeros(_native.bn_ws_floats(C), device='cuda')
        y, sm, si, rm, rv = torch.zeros_like(x), new(), new(), new(), new() + 1
        if direction == 'backward':
            K.bn_relu_forward(x, y, w, bb, None, None, sm, si, ws, eps=1e-5, momentum=0.0, held=True)
            torch.cuda.synchronize()
            assert not _sync_words(ws, C).any() and not torch.isnan(y).any()
        ws.view(torch.int32)[C * 512] = 1               # BnSync.ticket
        dx, dg, db = torch.zeros_like(x), new(), new()
        if direction == 'forward':
            K.bn_relu_forward(x, y, w, bb, rm, rv, sm, si, ws, eps=1e-5, momentum=0.1, held=True)
        else:
            K.bn_relu_backward(x, dy, dx, w, bb, sm, si, dg, db, ws, held=True)
        torch.cuda.synchronize()
        assert int(ws.view(torch.int32)[C * 512 + 33]) != 0, 'the err word was not raised'
        out = y if direction == 'forward' else dx
        assert torch.isnan(out[:, 0]).any(), 'a starved channel produced numbers'
        assert not torch.isnan(out[:, 1:]).any()        # the other channels drained normally
        if direction == 'forward':
            assert torch.isnan(sm[0]) and torch.isnan(si[0]) and torch.isnan(rm[0]) and torch.isnan(rv[0])
        else:
            assert torch.isnan(dg[0]) and torch.isnan(db[0])


def test_several_streams_context_switches_the_held_form_off():
    """fused_bn.several_streams(): what ChainGroup's branches and bn_update_many's member streams run under - a large
    activation then takes the two-launch form (its scratch is not the zeroed kind the held form needs)."""
    from ursabench_amd import fused_bn
    old = fused_bn.held(True)
    try:
        assert fused_bn.held_allowed() is True
        with fused_bn.several_streams():
            assert fused_bn.held_allowed() is False
            with fused_bn.several_streams():
                assert fused_bn.held_allowed() is False
            assert fused_bn.held_allowed() is False
        assert fused_bn.held_allowed() is True
    finally:
        fused_bn.held(old)


def test_held_form_on_parallel_streams_and_through_the_module_path():
    """(a) Four held launches in flight at once on four streams, each in 4 pieces per channel - inside the general bound
    of csrc/ursa_bn.hip (sum of (S_k - 1) = 12 < the 32 workgroups one XCD holds): all drain, all correct. (The product
    never relies on it: overlapping launches run under fused_bn.several_streams().) (b) fused_bn picks the held
    form by itself for a large activation (zeroed scratch) and the layer's output / gradients equal the two-launch run."""
    from ursabench_amd import _native, fused_bn
    K = _native.default_kernels()
    shape, C = (512, 64, 32, 32), 64            # 134 MB: the forward takes the held form
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(shape, generator=g).cuda() for _ in range(4)]
    w, bb = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    ref = []
    for x in xs:
        y, sm, si = torch.empty_like(x), torch.empty(C, device='cuda'), torch.empty(C, device='cuda')
        K.bn_relu_forward(x, y, w, bb, None, None, sm, si, torch.empty(_native.bn_ws_floats(C), device='cuda'), eps=1e-5,
                          momentum=0.0, two_launch=True)
        ref.append(y)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    wss = [torch.zeros(_native.bn_ws_floats(C), device='cuda') for _ in range(4)]
    got = [torch.empty_like(x) for x in xs]
    stats = [(torch.empty(C, device='cuda'), torch.empty(C, device='cuda')) for _ in range(4)]
    torch.cuda.synchronize()
    for rep in range(3):
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                K.bn_relu_forward(xs[k], got[k], w, bb, None, None, stats[k][0], stats[k][1], wss[k], eps=1e-5, momentum=0.0,
                                  held=True)
    torch.cuda.synchronize()
    for k in range(4):
        assert not _sync_words(wss[k], C).any()
        assert torch.equal(got[k], ref[k])
    # (b) module path
    bn = nn.BatchNorm2d(C).cuda().train()
    x = xs[0].clone().requires_grad_(True)
    dy = torch.randn(shape, generator=g).cuda()
    res = {}
    for mode in (True, False):
        old = fused_bn.held(mode)
        try:
            bn.reset_running_stats()
            y = fused_bn.bn_relu(bn, x)
            gx, gw, gb = torch.autograd.grad(y, (x, bn.weight, bn.bias), dy)
            res[mode] = (y.detach(), gx, gw, gb, bn.running_mean.clone(), bn.running_var.clone())
        finally:
            fused_bn.held(old)
    for t1, t2 in zip(res[True], res[False]):
        assert torch.allclose(t1, t2, rtol=2e-7, atol=0) and float((t1 != t2).float().mean()) < 1e-3


@pytest.mark.parametrize('shape,form', [((128, 64, 8, 8), 'one-pass'), ((128, 64, 16, 16), 'one-pass'), ((16, 640, 8, 8), 'one-pass'),
                                        ((512, 16, 32, 32), 'held'), ((128, 160, 32, 32), 'held')])
@pytest.mark.parametrize('residual', [False, True])
def test_product_backward_forms_equal_the_gate_instrument_given_their_own_gates(K, shape, form, residual):
    """VERDICT r4 #5 ii. The parity instrument (ursa_bn_relu_bwd_gated_f32) always runs the two-launch kernels, while the timed
    path takes the ONE-PASS backward at the 8x8 / 16x16 maps and - opted in - the HELD backward from 24 MiB on. On identical
    inputs, with the gates the forward itself took listed as given, the instrument must equal those product forms: dx the same
    floats in every channel whose two sums round alike (both forms round exact double sums once; different grouping of the double
    additions may move one channel's sum by one unit in the last place), dgamma / dbeta equal up to that."""
    from ursabench_amd import _native
    g = torch.Generator().manual_seed(17 + sum(shape))
    C = shape[1]
    x, dy, dz = (torch.randn(shape, generator=g).cuda() for _ in range(3))
    w, bb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    new = lambda: torch.empty(C, device='cuda')
    y, sm, si = torch.empty_like(x), new(), new()
    ws = torch.zeros(_native.bn_ws_floats(C), device='cuda')
    K.bn_relu_forward(x, y, w, bb, None, None, sm, si, ws, eps=1e-5, momentum=0.0, two_launch=True)
    n = x.numel()
    pick = torch.sort(torch.randperm(n, generator=g)[:min(n, 4001)]).values.to(torch.int32).cuda()
    own = (y.reshape(-1)[pick.long()] > 0).to(torch.uint8)
    pad = torch.full((63,), 2 ** 31 - 1, dtype=torch.int32, device='cuda')
    gates = (torch.cat([pick, pad]).contiguous(), torch.cat([own, torch.zeros(63, dtype=torch.uint8, device='cuda')]).contiguous())
    res = {}
    ws_p = torch.zeros(_native.bn_ws_floats(C), device='cuda')          # the product call's own scratch (zeroed: the held form may run)
    for name, kw, wsk in (('product', dict(held=(form == 'held')), ws_p), ('instrument', dict(gates=gates), ws)):
        dx, dg, db = torch.full_like(x, float('nan')), new(), new()
        K.bn_relu_backward(x, dy, dx, w, bb, sm, si, dg, db, wsk, relu=True, dz=dz if residual else None, **kw)
        res[name] = (dx, dg, db)
    torch.cuda.synchronize()
    if form == 'held':
        assert not _sync_words(ws_p, C).any() and not ws_p[:C * 256].any(), 'the held form did not run (or left its sync words dirty)'
    (dx1, dg1, db1), (dx2, dg2, db2) = res['product'], res['instrument']
    assert not torch.isnan(dx1).any() and not torch.isnan(dx2).any()
    assert int((dg1 != dg2).sum()) <= 1 and int((db1 != db2).sum()) <= 1
    assert torch.allclose(dg1, dg2, rtol=2e-7, atol=0) and torch.allclose(db1, db2, rtol=2e-7, atol=0)
    same = (dg1 == dg2) & (db1 == db2)
    assert torch.equal(dx1[:, same], dx2[:, same])


def _net_step(net, x, y):
    for p in net.parameters():
        p.grad = None
    logits = net(x)
    loss = torch.nn.functional.cross_entropy(logits, y, reduction='sum')
    loss.backward()
    return loss.detach().clone(), [p.grad.detach().clone() for p in net.parameters()], logits.detach().clone()


@pytest.mark.parametrize('name,classes,batch', [('PreResNet164', 100, 256), ('WideResNet28x10', 100, 128)])
def test_held_form_in_a_real_network_step_equals_two_launch_and_leaves_no_error(name, classes, batch):
    """VERDICT r4 missing #5 / next #4: the held launches on REAL network steps at the sizes that take them - PreResNet-164 at
    batch 256 (67 MB bottleneck outputs; models/preresnet.py:76-90) and WideResNet-28-10 at batch 128 (84 MB; wideresnet.py:47-51)
    - opted in, 20 repeats: every repeat's error words clean (check_held), no NaN, loss within 1e-6 and logits within 1e-4 of
    the default (two-launch) run's - or within ten times what two default runs differ by themselves - and the gradients at the
    scale a flipped ReLU gate moves them by (the per-layer tests above carry the bit-for-bit claims: same inputs, same floats)."""
    from ursabench_amd import fused_bn, models
    cfg = getattr(models, name)
    torch.manual_seed(3)
    net = cfg.base(num_classes=classes, **cfg.kwargs).cuda().train()
    g = torch.Generator().manual_seed(4)
    x, y = torch.randn(batch, 3, 32, 32, generator=g).cuda(), torch.randint(0, classes, (batch,), generator=g).cuda()
    state = {k: v.clone() for k, v in net.state_dict().items()}

    def fresh():
        net.load_state_dict(state)

    assert fused_bn.held() is False
    fresh()
    _net_step(net, x, y)                                     # MIOpen's per-layer solver search runs inside the first call: not a reference
    fresh()
    loss_a, grads_a, logits_a = _net_step(net, x, y)
    fresh()
    loss_b, grads_b, logits_b = _net_step(net, x, y)
    # Two runs of the DEFAULT path from the same weights: the yardstick. (Their logits are not always bit-equal: MIOpen runs some of
    # these convolutions - PreResNet-164's 1x1 layers at batch 256 - with split-K atomics, observed on the GPU box; a last-bit
    # difference upstream can flip a ReLU gate downstream, which moves the gradients it touches at the 1e-2 scale,
    # test_networks_fused_vs_stock.)
    scale_l = float(logits_a.abs().max())
    noise_l = float((logits_a - logits_b).abs().max())
    noise = [float((a - b).abs().max()) for a, b in zip(grads_a, grads_b)]
    assert abs(float(loss_a) - float(loss_b)) <= 1e-6 * abs(float(loss_a))
    assert not fused_bn.held_in_use()
    old = fused_bn.held(True)
    try:
        for rep in range(20):
            fresh()
            loss_h, grads_h, logits_h = _net_step(net, x, y)
            fused_bn.check_held()                            # raises if any held launch ran into its bounded wait
            assert torch.isfinite(loss_h) and torch.isfinite(logits_h).all()
            assert abs(float(loss_h) - float(loss_a)) <= 1e-6 * abs(float(loss_a))
            assert float((logits_h - logits_a).abs().max()) <= 10 * noise_l + 1e-4 * scale_l, rep
            if rep in (0, 19):
                for k, (gh, ga, nz) in enumerate(zip(grads_h, grads_a, noise)):
                    assert torch.isfinite(gh).all()
                    assert float((gh - ga).abs().max()) <= 10 * nz + 3e-2 * float(ga.abs().max()) + 1e-12, k
        print(f'{name}: default-path logits run to run differ by {noise_l:.2e} (scale {scale_l:.2f}); held vs default {float((logits_h - logits_a).abs().max()):.2e}')
        assert fused_bn.held_in_use(), 'no layer took the held form: the test does not test what it says'
        n_layers = len(fused_bn._held_ws)
        assert n_layers >= 10, n_layers
    finally:
        fused_bn.held(old)


def test_hmc_and_chain_engine_check_the_held_error_word_at_their_host_syncs():
    """The samplers read the held launches' error words where they sync with the host anyway (HMC: the MH test; ChainEngine:
    end of epoch; bn_update: its end) - a no-op by default, one small read when the held form is opted into. Here: PreResNet-164
    HMC at batch 256 (potential captured into a hipGraph after two eager evaluations), opted in, runs clean; then a starved
    launch's mark in one layer's scratch makes the NEXT proposal raise."""
    import ursabench_amd.inference as inference
    from ursabench_amd import fused_bn, models
    from ursabench_amd.data import synthetic
    dev = torch.device('cuda', 0)
    train = synthetic(256, (3, 32, 32), 100, seed=0, device=dev, batch_size=128)
    old = fused_bn.held(True)
    try:
        torch.manual_seed(0)
        h = inference.HMC({'step_size': 2e-4, 'num_samples': 1, 'L': 1, 'tau': 1.0, 'burn': 0, 'mass': 1.0},
                          models.PreResNet(100, 164).to(dev), train, device=dev, seed=0)
        for _ in range(3):                                   # 2 evaluations each: eager, eager, then hipGraph replays
            h.sample()
        assert h._graph is not None and fused_bn.held_in_use()
        ws, C = next(iter(fused_bn._held_ws.values()))
        ws.view(torch.int32)[C * 512 + 33] = 1
        with pytest.raises(RuntimeError, match='starved'):
            h.sample()
        h.sample()                                           # the word was cleared by the raise
    finally:
        fused_bn.held(old)


@pytest.mark.parametrize('shape,held', [((128, 16, 32, 32), False), ((128, 64, 8, 8), False), ((7, 5, 3, 3), False), ((512, 16, 32, 32), True)])
def test_backward_takes_its_gates_from_the_forwards_saved_scalars(K, shape, held):
    """ADVICE r3 low / VERDICT r4 weak #11: the backward used to recompute the ReLU gate from the LIVE gamma / beta. The
    parameters are views of the flat arena that K1 updates through raw pointers - a write autograd's version counters cannot
    see - so a parameter changed between forward and backward would have moved gates silently. The forward now stores the
    scale / shift it applied (`save_gate`), and the backward handed them (`gate`) recomputes the gates from those:
      * unchanged parameters: bit for bit the backward without `gate`;
      * beta changed in place after the forward (it enters the backward ONLY through the gate): with `gate` the result is
        unchanged bit for bit, without it gates move;
      * the saved scalars are the forward's: alpha = invstd * gamma, beta' = fma(-mean, alpha, beta)."""
    from ursabench_amd import _native
    g = torch.Generator().manual_seed(23 + sum(shape))
    C = shape[1]
    x, dy = (torch.randn(shape, generator=g).cuda() for _ in range(2))
    w, bb = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.3).cuda()
    new = lambda: torch.empty(C, device='cuda')
    y, sm, si, gate = torch.empty_like(x), new(), new(), torch.empty(2, C, device='cuda')
    ws = torch.zeros(_native.bn_ws_floats(C), device='cuda')
    K.bn_relu_forward(x, y, w, bb, None, None, sm, si, ws, eps=1e-5, momentum=0.0, save_gate=gate, held=held)
    alpha = si * w
    assert torch.equal(gate[0], alpha) and torch.equal(gate[1], torch.addcmul(bb, -sm, alpha)) or \
        torch.allclose(gate[1], bb - sm * alpha, rtol=0, atol=1e-6)                  # (addcmul is not guaranteed to fuse: the fma form is the kernel's)

    def backward(beta, gt):
        dx, dg, db = torch.full_like(x, float('nan')), new(), new()
        K.bn_relu_backward(x, dy, dx, w, beta, sm, si, dg, db, ws, relu=True, held=held, gate=gt)
        return dx, dg, db
    ref = backward(bb, None)
    for a, b in zip(ref, backward(bb, gate)):
        assert torch.equal(a, b)
    moved = bb + 0.25                                                                    # what an in-place update of beta would leave
    for a, b in zip(ref, backward(moved, gate)):
        assert torch.equal(a, b), 'with the saved scalars a changed beta must not matter'
    assert not torch.equal(ref[0], backward(moved, None)[0]), 'the control: without them gates do move'


@pytest.mark.parametrize('shape', [(1030, 20, 16, 16), (220, 50, 20, 20), (300, 50, 12, 12), (4100, 16, 8, 8), (128, 160, 32, 32), (67, 33, 40, 40)])
@pytest.mark.parametrize('relu', [True, False])
def test_linear_evaluation_launch_equals_the_oracle_bit_for_bit(K, shape, relu):
    """Round 5: from 16 MiB of activation on (8 MiB with an addend) the evaluation launch streams the tensor front to back - one
    contiguous 16 KB span per workgroup, the channel of every float4 looked up in a per-workgroup table - instead of walking
    channel by channel (csrc/ursa_bn.hip k_bn_eval_lin; profiles/r05_bn_eval_lin_ab.json). Same arithmetic per element, so the
    same bits as the oracle (and as torch's CPU kernel): shapes whose H*W/4 is and is not a power of two, channel counts that are
    not, a last workgroup that is not full, plain and residual forms, with and without ReLU."""
    import oracle_lib as O
    g = torch.Generator().manual_seed(41 + sum(shape))
    C = shape[1]
    x, a = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    rm, rv = torch.randn(C, generator=g) * 0.2, torch.rand(C, generator=g) + 0.5
    dw, db, drm, drv = (t.cuda() for t in (w, b, rm, rv))
    big_enough_plain = x.numel() * 4 >= (16 << 20)
    want = O.bn_relu_eval(x.numpy(), w.numpy(), b.numpy(), rm.numpy(), rv.numpy(), eps=1e-5, relu=relu)
    y = torch.full(shape, float('nan'), device='cuda')
    K.bn_relu_eval(x.cuda(), y, dw, db, drm, drv, eps=1e-5, relu=relu)
    assert np.array_equal(y.cpu().numpy(), want), ('plain', big_enough_plain)
    z_ref = x + a                                                   # one fp32 add, as torch's (and the launch's)
    want_r = O.bn_relu_eval(z_ref.numpy(), w.numpy(), b.numpy(), rm.numpy(), rv.numpy(), eps=1e-5, relu=relu)
    y, z = torch.full(shape, float('nan'), device='cuda'), torch.full(shape, float('nan'), device='cuda')
    K.bn_relu_eval(x.cuda(), y, dw, db, drm, drv, eps=1e-5, relu=relu, addend=a.cuda(), z_out=z)
    assert torch.equal(z.cpu(), z_ref) and np.array_equal(y.cpu().numpy(), want_r), 'residual'
    assert x.numel() * 4 >= (8 << 20), 'the shape does not reach the linear form: the test does not test what it says'
