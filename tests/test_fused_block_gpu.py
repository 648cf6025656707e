"""GPU: K10 (ursa_preact_conv3x3_f32 & co, include/ursa_hip.h) - the pre-activation unit conv(relu(bn(x))) (+ residual) as one
launch each way. Three yardsticks:
  * the UNFUSED launches it replaces (K6 statistics / normalise, K8 convolution, K6 backward), bit for bit on random data: the
    fused forms are the same arithmetic in fewer launches, and everything already pinned about K6 / K8 / K7 (torch-CPU-bitwise
    BatchNorm forward, exact integer convolutions, the gate instrument) carries over;
  * the oracle's restatement (oracle_preact_*), exactly on integer-valued data and to rounding on random data;
  * the whole network: `fused_block.trunk` against the K6 / K8 path - logits, every gradient, running statistics - eagerly and
    inside a replayed hipGraph.
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle_lib as O
from ursabench_amd import _native, fused_block, fused_bn, fused_conv, models

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)
# (Cin, Cout, H = W of x, stride) of the units (BatchNorm in front): preresnet.py's BasicBlock layers
UNITS = [(16, 16, 32, 1), (32, 32, 16, 1), (64, 64, 8, 1), (16, 32, 32, 2), (32, 64, 16, 2)]
BATCHES = [1, 2, 5, 80, 128]


def _K():
    return _native.default_kernels()


def _sums(x):
    """[C, 1, 2] float64 (sum x, sum x^2): what a producer's epilogue leaves, as one partial per channel."""
    xd = x.double()
    return torch.stack([xd.sum((0, 2, 3)), (xd * xd).sum((0, 2, 3))], -1)[:, None, :].contiguous()


def _scratch(geo):
    return torch.zeros(geo[1], dtype=torch.uint8, device=DEV) if geo[1] else None


def _k10_fwd(x, w, stride, bn=None, add=None, in_partial=None, running=None, eps=1e-5, momentum=0.1):
    """One forward launch; returns (y, out_partial, save | None, scratch)."""
    K = _K()
    cout = w.shape[0]
    geo = K.preact_geometry(x.shape, cout, stride=stride, bn=bn is not None, add=add is not None)
    assert geo is not None
    y = torch.full((x.shape[0], cout, x.shape[2] // stride, x.shape[3] // stride), float('nan'), device=DEV)
    part = torch.full((cout, geo[0], 2), float('nan'), dtype=torch.float64, device=DEV)
    sc = _scratch(geo)
    save = None
    bnarg = None
    if bn is not None:
        save = torch.full((4, x.shape[1]), float('nan'), device=DEV)
        rm, rv = running if running is not None else (None, None)
        bnarg = (_sums(x) if in_partial is None else in_partial, bn[0], bn[1], rm, rv, save, eps, momentum)
    K.preact_conv3x3(x, w, y, part, sc, stride=stride, bn=bnarg, add=add)
    return y, part, save, sc


def _unfused_fwd(x, w, stride, gamma, beta, add=None, running=None, eps=1e-5, momentum=0.1):
    K = _K()
    C = x.shape[1]
    h = torch.empty_like(x)
    st = torch.empty(4, C, device=DEV)
    rm, rv = running if running is not None else (None, None)
    K.bn_relu_forward(x, h, gamma, beta, rm, rv, st[0], st[1], torch.empty(_native.bn_ws_floats(C), device=DEV), eps=eps,
                      momentum=momentum, relu=True, save_gate=st[2:])
    y = K.conv3x3(h, w, stride=stride)
    return (y if add is None else y + add), h, st


def _rand_unit(cin, cout, hw, stride, n, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    x = (torch.randn(n, cin, hw, hw, generator=g) * 1.3 + 0.2).to(DEV)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cout)) ** 0.5).to(DEV)
    gamma, beta = (torch.rand(cin, generator=g) + 0.5).to(DEV), (torch.randn(cin, generator=g) * 0.3).to(DEV)
    add = torch.randn(n, cout, hw // stride, hw // stride, generator=g).to(DEV)
    dy = torch.randn(n, cout, hw // stride, hw // stride, generator=g).to(DEV)
    return x, w, gamma, beta, add, dy


@pytest.mark.parametrize('cin,cout,hw,stride', UNITS)
@pytest.mark.parametrize('n', BATCHES)
def test_forward_unit_equals_the_unfused_launches_bit_for_bit(cin, cout, hw, stride, n):
    x, w, gamma, beta, add, _ = _rand_unit(cin, cout, hw, stride, n, 100 * n + cin + stride)
    for use_add in ([False, True] if stride == 1 else [False]):
        rm0, rv0 = torch.randn(cin, device=DEV), torch.rand(cin, device=DEV) + 0.5
        ra, rb = (rm0.clone(), rv0.clone()), (rm0.clone(), rv0.clone())
        want, _, st = _unfused_fwd(x, w, stride, gamma, beta, add if use_add else None, running=ra)
        y, part, save, sc = _k10_fwd(x, w, stride, bn=(gamma, beta), add=add if use_add else None, running=rb)
        assert torch.equal(save, st), 'mean / invstd / alpha / beta\' differ from K6\'s'
        assert torch.equal(y, want)
        assert torch.equal(rb[0], ra[0]) and torch.equal(rb[1], ra[1])
        # the statistics of what was stored: sum of the nl partial sums == the double sum, to double rounding
        got = part.sum(1)
        ref = _sums(want)[:, 0]
        assert torch.isfinite(part).all()
        assert torch.allclose(got, ref, rtol=1e-12, atol=1e-9)
        assert not sc.any(), 'slots / counters must be zero again when the launch has drained'


@pytest.mark.parametrize('n', BATCHES)
def test_stem_equals_k8_and_leaves_the_statistics(n):
    g = torch.Generator().manual_seed(n)
    x, w = torch.randn(n, 3, 32, 32, generator=g).to(DEV), (torch.randn(16, 3, 3, 3, generator=g) * 0.3).to(DEV)
    y, part, _, sc = _k10_fwd(x, w, 1)
    assert torch.equal(y, _K().conv3x3(x, w))
    assert torch.allclose(part.sum(1), _sums(y)[:, 0], rtol=1e-12, atol=1e-9)
    assert not sc.any()


@pytest.mark.parametrize('cin,cout,hw,stride', UNITS)
@pytest.mark.parametrize('n', BATCHES)
def test_partial_sums_feed_the_next_unit_like_k6s_own_statistics(cin, cout, hw, stride, n):
    """The producer's out_partial handed to the consumer (as the network does) gives the consumer the scalars K6 computes from
    the tensor itself: both are the correctly rounded mean / invstd of the same numbers."""
    x, w, gamma, beta, _, _ = _rand_unit(cin, cout, hw, stride, n, 7 * n + cout)
    y, part, _, _ = _k10_fwd(x, w, stride, bn=(gamma, beta))
    g2, b2 = torch.rand(cout, device=DEV) + 0.5, torch.randn(cout, device=DEV) * 0.2
    w2 = torch.randn(cout, cout, 3, 3, device=DEV) * 0.05
    want, _, st = _unfused_fwd(y, w2, 1, g2, b2)
    y2, _, save2, _ = _k10_fwd(y, w2, 1, bn=(g2, b2), in_partial=part)
    assert torch.equal(save2, st)
    assert torch.equal(y2, want)
    # and the network's last BatchNorm: K6's normalise launch alone from the same partial sums == K6's two launches
    K = _K()
    h, save = torch.empty_like(y), torch.empty(4, cout, device=DEV)
    K.bn_apply(y, h, part, g2, b2, None, None, save, eps=1e-5, momentum=0.0, relu=True)
    h_ref, st_ref = torch.empty_like(y), torch.empty(4, cout, device=DEV)
    K.bn_relu_forward(y, h_ref, g2, b2, None, None, st_ref[0], st_ref[1], torch.empty(_native.bn_ws_floats(cout), device=DEV), eps=1e-5,
                      momentum=0.0, relu=True, save_gate=st_ref[2:])
    assert torch.equal(save, st_ref) and torch.equal(h, h_ref)


@pytest.mark.parametrize('cin,cout,hw,stride', UNITS)
@pytest.mark.parametrize('n', BATCHES)
def test_backward_unit_equals_the_unfused_launches(cin, cout, hw, stride, n):
    x, w, gamma, beta, _, dy = _rand_unit(cin, cout, hw, stride, n, 31 * n + cin)
    K = _K()
    _, h, st = _unfused_fwd(x, w, stride, gamma, beta)
    dz = torch.randn_like(x)
    # unfused: K8 input gradient, K6 backward (two launches), K7 on the stored activation
    dh = K.conv3x3(dy, w, flip=True, stride=stride)
    dx_ref, dgb_ref = torch.empty_like(x), torch.empty(2, cin, device=DEV)
    K.bn_relu_backward(x, dh, dx_ref, gamma, beta, st[0], st[1], dgb_ref[0], dgb_ref[1], torch.empty(_native.bn_ws_floats(cin), device=DEV),
                       relu=True, dz=dz, gate=st[2:], two_launch=True)
    ws = torch.empty(K.conv_wgrad_ws_floats(x.shape, cout, 3, stride), device=DEV)
    dw_ref = torch.empty_like(w)
    K.conv_wgrad(h, dy, dw_ref, ws, stride)
    # fused
    geo = K.preact_geometry(dy.shape, cin, flip=True, stride=stride)
    g = torch.full_like(x, float('nan'))
    pb = torch.full((cin, geo[0], 2), float('nan'), dtype=torch.float64, device=DEV)
    sc = _scratch(geo)
    assert sc is None, 'the input-gradient forms hand their sums to the dx launch: no scratch'
    K.preact_conv3x3(dy, w, g, pb, sc, stride=stride, flip=True, bwd=(x, st))
    assert torch.equal(g, torch.where(h > 0, dh, torch.zeros_like(dh))), 'gated input gradient'
    dx, dgb = torch.empty_like(x), torch.empty(2, cin, device=DEV)
    K.bn_bwd_dx(x, g, dx, gamma, st, pb, dgb[0], dgb[1], dz=dz)
    assert torch.equal(dgb, dgb_ref), 'dgamma / dbeta'
    assert torch.equal(dx, dx_ref)
    dw = torch.empty_like(w)
    K.conv_wgrad_reduce([(K.preact_wgrad_partial(x, st, dy, w.shape, torch.empty_like(ws), stride), dw)])
    assert torch.equal(dw, dw_ref), 'K7 with the x operand rebuilt while staged'


@pytest.mark.parametrize('cin,cout,hw,stride', UNITS)
@pytest.mark.parametrize('n', BATCHES + [300])
def test_paired_backward_launch_equals_its_two_launches_bit_for_bit(cin, cout, hw, stride, n):
    """ursa_preact_bwd_pair_f32: the input-gradient and the weight-gradient workgroups of a unit interleaved in one grid - the same
    programs on the same operands (n = 300: K7 takes several images per workgroup there, so the two roles differ in number)."""
    x, w, gamma, beta, _, dy = _rand_unit(cin, cout, hw, stride, n, 17 * n + cout)
    K = _K()
    _, _, st = _unfused_fwd(x, w, stride, gamma, beta)
    geo = K.preact_geometry(dy.shape, cin, flip=True, stride=stride)
    g_ref = torch.empty_like(x)
    pb_ref = torch.empty(cin, geo[0], 2, dtype=torch.float64, device=DEV)
    K.preact_conv3x3(dy, w, g_ref, pb_ref, None, stride=stride, flip=True, bwd=(x, st))
    wsf = K.conv_wgrad_ws_floats(x.shape, cout, 3, stride)
    ws_ref = torch.zeros(wsf, device=DEV)
    K.preact_wgrad_partial(x, st, dy, w.shape, ws_ref, stride)
    g = torch.full_like(x, float('nan'))
    pb = torch.full_like(pb_ref, float('nan'))
    ws = torch.zeros(wsf, device=DEV)
    rec = K.preact_bwd_pair(dy, w, g, x, st, pb, ws, stride)
    assert torch.equal(g, g_ref) and torch.equal(pb, pb_ref) and torch.equal(ws, ws_ref)
    dw, dw_ref = torch.empty_like(w), torch.empty_like(w)
    K.conv_wgrad_reduce([(rec, dw)])
    K.conv_wgrad_reduce([((ws_ref,) + rec[1:], dw_ref)])
    assert torch.equal(dw, dw_ref) and torch.isfinite(dw).all()


@pytest.mark.parametrize('cin,cout,hw,stride', UNITS)
def test_integer_inputs_equal_the_oracle_exactly(cin, cout, hw, stride):
    """x = +-1 with every channel balanced: mean 0, variance 1 exactly, eps = 0 -> invstd = 1, alpha = gamma, beta' = beta: with
    integer gamma / beta / w / addend / dy every product and sum is an integer below 2^24, so ANY slip - a tap, a halo column, a
    band edge, a channel, the gate of one element, one element missing from a sum - shows, and the device must equal the oracle's
    restatement bit for bit, sums included."""
    n = 4
    rng = np.random.default_rng(cin + 10 * stride)
    x = np.empty((n, cin, hw, hw), np.float32)
    for c in range(cin):
        v = np.repeat(np.array([1.0, -1.0], np.float32), n * hw * hw // 2)
        rng.shuffle(v)
        x[:, c] = v.reshape(n, hw, hw)
    gamma = rng.integers(1, 4, cin).astype(np.float32)
    beta = rng.integers(-2, 3, cin).astype(np.float32)
    w = rng.integers(-2, 3, (cout, cin, 3, 3)).astype(np.float32)
    ho = hw // stride
    add = rng.integers(-5, 6, (n, cout, ho, ho)).astype(np.float32) if stride == 1 else None
    dy = rng.integers(-2, 3, (n, cout, ho, ho)).astype(np.float32)
    want_y, want_s, want_save = O.preact_fwd(x, w, bn=(gamma, beta), addend=add, stride=stride, eps=0.0, momentum=0.0)
    assert np.array_equal(want_save[0], np.zeros(cin, np.float32)) and np.array_equal(want_save[1], np.ones(cin, np.float32))
    tx, tw, tg, tb, tdy = (torch.from_numpy(a).to(DEV) for a in (x, w, gamma, beta, dy))
    tadd = None if add is None else torch.from_numpy(add).to(DEV)
    y, part, save, _ = _k10_fwd(tx, tw, stride, bn=(tg, tb), add=tadd, eps=0.0, momentum=0.0)
    assert np.array_equal(save.cpu().numpy(), want_save)
    assert np.array_equal(y.cpu().numpy(), want_y)
    assert np.array_equal(part.sum(1).cpu().numpy(), want_s)
    # backward
    want_g, want_bs = O.preact_bwd(dy, w, x, want_save, stride=stride)
    want_dx, want_dg, want_db = O.bn_bwd_dx(x, want_g, gamma, want_save, want_bs)
    K = _K()
    geo = K.preact_geometry(tdy.shape, cin, flip=True, stride=stride)
    g = torch.empty_like(tx)
    pb = torch.empty(cin, geo[0], 2, dtype=torch.float64, device=DEV)
    K.preact_conv3x3(tdy, tw, g, pb, _scratch(geo), stride=stride, flip=True, bwd=(tx, save))
    assert np.array_equal(g.cpu().numpy(), want_g)
    assert np.array_equal(pb.sum(1).cpu().numpy(), want_bs)
    dx, dgb = torch.empty_like(tx), torch.empty(2, cin, device=DEV)
    K.bn_bwd_dx(tx, g, dx, tg, save, pb, dgb[0], dgb[1])
    assert np.array_equal(dgb[0].cpu().numpy(), want_dg) and np.array_equal(dgb[1].cpu().numpy(), want_db)
    assert np.array_equal(dx.cpu().numpy(), want_dx)
    # weight gradient with the staged transform: integers again
    ws = torch.empty(K.conv_wgrad_ws_floats(tx.shape, cout, 3, stride), device=DEV)
    dw = torch.empty_like(tw)
    K.conv_wgrad_reduce([(K.preact_wgrad_partial(tx, save, tdy, tw.shape, ws, stride), dw)])
    h = np.maximum(x * gamma[None, :, None, None] + beta[None, :, None, None], 0).astype(np.float32)
    assert np.array_equal(dw.cpu().numpy(), O.conv_wgrad(h, dy, 3, stride))


@pytest.mark.parametrize('cin,cout,hw,stride', UNITS)
def test_random_inputs_against_the_oracle(cin, cout, hw, stride):
    n = 3
    x, w, gamma, beta, add, dy = _rand_unit(cin, cout, hw, stride, n, 5 + cin)
    add = add if stride == 1 else None
    a = [None if t is None else t.cpu().numpy() for t in (x, w, gamma, beta, add, dy)]
    want_y, want_s, want_save = O.preact_fwd(a[0], a[1], bn=(a[2], a[3]), addend=a[4], stride=stride)
    y, part, save, _ = _k10_fwd(x, w, stride, bn=(gamma, beta), add=add)
    assert np.array_equal(save.cpu().numpy(), want_save), 'statistics: correctly rounded on both sides'
    scale = np.abs(want_y).max()
    assert np.abs(y.cpu().numpy() - want_y).max() <= 2e-6 * scale          # fp32 fma chains vs the double sum rounded once
    assert np.allclose(part.sum(1).cpu().numpy(), want_s, rtol=1e-5, atol=1e-3)
    want_g, want_bs = O.preact_bwd(a[5], a[1], a[0], want_save, stride=stride)
    K = _K()
    geo = K.preact_geometry(dy.shape, cin, flip=True, stride=stride)
    g, pb = torch.empty_like(x), torch.empty(cin, geo[0], 2, dtype=torch.float64, device=DEV)
    K.preact_conv3x3(dy, w, g, pb, _scratch(geo), stride=stride, flip=True, bwd=(x, save))
    assert np.abs(g.cpu().numpy() - want_g).max() <= 2e-6 * np.abs(want_g).max()
    assert np.array_equal(g.cpu().numpy() == 0, want_g == 0) or np.abs((g.cpu().numpy() == 0).sum() - (want_g == 0).sum()) <= 2
    dx, dgb = torch.empty_like(x), torch.empty(2, cin, device=DEV)
    K.bn_bwd_dx(x, g, dx, gamma, save, pb, dgb[0], dgb[1])
    want_dx, want_dg, want_db = O.bn_bwd_dx(a[0], want_g, a[2], want_save, want_bs)
    assert np.abs(dx.cpu().numpy() - want_dx).max() <= 5e-6 * np.abs(want_dx).max()
    assert np.allclose(dgb[0].cpu().numpy(), want_dg, rtol=1e-4, atol=1e-4 * np.abs(want_dg).max())
    assert np.allclose(dgb[1].cpu().numpy(), want_db, rtol=1e-4, atol=1e-4 * np.abs(want_db).max())


@pytest.mark.parametrize('cin,cout,hw,stride', UNITS)
def test_sums_are_bit_reproducible_and_the_scratch_is_reusable(cin, cout, hw, stride):
    """Fixed summation order whatever the arrival order of the workgroups: 20 launches on ONE scratch, identical doubles."""
    x, w, gamma, beta, _, dy = _rand_unit(cin, cout, hw, stride, 128, 3)
    K = _K()
    geo = K.preact_geometry(x.shape, cout, stride=stride, bn=True)
    sc, ip = _scratch(geo), _sums(x)
    first = None
    for _ in range(20):
        y = torch.empty(128, cout, hw // stride, hw // stride, device=DEV)
        part = torch.full((cout, geo[0], 2), float('nan'), dtype=torch.float64, device=DEV)
        K.preact_conv3x3(x, w, y, part, sc, stride=stride, bn=(ip, gamma, beta, None, None, torch.empty(4, cin, device=DEV), 1e-5, 0.0))
        first = part if first is None else first
        assert torch.equal(part, first)
    assert not sc.any()


def test_wrapper_refuses_what_the_library_does_not_cover():
    K = _K()
    assert K.preact_geometry((4, 5, 32, 32), 16, bn=True) is None
    assert K.preact_geometry((4, 3, 32, 32), 16, bn=True) is None           # the stem has no BatchNorm in front
    assert K.preact_geometry((4, 16, 32, 32), 32, stride=2, bn=True, add=True) is None
    x, w = torch.randn(4, 16, 32, 32, device=DEV), torch.randn(16, 16, 3, 3, device=DEV)
    geo = K.preact_geometry(x.shape, 16, bn=True)
    y, part = torch.empty_like(x), torch.empty(16, geo[0], 2, dtype=torch.float64, device=DEV)
    bn = (_sums(x), torch.ones(16, device=DEV), torch.zeros(16, device=DEV), None, None, torch.empty(4, 16, device=DEV), 1e-5, 0.1)
    with pytest.raises(ValueError, match='scratch'):
        K.preact_conv3x3(x, w, y, part, torch.zeros(64, dtype=torch.uint8, device=DEV), bn=bn)
    with pytest.raises(ValueError, match='out_partial'):
        K.preact_conv3x3(x, w, y, part[:, :1].contiguous() if geo[0] > 1 else part.float(), _scratch(geo), bn=bn)
    with pytest.raises(ValueError, match='flip goes with bwd'):
        K.preact_conv3x3(x, w, y, part, _scratch(geo), flip=True)


# ---- the network ------------------------------------------------------------------------------------------------------------
def _step(net, x, y):
    for p in net.parameters():
        p.grad = None
    logits = net(x)
    nn.functional.cross_entropy(logits, y).backward()
    return logits.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}


@pytest.mark.parametrize('pair', [True, False])
@pytest.mark.parametrize('depth,n', [(8, 5), (20, 128), (20, 80), (32, 16)])
def test_network_fused_equals_the_k6_k8_path_bit_for_bit(depth, n, pair, request):
    request.addfinalizer(lambda old=fused_block.paired(pair): fused_block.paired(old))
    torch.manual_seed(depth + n)
    net = models.PreResNet(10, depth).to(DEV).train()
    for m in net.modules():                        # non-trivial affine parameters and running statistics
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    ref = copy.deepcopy(net)
    x, y = torch.randn(n, 3, 32, 32, device=DEV), torch.randint(0, 10, (n,), device=DEV)
    assert fused_block.eligible(net, x)
    old = fused_block.enabled(False)
    try:
        assert not fused_block.eligible(ref, x)
        want_logits, want = _step(ref, x, y)
    finally:
        fused_block.enabled(old)
    logits, got = _step(net, x, y)
    fused_block.check(DEV)
    assert torch.equal(logits, want_logits)
    bad = [k for k in want if not torch.equal(got[k], want[k])]
    assert not bad, f'gradients differ: {bad}'
    for (k, a), (_, b) in zip(net.named_buffers(), ref.named_buffers()):
        assert torch.equal(a, b), k
    # a second step on the same scratch buffers (they must have come back to zero)
    x2 = torch.randn(n, 3, 32, 32, device=DEV)
    old = fused_block.enabled(False)
    try:
        want_logits, want = _step(ref, x2, y)
    finally:
        fused_block.enabled(old)
    logits, got = _step(net, x2, y)
    assert torch.equal(logits, want_logits) and all(torch.equal(got[k], want[k]) for k in want)


def test_network_is_not_fused_where_it_must_not_be():
    net = models.PreResNet(10, 8).to(DEV)
    x = torch.randn(4, 3, 32, 32, device=DEV)
    assert fused_block.eligible(net.train(), x)
    assert not fused_block.eligible(net.eval(), x)                          # running statistics: K6's evaluation launch
    net.train()
    with torch.no_grad():
        assert not fused_block.eligible(net, x)                             # bn_update: no gradient recorded
    with fused_bn.probing(fused_bn.GateProbe(7, 16, DEV, force=False)):
        assert not fused_block.eligible(net, x)                             # the parity instrument observes relu(bn(x))
    assert not fused_block.eligible(net, x.double())
    assert not fused_block.eligible(net, x.cpu())
    assert not fused_block.eligible(net, torch.randn(4, 3, 16, 16, device=DEV))
    old = fused_conv.enabled(False)
    try:
        assert not fused_block.eligible(net, x)
    finally:
        fused_conv.enabled(old)
    assert not fused_block.eligible(models.PreResNet(10, 47).to(DEV).train(), x)     # Bottleneck blocks: 1x1 layers


def test_network_inside_a_replayed_graph_equals_eager():
    """The engine's path: capture one training step, replay it on fresh inputs; the slots / counters come back to zero inside
    every replay, so replays equal eager steps bit for bit."""
    from ursabench_amd._capture import capture
    torch.manual_seed(3)
    net = models.PreResNet(10, 20).to(DEV).train()
    ref = copy.deepcopy(net)
    xs = [torch.randn(128, 3, 32, 32, device=DEV) for _ in range(4)]
    y = torch.randint(0, 10, (128,), device=DEV)
    sx = torch.empty_like(xs[0])
    params = list(net.parameters())
    gbuf = [torch.zeros_like(p) for p in params]

    def step():
        for p in params:
            p.grad = None
        out = net(sx)
        nn.functional.cross_entropy(out, y).backward()
        torch._foreach_copy_(gbuf, [p.grad for p in params])
        return out

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        sx.copy_(xs[0])
        for _ in range(2):
            step()                                  # warm-up: scratch buffers are allocated (and zeroed) outside the capture
    torch.cuda.current_stream().wait_stream(s)
    for (_, a), (_, b) in zip(net.named_buffers(), ref.named_buffers()):
        a.copy_(b)                                  # running statistics back to the start
    g = torch.cuda.CUDAGraph()
    with capture(g):
        out = step()
    for (_, a), (_, b) in zip(net.named_buffers(), ref.named_buffers()):
        a.copy_(b)
    for x in xs:
        sx.copy_(x)
        g.replay()
        want_logits, want = _step(ref, x, y)
        assert torch.equal(out, want_logits)
        for p, gb, (k, _) in zip(params, gbuf, net.named_parameters()):
            assert torch.equal(gb, want[k]), k
    fused_block.check(DEV)


def test_side_branch_of_the_weight_gradients_changes_no_bit():
    """The engine runs K7's first launches on a side stream beside the backward pass (fused_conv.Sink; inside the captured step:
    a parallel branch of the hipGraph). Same launches, same operands, only their place in time differs: a chain sampled with and
    without the side branch ends on identical parameters - eager steps, the captured step and its replays."""
    import ursabench_amd.inference as inference
    from ursabench_amd.data import synthetic
    finals = []
    for side in (True, False):
        torch.manual_seed(11)
        train = synthetic(128 * 6 + 40, (3, 32, 32), 10, seed=5, device=DEV, batch_size=128)      # 6 full batches + a ragged one
        s = inference.SGHMC({'lr': 0.1, 'prior_std': 0.5, 'num_samples': 2, 'alpha': 0.5, 'burn_in_epochs': 0},
                            models.PreResNet(10, 20).to(DEV), train, device=DEV)
        s.engine.wgrad_side = side
        ens = s.sample()
        assert s.engine.stats['graph_replays'] > 0
        finals.append(torch.cat([p.detach().reshape(-1) for p in ens[-1].parameters()]).clone())
    assert torch.isfinite(finals[0]).all()
    assert torch.equal(finals[0], finals[1])
    fused_block.check(DEV)


# ---- evaluation mode: an ensemble member's forward -------------------------------------------------------------------------
@pytest.mark.parametrize('cin,cout,hw,stride', UNITS)
@pytest.mark.parametrize('n', [1, 5, 128, 1000])
def test_evaluation_unit_equals_k6_eval_then_k8_bit_for_bit(cin, cout, hw, stride, n):
    """ursa_preact_conv3x3_f32(URSA_PREACT_EVAL): the running-statistics transform + ReLU while the tile is staged, then K8's
    convolution (+ the residual add): the same floats as ursa_bn_relu_eval_f32 followed by ursa_conv3x3_f32 (and torch's add)."""
    x, w, gamma, beta, add, _ = _rand_unit(cin, cout, hw, stride, n, 13 * n + cin)
    rm, rv = torch.randn(cin, device=DEV) * 0.3, torch.rand(cin, device=DEV) + 0.4
    K = _K()
    assert K.preact_eval_supported(x.shape, cout, stride=stride)
    h = torch.empty_like(x)
    K.bn_relu_eval(x, h, gamma, beta, rm, rv, eps=1e-5, relu=True)
    want = K.conv3x3(h, w, stride=stride)
    y = torch.full_like(want, float('nan'))
    K.preact_eval(x, w, y, gamma, beta, rm, rv, eps=1e-5, stride=stride)
    assert torch.equal(y, want)
    if stride == 1:
        assert K.preact_eval_supported(x.shape, cout, add=True)
        K.preact_eval(x, w, y, gamma, beta, rm, rv, eps=1e-5, add=add)
        assert torch.equal(y, want + add)
    else:
        assert not K.preact_eval_supported(x.shape, cout, stride=2, add=True)


@pytest.mark.parametrize('depth,n', [(8, 7), (20, 128), (20, 1024)])
def test_network_evaluation_forward_fused_vs_the_k6_miopen_path(depth, n):
    """models.PreResNet in eval mode under no_grad: the fused units against K6's evaluation launches + MIOpen's convolutions (the
    path of rounds 2-5): the same function up to the convolutions' summation order (1e-5 of the logits' scale); and the path is
    not taken where it must not be (gradients recorded, training mode)."""
    torch.manual_seed(depth + n)
    net = models.PreResNet(10, depth).to(DEV)
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
            m.running_mean.normal_(0, 0.3)
            m.running_var.uniform_(0.5, 1.5)
    net.eval()
    x = torch.randn(n, 3, 32, 32, device=DEV)
    with torch.no_grad():
        assert fused_block.eval_eligible(net, x)
        got = net(x)
        old = fused_block.eval_fused(False)
        try:
            assert not fused_block.eval_eligible(net, x)
            want = net(x)
        finally:
            fused_block.eval_fused(old)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert not fused_block.eval_eligible(net, x), 'gradients recorded: the stock evaluation path (autograd through MIOpen / ATen)'
    with torch.no_grad():
        assert not fused_block.eval_eligible(net.train(), x)
