"""GPU: K7 (ursa_conv_wgrad_f32) through the C ABI against the oracle (the sum taken in double, rounded once: what
every fp32 summation order approximates - pinned against torch's CPU op in tests/test_fused_conv_cpu.py), and
`fused_conv.Conv2d` inside the benchmark network against the stock launches and against the reference's CPU path."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn

import oracle_lib
from ursabench_amd import _native, fused_conv, models

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)
# (Cin, Cout, H = W, ksize, stride): every convolution of the BasicBlock pre-activation ResNets
LAYERS = [(3, 16, 32, 3, 1), (16, 16, 32, 3, 1), (32, 32, 16, 3, 1), (64, 64, 8, 3, 1), (16, 32, 32, 3, 2), (32, 64, 16, 3, 2),
          (16, 32, 32, 1, 2), (32, 64, 16, 1, 2)]
# fp32 fma chains over K = N*OH*OW products per element, split in S ordered partial sums: |err| <= ~sqrt(K) eps sum|a b|; measured
# 2.5e-7 of max|dw| at K = 131,072 (tools/exp/conv_wgrad_probe.py; MIOpen's own launch: 4e-7..8e-7). Bound used: 2e-6.
RTOL_OF_MAX = 2e-6


def _k7(x, dy, cout, ksize=3, stride=1):
    k = _native.default_kernels()
    ws = torch.empty(k.conv_wgrad_ws_floats(x.shape, cout, ksize, stride), device=DEV)
    dw = torch.full((cout, x.shape[1], ksize, ksize), float('nan'), device=DEV)
    k.conv_wgrad(x, dy, dw, ws, stride)
    return dw


@pytest.mark.parametrize('cin,cout,hw,ksize,stride', LAYERS)
@pytest.mark.parametrize('n', [1, 2, 3, 5, 80, 128])
def test_k7_equals_the_oracle(cin, cout, hw, ksize, stride, n):
    rng = np.random.default_rng(1000 * n + cin)
    x = rng.standard_normal((n, cin, hw, hw), dtype=np.float32)
    dy = rng.standard_normal((n, cout, hw // stride, hw // stride), dtype=np.float32)
    want = oracle_lib.conv_wgrad(x, dy, ksize, stride)
    got = _k7(torch.from_numpy(x).to(DEV), torch.from_numpy(dy).to(DEV), cout, ksize, stride).cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= RTOL_OF_MAX * np.abs(want).max()


@pytest.mark.parametrize('cin,cout,hw,ksize,stride', LAYERS)
def test_k7_structured_inputs_exactly(cin, cout, hw, ksize, stride):
    """Small integers: every product and partial sum is exact in fp32, so ANY indexing slip (a tap, a halo column, a band edge, a
    channel pair, a padded channel) shows as a wrong integer, and the result must equal the oracle bit for bit."""
    rng = np.random.default_rng(7)
    n, ho = 3, hw // stride
    x = rng.integers(-3, 4, (n, cin, hw, hw)).astype(np.float32)
    dy = rng.integers(-2, 3, (n, cout, ho, ho)).astype(np.float32)
    want = oracle_lib.conv_wgrad(x, dy, ksize, stride)
    got = _k7(torch.from_numpy(x).to(DEV), torch.from_numpy(dy).to(DEV), cout, ksize, stride).cpu().numpy()
    assert np.array_equal(got, want)
    # one hot: a single (n, co, oh, ow) and a single (ci, ih, iw) set - exactly one tap of one (co, ci) is non-zero
    x = np.zeros((n, cin, hw, hw), np.float32)
    dy = np.zeros((n, cout, ho, ho), np.float32)
    oh, ow, pad = 1, ho - 1, ksize // 2
    kh, kw = (0, 1) if ksize == 3 else (0, 0)
    x[2, cin - 1, oh * stride + kh - pad, ow * stride + kw - pad] = 2.0
    dy[2, cout - 3, oh, ow] = 3.0
    got = _k7(torch.from_numpy(x).to(DEV), torch.from_numpy(dy).to(DEV), cout, ksize, stride).cpu().numpy()
    want = np.zeros((cout, cin, ksize, ksize), np.float32)
    want[cout - 3, cin - 1, kh, kw] = 6.0
    assert np.array_equal(got, want)


@pytest.mark.parametrize('cin,cout,hw,ksize,stride', LAYERS)
def test_k7_is_bit_reproducible_and_ignores_its_scratch(cin, cout, hw, ksize, stride):
    torch.manual_seed(0)
    x, dy = torch.randn(128, cin, hw, hw, device=DEV), torch.randn(128, cout, hw // stride, hw // stride, device=DEV)
    k = _native.default_kernels()
    ws = torch.full((k.conv_wgrad_ws_floats(x.shape, cout, ksize, stride),), float('nan'), device=DEV)
    a, b = torch.empty(cout, cin, ksize, ksize, device=DEV), torch.empty(cout, cin, ksize, ksize, device=DEV)
    k.conv_wgrad(x, dy, a, ws, stride)
    ws.fill_(123.0)
    k.conv_wgrad(x, dy, b, ws, stride)
    assert torch.equal(a, b) and torch.isfinite(a).all()


def test_deferred_second_launch_equals_the_immediate_one():
    """ursa_conv_wgrad_partial_f32 per layer + ONE ursa_conv_wgrad_reduce_f32 for all of them = ursa_conv_wgrad_f32 per layer,
    bit for bit (the same partial sums, the same ascending order); 60 layers: more than one launch's 48 items."""
    k = _native.default_kernels()
    torch.manual_seed(1)
    pend, want = [], []
    for rep in range(8):
        for cin, cout, hw, ksize, stride in LAYERS[:-1] + LAYERS[-1:] * (1 if rep else 4):
            n = (5, 128)[rep % 2]
            x, dy = torch.randn(n, cin, hw, hw, device=DEV), torch.randn(n, cout, hw // stride, hw // stride, device=DEV)
            ws = torch.empty(k.conv_wgrad_ws_floats(x.shape, cout, ksize, stride), device=DEV)
            dw = torch.empty(cout, cin, ksize, ksize, device=DEV)
            k.conv_wgrad(x, dy, dw, ws, stride)
            want.append(dw)
            ws2 = torch.empty_like(ws)
            pend.append((k.conv_wgrad_partial(x, dy, dw.shape, ws2, stride), torch.full_like(dw, float('nan'))))
    assert len(pend) > 48
    k.conv_wgrad_reduce(pend)
    for (_, got), w in zip(pend, want):
        assert torch.equal(got, w)


def test_wrapper_refuses_what_the_library_does_not_cover():
    k = _native.default_kernels()
    x, dy = torch.randn(4, 5, 32, 32, device=DEV), torch.randn(4, 16, 32, 32, device=DEV)
    dw, ws = torch.empty(16, 5, 3, 3, device=DEV), torch.empty(1 << 20, device=DEV)
    with pytest.raises(ValueError, match='ursa error -5'):
        k.conv_wgrad(x, dy, dw, ws, 1)
    x = torch.randn(4, 16, 32, 32, device=DEV)
    with pytest.raises(ValueError, match='ursa error -2'):                    # scratch too small
        k.conv_wgrad(x, dy, torch.empty(16, 16, 3, 3, device=DEV), torch.empty(64, device=DEV), 1)
    with pytest.raises(ValueError, match='shapes do not belong'):
        k.conv_wgrad(x, dy[:, :8].contiguous(), torch.empty(16, 16, 3, 3, device=DEV), ws, 1)


def _grads(net, x, y):
    for p in net.parameters():
        p.grad = None
    nn.functional.cross_entropy(net(x), y).backward()
    return {k: p.grad.detach().clone() for k, p in net.named_parameters()}


def test_network_gradients_k7_vs_stock_vs_cpu():
    """PreResNet-20 at the workload's batch: every parameter's gradient with K7's weight gradients against the same step with
    MIOpen's (URSA_FUSED_CONV=0's path) and against the reference's CPU path (the same module on the host). The forward is the
    same launches in both GPU runs, so the two differ by the weight gradients' rounding only."""
    torch.manual_seed(3)
    net = models.PreResNet(10, 20).to(DEV).train()
    x, y = torch.randn(128, 3, 32, 32, device=DEV), torch.randint(0, 10, (128,), device=DEV)
    host = copy.deepcopy(net).cpu()
    old = fused_conv.enabled(True), fused_conv.forward_enabled(False)       # K7 alone: forward and dx stay MIOpen's in both runs
    try:
        g7 = _grads(net, x, y)
        fused_conv.enabled(False)
        gs = _grads(net, x, y)
        fused_conv.enabled(True)
        gc = _grads(host, x.cpu(), y.cpu())
        taken = 0
        for k in g7:
            scale = float(gc[k].abs().max()) + 1e-12
            assert float((g7[k] - gs[k]).abs().max()) <= 2e-5 * scale, k           # same forward; K7 vs MIOpen's atomics order
            # vs the CPU path the forward itself differs in the last bits (MIOpen's Winograd vs oneDNN): the stock launches' own
            # distance to the CPU path is the yardstick, K7 must not be further away than 2x that (+ 1e-5)
            d7, ds = float((g7[k].cpu() - gc[k]).abs().max()), float((gs[k].cpu() - gc[k]).abs().max())
            assert d7 <= 2 * ds + 1e-5 * scale, (k, d7, ds, scale)
            taken += k.endswith('conv1.weight') or k.endswith('conv2.weight') or k.endswith('downsample.0.weight')
        assert taken == 21
        # the engine's form: first launches during backward(), ONE second launch for all 21 layers - the same bits
        with fused_conv.deferred() as pend:
            loss = nn.functional.cross_entropy(net(x), y)
        for p in net.parameters():
            p.grad = None
        loss.backward()
        assert len(pend) == 21 and all(w.grad is None for _, w in pend)
        out = {id(w): torch.empty_like(w) for _, w in pend}
        fused_conv.flush(pend, lambda w: out[id(w)])
        names = {id(p): k for k, p in net.named_parameters()}
        for i, t in out.items():
            assert torch.equal(t, g7[names[i]]), names[i]
        assert all((p.grad is None) == (id(p) in out) for p in net.parameters())
    finally:
        fused_conv.enabled(old[0])
        fused_conv.forward_enabled(old[1])


def test_conv2d_takes_k7_only_where_covered():
    calls = []
    k = _native.default_kernels()
    orig = k.conv_wgrad

    def spy(x, dy, dw, ws, stride=1):
        calls.append(tuple(x.shape))
        return orig(x, dy, dw, ws, stride)
    k.conv_wgrad = spy
    try:
        for cin, cout, hw, stride, ksz, covered in ((16, 16, 32, 1, 3, True), (3, 16, 32, 1, 3, True), (16, 32, 32, 2, 3, True), (32, 64, 16, 2, 3, True),
                                                    (16, 32, 32, 2, 1, True), (64, 64, 8, 1, 3, True), (16, 16, 16, 1, 3, False),
                                                    (16, 64, 32, 1, 1, True), (16, 32, 32, 1, 1, False), (8, 16, 32, 1, 3, False)):     # (1x1 / stride 1: K12's shapes)
            m = fused_conv.Conv2d(cin, cout, ksz, stride, ksz // 2, bias=False).to(DEV)
            ref = nn.Conv2d(cin, cout, ksz, stride, ksz // 2, bias=False).to(DEV)
            ref.load_state_dict(m.state_dict())
            x = torch.randn(4, cin, hw, hw, device=DEV)
            xa, xb = x.clone().requires_grad_(), x.clone().requires_grad_()
            n0 = len(calls)
            ya, yb = m(xa), ref(xb)
            ya.square().sum().backward()
            yb.square().sum().backward()
            assert (len(calls) > n0) == covered
            assert torch.allclose(ya, yb, rtol=1e-5, atol=1e-5) and torch.allclose(xa.grad, xb.grad, rtol=1e-4, atol=1e-4)
            scale = float(ref.weight.grad.abs().max())
            assert float((m.weight.grad - ref.weight.grad).abs().max()) <= 1e-5 * scale
        with torch.no_grad():                                   # no gradient recorded: the stock module
            n0 = len(calls)
            fused_conv.Conv2d(16, 16, 3, 1, 1, bias=False).to(DEV)(torch.randn(2, 16, 32, 32, device=DEV))
            assert len(calls) == n0
    finally:
        del k.conv_wgrad


# ---- K8: forward / input gradient -----------------------------------------------------------------------------------------
K8_LAYERS = [(3, 16, 32), (16, 16, 32), (32, 32, 16), (64, 64, 8)]
# two interleaved fp32 fma chains over Cin*9 <= 576 products: measured 2.6e-7 .. 8.2e-7 of max|y| (tools/exp/conv_fwd_probe.py;
# MIOpen's Winograd launch: 2.0e-7 .. 4.2e-7). Bound used: 3e-6.
K8_RTOL_OF_MAX = 3e-6


@pytest.mark.parametrize('cin,cout,hw', K8_LAYERS)
@pytest.mark.parametrize('n', [1, 2, 3, 5, 80, 128])
def test_k8_equals_the_oracle(cin, cout, hw, n):
    rng = np.random.default_rng(77 * n + cin)
    x = rng.standard_normal((n, cin, hw, hw), dtype=np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) * 0.2).astype(np.float32)
    k = _native.default_kernels()
    got = k.conv3x3(torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV)).cpu().numpy()
    want = oracle_lib.conv3x3(x, w)
    assert np.isfinite(got).all() and np.abs(got - want).max() <= K8_RTOL_OF_MAX * np.abs(want).max()
    if cin == cout:                                               # the input gradient of the same layer
        dy = rng.standard_normal((n, cout, hw, hw), dtype=np.float32)
        got = k.conv3x3(torch.from_numpy(dy).to(DEV), torch.from_numpy(w).to(DEV), flip=True).cpu().numpy()
        want = oracle_lib.conv3x3(dy, w, flip=True)
        assert np.abs(got - want).max() <= K8_RTOL_OF_MAX * np.abs(want).max()


@pytest.mark.parametrize('cin,cout,hw', K8_LAYERS)
def test_k8_structured_inputs_exactly(cin, cout, hw):
    """Small integers (every product and sum exact in fp32): equal to the oracle bit for bit, forward and flipped, with weights
    that are NOT symmetric under the flip / transpose - any slip in a tap, a halo, a band edge, a channel group shows."""
    rng = np.random.default_rng(11)
    n = 3
    x = rng.integers(-3, 4, (n, cin, hw, hw)).astype(np.float32)
    w = rng.integers(-2, 3, (cout, cin, 3, 3)).astype(np.float32)
    k = _native.default_kernels()
    got = k.conv3x3(torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV)).cpu().numpy()
    assert np.array_equal(got, oracle_lib.conv3x3(x, w))
    if cin == cout:
        dy = rng.integers(-3, 4, (n, cout, hw, hw)).astype(np.float32)
        got = k.conv3x3(torch.from_numpy(dy).to(DEV), torch.from_numpy(w).to(DEV), flip=True).cpu().numpy()
        assert np.array_equal(got, oracle_lib.conv3x3(dy, w, flip=True))


def test_k8_is_bit_reproducible_and_checks_its_operands():
    k = _native.default_kernels()
    torch.manual_seed(0)
    x, w = torch.randn(128, 32, 16, 16, device=DEV), torch.randn(32, 32, 3, 3, device=DEV)
    assert torch.equal(k.conv3x3(x, w), k.conv3x3(x, w))
    with pytest.raises(ValueError, match='not a 3x3 convolution'):
        k.conv3x3(x, torch.randn(32, 16, 3, 3, device=DEV))
    with pytest.raises(ValueError, match='ursa error -5'):
        k.conv3x3(torch.randn(2, 16, 32, 32, device=DEV), torch.randn(32, 16, 3, 3, device=DEV))
    with pytest.raises(ValueError, match='should be'):
        k.conv3x3(x, w, torch.empty(128, 32, 16, 8, device=DEV))
    # a weight view at storage offset 1 (4-byte aligned only): K8 reads w through 16-byte loads, so the library refuses it
    # (URSA_EALIGN) and the module takes the stock launch for it - same values to rounding
    w_odd = torch.empty(w.numel() + 1, device=DEV)[1:].view_as(w).copy_(w)
    assert w_odd.data_ptr() % 16 == 4 and w_odd.is_contiguous()
    with pytest.raises(ValueError, match='ursa error -3'):
        k.conv3x3(x, w_odd)
    conv = fused_conv.Conv2d(32, 32, 3, 1, 1, bias=False).to(DEV)
    conv.weight = nn.Parameter(w_odd)
    xg = x.clone().requires_grad_()
    y = conv(xg)
    assert conv.__dict__.get('_ursa_plan') is None, 'an odd-offset weight must not reach K8'
    assert torch.allclose(y, k.conv3x3(x, w), rtol=1e-4, atol=1e-4)


def test_network_forward_and_gradients_k8_vs_stock_vs_cpu():
    """PreResNet-20, workload batch: logits and every gradient with K8 + K7 against the all-MIOpen step and the CPU path. The
    forwards differ in the last bits (direct fp32 vs Winograd vs oneDNN), which BatchNorm + ReLU gates can amplify in single
    elements; the yardstick is again the stock launches' own distance to the CPU path."""
    torch.manual_seed(5)
    net = models.PreResNet(10, 20).to(DEV).train()
    x, y = torch.randn(128, 3, 32, 32, device=DEV), torch.randint(0, 10, (128,), device=DEV)
    host = copy.deepcopy(net).cpu()
    old = fused_conv.enabled(True), fused_conv.forward_enabled(True)
    try:
        l8 = net(x).detach()                                      # (gradients recorded: evaluation-mode forwards stay MIOpen's)
        g8 = _grads(net, x, y)
        fused_conv.enabled(False)
        ls = net(x).detach()
        gs = _grads(net, x, y)
    finally:
        fused_conv.enabled(old[0])
        fused_conv.forward_enabled(old[1])
    lc = host(x.cpu()).detach()
    gc = _grads(host, x.cpu(), y.cpu())
    sc = float(lc.abs().max())
    assert float((l8.cpu() - lc).abs().max()) <= 2 * float((ls.cpu() - lc).abs().max()) + 1e-5 * sc
    worse = 0
    for k in g8:
        scale = float(gc[k].abs().max()) + 1e-12
        d8, ds = float((g8[k].cpu() - gc[k]).abs().max()), float((gs[k].cpu() - gc[k]).abs().max())
        assert d8 <= 1e-2 * scale, (k, d8, ds, scale)             # a flipped ReLU gate moves single elements by O(dy) (5e-5 on the stem's
        #                                                           2e-2 seen): a loose cap - indexing is pinned exactly by the integer cases above
        worse += d8 > 2 * ds + 1e-5 * scale
    assert worse <= len(g8) // 4, worse                           # ... and no systematic loss against the stock launches


def test_network_gradients_with_the_references_gates_given_hold_1e5_of_scale():
    """The loose cap above exists because a ReLU gate that differs between the devices moves single elements by O(dy). With the
    CPU run's near-zero gates GIVEN to the backward launches (the parity instrument: tests/gate_lists.py records them on the
    host, fused_bn.GateProbe hands them to ursa_bn_relu_bwd_gated_f32) both devices differentiate the same piecewise-linear
    function, and what is left is rounding: EVERY gradient tensor of the PreResNet-20 step at the workload batch within 1e-5 of
    its scale of the CPU path's, K7 / K8 / K9 convolutions and K6 BatchNorm launches throughout (VERDICT r5 #7 i)."""
    import gate_lists as GL
    from ursabench_amd import fused_bn
    torch.manual_seed(5)
    net = models.PreResNet(10, 20).to(DEV).train()
    x, y = torch.randn(128, 3, 32, 32, device=DEV), torch.randint(0, 10, (128,), device=DEV)
    host = copy.deepcopy(net).cpu()
    log = GL.NearZeroGates(host)
    gc = _grads(host, x.cpu(), y.cpu())
    calls = log.take()
    log.remove()
    probe = fused_bn.GateProbe(len(calls), max(len(c['idx']) for c in calls) + 1, DEV, force=True)
    probe.load([(c['idx'], c['open']) for c in calls])
    with fused_bn.probing(probe):
        g8 = _grads(net, x, y)
    rec = probe.collect()
    assert rec['n_open_as_reference'] == [c['n_open'] for c in calls], 'a gate OUTSIDE the 1e-4 band differs'
    worst = {}
    for k in g8:
        scale = float(gc[k].abs().max()) + 1e-12
        worst[k] = float((g8[k].cpu() - gc[k]).abs().max()) / scale
    bad = {k: v for k, v in worst.items() if v > 1e-5}
    assert not bad, (bad, max(worst.values()))


# ---- K9: the 1x1 / stride 2 shortcuts -------------------------------------------------------------------------------------
@pytest.mark.parametrize('cin,cout,hw', [(16, 32, 32), (32, 64, 16)])
@pytest.mark.parametrize('n', [1, 3, 80, 128])
def test_k9_equals_the_oracle(cin, cout, hw, n):
    rng = np.random.default_rng(5 * n + cin)
    k = _native.default_kernels()
    x = rng.standard_normal((n, cin, hw, hw), dtype=np.float32)
    w = (rng.standard_normal((cout, cin, 1, 1)) * 0.3).astype(np.float32)
    dy = rng.standard_normal((n, cout, hw // 2, hw // 2), dtype=np.float32)
    got = k.conv1x1s2(torch.from_numpy(x).to(DEV), torch.from_numpy(w).to(DEV)).cpu().numpy()
    want = oracle_lib.conv1x1s2(x, w)
    assert np.abs(got - want).max() <= 1e-6 * np.abs(want).max()          # fma chains over 16 / 32 products
    dx = torch.full((n, cin, hw, hw), float('nan'), device=DEV)            # every element must be written (the zeros too)
    k.conv1x1s2(torch.from_numpy(dy).to(DEV), torch.from_numpy(w).to(DEV), dx, flip=True)
    want = oracle_lib.conv1x1s2(dy, w, flip=True)
    got = dx.cpu().numpy()
    assert np.isfinite(got).all() and np.abs(got - want).max() <= 1e-6 * np.abs(want).max()
    assert np.array_equal(got == 0, want == 0) or np.count_nonzero((got == 0) != (want == 0)) <= 2    # the zero pattern (exact zeros of a sum aside)
    # integers: exact, bit for bit
    xi = rng.integers(-3, 4, x.shape).astype(np.float32)
    wi = rng.integers(-2, 3, w.shape).astype(np.float32)
    di = rng.integers(-3, 4, dy.shape).astype(np.float32)
    assert np.array_equal(k.conv1x1s2(torch.from_numpy(xi).to(DEV), torch.from_numpy(wi).to(DEV)).cpu().numpy(), oracle_lib.conv1x1s2(xi, wi))
    assert np.array_equal(k.conv1x1s2(torch.from_numpy(di).to(DEV), torch.from_numpy(wi).to(DEV), flip=True).cpu().numpy(),
                          oracle_lib.conv1x1s2(di, wi, flip=True))


@pytest.mark.parametrize('cin,cout,hw', [(16, 32, 32), (32, 64, 16)])
@pytest.mark.parametrize('n', [1, 3, 80, 128])
def test_k8_stride2_equals_the_oracle(cin, cout, hw, n):
    rng = np.random.default_rng(9 * n + cin)
    k = _native.default_kernels()
    x = rng.standard_normal((n, cin, hw, hw), dtype=np.float32)
    w = (rng.standard_normal((cout, cin, 3, 3)) * 0.2).astype(np.float32)
    dy = rng.standard_normal((n, cout, hw // 2, hw // 2), dtype=np.float32)
    dw = torch.from_numpy(w).to(DEV)
    got = k.conv3x3(torch.from_numpy(x).to(DEV), dw, stride=2).cpu().numpy()
    want = oracle_lib.conv3x3(x, w, stride=2)
    assert got.shape == want.shape and np.abs(got - want).max() <= K8_RTOL_OF_MAX * np.abs(want).max()
    dx = torch.full((n, cin, hw, hw), float('nan'), device=DEV)
    k.conv3x3(torch.from_numpy(dy).to(DEV), dw, dx, flip=True, stride=2)
    want = oracle_lib.conv3x3(dy, w, flip=True, stride=2)
    got = dx.cpu().numpy()
    assert np.isfinite(got).all() and np.abs(got - want).max() <= K8_RTOL_OF_MAX * np.abs(want).max()
    # integers: exact, bit for bit (taps, parity classes, halos, band edges)
    xi = rng.integers(-3, 4, x.shape).astype(np.float32)
    wi = rng.integers(-2, 3, w.shape).astype(np.float32)
    di = rng.integers(-3, 4, dy.shape).astype(np.float32)
    dwi = torch.from_numpy(wi).to(DEV)
    assert np.array_equal(k.conv3x3(torch.from_numpy(xi).to(DEV), dwi, stride=2).cpu().numpy(), oracle_lib.conv3x3(xi, wi, stride=2))
    assert np.array_equal(k.conv3x3(torch.from_numpy(di).to(DEV), dwi, flip=True, stride=2).cpu().numpy(),
                          oracle_lib.conv3x3(di, wi, flip=True, stride=2))


@pytest.mark.parametrize('k10', [False, True])
def test_every_convolution_of_the_network_takes_a_hand_written_launch(k10):
    """PreResNet-20 training step with gradients recorded: no MIOpen convolution launch is left - with the K6 / K8 launches
    (fused_block off: 19 + 18 K8, 21 K7) and with K10's (19 fused forward launches, 18 paired backward launches = input gradient +
    weight gradient of a unit in one grid; the stem's and the two 1x1 shortcuts' weight gradients by K7's plain form)."""
    from ursabench_amd import fused_block
    k = _native.default_kernels()
    seen = dict(conv3x3=0, conv1x1s2=0, conv_wgrad=0, conv_wgrad_partial=0, preact_conv3x3=0, preact_wgrad_partial=0, preact_bwd_pair=0)
    origs = {n: getattr(k, n) for n in seen}

    def wrap(name):
        def f(*a, **kw):
            seen[name] += 1
            return origs[name](*a, **kw)
        return f
    for n in seen:
        setattr(k, n, wrap(n))
    old = fused_block.enabled(k10)
    try:
        torch.manual_seed(0)
        net = models.PreResNet(10, 20).to(DEV).train()
        x, y = torch.randn(16, 3, 32, 32, device=DEV), torch.randint(0, 10, (16,), device=DEV)
        _grads(net, x, y)
    finally:
        fused_block.enabled(old)
        for n in seen:
            delattr(k, n)
    if k10:          # the 18 units' input-gradient + weight-gradient launches go out as 18 paired launches
        assert seen == dict(conv3x3=0, conv1x1s2=2 + 2, conv_wgrad=2, conv_wgrad_partial=1, preact_conv3x3=19, preact_wgrad_partial=0,
                            preact_bwd_pair=18), seen
    else:
        assert seen == dict(conv3x3=19 + 18, conv1x1s2=2 + 2, conv_wgrad=21, conv_wgrad_partial=0, preact_conv3x3=0, preact_wgrad_partial=0,
                            preact_bwd_pair=0), seen


@pytest.mark.parametrize('cin,cout,hw,n', [(16, 16, 32, 400), (32, 32, 16, 500), (64, 64, 8, 1600)])
def test_large_batches_take_longer_k_slices_and_stay_exact(cin, cout, hw, n):
    """Beyond 768 K slices K7 gives a workgroup several images (HMC's 1,024-row chunks): the slice count stays bounded and
    the result - small integers, exact in fp32 - still equals the oracle bit for bit; K8 at the same batch likewise."""
    k = _native.default_kernels()
    per_copy = cout * cin * 9
    slices = k.conv_wgrad_ws_floats((n, cin, hw, hw), cout, 3, 1) // per_copy
    assert slices <= 768 and slices < k.conv_wgrad_ws_floats((128, cin, hw, hw), cout, 3, 1) // per_copy * n // 128
    rng = np.random.default_rng(n)
    x = rng.integers(-2, 3, (n, cin, hw, hw)).astype(np.float32)
    dy = rng.integers(-2, 3, (n, cout, hw, hw)).astype(np.float32)
    w = rng.integers(-2, 3, (cout, cin, 3, 3)).astype(np.float32)
    dx_, ddy, dw_ = (torch.from_numpy(a).to(DEV) for a in (x, dy, w))
    assert np.array_equal(_k7(dx_, ddy, cout).cpu().numpy(), oracle_lib.conv_wgrad(x, dy, 3, 1))
    assert np.array_equal(k.conv3x3(dx_, dw_).cpu().numpy(), oracle_lib.conv3x3(x, w))
    assert np.array_equal(k.conv3x3(ddy, dw_, flip=True).cpu().numpy(), oracle_lib.conv3x3(dy, w, flip=True))


# ---- K12: the 1x1 / stride 1 layers of the Bottleneck networks -------------------------------------------------------------
K12_SHAPES = [(64, 16, 32), (16, 64, 32), (128, 32, 16), (32, 128, 16), (256, 64, 8), (64, 256, 8), (16, 16, 32), (64, 32, 32), (128, 64, 16)]


@pytest.mark.parametrize('cin,cout,hw', K12_SHAPES)
@pytest.mark.parametrize('n', [1, 3, 80])
def test_k12_equals_the_oracle(cin, cout, hw, n):
    """Forward, input gradient and weight gradient of a 1x1 / stride 1 layer: random data against the oracle's double sums
    (fp32 fma chains over <= 256 products / over N*H*W products in ordered slices), small integers bit for bit."""
    rng = np.random.default_rng(17 * n + cin + cout)
    k = _native.default_kernels()
    x = rng.standard_normal((n, cin, hw, hw), dtype=np.float32)
    w = (rng.standard_normal((cout, cin, 1, 1)) * (1.0 / cin) ** 0.5).astype(np.float32)
    dy = rng.standard_normal((n, cout, hw, hw), dtype=np.float32)
    tx, tw, tdy = (torch.from_numpy(a).to(DEV) for a in (x, w, dy))
    assert k.conv1x1_supported(tx.shape, cout) and k.conv1x1_supported(tdy.shape, cin, flip=True)
    y = torch.full((n, cout, hw, hw), float('nan'), device=DEV)
    k.conv1x1(tx, tw, y)
    want = oracle_lib.conv1x1(x, w)
    assert np.abs(y.cpu().numpy() - want).max() <= 2e-6 * np.abs(want).max()
    dx = torch.full((n, cin, hw, hw), float('nan'), device=DEV)
    k.conv1x1(tdy, tw, dx, flip=True)
    want = oracle_lib.conv1x1(dy, w, flip=True)
    assert np.abs(dx.cpu().numpy() - want).max() <= 2e-6 * np.abs(want).max()
    wsf = k.conv_wgrad_ws_floats(tx.shape, cout, 1, 1)
    if (cin, cout) != (16, 16):
        assert wsf > 0
        dw = torch.full((cout, cin, 1, 1), float('nan'), device=DEV)
        k.conv_wgrad(tx, tdy, dw, torch.empty(wsf, device=DEV), 1)
        want = oracle_lib.conv_wgrad(x, dy, 1, 1)
        assert np.abs(dw.cpu().numpy() - want).max() <= RTOL_OF_MAX * np.abs(want).max()
    else:
        assert wsf == 0                                     # one 16 x 16 tile: left to the stock launch
    # integers: exact, bit for bit
    xi = rng.integers(-3, 4, x.shape).astype(np.float32)
    wi = rng.integers(-2, 3, w.shape).astype(np.float32)
    di = rng.integers(-2, 3, dy.shape).astype(np.float32)
    txi, twi, tdi = (torch.from_numpy(a).to(DEV) for a in (xi, wi, di))
    assert np.array_equal(k.conv1x1(txi, twi).cpu().numpy(), oracle_lib.conv1x1(xi, wi))
    assert np.array_equal(k.conv1x1(tdi, twi, flip=True).cpu().numpy(), oracle_lib.conv1x1(di, wi, flip=True))
    if wsf:
        dwi = torch.empty(cout, cin, 1, 1, device=DEV)
        k.conv_wgrad(txi, tdi, dwi, torch.empty(wsf, device=DEV), 1)
        assert np.array_equal(dwi.cpu().numpy(), oracle_lib.conv_wgrad(xi, di, 1, 1))


@pytest.mark.parametrize('cin,cout,hw', [(16, 64, 32), (256, 64, 8)])
def test_k12_at_the_hmc_batch_against_torch_and_in_the_module(cin, cout, hw):
    """1,024 rows (several images per workgroup, 256 K slices): against torch's own convolution on the device, through
    fused_conv.Conv2d with autograd; bit-reproducible."""
    torch.manual_seed(cin)
    conv = fused_conv.Conv2d(cin, cout, 1, bias=False).to(DEV)
    x = torch.randn(1024, cin, hw, hw, device=DEV, requires_grad=True)
    dy = torch.randn(1024, cout, hw, hw, device=DEV)
    y = conv(x)
    assert conv.__dict__['_ursa_plan'][2] and conv.__dict__['_ursa_plan'][3] and conv.__dict__['_ursa_plan'][1] > 0
    y.backward(dy)
    gx, gw = x.grad.clone(), conv.weight.grad.clone()
    xr = x.detach().clone().requires_grad_()
    wr = conv.weight.detach().clone().requires_grad_()
    yr = torch.nn.functional.conv2d(xr, wr)
    yr.backward(dy)
    assert torch.allclose(y, yr, rtol=1e-4, atol=1e-5 * float(yr.abs().max()))
    assert torch.allclose(gx, xr.grad, rtol=1e-4, atol=1e-5 * float(xr.grad.abs().max()))
    assert float((gw - wr.grad).abs().max()) <= 2e-5 * float(wr.grad.abs().max())
    x.grad = None
    conv.weight.grad = None
    conv(x).backward(dy)
    assert torch.equal(x.grad, gx) and torch.equal(conv.weight.grad, gw)
