"""GPU: the channels-last twins K6 stores for the 3x3 convolutions' weight gradients (csrc/ursa_bn.hip k_bn_fwd_apply_t /
k_bn_bwd_dx_t / one-pass scatter; include/ursa_hip.h ursa_bn_relu_*_nhwc_f32) and `fused_conv.conv2d`, whose backward takes
the weight gradient on them. The twin launches must produce the SAME floats as the channel-tiled launches they replace
(they merge the same partial sums in the same lane order), the twin must be the output transposed, and the networks'
gradients must equal the stock backward's up to the weight-gradient kernel's own summation order."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

SHAPES = [(128, 16, 32, 32), (128, 32, 16, 16), (128, 64, 8, 8),      # PreResNet-20's stages (two-launch, two-launch, one-pass)
          (16, 160, 32, 32), (16, 640, 8, 8),                          # WideResNet widths: 40 / 160 channel groups
          (7, 12, 6, 6), (3, 4, 2, 2), (5, 1024, 2, 2), (2, 20, 34, 30)]   # ragged tiles, one position tile, most channels, C/4 = 5


@pytest.fixture(scope='module')
def K():
    from ursabench_amd import _native
    return _native.default_kernels()


@pytest.mark.parametrize('shape', SHAPES)
@pytest.mark.parametrize('relu', [True, False])
@pytest.mark.parametrize('two_launch', [False, True])
def test_twin_launches_equal_plain_launches(K, shape, relu, two_launch):
    from ursabench_amd import _native
    g = torch.Generator().manual_seed(sum(shape) + relu)
    C = shape[1]
    a, b, dy, dz = (torch.randn(shape, generator=g).cuda() for _ in range(4))
    w, bb = (torch.rand(C, generator=g) + 0.5).cuda(), torch.randn(C, generator=g).cuda()
    assert _native.nhwc_twin_supported(a)
    ws = torch.empty(_native.bn_ws_floats(C), device='cuda')
    out = {}
    for twin in (False, True):
        for resid in (False, True):
            z, y, dx = (torch.full_like(a, float('nan')) for _ in range(3))
            yt = _native.nhwc_twin(a).fill_(float('nan')) if twin else None
            dxt = _native.nhwc_twin(a).fill_(float('nan')) if twin else None
            sm, si, dw, db = (torch.full((C,), float('nan'), device='cuda') for _ in range(4))
            rm, rv = torch.zeros(C, device='cuda'), torch.ones(C, device='cuda')
            K.bn_relu_forward(a, y, w, bb, rm, rv, sm, si, ws, eps=1e-5, momentum=0.1, relu=relu, two_launch=two_launch, y_nhwc=yt,
                              **(dict(addend=b, z_out=z) if resid else {}))
            K.bn_relu_backward(z if resid else a, dy, dx, w, bb, sm, si, dw, db, ws, relu=relu, two_launch=two_launch,
                               dz=dz if resid else None, dx_nhwc=dxt)
            out[(twin, resid)] = [y, sm, si, rm, rv, dx, dw, db] + ([z] if resid else [])
            if twin:
                assert yt.is_contiguous(memory_format=torch.channels_last) and torch.equal(yt, y), 'y twin'
                assert torch.equal(dxt, dx), 'dx twin'
                phys = yt.permute(0, 2, 3, 1)                                      # the memory really is [N, H, W, C]
                assert phys.is_contiguous() and torch.equal(phys, y.permute(0, 2, 3, 1))
    for resid in (False, True):
        for t1, t2, name in zip(out[(True, resid)], out[(False, resid)],
                                ('y', 'mean', 'invstd', 'running_mean', 'running_var', 'dx', 'dgamma', 'dbeta', 'z')):
            assert not torch.isnan(t1).any(), name
            assert torch.equal(t1, t2), (name, resid)                               # same floats, bit for bit


def test_twin_arguments(K):
    from ursabench_amd import _native
    x = torch.randn(4, 6, 4, 4, device='cuda')                                       # C % 4 != 0
    assert not _native.nhwc_twin_supported(x)
    C = 6
    v = torch.ones(C, device='cuda')
    ws = torch.empty(_native.bn_ws_floats(C), device='cuda')
    with pytest.raises(ValueError):
        K.bn_relu_forward(x, torch.empty_like(x), v, v, None, None, v.clone(), v.clone(), ws, eps=1e-5, momentum=0.0,
                          y_nhwc=_native.nhwc_twin(x))
    x = torch.randn(4, 8, 4, 4, device='cuda')
    with pytest.raises(ValueError):
        K.bn_relu_forward(x, torch.empty_like(x), torch.ones(8, device='cuda'), torch.ones(8, device='cuda'), None, None,
                          torch.ones(8, device='cuda'), torch.ones(8, device='cuda'),
                          torch.empty(_native.bn_ws_floats(8), device='cuda'), eps=1e-5, momentum=0.0,
                          y_nhwc=torch.empty_like(x))                                 # not channels-last


@pytest.mark.parametrize('cin,cout,hw,stride,bias', [(16, 16, 32, 1, False), (16, 32, 32, 2, False), (64, 64, 8, 1, False),
                                                      (160, 160, 16, 1, True)])
def test_fused_conv_backward_equals_stock(cin, cout, hw, stride, bias):
    """relu(bn(x)) -> conv 3x3 -> relu(bn(.)) with twins: the convolution's backward takes the weight gradient on the
    channels-last twins (counter asserted) and the result equals the stock backward on the same tensors."""
    from ursabench_amd import fused_bn, fused_conv
    from ursabench_amd.fused_bn import TWIN_DX, TWIN_Y
    torch.manual_seed(0)
    bn1, bn2 = nn.BatchNorm2d(cin).cuda(), nn.BatchNorm2d(cout).cuda()
    conv = nn.Conv2d(cin, cout, 3, stride, 1, bias=bias).cuda()
    x = torch.randn(32, cin, hw, hw, device='cuda', requires_grad=True)
    g = torch.randn(32, cout, hw // stride, hw // stride, device='cuda')
    res = {}
    for mode in (True, False):
        old = fused_bn.nhwc_twins(mode)
        fused_conv.stats.update(nhwc_wgrad=0, stock_backward=0)
        try:
            h = fused_bn.bn_relu(bn1, x, twins=TWIN_Y | TWIN_DX)
            assert (getattr(h, '_ursa_nhwc', None) is not None) == (mode and hw * hw >= fused_bn.TWIN_MIN_HW)
            y = fused_conv.conv2d(conv, h)
            out = fused_bn.bn_relu(bn2, y, twins=TWIN_Y | TWIN_DX)
            grads = torch.autograd.grad(out, [x, conv.weight] + ([conv.bias] if bias else []) + [bn1.weight, bn2.weight], g)
            res[mode] = [out.detach()] + list(grads)
            assert fused_conv.stats['nhwc_wgrad'] == (1 if mode and hw * hw >= fused_bn.TWIN_MIN_HW and (hw // stride) ** 2 >= fused_bn.TWIN_MIN_HW else 0), fused_conv.stats
        finally:
            fused_bn.nhwc_twins(old)
    names = ['out', 'dx', 'dw'] + (['dbias'] if bias else []) + ['dgamma1', 'dgamma2']
    for name, t1, t2 in zip(names, res[True], res[False]):
        assert t1.shape == t2.shape and t1.is_contiguous()
        # (the bias gradient of a convolution that feeds a BatchNorm is analytically zero: what is left is the rounding of a
        #  sum of N*H*W terms of size |dy|, so that - not its own near-zero value - is its scale)
        scale = float(t2.abs().max()) if name != 'dbias' else float(res[False][0].numel() / cout)
        assert float((t1 - t2).abs().max()) <= 2e-6 * scale, name


@pytest.mark.parametrize('name', ['PreResNet20', 'PreResNet164', 'WideResNet28x10'])
def test_networks_take_the_nhwc_weight_gradient(name):
    """One training step of the benchmark networks with and without the twins: same loss, and EVERY 3x3 convolution with a
    K6 input took the channels-last path. Gradients: MIOpen's weight-gradient kernel accumulates its split-K partial
    sums with atomics, so the stock backward itself differs from run to run (WideResNet-28-10 at batch 8: 3e-3 of a tensor's
    scale against float64, tools/exp/nhwc_wgrad_accuracy.py; the twins' path 1e-3); the two paths must agree to within
    that noise - 10x the stock path's own run-to-run difference plus a per-network floor - on every weight (convolution
    biases in front of a BatchNorm have an analytically zero gradient and are skipped); WideResNet-28-10, where one pair of
    stock runs is too noisy a yardstick, is judged against a float64 run of the same step: the twins' median error over
    the weight tensors must not exceed 3x the stock path's."""
    from ursabench_amd import fused_bn, fused_conv, models
    torch.manual_seed(0)
    cfg = getattr(models, name)
    classes = 10 if name == 'PreResNet20' else 100
    net = cfg.base(num_classes=classes, **cfg.kwargs).cuda().train()
    bs = 32 if name == 'PreResNet20' else 8
    x, y = torch.randn(bs, 3, 32, 32, device='cuda'), torch.randint(0, classes, (bs,), device='cuda')
    # the 3x3 convolutions whose input AND output maps are large enough for a twin (fused_bn.TWIN_MIN_HW) and whose input is
    # a K6 output (every one but the stem)
    shapes, hooks = [], []
    for m in net.modules():
        if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and m.in_channels > 3:
            hooks.append(m.register_forward_hook(lambda mod, i, o: shapes.append((i[0].shape[2] * i[0].shape[3], o.shape[2] * o.shape[3]))))
    with torch.no_grad():
        net(x)
    for h in hooks:
        h.remove()
    n3x3 = sum(1 for hi, ho in shapes if hi >= fused_bn.TWIN_MIN_HW and ho >= fused_bn.TWIN_MIN_HW)
    assert n3x3 >= 6
    res = []
    for mode in (True, False, False):
        old = fused_bn.nhwc_twins(mode)
        fused_conv.stats.update(nhwc_wgrad=0, stock_backward=0)
        try:
            for m in net.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.reset_running_stats()
            loss = torch.nn.functional.cross_entropy(net(x), y)
            grads = torch.autograd.grad(loss, list(net.parameters()))
            res.append((loss.detach(), grads))
            if mode:
                assert fused_conv.stats['nhwc_wgrad'] == n3x3, (fused_conv.stats, n3x3)
            else:
                assert fused_conv.stats['nhwc_wgrad'] == 0
        finally:
            fused_bn.nhwc_twins(old)
    # the forward pass computes the same statistics; where the plain path takes the one-pass form and the twin path the
    # two-launch form a channel's invstd may differ in the last place (test_fused_bn_gpu.py), hence not torch.equal
    assert torch.allclose(res[0][0], res[1][0], rtol=2e-6, atol=0)
    conv_bias = {k + '.bias' for k, m in net.named_modules() if isinstance(m, nn.Conv2d)}
    names = [k for k, _ in net.named_parameters()]
    if name == 'WideResNet28x10':
        # run-to-run spread of ONE pair of stock runs is too noisy a yardstick here: judge both paths against float64
        net64 = cfg.base(num_classes=classes, **cfg.kwargs).double()
        net64.load_state_dict({k: v.double().cpu() for k, v in net.state_dict().items()})
        net64.train()
        for m in net64.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.reset_running_stats()
        ref = torch.autograd.grad(torch.nn.functional.cross_entropy(net64(x.double().cpu()), y.cpu()), list(net64.parameters()))
        err = lambda gs: np.median([float((g.double().cpu() - r).abs().max()) / float(r.abs().max())
                                    for k, g, r in zip(names, gs, ref) if k not in conv_bias])
        e_twin, e_stock = err(res[0][1]), err(res[1][1])
        assert e_twin <= 3 * e_stock + 1e-4, (e_twin, e_stock)
        return
    for k, g1, g2, g3 in zip(names, res[0][1], res[1][1], res[2][1]):
        assert g1.is_contiguous() and g1.shape == g2.shape
        if k in conv_bias:
            continue
        scale = float(g2.abs().max())
        noise = float((g2 - g3).abs().max())
        floor = {'PreResNet20': 1e-4, 'PreResNet164': 1e-3}[name]
        assert float((g1 - g2).abs().max()) <= 10 * noise + floor * scale + 1e-12, (k, noise / max(scale, 1e-30))
