"""CPU: the host logic around K10 / K11 - who takes the fused launches and who does not - without a device. (The launches
themselves: tests/test_fused_block_gpu.py, tests/test_fused_head_gpu.py.)"""
import numpy as np
import torch
import torch.nn as nn

import oracle_lib as O
from ursabench_amd import _native, fused_block, fused_conv, models


def test_host_tensors_never_take_the_fused_launches():
    net = models.PreResNet(10, 8).train()
    x, y = torch.randn(4, 3, 32, 32), torch.randint(0, 10, (4,))
    assert not fused_block.eligible(net, x)
    assert net.forward_loss(x, y, nn.CrossEntropyLoss()) is None
    out = net(x)                                   # the stock ops, op for op the reference's
    assert out.shape == (4, 10) and torch.isfinite(out).all()


def test_geometry_is_a_host_computation_and_matches_the_header():
    """ursa_preact_geometry: lines (<= 16 partial sums per channel for the forward forms, one per workgroup for the input-gradient
    forms, which need no scratch), scratch bytes and the error word's offset - no device involved."""
    K = _native.HipKernels()
    nl, sbytes, wgs, err = K.preact_geometry((128, 16, 32, 32), 16, bn=True)
    assert (nl, wgs) == (16, 512) and sbytes == err + 128 + 16 * 512 * 16 and err == 16 * 128
    assert K.preact_geometry((128, 64, 8, 8), 64, bn=True, add=True)[:3] == (16, 4 * 16 * 128 + 128 + 64 * 64 * 16, 64)
    assert K.preact_geometry((80, 64, 8, 8), 64, bn=True)[0] == 14                   # 40 workgroups in lines of 3
    assert K.preact_geometry((1, 64, 8, 8), 64, bn=True)[0] == 1
    assert K.preact_geometry((128, 16, 32, 32), 16, flip=True)[:3] == (512, 0, 512)
    assert K.preact_geometry((128, 64, 8, 8), 32, flip=True, stride=2)[:3] == (128, 0, 128)
    for shape, cout, kw in [((4, 5, 32, 32), 16, dict(bn=True)), ((4, 3, 32, 32), 16, dict(bn=True)), ((4, 16, 32, 32), 32, dict(stride=2, bn=True, add=True)),
                            ((4, 16, 32, 32), 32, dict(flip=True)), ((4, 16, 16, 16), 16, dict(bn=True))]:
        assert K.preact_geometry(shape, cout, **kw) is None
    assert K.head_supported((128, 64, 8, 8), 10) and not K.head_supported((128, 64, 8, 8), 100)
    assert not K.head_supported((129, 64, 8, 8), 10) and not K.head_supported((128, 64, 4, 4), 10)


def test_sink_without_a_side_stream_runs_in_place():
    with fused_conv.deferred() as pend:
        assert isinstance(pend, fused_conv.Sink) and pend.side is None
        assert pend.launch(lambda: 'rec', 1, 2) == 'rec' and not pend.keep and not pend.forked
    pend.join()


def test_oracle_restatements_of_the_fused_units_against_torch_float64():
    """oracle_preact_* / oracle_bn_bwd_dx / oracle_fc_ce / oracle_bn_relu_pool*: the compositions the device launches are checked
    against, themselves checked against torch's float64 autograd of the reference's ops (preresnet.py:33-52, 146-150)."""
    rng = np.random.default_rng(0)
    N, Cin, Cout, H = 3, 16, 32, 16
    x = rng.standard_normal((N, Cin, H, H), dtype=np.float32)
    w = rng.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * 0.1
    gamma, beta = rng.random(Cin, dtype=np.float32) + 0.5, rng.standard_normal(Cin, dtype=np.float32) * 0.3
    add = rng.standard_normal((N, Cout, H // 2, H // 2), dtype=np.float32)
    y, sums, save = O.preact_fwd(x, w, bn=(gamma, beta), addend=add, stride=2)
    xt, wt = torch.from_numpy(x).double().requires_grad_(), torch.from_numpy(w).double().requires_grad_()
    gt, bt = torch.from_numpy(gamma).double().requires_grad_(), torch.from_numpy(beta).double().requires_grad_()
    h = torch.relu(torch.nn.functional.batch_norm(xt, None, None, gt, bt, True, 0.1, 1e-5))
    yt = torch.nn.functional.conv2d(h, wt, None, 2, 1) + torch.from_numpy(add).double()
    assert np.abs(y - yt.detach().numpy()).max() < 2e-6
    assert np.allclose(sums[:, 0], yt.sum((0, 2, 3)).detach().numpy(), rtol=1e-6, atol=1e-4)
    assert np.allclose(sums[:, 1], (yt * yt).sum((0, 2, 3)).detach().numpy(), rtol=1e-6)
    dy = rng.standard_normal(y.shape, dtype=np.float32)
    yt.backward(torch.from_numpy(dy).double())
    g, bs = O.preact_bwd(dy, w, x, save, stride=2)
    dx, dg, db = O.bn_bwd_dx(x, g, gamma, save, bs)
    assert np.abs(dx - xt.grad.numpy()).max() < 2e-6 and np.abs(dg - gt.grad.numpy()).max() < 1e-5 and np.abs(db - bt.grad.numpy()).max() < 1e-5
    hw = np.maximum(x * save[2][None, :, None, None] + save[3][None, :, None, None], 0).astype(np.float32)
    assert np.abs(O.conv_wgrad(hw, dy, 3, 2) - wt.grad.numpy()).max() < 1e-4
    # the head
    z = rng.standard_normal((6, 8, 8, 8), dtype=np.float32)
    g2, b2 = rng.random(8, dtype=np.float32) + 0.5, rng.standard_normal(8, dtype=np.float32) * 0.2
    W, b = rng.standard_normal((5, 8), dtype=np.float32) * 0.4, rng.standard_normal(5, dtype=np.float32) * 0.1
    t = np.array([0, 4, -100, 2, 1, 3])
    pooled, sv = O.bn_relu_pool(z, g2, b2)
    loss, logits, dW, dbias, dp = O.fc_ce(pooled, W, b, t)
    dz, dgam, dbet = O.bn_relu_pool_bwd(z, dp, g2, b2, sv)
    zt = torch.from_numpy(z).double().requires_grad_()
    g2t, b2t = torch.from_numpy(g2).double().requires_grad_(), torch.from_numpy(b2).double().requires_grad_()
    Wt, bt2 = torch.from_numpy(W).double().requires_grad_(), torch.from_numpy(b).double().requires_grad_()
    pt = torch.relu(torch.nn.functional.batch_norm(zt, None, None, g2t, b2t, True, 0.1, 1e-5)).mean((2, 3))
    lt = torch.nn.functional.cross_entropy(torch.nn.functional.linear(pt, Wt, bt2), torch.from_numpy(t))
    lt.backward()
    assert abs(loss - float(lt)) < 1e-6 and np.abs(pooled - pt.detach().numpy()).max() < 1e-6
    assert np.abs(dW - Wt.grad.numpy()).max() < 1e-6 and np.abs(dbias - bt2.grad.numpy()).max() < 1e-6
    assert np.abs(dz - zt.grad.numpy()).max() < 1e-6 and np.abs(dgam - g2t.grad.numpy()).max() < 1e-5 and np.abs(dbet - b2t.grad.numpy()).max() < 1e-5
