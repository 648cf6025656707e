"""GPU: K11 (ursa_bn_relu_pool_f32, ursa_fc_ce_f32, ursa_bn_relu_pool_bwd_f32: include/ursa_hip.h) - the head of a training step
(last BatchNorm + ReLU + AvgPool2d(8) + fc + CrossEntropyLoss and their backward) as three launches - against the oracle's
restatement (sums in double, rounded once), against torch's own ops, and inside the network / the chain engine against the
unfused head."""
import copy

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import oracle_lib as O
from ursabench_amd import _native, fused_block, models

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda', 0)


def _K():
    return _native.default_kernels()


def _sums(x):
    xd = x.double()
    return torch.stack([xd.sum((0, 2, 3)), (xd * xd).sum((0, 2, 3))], -1)[:, None, :].contiguous()


@pytest.mark.parametrize('n,c', [(128, 64), (80, 64), (5, 64), (1, 16), (33, 256)])
def test_bn_relu_pool_forward_and_backward_against_the_oracle(n, c):
    g = torch.Generator().manual_seed(n + c)
    z = torch.randn(n, c, 8, 8, generator=g) * 1.4 + 0.3
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    dp = torch.randn(n, c, generator=g)
    want_rm, want_rv = rm.numpy().copy(), rv.numpy().copy()
    want_p, want_save = O.bn_relu_pool(z.numpy(), gamma.numpy(), beta.numpy(), running=(want_rm, want_rv))
    K = _K()
    dz_, dgamma, dbeta, drm, drv = (t.to(DEV) for t in (z, gamma, beta, rm, rv))
    save, pooled = torch.empty(4, c, device=DEV), torch.full((n, c), float('nan'), device=DEV)
    K.bn_relu_pool(dz_, _sums(dz_), dgamma, dbeta, drm, drv, save, pooled, eps=1e-5, momentum=0.1)
    assert np.array_equal(save.cpu().numpy(), want_save), 'statistics: correctly rounded on both sides'
    assert np.array_equal(drm.cpu().numpy(), want_rm) and np.array_equal(drv.cpu().numpy(), want_rv)
    assert np.abs(pooled.cpu().numpy() - want_p).max() <= 1e-6 * np.abs(want_p).max()
    # ... and equal to K6's normalise launch + torch's mean to rounding
    h = torch.empty_like(dz_)
    K.bn_apply(dz_, h, _sums(dz_), dgamma, dbeta, None, None, torch.empty(4, c, device=DEV), eps=1e-5, momentum=0.0, relu=True)
    assert torch.allclose(pooled, h.mean((2, 3)), rtol=1e-6, atol=1e-7)
    if n <= 128:
        want_dz, want_dg, want_db = O.bn_relu_pool_bwd(z.numpy(), dp.numpy(), gamma.numpy(), beta.numpy(), want_save)
        out, dgb = torch.full_like(dz_, float('nan')), torch.empty(2, c, device=DEV)
        K.bn_relu_pool_bwd(dz_, dp.to(DEV), dgamma, save, out, dgb[0], dgb[1])
        assert np.abs(out.cpu().numpy() - want_dz).max() <= 2e-6 * np.abs(want_dz).max()
        assert np.allclose(dgb[0].cpu().numpy(), want_dg, rtol=2e-6, atol=2e-6 * np.abs(want_dg).max())
        assert np.allclose(dgb[1].cpu().numpy(), want_db, rtol=2e-6, atol=2e-6 * np.abs(want_db).max())


@pytest.mark.parametrize('n,c,k', [(128, 64, 10), (80, 64, 10), (1, 64, 10), (7, 16, 3), (128, 64, 16), (256, 32, 2)])
@pytest.mark.parametrize('bias', [True, False])
def test_fc_ce_against_the_oracle_and_torch(n, c, k, bias):
    g = torch.Generator().manual_seed(n * 7 + k)
    p = torch.rand(n, c, generator=g) * 2
    W, b = torch.randn(k, c, generator=g) * 0.3, (torch.randn(k, generator=g) * 0.2 if bias else None)
    t = torch.randint(0, k, (n,), generator=g)
    if n > 4:
        t[1], t[n - 2] = -100, -100                          # ignored rows (nn.CrossEntropyLoss's default ignore_index)
    want = O.fc_ce(p.numpy(), W.numpy(), None if b is None else b.numpy(), t.numpy())
    K = _K()
    dp_, dW_, db_ = (None if a is None else a.to(DEV) for a in (p, W, b))
    loss = torch.empty(1, device=DEV)
    logits = torch.empty(n, k, device=DEV)
    dW, db, dp = torch.empty_like(dW_), (None if b is None else torch.empty_like(db_)), torch.empty_like(dp_)
    K.fc_ce(dp_, dW_, db_, t.to(DEV), loss, dW, db, dp, logits=logits)
    assert abs(float(loss) - want[0]) <= 2e-6 * abs(want[0])
    assert np.abs(logits.cpu().numpy() - want[1]).max() <= 2e-6 * np.abs(want[1]).max()
    for got, ref, name in ((dW, want[2], 'dW'), (db, want[3], 'db'), (dp, want[4], 'dp')):
        if ref is not None:
            assert np.abs(got.cpu().numpy() - ref).max() <= 5e-6 * np.abs(ref).max() + 1e-9, name
    # torch's own ops on the device
    pt, Wt = dp_.clone().requires_grad_(), dW_.clone().requires_grad_()
    bt = None if b is None else db_.clone().requires_grad_()
    l = F.cross_entropy(F.linear(pt, Wt, bt), t.to(DEV))
    l.backward()
    assert torch.allclose(loss[0], l, rtol=1e-5)
    assert torch.allclose(dW, Wt.grad, rtol=1e-4, atol=1e-6) and torch.allclose(dp, pt.grad, rtol=1e-4, atol=1e-7)
    if bias:
        assert torch.allclose(db, bt.grad, rtol=1e-4, atol=1e-6)


def test_fc_ce_is_loud_about_a_label_out_of_range_and_refuses_what_it_does_not_cover():
    K = _K()
    p, W, b = torch.rand(8, 16, device=DEV), torch.randn(4, 16, device=DEV), torch.zeros(4, device=DEV)
    t = torch.tensor([0, 1, 2, 3, 4, 0, 1, 2], device=DEV)       # 4 is out of range
    loss, dW, db, dp = torch.zeros(1, device=DEV), torch.empty_like(W), torch.empty_like(b), torch.empty_like(p)
    K.fc_ce(p, W, b, t, loss, dW, db, dp)
    assert torch.isnan(loss).all() and torch.isnan(dW).all()
    assert not K.head_supported((128, 64, 8, 8), 100)            # CIFAR-100: the classifier does not fit the one workgroup
    assert not K.head_supported((128, 64, 4, 4), 10) and not K.head_supported((256, 64, 8, 8), 10)
    with pytest.raises(ValueError, match='ursa error -5'):
        K.fc_ce(torch.rand(8, 16, device=DEV), torch.randn(32, 16, device=DEV), None, t, loss, torch.empty(32, 16, device=DEV), None, dp)


def _step(net, x, y, fused):
    crit = nn.CrossEntropyLoss()
    for p in net.parameters():
        p.grad = None
    if fused:
        loss = net.forward_loss(x, y, crit)
        assert loss is not None
        loss.backward(fused_block.one(DEV))
    else:
        loss = crit(net(x), y)
        loss.backward()
    return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters()}


@pytest.mark.parametrize('depth,n', [(8, 5), (20, 128), (20, 80)])
def test_network_loss_and_gradients_with_the_fused_head(depth, n):
    torch.manual_seed(depth * 3 + n)
    net = models.PreResNet(10, depth).to(DEV).train()
    for m in net.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.2)
    ref = copy.deepcopy(net)
    x, y = torch.randn(n, 3, 32, 32, device=DEV), torch.randint(0, 10, (n,), device=DEV)
    want_loss, want = _step(ref, x, y, fused=False)
    loss, got = _step(net, x, y, fused=True)
    assert torch.allclose(loss, want_loss, rtol=2e-6)
    for k in want:
        scale = float(want[k].abs().max())
        assert float((got[k] - want[k]).abs().max()) <= 2e-5 * scale + 1e-9, (k, float((got[k] - want[k]).abs().max()), scale)
    for (k, a), (_, b) in zip(net.named_buffers(), ref.named_buffers()):
        assert torch.equal(a, b), k                              # running statistics and counters: the same launches' arithmetic
    # a caller's own grad_output goes through the scaling path
    for p in net.parameters():
        p.grad = None
    net.forward_loss(x, y, nn.CrossEntropyLoss()).backward(torch.tensor(2.0, device=DEV))
    for k, p in net.named_parameters():
        assert torch.allclose(p.grad, 2 * got[k], rtol=1e-5, atol=1e-8 * float(got[k].abs().max() + 1)), k


def test_forward_loss_only_where_it_applies():
    net = models.PreResNet(10, 8).to(DEV).train()
    x, y = torch.randn(4, 3, 32, 32, device=DEV), torch.randint(0, 10, (4,), device=DEV)
    assert net.forward_loss(x, y, nn.CrossEntropyLoss()) is not None
    assert net.forward_loss(x, y, nn.CrossEntropyLoss(reduction='sum')) is None
    assert net.forward_loss(x, y, nn.CrossEntropyLoss(label_smoothing=0.1)) is None
    assert net.forward_loss(x, y, nn.CrossEntropyLoss(weight=torch.ones(10, device=DEV))) is None
    assert net.forward_loss(x, y, nn.NLLLoss()) is None
    assert net.forward_loss(x, y.int(), nn.CrossEntropyLoss()) is None
    assert net.eval().forward_loss(x, y, nn.CrossEntropyLoss()) is None
    assert models.PreResNet(100, 8).to(DEV).train().forward_loss(x, y, nn.CrossEntropyLoss()) is None       # 100 classes
    assert models.PreResNet_dropout(10, 8).to(DEV).train().forward_loss(x, y, nn.CrossEntropyLoss()) is None  # dropout before fc


def test_engine_with_and_without_the_fused_head_samples_the_same_chain():
    """Two samples of one minibatch step each (the second a hipGraph replay). After the FIRST step no ReLU gate can have moved - the
    units in front of the head are the same launches bit for bit - so the two chains differ by the head's summation trees only
    (held to 2e-6 of the largest parameter); after the second a 1e-8 difference in a weight may have flipped a gate: loose bound."""
    import ursabench_amd.inference as inference
    from ursabench_amd.data import synthetic
    first, second, losses = [], [], []
    for head in (True, False):
        torch.manual_seed(11)
        train = synthetic(128, (3, 32, 32), 10, seed=5, device=DEV, batch_size=128)
        s = inference.SGHMC({'lr': 0.1, 'prior_std': 0.5, 'num_samples': 2, 'alpha': 0.5, 'burn_in_epochs': 0},
                            models.PreResNet(10, 20).to(DEV), train, device=DEV)
        s.engine.fused_head = head
        s.engine.WARMUP_STEPS = 1
        ens = s.sample()
        assert s.engine.stats['graph_replays'] == 1 and s.engine.stats['eager_steps'] == 1, s.engine.stats
        flat = lambda m: torch.cat([p.detach().reshape(-1) for p in m.parameters()]).clone()
        first.append(flat(ens[0]))
        second.append(flat(ens[1]))
        losses.append(float(s.engine.loss_acc))
    scale = float(first[1].abs().max())
    assert torch.isfinite(second[0]).all()
    assert float((first[0] - first[1]).abs().max()) <= 2e-6 * scale
    assert float((second[0] - second[1]).abs().max()) <= 1e-3 * scale
    assert abs(losses[0] - losses[1]) <= 1e-4 * abs(losses[1])
