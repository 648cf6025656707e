"""Host side of fused_bn (no GPU): on host tensors `bn_relu` / `add_bn_relu` ARE the reference's ops — the module's
forward, the in-place ReLU, torch's add — so the CPU replays of the reference's runs stay op for op; the pending-sum
plumbing of the networks computes what `out += residual` followed by the next block computes."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from ursabench_amd import fused_bn, models, util


def test_host_tensors_take_the_stock_ops():
    torch.manual_seed(0)
    a, b = torch.randn(4, 6, 5, 5), torch.randn(4, 6, 5, 5)
    m1, m2 = nn.BatchNorm2d(6), nn.BatchNorm2d(6)
    assert torch.equal(fused_bn.bn_relu(m1, a), F.relu(m2(a)))
    assert torch.equal(m1.running_var, m2.running_var) and int(m1.num_batches_tracked) == 1
    z, y = fused_bn.add_bn_relu(m1, (a, b))
    assert torch.equal(z, a + b) and torch.equal(y, F.relu(m2(a + b)))
    z, y = fused_bn.add_bn_relu(m1, a)                     # a plain tensor passes through
    assert z is a
    assert torch.equal(fused_bn.bn_relu(m1, a, relu=False), m2(a))
    m2.load_state_dict(m1.state_dict())
    m1.eval(), m2.eval()
    assert torch.equal(fused_bn.add_bn_relu(m1, (a, b))[1], F.relu(m2(a + b)))
    old = fused_bn.enabled(False)
    try:
        assert fused_bn.enabled() is False
    finally:
        fused_bn.enabled(old)
    assert fused_bn.enabled() is old


def _reference_style_forward(net, x):
    """PreResNet.forward written the reference's way (preresnet.py:76-90,139-151): add at the end of each block."""
    x = net.conv1(x)
    for layer in (net.layer1, net.layer2, net.layer3):
        for blk in layer:
            out = F.relu(blk.bn1(x))
            out = blk.conv1(out)
            out = blk.conv2(F.relu(blk.bn2(out)))
            x = out + (x if blk.downsample is None else blk.downsample(x))
    x = F.relu(net.bn(x))
    return net.fc(net.avgpool(x).flatten(1))


def test_pending_sums_compute_the_reference_forward_and_gradients():
    torch.manual_seed(1)
    net = models.PreResNet(10, 8)
    twin = models.PreResNet(10, 8)
    twin.load_state_dict(net.state_dict())
    x = torch.randn(6, 3, 32, 32)
    y1 = net(x)
    y2 = _reference_style_forward(twin, x)
    assert torch.equal(y1, y2)
    y1.square().sum().backward()
    y2.square().sum().backward()
    for (k, p), q in zip(net.named_parameters(), twin.parameters()):
        assert torch.equal(p.grad, q.grad), k
    for (k, b), c in zip(net.named_buffers(), twin.buffers()):
        assert torch.equal(b, c), k


def test_wide_and_bottleneck_blocks_return_pending_sums():
    torch.manual_seed(2)
    w = models.WideResNet(10, 10, 1)
    out = w.layer1(w.conv1(torch.randn(2, 3, 32, 32)))
    assert isinstance(out, tuple) and len(out) == 2 and out[0].shape == out[1].shape
    assert w(torch.randn(2, 3, 32, 32)).shape == (2, 10)
    p = models.PreResNet(100, 47)                          # 9n + 2: bottleneck blocks
    assert p(torch.randn(2, 3, 32, 32)).shape == (2, 100)


def test_held_form_is_opt_in_and_several_streams_nests_and_restores():
    """The held form (ONE launch in flight per device with nothing beside it, csrc/ursa_bn.hip) is OFF unless opted in
    (VERDICT r4 #4 / ADVICE r4 high). Opted in, `several_streams()` - what ChainGroup's branches and bn_update_many's member
    streams run under - switches it off for THIS thread's forwards, whatever the nesting, restores the previous state also
    after an exception, and does not leak into other threads (a module-global flag did)."""
    import threading
    assert fused_bn.held() is False and fused_bn.held_allowed() is False
    with fused_bn.several_streams():
        assert fused_bn.held_allowed() is False
    old = fused_bn.held(True)
    try:
        assert fused_bn.held() is True and fused_bn.held_allowed() is True
        with fused_bn.several_streams():
            assert fused_bn.held_allowed() is False and fused_bn.held() is True
            with fused_bn.several_streams():
                assert fused_bn.held_allowed() is False
            assert fused_bn.held_allowed() is False
            seen = []
            t = threading.Thread(target=lambda: seen.append(fused_bn.held_allowed()))
            t.start()
            t.join()
            assert seen == [True]                     # another thread's forwards are not inside this thread's context
        assert fused_bn.held_allowed() is True
        try:
            with fused_bn.several_streams():
                raise KeyError('x')
        except KeyError:
            pass
        assert fused_bn.held_allowed() is True
    finally:
        fused_bn.held(old)
    assert fused_bn.held_allowed() is False
    assert fused_bn.held_in_use() is False
    fused_bn.check_held()                             # no held scratch registered: no device access, no error


def test_bn_update_many_gives_every_model_bn_updates_statistics_on_cpu():
    """fused_bn never touches the native library for host tensors; bn_update_many without streams (the CPU path) gives
    every model the statistics bn_update gives it alone."""
    torch.manual_seed(0)
    bn = nn.BatchNorm2d(4).train()
    x = torch.randn(8, 4, 5, 5)
    y = fused_bn.bn_relu(bn, x)
    bn2 = nn.BatchNorm2d(4).train()
    assert torch.equal(y, torch.relu(bn2(x)))
    nets = [nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4)) for _ in range(2)]
    one = nn.Sequential(nn.Conv2d(3, 4, 3), nn.BatchNorm2d(4))
    one.load_state_dict(nets[0].state_dict())
    data = [(torch.randn(6, 3, 8, 8), torch.zeros(6)) for _ in range(3)]
    util.bn_update_many(data, nets)
    util.bn_update(data, one)
    assert torch.equal(nets[0][1].running_mean, one[1].running_mean) and torch.equal(nets[0][1].running_var, one[1].running_var)


def test_check_held_reads_raises_and_clears_the_layers_error_words():
    """fused_bn.check_held(): one batched read of every registered layer's error word (BnSync.err at float offset 512 C + 33 of
    the layer's scratch, csrc/ursa_bn.hip), raises naming the layer's device / width / word, and clears the word so that the next
    check speaks of the next launches. Host tensors stand in for the scratch here (the bookkeeping, not the launches)."""
    import torch.nn as nn
    a, b = nn.BatchNorm2d(4), nn.BatchNorm2d(8)
    wa, wb = torch.zeros(4 * 512 + 6 * 32), torch.zeros(8 * 512 + 10 * 32)
    assert not fused_bn.held_in_use()
    fused_bn._held_ws[a], fused_bn._held_ws[b] = (wa, 4), (wb, 8)
    try:
        assert fused_bn.held_in_use()
        fused_bn.check_held()
        wb.view(torch.int32)[8 * 512 + 33] = 1
        with pytest.raises(RuntimeError, match='starved'):
            fused_bn.check_held()
        assert int(wb.view(torch.int32)[8 * 512 + 33]) == 0
        fused_bn.check_held()
    finally:
        del fused_bn._held_ws[a], fused_bn._held_ws[b]
    assert not fused_bn.held_in_use()
