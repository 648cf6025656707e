"""CPU: host-side switches of ursabench_amd/fused_bn.py (no kernels are launched here)."""
import torch
import torch.nn as nn

from ursabench_amd import fused_bn, util


def test_several_streams_context_nests_and_restores():
    """What ChainGroup's branches and bn_update_many's member streams run under: the held form (ONE launch in flight per
    device, csrc/ursa_bn.hip) is not taken inside, whatever the nesting, and the previous setting comes back - also after
    an exception."""
    assert fused_bn.held() is True
    with fused_bn.several_streams():
        assert fused_bn.held() is False
        with fused_bn.several_streams():
            assert fused_bn.held() is False
        assert fused_bn.held() is False
    assert fused_bn.held() is True
    try:
        with fused_bn.several_streams():
            raise KeyError('x')
    except KeyError:
        pass
    assert fused_bn.held() is True
    old = fused_bn.held(False)
    try:
        with fused_bn.several_streams():
            pass
        assert fused_bn.held() is False            # an outer "off" survives the context
    finally:
        fused_bn.held(old)


def test_host_tensors_take_the_stock_ops_and_bn_update_many_is_unchanged_on_cpu():
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
