"""GPU: the PARKED experiment of DESIGN.md §10 - K6 launches that also store their output channels-last for the 3x3
convolutions' weight gradients (csrc/ursa_bn.hip k_bn_fwd_apply_4 / k_bn_bwd_dx_4; include/ursa_hip.h
ursa_bn_relu_*_nhwc_f32 under URSA_DEBUG_KNOBS). Measured -2 % on the workload, so since round 5 it is compiled into
csrc/libursa_hip_knobs.so only: the product library does not export it and nothing in ursabench_amd/ calls it
(tests/test_abi_exports.py). What stays tested is the kernels' claim: the twin launches produce the SAME floats as the
channel-tiled launches (they merge the same partial sums in the same lane order) and the twin is the output transposed."""
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
    return _native.knobs_kernels()


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
