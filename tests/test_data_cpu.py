"""CPU: the synthetic device loaders duck-type what the reference's samplers/tasks read from a DataLoader."""
import pytest
import torch

from ursabench_amd import datasets
from ursabench_amd.data import DeviceLoader, synthetic


def test_device_loader_contract():
    l = synthetic(300, (3, 8, 8), 10, seed=4, device='cpu', batch_size=128)
    assert len(l.dataset) == 300 and l.batch_size == 128 and len(l) == 3
    sizes = [len(x) for x, _ in l]
    assert sizes == [128, 128, 44]
    a = torch.cat([x for x, _ in l])
    b = torch.cat([x for x, _ in synthetic(300, (3, 8, 8), 10, seed=4, device='cpu', batch_size=64)])
    assert torch.equal(a, b) and torch.equal(a, l.dataset.x)          # same seed -> same data, order preserved
    x0, y0 = l.dataset[5]
    assert x0.shape == (3, 8, 8) and 0 <= int(y0) < 10
    s = DeviceLoader(l.dataset.x, l.dataset.y, 100, shuffle=True, seed=1)
    e1 = torch.cat([y for _, y in s])
    e2 = torch.cat([y for _, y in s])
    assert sorted(e1.tolist()) == sorted(l.dataset.y.tolist()) and not torch.equal(e1, e2)   # a new permutation per epoch


def test_named_synthetic_datasets():
    loaders, c = datasets.loaders('CIFAR100', batch_size=32, device='cpu', train_size=96, test_size=40)
    assert c == 100 and type(loaders['train'].dataset).__name__ == 'CIFAR100'
    assert loaders['train'].dataset.x.shape == (96, 3, 32, 32) and len(loaders['test'].dataset) == 40
    v, _ = datasets.loaders('MNIST', batch_size=32, device='cpu', train_size=100, use_validation=True, val_size=0.2)
    assert len(v['train'].dataset) == 80 and len(v['test'].dataset) == 20 and v['train'].dataset.x.shape[1:] == (1, 28, 28)
    ood, _ = datasets.loaders('SVHN', batch_size=32, device='cpu', train_size=64, test_size=64)
    assert abs(float(ood['test'].dataset.x.mean()) - 0.5) < 0.1          # OOD sets are shifted/scaled
    import pytest
    with pytest.raises(NotImplementedError):
        datasets.loaders('ImageNet')


def test_deferred_bn_counters_are_the_same_integers():
    """util.deferred_bn_counters: one multi-tensor add instead of one counter kernel per BatchNorm layer; layers with
    momentum=None (cumulative average, the counter feeds the update) keep their own increment."""
    import torch
    from ursabench_amd import models, util
    torch.manual_seed(0)
    a, b = models.PreResNet(10, 8), models.PreResNet(10, 8)
    b.load_state_dict(a.state_dict())
    b.bn.momentum = None                                   # cumulative moving average layer
    a.bn.momentum = None
    x = torch.randn(4, 3, 32, 32)
    a.train(); b.train()
    for _ in range(3):
        a(x)
        with util.deferred_bn_counters(b):
            assert b.layer1[0].bn1.num_batches_tracked is None and b.bn.num_batches_tracked is not None
            b(x)
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    assert int(b.layer1[0].bn1.num_batches_tracked) == 3 and int(b.bn.num_batches_tracked) == 3
    with pytest.raises(RuntimeError):
        with util.deferred_bn_counters(b):
            raise RuntimeError('forward failed')           # counters restored, not bumped
    assert int(b.layer1[0].bn1.num_batches_tracked) == 3
