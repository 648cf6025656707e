"""Import the upstream reference (read-only, /root/reference) with stubbed third-party
packages so its inference/ and tasks/ hot path can be *run* on CPU in the build container.

Only tools/gen_golden.py uses this. Nothing here (and nothing under /root/reference) exists
on the GPU box: the outputs travel as small fixtures under tests/golden/.

Recipe: SURVEY.md Appendix B.
"""
import importlib
import sys
import types

REF_ROOT = '/root/reference'


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _AcceptAnything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, x):
        return x


def _make_dataset_stub(name):
    # Decision dispatches on *class identity* (tasks/decision_making.py:90-95), so the
    # synthetic dataset handed to the reference must BE this class.
    def __init__(self, x, y):
        self.x, self.y = x, y

    def __len__(self):
        return len(self.x)

    def __getitem__(self, i):
        return self.x[i], self.y[i]

    return type(name, (), {'__init__': __init__, '__len__': __len__, '__getitem__': __getitem__})


def import_reference():
    """Returns (util, models, inference, tasks) modules of the reference."""
    if 'URSABench.inference' in sys.modules:
        U = sys.modules
        return U['URSABench.util'], U['URSABench.models'], U['URSABench.inference'], U['URSABench.tasks']
    import torch

    pkg = _mod('URSABench')
    pkg.__path__ = [REF_ROOT + '/URSABench']          # skip URSABench/__init__.py (pulls botorch)
    _mod('wandb', log=lambda *a, **k: None)
    ht = _mod('hamiltorch')
    ht.util = _mod('hamiltorch.util')
    tv = _mod('torchvision')
    tv.__path__ = []
    tr = _mod('torchvision.transforms', Compose=_AcceptAnything, Resize=_AcceptAnything,
              RandomCrop=_AcceptAnything, RandomHorizontalFlip=_AcceptAnything,
              ToTensor=_AcceptAnything, Normalize=_AcceptAnything)
    tr.transforms = tr
    tv.transforms = tr
    ds = _mod('torchvision.datasets')
    ds.__path__ = []
    ds.mnist = _mod('torchvision.datasets.mnist', MNIST=_make_dataset_stub('MNIST'))
    ds.cifar = _mod('torchvision.datasets.cifar', CIFAR10=_make_dataset_stub('CIFAR10'),
                    CIFAR100=_make_dataset_stub('CIFAR100'))
    tv.datasets = ds
    import sklearn.decomposition  # noqa: F401
    _mod('sklearn.decomposition.pca', _assess_dimension_=lambda *a, **k: 0.0)
    # bn_update hard-codes input.cuda() (util.py:236): identity on the CPU-only container
    torch.Tensor.cuda = lambda self, *a, **k: self
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):   # "You have not imported hamiltorch" prints
        out = [importlib.import_module('URSABench.' + n) for n in ('util', 'models', 'inference', 'tasks')]
    return tuple(out)


if __name__ == '__main__':
    util, models, inference, tasks = import_reference()
    print('reference imported:', [m.__name__ for m in (util, models, inference, tasks)])
    net = models.PreResNet8.base(num_classes=10, depth=20)
    print('PreResNet-20 params', sum(p.numel() for p in net.parameters()), 'tensors', len(list(net.parameters())))
