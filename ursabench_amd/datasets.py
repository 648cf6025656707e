"""Synthetic stand-ins for URSABench/datasets.py `loaders(...)` (datasets.py:138-261): same return
shape — `({'train': loader, 'test': loader}, num_classes)` — with device-resident synthetic tensors of
the named dataset's shape (there is no network on the benchmark boxes, and the hot path does not
depend on pixel values). x ~ N(0,1) (OOD sets: shifted and scaled), y ~ U{0..C-1}."""
import torch

from .data import DeviceLoader, TensorSet

# name -> (sample shape, classes, train size, test size)
SHAPES = {
    'MNIST': ((1, 28, 28), 10, 60000, 10000), 'FashionMNIST': ((1, 28, 28), 10, 60000, 10000),
    'KMNIST': ((1, 28, 28), 10, 60000, 10000), 'CIFAR10': ((3, 32, 32), 10, 50000, 10000),
    'CIFAR100': ((3, 32, 32), 100, 50000, 10000), 'SVHN': ((3, 32, 32), 10, 73257, 10000),   # test cut to 10k: datasets.py:76-77
    'STL10': ((3, 32, 32), 10, 5000, 8000),
}


def _named_dataset_cls(name):
    # tasks.Decision picks its cost matrix from the dataset class NAME (decision_making.py:90-97 uses
    # torchvision class identity)
    return type(name, (TensorSet,), {})


# classes thinned for the imbalanced decision-making task and the fraction removed (datasets.py:173-195)
IMBALANCE = {'MNIST': ([3, 7], 0.99), 'CIFAR10': ([0, 1, 8, 9], 0.9), 'CIFAR100': ([58, 69, 85], 0.9)}


def increase_data_imbalance(label, x, y, remove_frac=0.9):
    """util.py:356-377 (deterministic form): keep the first int(N - remove_frac*N) rows of class `label`."""
    ind = torch.where(y == label)[0]
    keep_n = int(len(ind) - remove_frac * len(ind))
    drop = torch.zeros(len(y), dtype=torch.bool)
    drop[ind[keep_n:]] = True
    return x[~drop], y[~drop]


# the two CIFAR-10 class halves of `split_classes` (datasets.py:11-14)
C10_CLASSES = ([0, 1, 2, 8, 9], [3, 4, 5, 6, 7])


def _split(x, y, split_classes):
    """datasets.py:224-242: keep the rows of the chosen class half and relabel them 0..4 by position in the half
    (num_classes stays the dataset's: it is computed before the split, datasets.py:168)."""
    half = torch.tensor(C10_CLASSES[split_classes])
    mask = torch.isin(y, half)
    y = (y[mask][:, None] == half[None, :]).nonzero()[:, 1]
    return x[mask], y


def loaders(dataset, path=None, batch_size=128, num_workers=0, transform_train=None, transform_test=None,
            use_validation=False, val_size=0.2, split_classes=None, shuffle_train=True, imbalance=False,
            device='cuda', train_size=None, test_size=None, seed=0, **kwargs):
    if dataset not in SHAPES:
        raise NotImplementedError(dataset)
    if split_classes is not None:
        assert dataset == 'CIFAR10'                       # datasets.py:225-226
        assert split_classes in {0, 1}
    shape, classes, n_train, n_test = SHAPES[dataset]
    n_train, n_test = train_size or n_train, test_size or n_test
    cls = _named_dataset_cls(dataset)
    ood = dataset in ('FashionMNIST', 'KMNIST', 'SVHN', 'STL10')
    out = {}
    for split, n, s in (('train', n_train, seed), ('test', n_test, seed + 1)):
        g = torch.Generator().manual_seed(s + (1000 if ood else 0))
        x = torch.randn((n,) + shape, generator=g)
        if ood:
            x = x * 2.0 + 0.5
        y = torch.randint(0, classes, (n,), generator=g)
        if split == 'train' and imbalance and dataset in IMBALANCE:
            labels, frac = IMBALANCE[dataset]
            for l in labels:
                x, y = increase_data_imbalance(l, x, y, frac)
            n = len(y)
        if split == 'train' and use_validation:
            n_val = int(n * val_size)
            (xt, yt), (xv, yv) = (x[:-n_val], y[:-n_val]), (x[-n_val:], y[-n_val:])
            if split_classes is not None:
                (xt, yt), (xv, yv) = _split(xt, yt, split_classes), _split(xv, yv, split_classes)
            out['train'] = DeviceLoader(xt.to(device), yt.to(device), batch_size, shuffle_train, s, cls)
            out['test'] = DeviceLoader(xv.to(device), yv.to(device), batch_size, False, s, cls)
            return out, classes
        if split_classes is not None:
            x, y = _split(x, y, split_classes)
        out[split] = DeviceLoader(x.to(device), y.to(device), batch_size, shuffle_train and split == 'train', s, cls)
    return out, classes
