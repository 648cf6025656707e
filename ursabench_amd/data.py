"""Device-resident loader for the synthetic benchmark inputs.

Duck-types the bits of torch.utils.data.DataLoader the reference's samplers and tasks touch:
`len(loader.dataset)` (sghmc.py:37, prediction.py:24), `loader.batch_size` (csghmc.py:29),
`len(loader)`, iteration yielding (inputs, labels). Batches are slices of tensors that already
sit in HBM, so `.to(device)` in the loops is a no-op and nothing crosses PCIe."""
import torch


class TensorSet:
    def __init__(self, x, y):
        assert len(x) == len(y)
        self.x, self.y = x, y

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.y[i]


class DeviceLoader:
    def __init__(self, x, y, batch_size, shuffle=False, seed=0, dataset_cls=TensorSet):
        self.dataset = dataset_cls(x, y)
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self._gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        n = len(self.dataset)
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        x, y, n, b = self.dataset.x, self.dataset.y, len(self.dataset), self.batch_size
        if self.shuffle:
            perm = torch.randperm(n, generator=self._gen).to(x.device)
            for i in range(0, n, b):
                idx = perm[i:i + b]
                yield x[idx], y[idx]
        else:
            for i in range(0, n, b):
                yield x[i:i + b], y[i:i + b]


def synthetic(n, shape, num_classes, seed, device, batch_size, shuffle=False):
    """x ~ N(0,1), y ~ U{0..C-1} from a CPU generator (same values on every box), then moved
    to `device` once (BASELINE.md §3 'Synthetic inputs')."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((n,) + tuple(shape), generator=g)
    y = torch.randint(0, num_classes, (n,), generator=g)
    return DeviceLoader(x.to(device), y.to(device), batch_size, shuffle=shuffle, seed=seed)
