"""PCIe-inclusive rate: the same PreResNet-20 SGHMC sample with the training set left in (pinned / pageable)
HOST memory, every minibatch crossing PCIe, vs resident in HBM."""
import os, sys, tempfile, time
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='ursa_hd_'))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import inference, models, util
from ursabench_amd.data import DeviceLoader, synthetic

dev = torch.device('cuda', 0)
hyp = {'lr': 0.1, 'prior_std': 0.5, 'num_samples': 3, 'alpha': 0.5, 'burn_in_epochs': 0}
base = synthetic(50000, (3, 32, 32), 10, seed=0, device='cpu', batch_size=128)
for name, x, y in (('HBM-resident', base.dataset.x.to(dev), base.dataset.y.to(dev)),
                   ('host pinned', base.dataset.x.pin_memory(), base.dataset.y.pin_memory()),
                   ('host pageable', base.dataset.x, base.dataset.y)):
    util.set_random_seed(0)
    s = inference.SGHMC(dict(hyp), models.PreResNet(10, 20).to(dev), DeviceLoader(x, y, 128), device=dev)
    s.sample_iterative()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.sample_iterative(); s.sample_iterative()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 2
    print(f'{name:14s}: {1 / dt:.3f} posterior-samples/s ({dt * 1e3:.0f} ms/sample)', flush=True)
