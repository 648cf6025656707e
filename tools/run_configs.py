"""Reduced runs of BASELINE.json configs[3] (WideResNet-28-10 / CIFAR-100-shaped, SWAG BMA) and
configs[4] (PreResNet-164 / CIFAR-100-shaped, HMC) on one GPU: end-to-end through the product path,
with timings. Synthetic data; sizes trimmed so the whole script takes a few minutes.
    python tools/run_configs.py [c4] [c5]
"""
import json
import os
import sys
import tempfile
import time

os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='ursa_cfg_miopen_'))
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ursabench_amd import inference, models, tasks, util
from ursabench_amd.data import synthetic

dev = torch.device('cuda', 0)


def sync_time(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return out, time.perf_counter() - t0


def c4(n_train=5120, n_test=10000, members=30):
    """hyper-parameters: URSABench/hyperparams/WideResNet28x10CIFAR100/swag_hyperparams.json shape (lr_init,
    swag_lr, swag_wd, momentum); burn-in / iterates cut to 1 + 2 epochs of a 5,120-image synthetic set."""
    util.set_random_seed(0)
    train = synthetic(n_train, (3, 32, 32), 100, seed=0, device=dev, batch_size=128)
    test = synthetic(n_test, (3, 32, 32), 100, seed=1, device=dev, batch_size=128)
    hyp = {'swag_lr': 0.01, 'swag_wd': 3e-4, 'lr_init': 0.1, 'num_samples': members, 'momentum': 0.9,
           'burn_in_epochs': 1, 'num_iterates': 2}
    net = models.WideResNet(100, 28, 10).to(dev)
    s = inference.SWAG(hyp, net, train, device=dev, reference_quirks=False)
    first, t_first = sync_time(lambda: s.sample_iterative())          # trajectory + first draw + bn_update
    rest, t_rest = sync_time(lambda: [s.sample_iterative() for _ in range(members - 1)])
    ens = [first] + rest
    pred = tasks.Prediction({'in_distribution_test': test}, 100, dev, 'ALL')
    _, t_warm = sync_time(lambda: pred.update_statistics(ens[:1], output_performance=False))
    pred.reset()
    _, t_bma = sync_time(lambda: pred.update_statistics(ens, output_performance=False))
    m = pred.get_performance_metrics()
    return {'config': 'C4 WideResNet-28-10 / CIFAR-100-shaped, SWAG (corrected mode), 1 GPU', 'params': s.num_parameters,
            'train_epochs': 3, 'train_images': n_train, 'trajectory_plus_first_sample_s': round(t_first, 2),
            'members': members, 'draw_plus_bn_update_s_per_member': round(t_rest / (members - 1), 3),
            'bma_seconds': round(t_bma, 2), 'bma_preds_per_s': round(n_test / t_bma, 1),
            'bma_member_forwards_per_s': round(members * n_test / t_bma, 1), 'nll': round(float(m['nll']), 4),
            'engine': s.engine.stats}


def c5(n_train=256, num_samples=3, L=3):
    """full-batch HMC; N sized so one forward/backward of PreResNet-164 fits comfortably;
    tau/mass from URSABench/hyperparams/MLP200MNIST/HMC_BO.json's shape."""
    util.set_random_seed(0)
    train = synthetic(n_train, (3, 32, 32), 100, seed=0, device=dev, batch_size=128)
    net = models.PreResNet(100, 164).to(dev)
    s = inference.HMC({'step_size': 2e-4, 'num_samples': num_samples, 'L': L, 'tau': 1.0, 'burn': 0, 'mass': 1.0},
                      net, train, device=dev)
    _, t_warm = sync_time(lambda: s.sample())          # MIOpen solver search, eager warm-ups, graph capture
    s.accepted = 0
    out, t = sync_time(lambda: s.sample())
    return {'first_call_seconds': round(t_warm, 2),'config': 'C5 PreResNet-164 / CIFAR-100-shaped, HMC, 1 chain on 1 GPU', 'params': s.arena.num_parameters,
            'full_batch': n_train, 'proposals': num_samples, 'L': L, 'accepted': s.accepted, 'returned': len(out),
            'seconds': round(t, 2), 'leapfrog_steps_per_s': round(num_samples * L / t, 3)}


if __name__ == '__main__':
    which = sys.argv[1:] or ['c4', 'c5']
    res = [dict(c4=c4, c5=c5)[w]() for w in which]
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(res, open('gpurun_out/r01_configs.json', 'w'), indent=1)
    print(json.dumps(res, indent=1))
