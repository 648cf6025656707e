"""Timing harness with the protocol of URSABench/time_script.py:70-125: for each method, T trials of
the wall time of `sample()` for S samples with burn-in forced to 0, mean and std to
`<save_path>.json` as `{<method>_mean, <method>_std}`. Hyper-parameters come from
`<hyperparams_path><method>_BO.json` like the reference, or from built-in defaults.

    python -m ursabench_amd.time_script --dataset CIFAR10 --model PreResNet20 --save_path timing \
        --methods SGLD SGHMC cSGHMC SWAG --samples 3 --trials 10
"""
import argparse
import json
import os
import time

import torch

from . import datasets, inference, models, util

DEFAULTS = {
    'SGLD': {'lr': 0.1, 'prior_std': 0.5, 'num_samples': 3, 'alpha': 1.0, 'burn_in_epochs': 0},
    'SGHMC': {'lr': 0.1, 'prior_std': 0.5, 'num_samples': 3, 'alpha': 0.5, 'burn_in_epochs': 0},
    'cSGLD': {'lr_0': 0.1, 'prior_std': 0.5, 'num_samples_per_cycle': 3, 'cycle_length': 4, 'burn_in_epochs': 0,
              'num_cycles': 1, 'alpha': 1.0},
    'cSGHMC': {'lr_0': 0.1, 'prior_std': 0.5, 'num_samples_per_cycle': 3, 'cycle_length': 4, 'burn_in_epochs': 0,
               'num_cycles': 1, 'alpha': 0.5},
    'SWAG': {'swag_lr': 0.01, 'swag_wd': 3e-4, 'lr_init': 0.1, 'num_samples': 3, 'momentum': 0.9,
             'burn_in_epochs': 1, 'num_iterates': 1},
    'HMC': {'step_size': 2e-4, 'num_samples': 3, 'L': 3, 'tau': 1.0, 'burn': -1, 'mass': 1.0},
    'MCdropout': {'lr': 0.01, 'epochs': 0, 'dropout': 0.2, 'lengthscale': 0.01, 'num_samples': 3, 'momentum': 0.9,
                  'weight_decay': 0},
    'SGD': {'lr': 0.1, 'epochs': 0, 'momentum': 0.9, 'weight_decay': 5e-4},
}


def prepare(method, hyp, S):
    """time_script.py:82-98."""
    hyp = dict(hyp)
    if method == 'HMC':
        hyp['burn'] = -1
    if method == 'SWAG':
        hyp['burn_in_epochs'] = 1
    if method in ('SGHMC', 'SGLD'):
        hyp['burn_in_epochs'] = 0
    if method in ('cSGHMC', 'cSGLD'):
        hyp['burn_in_epochs'] = 0
        hyp['num_cycles'] = 1
        hyp['num_samples_per_cycle'] = S
    if method in ('MCdropout', 'SGD'):
        hyp['epochs'] = 0
    hyp['num_samples'] = S
    return hyp


def run(args, device=None, kernels=None):
    kw = {} if kernels is None else {'kernels': kernels}
    if device is None:
        torch.cuda.set_device(args.device_num)
        device = torch.device('cuda', args.device_num)
    util.set_random_seed(args.seed)
    model_cfg = getattr(models, args.model)
    loaders, num_classes = datasets.loaders(args.dataset, batch_size=args.batch_size, device=device,
                                            train_size=args.train_size, test_size=args.test_size or 128)
    timer = {}
    for method in args.methods:
        if args.hyperparams_path and os.path.exists(args.hyperparams_path + method + '_BO.json'):
            hyp = json.load(open(args.hyperparams_path + method + '_BO.json'))
        else:
            hyp = DEFAULTS[method]
        hyp = prepare(method, hyp, args.samples)
        t = torch.zeros(args.trials)
        for k in range(-1 if getattr(args, 'discard_first', False) else 0, args.trials):
            model = model_cfg.base(*model_cfg.args, num_classes=num_classes, **model_cfg.kwargs).to(device)
            sampler = getattr(inference, method)(hyperparameters=hyp, model=model, train_loader=loaders['train'],
                                                 device=device, **kw)
            fn = util.silent(sampler.sample)
            if device.type == 'cuda':
                torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            fn()
            if device.type == 'cuda':
                torch.cuda.synchronize(device)
            if k >= 0:                       # k = -1: the discarded trial that pays MIOpen's one-time solver search
                t[k] = time.perf_counter() - t0
        timer[method + '_mean'], timer[method + '_std'] = float(t.mean()), float(t.std()) if args.trials > 1 else 0.0
        print(method, 'time for', args.samples, 'samples:', timer[method + '_mean'], '+-', timer[method + '_std'])
    with open(args.save_path + '.json', 'w') as f:
        json.dump(timer, f)
    return timer


def main(argv=None):
    from .tuning import use_shipped_miopen_db
    use_shipped_miopen_db()                      # tuned MIOpen solver choices for the benchmark networks (tuning.py)
    p = argparse.ArgumentParser()
    p.add_argument('--dataset', type=str, default='CIFAR10')
    p.add_argument('--model', type=str, required=True)
    p.add_argument('--seed', type=int, default=1)
    p.add_argument('--data_path', type=str, default=None, help='ignored: data is synthetic (time_script.py:16)')
    p.add_argument('--num_workers', type=int, default=0, help='ignored (time_script.py:18)')
    p.add_argument('--validation', type=float, default=0.2, help='ignored (time_script.py:28)')
    p.add_argument('--use_val', action='store_true', help='ignored (time_script.py:30)')
    p.add_argument('--hyperparams_path', type=str, default=None)
    p.add_argument('--batch_size', type=int, default=128)
    p.add_argument('--save_path', type=str, required=True)
    p.add_argument('--device_num', type=int, default=0)
    p.add_argument('--methods', nargs='+', default=['SGLD', 'SGHMC', 'cSGLD', 'cSGHMC', 'SWAG'])
    p.add_argument('--samples', type=int, default=3)      # S = 3  (time_script.py:73)
    p.add_argument('--trials', type=int, default=10)      # T = 10 (time_script.py:74)
    p.add_argument('--train_size', type=int, default=None)
    p.add_argument('--test_size', type=int, default=None)
    p.add_argument('--discard_first', action='store_true',
                   help='run one extra, untimed trial per method first (MIOpen searches its solvers once per process)')
    run(p.parse_args(argv))


if __name__ == '__main__':
    main()
