"""Benchmark driver with the contract of URSABench/experiment.py:166-266: S trials x (sample ->
Prediction -> Decision -> OOD), one CSV row `dataset, model, seed, method, task, batch_size,
*sorted hyper-parameters, *sorted result means/stds` appended to `<save_path>results.csv` and the
result dict saved to `<save_path>_tests.npy`. Same flag names as experiment.py:14-36 (data comes
from ursabench_amd.datasets: synthetic, device-resident).

    python -m ursabench_amd.experiment --dataset CIFAR10 --model PreResNet20 --inference_method SGHMC \
        --hyperparams '{"lr":0.1,"prior_std":0.5,"num_samples":3,"alpha":0.5,"burn_in_epochs":0}' \
        --save_path out/ --num_trials 2
"""
import argparse
import csv
import json
import os

import torch

from . import datasets, inference, models, tasks, util

OOD_SETS = {'MNIST': ['FashionMNIST', 'KMNIST'], 'CIFAR10': ['STL10', 'SVHN'], 'CIFAR100': ['STL10', 'SVHN']}


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--dataset', type=str, default='CIFAR10')
    p.add_argument('--data_path', type=str, default=None, help='ignored: data is synthetic')
    p.add_argument('--num_workers', type=int, default=0)
    p.add_argument('--model', type=str, required=True)
    p.add_argument('--seed', type=int, default=1)
    p.add_argument('--inference_method', type=str, default='SGHMC')
    p.add_argument('--hyperparams', type=str, default=None, help='hyper-parameters as inline JSON')
    p.add_argument('--hyperparams_path', type=str, default=None, help='path to a hyper-parameter JSON file')
    p.add_argument('--task', type=str, default='Prediction')
    p.add_argument('--batch_size', type=int, default=128)
    p.add_argument('--save_path', type=str, required=True)
    p.add_argument('--device_num', type=int, default=0)
    p.add_argument('--num_trials', type=int, default=5)
    p.add_argument('--pretrained_model_path', type=str, default=None)
    p.add_argument('--split_classes', type=int, default=None)
    p.add_argument('--validation', type=float, default=0.2, help='proportion of training used as validation')
    p.add_argument('--use_val', dest='use_val', action='store_true', help='use a validation split instead of the test set')
    p.add_argument('--use_dm_imbalance', dest='use_dm_imbalance', action='store_true',
                   help='decision making on a model retrained on class-imbalanced data (experiment.py:218-247)')
    p.add_argument('--train_size', type=int, default=None, help='synthetic train-set size (default: the dataset\'s)')
    p.add_argument('--test_size', type=int, default=None)
    return p


def run(args, device=None, kernels=None):
    """The trial loop of experiment.py:166-266. `kernels` is for tests only (default: the HIP library)."""
    kw = {} if kernels is None else {'kernels': kernels}
    if device is None:
        torch.cuda.set_device(args.device_num)
        device = torch.device('cuda', args.device_num)
    util.set_random_seed(args.seed)
    if args.hyperparams is not None:
        hyperparams = json.loads(args.hyperparams)
    else:
        hyperparams = json.load(open(args.hyperparams_path))
    model_cfg = getattr(models, args.model)
    ds = dict(batch_size=args.batch_size, device=device, train_size=args.train_size, test_size=args.test_size)
    loaders, num_classes = datasets.loaders(args.dataset, use_validation=getattr(args, 'use_val', False),
                                            val_size=getattr(args, 'validation', 0.2),
                                            split_classes=getattr(args, 'split_classes', None), **ds)
    train_loader, test_loader = loaders['train'], loaders['test']
    model = model_cfg.base(*model_cfg.args, num_classes=num_classes, **model_cfg.kwargs).to(device)
    if args.pretrained_model_path is not None:
        model.load_state_dict(torch.load(args.pretrained_model_path))
    inference_method = getattr(inference, args.inference_method)
    task_method = getattr(tasks, args.task)
    use_val, dm_imbalance = getattr(args, 'use_val', False), getattr(args, 'use_dm_imbalance', False)
    os.makedirs(os.path.dirname(args.save_path) or '.', exist_ok=True)
    if args.task == 'Prediction' and use_val:
        # the hyper-optimisation row of experiment.py:84-109: Prediction metrics on the validation split, appended to
        # `results.csv` in the WORKING directory (the reference's literal path). The reference reads an ensemble it
        # never sampled there (NameError); here it is sampled first.
        util.set_random_seed(args.seed)
        sampler = inference_method(hyperparameters=hyperparams, model=model, train_loader=train_loader, device=device, **kw)
        pred = task_method(dataloader={'in_distribution_test': test_loader}, num_classes=num_classes, device=device,
                           metric_list='ALL', **kw)
        pred.update_statistics(models=sampler.sample(), output_performance=False, smoothing=True)
        perf = pred.get_performance_metrics()
        with open('results.csv', 'a+') as f:
            csv.writer(f, dialect='excel').writerow([args.dataset, args.model, args.seed, args.inference_method, args.task,
                                                     args.batch_size, *[hyperparams[k] for k in sorted(hyperparams)],
                                                     *[perf[k] for k in sorted(perf)]])
    if use_val:
        # experiment.py:113: the trial loop, the OOD / Decision tasks and the <save_path>results.csv / _tests.npy writes
        # all sit under `if not args.use_val:` — a hyper-optimisation call costs ONE sampler run and writes ONE row
        return perf if args.task == 'Prediction' else {}
    ood_loaders = []
    if args.dataset not in OOD_SETS:                              # experiment.py:113-160
        raise NotImplementedError
    for name in OOD_SETS[args.dataset]:
        l, _ = datasets.loaders(name, **ds)
        ood_loaders.append({'data': name, 'in_distribution_test': test_loader, 'out_distribution_test': l['test']})

    S = args.num_trials
    results, temp, costs = {}, {}, []
    for s in range(S):
        util.set_random_seed(s)
        sampler = inference_method(hyperparameters=hyperparams, model=model, train_loader=train_loader, device=device,
                                   **kw)
        ensemble = sampler.sample()
        pred = task_method(dataloader={'in_distribution_test': test_loader}, num_classes=num_classes, device=device,
                           metric_list='ALL', **kw)
        pred.update_statistics(models=ensemble, output_performance=False, smoothing=True)
        perf = pred.get_performance_metrics()
        if not dm_imbalance:
            dec = tasks.Decision(dataloader={'decision_data_test': test_loader}, num_classes=num_classes, device=device, **kw)
            dec.update_statistics(models=ensemble, output_performance=False, smoothing=True)
            costs.append(dec.get_performance_metrics()['True_Cost'])
        for ood in ood_loaders:
            o = tasks.OODDetection(data_loader=ood, num_classes=num_classes, device=device, **kw)
            for key, val in o.update_statistics(ensemble, output_performance=True).items():
                temp.setdefault(key + '_' + ood['data'], []).append(val)
        for key in pred.required_metric_list:
            temp.setdefault(key, []).append(perf[key])
    for key, vals in temp.items():
        t = torch.tensor(vals).float()
        results[key + '_mean'], results[key + '_std'] = torch.mean(t), torch.std(t)
    if dm_imbalance:
        # experiment.py:218-247: per seed, retrain on the class-imbalanced training set and decide on its test set
        for s in range(S):
            util.set_random_seed(s)
            l, _ = datasets.loaders(args.dataset, imbalance=True, split_classes=getattr(args, 'split_classes', None), **ds)
            sampler = inference_method(hyperparameters=hyperparams, model=model, train_loader=l['train'], device=device, **kw)
            dec = tasks.Decision(dataloader={'decision_data_test': l['test']}, num_classes=num_classes, device=device, **kw)
            dec.update_statistics(models=sampler.sample(), output_performance=False, smoothing=True)
            costs.append(dec.get_performance_metrics()['True_Cost'])
    results['cost_mean'] = torch.mean(torch.tensor(costs))
    results['cost_std'] = torch.std(torch.tensor(costs))

    row = [args.dataset, args.model, args.seed, args.inference_method, args.task, args.batch_size,
           *[hyperparams[k] for k in sorted(hyperparams)], *[results[k] for k in sorted(results)]]
    with open(args.save_path + 'results.csv', 'a+') as f:
        csv.writer(f, dialect='excel').writerow(row)
    torch.save(results, args.save_path + '_tests.npy')
    return results


def main(argv=None):
    from .tuning import use_shipped_miopen_db
    use_shipped_miopen_db()                      # tuned MIOpen solver choices for the benchmark networks (tuning.py)
    res = run(build_parser().parse_args(argv))
    print(sorted(res.keys()))
    print(res)


if __name__ == '__main__':
    main()
