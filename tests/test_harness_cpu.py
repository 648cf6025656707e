"""CPU: the experiment / time_script drivers (row format of URSABench/experiment.py:249-266, JSON of
time_script.py:117-125) through the oracle kernel set on tiny synthetic sets."""
import csv
import json
import os

import pytest
import torch

from ursabench_amd import experiment, time_script
from oracle_kernels import OracleKernels


def test_experiment_row_and_npy(tmp_path, golden_dir):
    hyp = {'lr': 0.05, 'prior_std': 1.0, 'num_samples': 2, 'alpha': 1.0, 'burn_in_epochs': 0}
    args = experiment.build_parser().parse_args([
        '--dataset', 'MNIST', '--model', 'MLP200MNIST', '--inference_method', 'SGLD', '--hyperparams', json.dumps(hyp),
        '--save_path', str(tmp_path) + '/', '--num_trials', '2', '--batch_size', '32', '--train_size', '64',
        '--test_size', '48'])
    res = experiment.run(args, device=torch.device('cpu'), kernels=OracleKernels())
    keys = sorted(res)
    assert 'cost_mean' in keys and 'nll_mean' in keys and 'error_rate_std' in keys
    assert 'total_uncertainty_auroc_FashionMNIST_mean' in keys and 'model_uncertainty_auroc_KMNIST_std' in keys
    assert len(keys) == 2 * (11 + 2 * 2) + 2
    gold = json.load(open(os.path.join(golden_dir, 'experiment_columns.json')))       # G11: derived from the reference's task objects
    assert keys == gold['datasets']['MNIST']
    row = next(csv.reader(open(str(tmp_path) + '/results.csv')))
    assert row[:6] == ['MNIST', 'MLP200MNIST', '1', 'SGLD', 'Prediction', '32']
    assert len(row) == 6 + len(hyp) + len(keys)
    assert [float(v) for v in row[6:6 + len(hyp)]] == [hyp[k] for k in sorted(hyp)]
    saved = torch.load(str(tmp_path) + '/_tests.npy')
    assert sorted(saved) == keys


def test_time_script_json(tmp_path):
    p = str(tmp_path / 'timing')
    class A:  # noqa: E701
        dataset, model, seed, hyperparams_path, batch_size, save_path = 'MNIST', 'MLP200MNIST', 1, None, 32, p
        device_num, methods, samples, trials, train_size, test_size = 0, ['SGLD', 'cSGHMC', 'SWAG', 'MCdropout', 'SGD'], 2, 2, 64, 32
    out = time_script.run(A, device=torch.device('cpu'), kernels=OracleKernels())
    assert sorted(out) == sorted(m + sfx for m in A.methods for sfx in ('_mean', '_std'))
    assert json.load(open(p + '.json')) == out and all(v >= 0 for v in out.values())
    h = time_script.prepare('cSGHMC', time_script.DEFAULTS['cSGHMC'], 3)
    assert h['burn_in_epochs'] == 0 and h['num_cycles'] == 1 and h['num_samples_per_cycle'] == 3
    assert time_script.prepare('MCdropout', dict(time_script.DEFAULTS['MCdropout'], epochs=7), 3)['epochs'] == 0   # time_script.py:96-97


def test_experiment_accepts_the_reference_flags(tmp_path, monkeypatch):
    """--use_val / --validation / --split_classes / --use_dm_imbalance of URSABench/experiment.py:27-31. With --use_val
    the reference writes ONLY the hyper-optimisation row to ./results.csv: its trial loop, OOD / Decision tasks and the
    <save_path> outputs all sit under `if not args.use_val:` (experiment.py:113-266) — one sampler run, one
    row. The imbalance path retrains per seed on the thinned training set (classes 3 and 7 of MNIST cut by 99 %).
    --split_classes keeps one CIFAR-10 class half, relabelled 0..4 (datasets.py:224-242), CIFAR-10 only."""
    from ursabench_amd import datasets
    monkeypatch.chdir(tmp_path)
    hyp = {'lr': 0.05, 'prior_std': 1.0, 'num_samples': 2, 'alpha': 1.0, 'burn_in_epochs': 0}
    base = ['--dataset', 'MNIST', '--model', 'MLP200MNIST', '--inference_method', 'SGLD', '--hyperparams', json.dumps(hyp),
            '--save_path', str(tmp_path) + '/out_', '--num_trials', '2', '--batch_size', '32', '--train_size', '200',
            '--test_size', '48', '--data_path', 'ignored', '--num_workers', '4']
    K = OracleKernels()
    args = experiment.build_parser().parse_args(base + ['--use_val', '--validation', '0.25'])
    res = experiment.run(args, device=torch.device('cpu'), kernels=K)
    assert len(res) == 11                                # the Prediction metrics of the one validation run
    assert 'nll' in res and 'cost_mean' not in res
    hrow = next(csv.reader(open(tmp_path / 'results.csv')))
    assert hrow[:6] == ['MNIST', 'MLP200MNIST', '1', 'SGLD', 'Prediction', '32'] and len(hrow) == 6 + len(hyp) + 11
    assert len(list(csv.reader(open(tmp_path / 'results.csv')))) == 1
    assert not os.path.exists(str(tmp_path) + '/out_results.csv') and not os.path.exists(str(tmp_path) + '/out__tests.npy')
    assert len(K.step_log) == 2 * 5                      # ONE sampler run: 2 samples x ceil(150 / 32) minibatch steps
    with pytest.raises(AssertionError):                  # datasets.py:225: CIFAR-10 only
        experiment.run(experiment.build_parser().parse_args(base + ['--split_classes', '0']), device=torch.device('cpu'), kernels=K)
    l, c = datasets.loaders('CIFAR10', batch_size=32, device='cpu', train_size=500, test_size=100, split_classes=1)
    full, _ = datasets.loaders('CIFAR10', batch_size=32, device='cpu', train_size=500, test_size=100)
    yf, ys = full['train'].dataset.y, l['train'].dataset.y
    keep = torch.isin(yf, torch.tensor([3, 4, 5, 6, 7]))
    assert c == 10 and torch.equal(ys, yf[keep] - 3) and torch.equal(l['train'].dataset.x, full['train'].dataset.x[keep])
    assert int(l['test'].dataset.y.max()) <= 4
    args = experiment.build_parser().parse_args(base + ['--use_dm_imbalance'])
    res = experiment.run(args, device=torch.device('cpu'), kernels=OracleKernels())
    assert 'cost_mean' in res and torch.isfinite(res['cost_mean'])
    l, _ = datasets.loaders('MNIST', batch_size=32, device='cpu', train_size=2000, test_size=10, imbalance=True)
    y = l['train'].dataset.y
    full, _ = datasets.loaders('MNIST', batch_size=32, device='cpu', train_size=2000, test_size=10)
    yf = full['train'].dataset.y
    for c in range(10):
        n, nf = int((y == c).sum()), int((yf == c).sum())
        assert n == (int(nf - 0.99 * nf) if c in (3, 7) else nf)


def test_shipped_miopen_db_is_copied_privately(monkeypatch, tmp_path):
    """ursabench_amd/tuning.py: every process gets a PRIVATE writable copy of the shipped tuned MIOpen databases (or an
    empty private one), and a path the caller set is respected."""
    from ursabench_amd import tuning
    shipped = tmp_path / 'shipped'
    shipped.mkdir()
    (shipped / 'gfx950.udb.txt').write_text('x=y\n')
    (shipped / 'README.md').write_text('not a database')
    monkeypatch.setattr(tuning, 'SHIPPED', str(shipped))
    monkeypatch.delenv('MIOPEN_USER_DB_PATH', raising=False)
    monkeypatch.delenv('URSA_NO_SHIPPED_MIOPEN_DB', raising=False)
    d = tuning.use_shipped_miopen_db()
    # (besides the databases: the owner marker by which a later process recognises a dead owner's directory)
    assert os.environ['MIOPEN_USER_DB_PATH'] == d and sorted(os.listdir(d)) == ['gfx950.udb.txt', tuning._OWNER] and d != str(shipped)
    assert tuning.use_shipped_miopen_db() == d                         # already set: respected
    monkeypatch.delenv('MIOPEN_USER_DB_PATH')
    monkeypatch.setenv('URSA_NO_SHIPPED_MIOPEN_DB', '1')
    assert os.listdir(tuning.use_shipped_miopen_db()) == [tuning._OWNER]
