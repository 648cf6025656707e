"""CPU: MCdropout (URSABench/inference/vi_dropout.py) against the reference's own run (golden G10,
tools/gen_golden.py gen_mcdropout): the model swap, the per-minibatch OneCycleLR / cosine (lr, momentum) pairs
walked by the device schedule table, FlatSGD bit-exact to torch.optim.SGD, always-on dropout masks."""
import json
import os

import numpy as np
import pytest
import torch

import ursabench_amd.inference as inference
from ursabench_amd import models, tasks
from ursabench_amd._native import STEP_SGD
from oracle_kernels import OracleKernels
from test_samplers_cpu import tiny_loader


def flat(m):
    return torch.cat([p.detach().reshape(-1) for p in m.parameters()]).numpy()


def test_mcdropout_equals_the_reference_run(golden_dir):
    g = np.load(os.path.join(golden_dir, 'mcdropout.npz'))
    hyp, hyp2 = json.loads(str(g['hyper'])), json.loads(str(g['hyper2']))
    K = OracleKernels()
    torch.manual_seed(21)
    s = inference.MCdropout(dict(hyp), models.MLP(16, 12, 4), tiny_loader(), kernels=K, use_graph=False)
    assert type(s.model).__name__ == 'MLP_dropout' and s.model.dropout == 0.2
    assert s.weight_decay == 0.01 ** 2 * (1 - 0.2) / (2. * 64)          # vi_dropout.py:53
    assert np.array_equal(flat(s.model), g['theta0'])                # fresh init from the same generator state
    for k in range(2):
        m = s.sample_iterative()
        assert m is s.model                                          # the live model, every time (vi_dropout.py:119)
        assert np.array_equal(flat(m), g['samples'][k]), k           # bit-exact: same masks, same SGD arithmetic
    used = np.array([(lr, mu) for lr, mu, fl, _ in K.step_log], np.float64)
    assert all(fl & STEP_SGD for _, _, fl, _ in K.step_log)
    np.testing.assert_allclose(used, g['lr_mom'], rtol=1e-7)         # float32 control block vs float64 scheduler
    xt = torch.tensor(g['x_test'])
    s.model.eval()
    with torch.no_grad():
        mc = np.stack([s.model(xt).numpy() for _ in range(3)])
    assert np.array_equal(mc, g['mc_logits']) and not np.array_equal(mc[0], mc[1])
    # update_hyp: re-init in place, cosine schedule stepped per minibatch, control block re-used
    ctl = s.optimizer._ctl.data_ptr()
    s.update_hyp(dict(hyp2))
    assert s.optimizer._ctl.data_ptr() == ctl and s.weight_decay == float(g['weight_decay'])
    assert np.array_equal(flat(s.model), g['theta1'])
    n0 = len(K.step_log)
    assert np.array_equal(flat(s.sample_iterative()), g['sample2'])
    np.testing.assert_allclose(np.array([(lr, mu) for lr, mu, _, _ in K.step_log[n0:]]), g['lr_mom2'], rtol=1e-7)


def test_mcdropout_members_through_the_tasks():
    """T copies of ONE live model: every forward draws new masks, the tasks evaluate it eagerly."""
    torch.manual_seed(3)
    hyp = {'lr': 0.05, 'epochs': 0, 'dropout': 0.2, 'lengthscale': 0.01, 'num_samples': 4, 'momentum': 0.9, 'weight_decay': 0}
    s = inference.MCdropout(dict(hyp), models.MLP(16, 12, 4), tiny_loader(), kernels=OracleKernels(), use_graph=False)
    ens = s.sample()
    assert len(ens) == 4 and all(m is s.model for m in ens)
    pred = tasks.Prediction({'in_distribution_test': tiny_loader(seed=5)}, 4, torch.device('cpu'), 'ALL', kernels=OracleKernels())
    pred.update_statistics(ens, output_performance=False)
    assert pred.num_samples_collected == 4 and pred._acc.stats['eager_forwards'] == 4 * 1      # 64 rows: one evaluation batch
    np.testing.assert_allclose(pred.ensemble_proba.sum(1).numpy(), np.full(64, 4.0, np.float32), rtol=1e-5)
    single = tasks.Prediction({'in_distribution_test': tiny_loader(seed=5)}, 4, torch.device('cpu'), 'ALL', kernels=OracleKernels())
    single.update_statistics(ens[:1], output_performance=False)
    assert not np.allclose(pred.ensemble_proba.numpy() / 4, single.ensemble_proba.numpy())   # the masks differ per forward


def test_preresnet_dropout_variant_and_dispatch():
    m = inference.vi_dropout.change_to_dropout_model(models.PreResNet(10, 8), 0.5)
    assert type(m).__name__ == 'PreResNet_dropout' and m.dropout == 0.2 and m.depth == 8      # quirk: argument ignored
    with pytest.raises(AttributeError):
        inference.vi_dropout.change_to_dropout_model(models.LeNet5(10), 0.2)                  # no LeNet5_dropout, as in the reference
    assert getattr(inference, 'MCdropout') is inference.vi_dropout.MCdropout


def test_host_drawn_dropout_masks_are_atens(golden_dir, monkeypatch):
    """The GPU replay of G10 (tests/test_samplers_gpu.py) feeds the device run the reference's dropout masks by drawing
    them on the host the way ATen's CPU dropout does. Here, on CPU, that replacement must leave the bit-exact replay of
    the reference's run intact."""
    import torch.nn.functional as F

    def cpu_stream_dropout(x, p=0.5, training=True, inplace=False):
        if not training or p == 0:
            return x
        return x * torch.empty(x.shape, dtype=x.dtype).bernoulli_(1 - p).div_(1 - p).to(x.device)
    monkeypatch.setattr(F, 'dropout', cpu_stream_dropout)
    g = np.load(os.path.join(golden_dir, 'mcdropout.npz'))
    torch.manual_seed(21)
    s = inference.MCdropout(dict(json.loads(str(g['hyper']))), models.MLP(16, 12, 4), tiny_loader(), kernels=OracleKernels(),
                            use_graph=False)
    for k in range(2):
        assert np.array_equal(flat(s.sample_iterative()), g['samples'][k]), k
