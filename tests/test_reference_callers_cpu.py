"""CPU, build container only (skipped where /root/reference is absent, e.g. on the GPU box): the
REFERENCE'S OWN caller code drives our drop-in classes unmodified — its random-search hyper-optimiser
(URSABench/hyperopt/hyper_optimization.py:78-170 -> _HypOpt.inference_step :51-73) with our SGHMC
sampler and our Prediction task (oracle kernel set)."""
import os
import sys
import types

import pytest
import torch

REF = '/root/reference'
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='reference sources are only present in the build container')


def _import_reference_hyperopt():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from ref_import import import_reference
    import_reference()

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    anything = type('Anything', (), {'__init__': lambda self, *a, **k: None})
    for mod, names in {'botorch': [], 'botorch.acquisition': ['UpperConfidenceBound'], 'botorch.fit': ['fit_gpytorch_model'],
                       'botorch.models': ['SingleTaskGP'], 'botorch.optim': ['initializers', 'optimize_acqf'],
                       'botorch.utils': ['standardize'], 'gpytorch': [], 'gpytorch.constraints': [],
                       'gpytorch.constraints.constraints': ['GreaterThan'], 'gpytorch.likelihoods': [],
                       'gpytorch.likelihoods.gaussian_likelihood': ['GaussianLikelihood'], 'gpytorch.mlls': ['ExactMarginalLogLikelihood'],
                       'gpytorch.priors': [], 'gpytorch.priors.torch_priors': ['GammaPrior']}.items():
        stub(mod, **{n: anything for n in names})
    import importlib
    return importlib.import_module('URSABench.hyperopt.hyper_optimization')


def test_reference_random_search_drives_our_sampler_and_task():
    import ursabench_amd.inference as inference
    from ursabench_amd import tasks
    from oracle_kernels import OracleKernels
    from test_samplers_cpu import tiny_loader, tiny_net
    hypopt = _import_reference_hyperopt()
    K = OracleKernels()
    sampler = inference.SGHMC({'lr': 0.05, 'prior_std': 1.0, 'num_samples': 2, 'alpha': 0.5, 'burn_in_epochs': 0},
                              tiny_net(), tiny_loader(), kernels=K, use_graph=False)
    objective = tasks.Prediction({'in_distribution_test': tiny_loader(seed=3)}, 4, torch.device('cpu'), ['ll'], kernels=K)
    domain = [{'name': 'lr', 'type': 'continuous', 'option': 'logspace', 'domain': (1e-3, 1e-1)},
              {'name': 'prior_std', 'type': 'constant', 'domain': 1.0},
              {'name': 'num_samples', 'type': 'constant', 'domain': 2},
              {'name': 'alpha', 'type': 'constant', 'domain': 0.5},
              {'name': 'burn_in_epochs', 'type': 'constant', 'domain': 0}]
    search = hypopt.RandomSearch(objective, domain, sampler, N_evaluations=3, seed=7)
    best, best_obj, hyps, objs = search.run(verbose=1, return_all=1)
    assert len(objs) == 3 and all(isinstance(o, float) for o in objs) and len(search.time) == 3
    assert float(best_obj) == pytest.approx(max(objs), rel=1e-6) and set(best) == set(hyps[0])
    assert all(1e-3 <= float(h['lr']) <= 1e-1 for h in hyps)


def test_reference_tasks_accept_our_members_and_agree_with_ours():
    """Members returned by our sampler (bank views) go through the REFERENCE'S Prediction task
    (model.to(device) / .eval() / model.to('cpu') per batch) and give the same accumulators as ours."""
    import numpy as np
    import ursabench_amd.inference as inference
    from ursabench_amd import tasks
    from oracle_kernels import OracleKernels
    from test_samplers_cpu import tiny_loader, tiny_net
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    from ref_import import import_reference
    _, _, _, ref_tasks = import_reference()
    K = OracleKernels()
    s = inference.SGHMC({'lr': 0.05, 'prior_std': 1.0, 'num_samples': 3, 'alpha': 0.5, 'burn_in_epochs': 0},
                        tiny_net(), tiny_loader(), kernels=K, use_graph=False)
    ens = s.sample()
    test = tiny_loader(n=50, b=16, seed=9)
    theirs = ref_tasks.Prediction({'in_distribution_test': test}, 4, torch.device('cpu'), 'ALL')
    theirs.update_statistics(ens, output_performance=False)
    ours = tasks.Prediction({'in_distribution_test': test}, 4, torch.device('cpu'), 'ALL', kernels=K)
    ours.update_statistics(ens, output_performance=False)
    np.testing.assert_allclose(ours.ensemble_proba.numpy(), theirs.ensemble_proba.numpy(), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(ours.expected_data_uncertainty.numpy(), theirs.expected_data_uncertainty.numpy(), rtol=1e-5, atol=1e-7)
    a, b = ours.get_performance_metrics(), theirs.get_performance_metrics()
    assert list(a) == list(b)
    for k in a:
        assert a[k] == pytest.approx(b[k], rel=2e-5, abs=1e-7, nan_ok=True), k
