import os, sys, json, tempfile
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp())
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch
import ursabench_amd.inference as inference
from ursabench_amd import tasks
from test_samplers_cpu import _load_preresnet8, _preresnet8_inputs
DEV = torch.device('cuda', 0)
g = np.load('tests/golden/e2e_preresnet8.npz')
hyp = json.loads(str(g['hyper']))
for rep in range(4):
    train, test = _preresnet8_inputs(g)
    s = inference.SGHMC(dict(hyp), _load_preresnet8(g), train, device=DEV)
    def eps(k):
        e = torch.zeros(s.arena.n, device=DEV)
        e[s.arena.layout.gather_index(DEV)] = torch.tensor(g['eps'][k], device=DEV)
        return e
    s.eps_provider = eps
    ens = s.sample()
    pred = tasks.Prediction({'in_distribution_test': test}, 10, DEV, 'ALL')
    pred.update_statistics(ens, output_performance=False)
    p, ref = pred.ensemble_proba.numpy(), g['proba_sum']
    print('rep', rep, 'max rel err proba', float(np.max(np.abs(p - ref) / np.abs(ref))), 'ent', float(np.max(np.abs(pred.expected_data_uncertainty.numpy() - g['ent_sum']) / np.abs(g['ent_sum']))))
