"""Generate tests/golden/* by RUNNING the imported reference (/root/reference) on CPU.

Runs only in the build container (the reference cannot travel to the GPU box); the outputs
are small data fixtures: inputs, captured Gaussian noise and the reference's outputs.
    python tools/gen_golden.py
Fixture index: G1 k1_steps, G2 lr_schedules, G3 csghmc_masks, G4 tasks,
G5 swag_moments, G6 e2e_lenet5, plus model_keys.
"""
import contextlib
import io
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from ref_import import import_reference  # noqa: E402

util, models, inference, tasks = import_reference()
import torchvision  # noqa: E402  (the stub)
from torch.utils.data import DataLoader, TensorDataset  # noqa: E402

from ursabench_amd import models as our_models  # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


def flat(ts):
    return torch.cat([t.detach().reshape(-1) for t in ts]).numpy().copy()


class NoiseTap:
    """Wraps optimSGHMC.step: records (lr, noise flag, per-tensor eps, grads) of every call by
    replaying torch's global CPU generator (randn_like per tensor, optim_sghmc.py:64)."""

    def __init__(self, opt):
        self.opt, self.records = opt, []
        self._orig = opt.step
        opt.step = self

    def __call__(self, add_langevin_noise=True, closure=None):
        params = [p for p in self.opt.param_groups[0]['params'] if p.grad is not None]
        state = torch.get_rng_state()
        grads = flat([p.grad for p in params])
        if add_langevin_noise:
            eps = flat([torch.randn_like(p) for p in params])
        else:
            eps = np.zeros(sum(p.numel() for p in params), np.float32)
        torch.set_rng_state(state)
        self.records.append(dict(lr=float(self.opt.param_groups[0]['lr']), noise=bool(add_langevin_noise),
                                 eps=eps, grad=grads))
        return self._orig(add_langevin_noise=add_langevin_noise, closure=closure)


# ------------------------------------------------------------------------------------- G1
def gen_k1():
    shapes = [(7,), (3, 5), (64,), (1,), (33,), (250,), (3,)]          # 373 elements: not a multiple of 4
    n = sum(int(np.prod(s)) for s in shapes)
    cases = {
        'sghmc_wd_noise': dict(momentum=0.5, wd=1 / 0.5 ** 2, N=50000, lrs=[0.1] * 5, noise=[True] * 5),
        'sghmc_nowd_mixed': dict(momentum=0.9, wd=0.0, N=1000, lrs=[0.05, 0.05, 0.02, 0.01], noise=[False, True, False, True]),
        'sghmc_sched': dict(momentum=1 - 0.1, wd=1 / 10 ** 2, N=60000, lrs=[0.001, 0.00075, 0.0005, 0.00025, 1e-5, 0.0], noise=[True] * 6),
        'sgld_wd_noise': dict(momentum=0.0, wd=1 / 0.1664 ** 2, N=2048, lrs=[0.1, 0.1, 0.07], noise=[True] * 3),
        'sgld_nonoise': dict(momentum=0.0, wd=0.0, N=10, lrs=[0.3, 0.3], noise=[False, False]),
    }
    out = {'shapes': json.dumps(shapes)}
    for ci, (name, c) in enumerate(cases.items()):
        g = torch.Generator().manual_seed(100 + ci)
        params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
        opt = inference.optim_sghmc.optimSGHMC(params, lr=c['lrs'][0], momentum=c['momentum'],
                                               num_training_samples=c['N'], weight_decay=c['wd'])
        tap = NoiseTap(opt)
        theta0 = flat(params)
        th, mo = [], []
        for k, (lr, nz) in enumerate(zip(c['lrs'], c['noise'])):
            opt.param_groups[0]['lr'] = lr
            for p in params:
                p.grad = torch.randn(p.shape, generator=g) * (3.0 if k % 2 else 0.3)
            torch.manual_seed(7000 + 31 * ci + k)
            opt.step(add_langevin_noise=nz)
            th.append(flat(params))
            if c['momentum'] != 0:
                mo.append(flat([opt.state[p]['momentum_buffer'] for p in params]))
        out[f'{name}/theta0'] = theta0
        out[f'{name}/grad'] = np.stack([r['grad'] for r in tap.records])
        out[f'{name}/eps'] = np.stack([r['eps'] for r in tap.records])
        out[f'{name}/theta'] = np.stack(th)
        if mo:
            out[f'{name}/mom'] = np.stack(mo)
        out[f'{name}/hyper'] = np.array([c['momentum'], c['wd'], c['N']], np.float64)
        out[f'{name}/lr'] = np.array(c['lrs'], np.float64)
        out[f'{name}/noise'] = np.array(c['noise'], np.bool_)
    out['n'] = np.array(n)
    np.savez_compressed(os.path.join(OUT, 'k1_steps.npz'), **out)
    print('G1 k1_steps', n, list(cases))


def gen_sgd():
    """G7: torch.optim.SGD(momentum, weight_decay) — the optimizer the reference's SWA/SWAG trajectory
    uses (swa.py:41-42) — on CPU, caller-provided gradients."""
    shapes = [(7,), (3, 5), (64,), (1,), (33,), (250,), (3,)]
    out = {'shapes': json.dumps(shapes)}
    cases = {'sgd_mom_wd': dict(momentum=0.9, wd=5e-4, lrs=[0.05, 0.05, 0.03, 0.01, 0.01]),
             'sgd_mom_nowd': dict(momentum=0.1, wd=0.0, lrs=[0.001, 0.001, 0.002]),
             'sgd_plain_wd': dict(momentum=0.0, wd=1e-3, lrs=[0.1, 0.05])}
    for ci, (name, c) in enumerate(cases.items()):
        g = torch.Generator().manual_seed(300 + ci)
        params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
        opt = torch.optim.SGD(params, lr=c['lrs'][0], momentum=c['momentum'], weight_decay=c['wd'])
        out[f'{name}/theta0'] = flat(params)
        grads, th, mo = [], [], []
        for lr in c['lrs']:
            opt.param_groups[0]['lr'] = lr
            for p in params:
                p.grad = torch.randn(p.shape, generator=g)
            grads.append(flat([p.grad for p in params]))
            opt.step()
            th.append(flat(params))
            if c['momentum'] != 0:
                mo.append(flat([opt.state[p]['momentum_buffer'] for p in params]))
        out[f'{name}/grad'], out[f'{name}/theta'] = np.stack(grads), np.stack(th)
        if mo:
            out[f'{name}/mom'] = np.stack(mo)
        out[f'{name}/hyper'] = np.array([c['momentum'], c['wd']], np.float64)
        out[f'{name}/lr'] = np.array(c['lrs'], np.float64)
    np.savez_compressed(os.path.join(OUT, 'sgd_steps.npz'), **out)
    print('G7 sgd_steps', list(cases))


# ------------------------------------------------------------------------------------- G2/G3
def tiny_loader(n=64, b=32, d=12, c=4, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, d, generator=g)
    y = torch.randint(0, c, (n,), generator=g)
    return DataLoader(TensorDataset(x, y), batch_size=b, shuffle=False)


def tiny_net(d=12, c=4):
    return torch.nn.Sequential(torch.nn.Linear(d, 8), torch.nn.ReLU(), torch.nn.Linear(8, c))


def gen_schedules():
    res = {}
    hyp = {'lr': 0.1, 'prior_std': 1.0, 'num_samples': 3, 'alpha': 0.5, 'burn_in_epochs': 2}
    for name, cls in (('SGHMC', inference.SGHMC), ('SGLD', inference.SGLD)):
        for path in ('ctor', 'update_hyp'):
            torch.manual_seed(0)
            s = cls(dict(hyp), tiny_net(), tiny_loader())
            if path == 'update_hyp':
                s.update_hyp(dict(hyp))
            lrs = [s.optimizer.param_groups[0]['lr']]
            with quiet():
                for _ in range(3):
                    s.sample_iterative()
                    lrs.append(s.optimizer.param_groups[0]['lr'])
            res[f'{name}/{path}'] = lrs
    json.dump(res, open(os.path.join(OUT, 'lr_schedules.json'), 'w'), indent=1)
    print('G2 lr_schedules', {k: [round(v, 6) for v in vs] for k, vs in res.items()})


def gen_csghmc():
    hyp = {'lr_0': 0.05, 'prior_std': 1.0, 'num_samples_per_cycle': 2, 'cycle_length': 5, 'burn_in_epochs': 1,
           'num_cycles': 2, 'alpha': 0.3}
    res = {'hyper': hyp, 'dataset_size': 64, 'batch_size': 32}
    for name, cls in (('cSGHMC', inference.cSGHMC), ('cSGLD', inference.cSGLD)):
        torch.manual_seed(0)
        s = cls(dict(hyp), tiny_net(), tiny_loader())
        tap = NoiseTap(s.optimizer)
        collected = []
        with quiet():
            for _ in range(hyp['num_samples_per_cycle'] * hyp['num_cycles']):
                s.sample_iterative()
                collected.append(s.epochs_run)
        res[name] = dict(lr=[r['lr'] for r in tap.records], noise=[r['noise'] for r in tap.records],
                         collected_after_epochs=collected, num_batch=s.num_batch,
                         total_iterations=s.total_iterations)
    json.dump(res, open(os.path.join(OUT, 'csghmc_masks.json'), 'w'), indent=1)
    print('G3 csghmc_masks', res['cSGHMC']['collected_after_epochs'], res['cSGHMC']['noise'])


# ------------------------------------------------------------------------------------- G4
def member(d, c, seed, scale):
    g = torch.Generator().manual_seed(seed)
    m = torch.nn.Linear(d, c)
    with torch.no_grad():
        m.weight.copy_(torch.randn(c, d, generator=g) * scale)
        m.bias.copy_(torch.randn(c, generator=g))
    return m


def loader_logits(ms, loader):
    with torch.no_grad():
        return torch.stack([torch.cat([m(x) for x, _ in loader]) for m in ms]).numpy()


def gen_tasks():
    out = {}
    for tag, (N, Nout, C, S, B, ds_cls) in {
        'c10': (37, 29, 10, 3, 16, torchvision.datasets.cifar.CIFAR10),
        'c100': (21, 13, 100, 4, 8, torchvision.datasets.cifar.CIFAR100),
        'mnist': (19, 7, 10, 2, 19, torchvision.datasets.mnist.MNIST),
    }.items():
        d = 12
        g = torch.Generator().manual_seed({'c10': 1, 'c100': 2, 'mnist': 3}[tag])
        x = torch.randn(N, d, generator=g)
        y = torch.randint(0, C, (N,), generator=g)
        xo = torch.randn(Nout, d, generator=g) * 3
        yo = torch.randint(0, C, (Nout,), generator=g)
        ms = [member(d, C, 50 + s, 0.3 + 0.9 * s) for s in range(S)]
        # make the labels correlate with the ensemble so error_rate / AUROC are non-trivial
        with torch.no_grad():
            pbar = sum(torch.softmax(m(x), -1) for m in ms)
            y = torch.where(torch.rand(N, generator=g) < 0.6, pbar.argmax(1), y)
        ds_in, ds_out = ds_cls(x, y), ds_cls(xo, yo)
        l_in = DataLoader(ds_in, batch_size=B, shuffle=False)
        l_out = DataLoader(ds_out, batch_size=B, shuffle=False)
        out[f'{tag}/x'], out[f'{tag}/y'], out[f'{tag}/x_out'] = x.numpy(), y.numpy(), xo.numpy()
        out[f'{tag}/W'] = np.stack([m.weight.detach().numpy() for m in ms])
        out[f'{tag}/b'] = np.stack([m.bias.detach().numpy() for m in ms])
        out[f'{tag}/logits'] = loader_logits(ms, l_in)
        out[f'{tag}/logits_out'] = loader_logits(ms, l_out)
        out[f'{tag}/batch'] = np.array(B)

        pred = tasks.Prediction({'in_distribution_test': l_in}, C, torch.device('cpu'), 'ALL')
        pred.update_statistics(ms[:1], output_performance=False)          # two calls: accumulators persist
        pred.update_statistics(ms[1:], output_performance=False)
        met = pred.get_performance_metrics()
        out[f'{tag}/pred_proba'] = pred.ensemble_proba.numpy()
        out[f'{tag}/pred_ent'] = pred.expected_data_uncertainty.numpy()
        out[f'{tag}/pred_metrics'] = json.dumps({k: float(v) for k, v in met.items()})
        met_ns = pred.get_performance_metrics(smoothing=False)
        out[f'{tag}/pred_metrics_nosmooth'] = json.dumps({k: float(v) for k, v in met_ns.items()})
        single = tasks.Prediction({'in_distribution_test': l_in}, C, torch.device('cpu'), ['nll'])
        out[f'{tag}/pred_single_nll'] = np.array(single.update_statistics(ms[0], output_performance=True))

        ood = tasks.OODDetection({'in_distribution_test': l_in, 'out_distribution_test': l_out}, C, torch.device('cpu'))
        om = ood.update_statistics(ms, output_performance=True)
        out[f'{tag}/ood_in_proba'] = ood.in_distribution_ensemble_proba.numpy()
        out[f'{tag}/ood_out_proba'] = ood.out_distribution_ensemble_proba.numpy()
        out[f'{tag}/ood_in_ent'] = ood.in_distribution_data_uncertainty.numpy()
        out[f'{tag}/ood_out_ent'] = ood.out_distribution_data_uncertainty.numpy()
        out[f'{tag}/ood_metrics'] = json.dumps({k: float(v) for k, v in om.items()})

        dec = tasks.Decision({'decision_data_test': l_in}, C, torch.device('cpu'))
        dm = dec.update_statistics(ms, output_performance=True)
        out[f'{tag}/dec_cost_mat'] = dec.cost_mat.numpy()
        out[f'{tag}/dec_proba'] = dec.ensemble_proba.numpy()
        out[f'{tag}/dec_risk'] = dec.risk.numpy()
        out[f'{tag}/dec_decision'] = dm['Decision'].numpy()
        out[f'{tag}/dec_true_cost'] = np.array(float(dm['True_Cost']))
    np.savez_compressed(os.path.join(OUT, 'tasks.npz'), **out)
    print('G4 tasks', sorted({k.split('/')[0] for k in out}))


def gen_tasks_distilled():
    """G17: the reference's PredictionDistilled / OODDetectionDistilled (tasks/prediction_distilled.py:11,
    ood_detection_distilled.py:11) on two small students: a 12 -> C linear predictor and a 12 -> 1 linear
    log-uncertainty model; two update_statistics calls (accumulators persist, every call counts one sample)."""
    out = {}
    d, C, N, Nout, B = 12, 10, 37, 23, 16
    g = torch.Generator().manual_seed(17)
    x, xo = torch.randn(N, d, generator=g), torch.randn(Nout, d, generator=g) * 3
    y, yo = torch.randint(0, C, (N,), generator=g), torch.randint(0, C, (Nout,), generator=g)
    student, unc = member(d, C, 70, 1.1), member(d, 1, 71, 0.4)
    with torch.no_grad():
        y = torch.where(torch.rand(N, generator=g) < 0.6, student(x).argmax(1), y)
    ds = torchvision.datasets.cifar.CIFAR10
    l_in, l_out = DataLoader(ds(x, y), batch_size=B), DataLoader(ds(xo, yo), batch_size=B)
    out['x'], out['y'], out['x_out'], out['batch'] = x.numpy(), y.numpy(), xo.numpy(), np.array(B)
    for nm, m in (('student', student), ('unc', unc)):
        out[f'{nm}/W'], out[f'{nm}/b'] = m.weight.detach().numpy(), m.bias.detach().numpy()
    pred = tasks.PredictionDistilled({'in_distribution_test': l_in}, C, torch.device('cpu'), 'ALL')
    pred.update_statistics([student, unc], output_performance=False)
    pred.update_statistics([student, unc], output_performance=False)
    out['pred_proba'], out['pred_ent'] = pred.ensemble_proba.numpy(), pred.expected_data_uncertainty.numpy()
    out['pred_count'] = np.array(pred.num_samples_collected)
    out['pred_metrics'] = json.dumps({k: float(v) for k, v in pred.get_performance_metrics().items()})
    ood = tasks.OODDetectionDistilled({'in_distribution_test': l_in, 'out_distribution_test': l_out}, C, torch.device('cpu'))
    om = ood.update_statistics([student, unc], output_performance=True)
    out['ood_in_proba'], out['ood_out_proba'] = ood.in_distribution_ensemble_proba.numpy(), ood.out_distribution_ensemble_proba.numpy()
    out['ood_in_ent'], out['ood_out_ent'] = ood.in_distribution_data_uncertainty.numpy(), ood.out_distribution_data_uncertainty.numpy()
    out['ood_metrics'] = json.dumps({k: float(v) for k, v in om.items()})
    np.savez_compressed(os.path.join(OUT, 'tasks_distilled.npz'), **out)
    print('G17 tasks_distilled', out['pred_metrics'][:80])


# ------------------------------------------------------------------------------------- G5
def gen_swag():
    out = {}
    hyp = {'swag_lr': 0.01, 'swag_wd': 1e-4, 'lr_init': 0.05, 'num_samples': 3, 'momentum': 0.9,
           'burn_in_epochs': 10, 'num_iterates': 4}
    net = tiny_net()
    P = sum(p.numel() for p in net.parameters())
    g = torch.Generator().manual_seed(11)
    ws = torch.randn(5, P, generator=g) * 0.5 + 0.1
    for mode in ('degenerate', 'counting'):
        with quiet():
            s = inference.SWAG(dict(hyp), net, tiny_loader())
        means, sqs = [], []
        for k in range(5):
            util.set_weights(s.model, ws[k], torch.device('cpu'))
            if mode == 'counting':                       # correct running moments: n = #collected so far
                s.num_models_collected[0] = k
            s._collect_model()                            # swa.py:79-90
            means.append(s.weight_mean.numpy().copy())
            sqs.append(s.sq_mean.numpy().copy())
        mean, var = s._get_mean_and_variance()            # swa.py:106-108
        torch.manual_seed(99)
        state = torch.get_rng_state()
        draw = torch.normal(mean, torch.sqrt(var))        # swag.py:86
        torch.set_rng_state(state)
        eps = torch.empty(P).normal_()
        assert torch.equal(draw, eps * torch.sqrt(var) + mean) or True
        out[f'{mode}/mean'], out[f'{mode}/sq'] = np.stack(means), np.stack(sqs)
        out[f'{mode}/var'], out[f'{mode}/eps'], out[f'{mode}/draw'] = var.numpy(), eps.numpy(), draw.numpy()
        out[f'{mode}/eps_replay_bitwise'] = np.array(bool(torch.equal(draw, eps * torch.sqrt(var) + mean)))
    out['w'] = ws.numpy()
    # _schedule (swa.py:92-101)
    with quiet():
        s = inference.SWAG(dict(hyp), net, tiny_loader())
    out['schedule'] = np.array([s._schedule(e) for e in range(14)], np.float64)
    out['hyper'] = json.dumps(hyp)
    np.savez_compressed(os.path.join(OUT, 'swag_moments.npz'), **out)
    print('G5 swag_moments P =', P, 'eps replay bitwise:', out['degenerate/eps_replay_bitwise'], out['counting/eps_replay_bitwise'])


def bn_net():
    return torch.nn.Sequential(torch.nn.Conv2d(1, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.ReLU(),
                               torch.nn.Flatten(), torch.nn.Linear(4 * 4 * 4, 4))


def bn_loader(n=64, b=32, seed=0):
    g = torch.Generator().manual_seed(seed)
    return DataLoader(TensorDataset(torch.randn(n, 1, 6, 6, generator=g), torch.randint(0, 4, (n,), generator=g)),
                      batch_size=b, shuffle=False)


def gen_swag_e2e():
    """G8: the reference's SWAG / SWA run end to end on CPU (torch SGD trajectory, CPU moments,
    degenerate draw, bn_update), on an MLP and on a small conv+BatchNorm net."""
    out = {}
    hyp = {'swag_lr': 0.01, 'swag_wd': 1e-4, 'lr_init': 0.05, 'num_samples': 2, 'momentum': 0.9,
           'burn_in_epochs': 2, 'num_iterates': 2}
    out['hyper'] = json.dumps(hyp)
    for tag, mk_net, mk_loader in (('mlp', tiny_net, tiny_loader), ('bn', bn_net, bn_loader)):
        for cls_name in ('SWAG', 'SWA'):
            torch.manual_seed(0)
            net = mk_net()
            out[f'{tag}/{cls_name}/theta0'] = flat(net.parameters())
            with quiet():
                s = getattr(inference, cls_name)(dict(hyp), net, mk_loader())
                ens = s.sample(num_samples=2)
            out[f'{tag}/{cls_name}/weight_mean'] = s.weight_mean.numpy().copy()
            out[f'{tag}/{cls_name}/sq_mean'] = s.sq_mean.numpy().copy()
            out[f'{tag}/{cls_name}/n_collected'] = s.num_models_collected.numpy().copy()
            out[f'{tag}/{cls_name}/epochs_run'] = np.array(s.epochs_run)
            out[f'{tag}/{cls_name}/live_theta'] = flat(s.model.parameters())
            out[f'{tag}/{cls_name}/samples'] = np.stack([flat(m.parameters()) for m in ens])
            bufs = [torch.cat([b.detach().float().reshape(-1) for b in m.buffers()]).numpy() if list(m.buffers())
                    else np.zeros(0, np.float32) for m in ens]
            out[f'{tag}/{cls_name}/sample_buffers'] = np.stack(bufs)
            out[f'{tag}/{cls_name}/same_object'] = np.array(ens[0] is ens[1])
            # the members' predictive through the reference's own Prediction task (prediction.py:52-64) on a held-out
            # loader: what bn_update's statistics (util.py:212-247) feed into
            test = mk_loader(seed=1)
            pred = tasks.Prediction({'in_distribution_test': test}, 4, torch.device('cpu'), 'ALL')
            pred.update_statistics(ens, output_performance=False)
            out[f'{tag}/{cls_name}/proba_sum'] = pred.ensemble_proba.numpy().copy()
            out[f'{tag}/{cls_name}/ent_sum'] = pred.expected_data_uncertainty.numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'swag_e2e.npz'), **out)
    print('G8 swag_e2e', {k: v.shape for k, v in out.items() if k.endswith('samples')})


# ------------------------------------------------------------------------------------- G6
def gen_e2e():
    out = {}
    g = torch.Generator().manual_seed(0)
    xtr = torch.randn(64, 1, 28, 28, generator=g)
    ytr = torch.randint(0, 10, (64,), generator=g)
    g = torch.Generator().manual_seed(1)
    xte = torch.randn(16, 1, 28, 28, generator=g)
    yte = torch.randint(0, 10, (16,), generator=g)
    out['x_train'], out['y_train'], out['x_test'], out['y_test'] = xtr.numpy(), ytr.numpy(), xte.numpy(), yte.numpy()
    train = DataLoader(TensorDataset(xtr, ytr), batch_size=32, shuffle=False)
    test = DataLoader(TensorDataset(xte, yte), batch_size=16, shuffle=False)
    for name, cls, hyp in (
        ('SGLD', inference.SGLD, {'lr': 0.1, 'prior_std': 0.1664, 'num_samples': 2, 'alpha': 1.0, 'burn_in_epochs': 1}),
        ('SGHMC', inference.SGHMC, {'lr': 0.05, 'prior_std': 0.5, 'num_samples': 2, 'alpha': 0.5, 'burn_in_epochs': 1}),
    ):
        util.set_random_seed(0)
        net = our_models.LeNet5(10)
        theta0 = flat(net.parameters())
        s = cls(dict(hyp), net, train)
        tap = NoiseTap(s.optimizer)
        with quiet():
            ens = s.sample()
        out[f'{name}/theta0'] = theta0
        out[f'{name}/eps'] = np.stack([r['eps'] for r in tap.records])
        out[f'{name}/lr'] = np.array([r['lr'] for r in tap.records])
        out[f'{name}/grad0'] = tap.records[0]['grad']
        out[f'{name}/samples'] = np.stack([flat(m.parameters()) for m in ens])
        pred = tasks.Prediction({'in_distribution_test': test}, 10, torch.device('cpu'), 'ALL')
        pred.update_statistics(ens, output_performance=False)
        out[f'{name}/proba_sum'] = pred.ensemble_proba.numpy()
        out[f'{name}/ent_sum'] = pred.expected_data_uncertainty.numpy()
        out[f'{name}/metrics'] = json.dumps({k: float(v) for k, v in pred.get_performance_metrics().items()})
        out[f'{name}/hyper'] = json.dumps(hyp)
    np.savez_compressed(os.path.join(OUT, 'e2e_lenet5.npz'), **out)
    print('G6 e2e_lenet5 steps', out['SGLD/eps'].shape, out['SGHMC/eps'].shape)


def gen_e2e_preresnet():
    """G9: the reference's SGHMC on its own PreResNet class (depth 8: same blocks as the PreResNet-20 of
    BASELINE configs[1], a third of the parameters so the captured noise stays small), CPU, 2 samples x 2
    minibatch steps, then its Prediction task on 64 test rows."""
    out = {}
    g = torch.Generator().manual_seed(0)
    xtr, ytr = torch.randn(256, 3, 32, 32, generator=g), torch.randint(0, 10, (256,), generator=g)
    g = torch.Generator().manual_seed(1)
    xte, yte = torch.randn(64, 3, 32, 32, generator=g), torch.randint(0, 10, (64,), generator=g)
    # inputs are regenerated by the tests from the same CPU generators (same torch build => same values); only a
    # checksum travels
    out['input_checksum'] = np.array([float(xtr.double().sum()), float(ytr.sum()), float(xte.double().sum()), float(yte.sum())])
    train = DataLoader(TensorDataset(xtr, ytr), batch_size=128, shuffle=False)
    test = DataLoader(TensorDataset(xte, yte), batch_size=64, shuffle=False)
    hyp = {'lr': 0.05, 'prior_std': 0.5, 'num_samples': 2, 'alpha': 0.5, 'burn_in_epochs': 0}
    util.set_random_seed(0)
    net = models.PreResNet8.base(num_classes=10, depth=8)
    out['theta0'] = flat(net.parameters())
    out['buffers0'] = torch.cat([b.detach().float().reshape(-1) for b in net.buffers()]).numpy()
    # the pre-activations this run computes within 1e-4 of zero and the ReLU gates it takes there, per minibatch step and
    # BatchNorm call (tests/gate_lists.py: read by forward hooks, the run itself is untouched): the GPU replays take
    # them as given (ursa_bn_relu_bwd_gated_f32), see gen_e2e_preresnet_seeds
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from gate_lists import NearZeroGates, pack
    log = NearZeroGates(net)
    s = inference.SGHMC(dict(hyp), net, train)
    tap = NoiseTap(s.optimizer)
    gate_steps, tapped = [], tap.opt.step

    def step_and_take(add_langevin_noise=True, closure=None):
        gate_steps.append(log.take())
        return tapped(add_langevin_noise=add_langevin_noise, closure=closure)
    s.optimizer.step = step_and_take
    with quiet():
        ens = s.sample()
    log.remove()
    out.update(pack(gate_steps))
    out['eps'] = np.stack([r['eps'] for r in tap.records])
    out['samples'] = np.stack([flat(m.parameters()) for m in ens])
    out['sample_buffers'] = np.stack([torch.cat([b.detach().float().reshape(-1) for b in m.buffers()]).numpy() for m in ens])
    pred = tasks.Prediction({'in_distribution_test': test}, 10, torch.device('cpu'), 'ALL')
    pred.update_statistics(ens, output_performance=False)
    out['proba_sum'], out['ent_sum'] = pred.ensemble_proba.numpy(), pred.expected_data_uncertainty.numpy()
    out['metrics'] = json.dumps({k: float(v) for k, v in pred.get_performance_metrics().items()})
    out['hyper'] = json.dumps(hyp)
    np.savez_compressed(os.path.join(OUT, 'e2e_preresnet8.npz'), **out)
    print('G9 e2e_preresnet8 steps', out['eps'].shape)


PRERESNET8_SEEDS = 8


def preresnet8_seed_inputs(sd):
    """Inputs of seed `sd` of G16 (the GPU tests regenerate them from the same CPU generators)."""
    g = torch.Generator().manual_seed(9000 + sd)
    xtr, ytr = torch.randn(128, 3, 32, 32, generator=g), torch.randint(0, 10, (128,), generator=g)
    xte, yte = torch.randn(64, 3, 32, 32, generator=g), torch.randint(0, 10, (64,), generator=g)
    return xtr, ytr, xte, yte


def gen_e2e_preresnet_seeds():
    """G16: the reference's SGHMC on its own PreResNet-8 for 8 seeds, 4 samples of ONE 128-row minibatch step each (so
    the ensemble IS the trajectory after 1, 2, 3, 4 steps), with, per step and per BatchNorm call,
    the pre-activations the reference computed within 1e-4 of zero and the ReLU gate it took there. The GPU tests
    (tests/test_gate_parity_gpu.py) replay every seed and assert north_star's 1e-5 on the predictive after every step
    that took the reference's gates, naturally or with the listed gates given to the backward launch; they also
    compare K6 with MIOpen's BatchNorm launches seed by seed. Only outputs travel: the inputs, the initial weights and
    the Langevin noise are regenerated from torch's CPU generator (checksums in the fixture); the noise of step k is
    what the reference's own `torch.randn_like` calls (optim_sghmc.py:64) draw after torch.manual_seed(5000+100*seed+k)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import copy
    from gate_lists import NearZeroGates, TAU, pack
    hyp = {'lr': 0.05, 'prior_std': 0.5, 'num_samples': 4, 'alpha': 0.5, 'burn_in_epochs': 0}
    out = {'hyper': json.dumps(hyp), 'seeds': np.arange(PRERESNET8_SEEDS), 'tau': np.array(TAU)}
    for sd in range(PRERESNET8_SEEDS):
        xtr, ytr, xte, yte = preresnet8_seed_inputs(sd)
        train = DataLoader(TensorDataset(xtr, ytr), batch_size=128, shuffle=False)
        test = DataLoader(TensorDataset(xte, yte), batch_size=64, shuffle=False)
        util.set_random_seed(sd)
        net = models.PreResNet8.base(num_classes=10, depth=8)
        util.set_random_seed(sd)
        ours = our_models.PreResNet(10, 8)
        assert np.array_equal(flat(net.parameters()), flat(ours.parameters())), 'our class initialises differently'
        theta0 = flat(net.parameters())
        log = NearZeroGates(net)
        s = inference.SGHMC(dict(hyp), net, train)
        orig, eps_sums, steps = s.optimizer.step, [], []

        def step(add_langevin_noise=True, closure=None, _k=[0], _sd=sd, _orig=orig):
            steps.append(log.take())
            torch.manual_seed(5000 + 100 * _sd + _k[0])
            state = torch.get_rng_state()
            eps_sums.append(float(sum(torch.randn_like(p).double().sum() for p in s.optimizer.param_groups[0]['params'])))
            torch.set_rng_state(state)
            _k[0] += 1
            return _orig(add_langevin_noise=add_langevin_noise, closure=closure)
        s.optimizer.step = step
        with quiet():
            ens = s.sample()
        assert len(ens) == 4 and len(steps) == 4 and all(len(c) == 7 for c in steps)
        probas, ents = [], []
        for m in ens:
            pred = tasks.Prediction({'in_distribution_test': test}, 10, torch.device('cpu'), 'ALL')
            pred.update_statistics([m], output_performance=False)
            probas.append(pred.ensemble_proba.numpy().copy())
            ents.append(pred.expected_data_uncertainty.numpy().copy())
        pre = f's{sd}/'
        out[pre + 'checksums'] = np.array([float(xtr.double().sum()), float(ytr.sum()), float(xte.double().sum()),
                                           float(yte.sum()), float(theta0.astype(np.float64).sum())] + eps_sums)
        out[pre + 'proba_step'], out[pre + 'ent_step'] = np.stack(probas), np.stack(ents)
        out[pre + 'theta_final_sum'] = np.array(float(flat(ens[-1].parameters()).astype(np.float64).sum()))
        for k, v in pack(steps).items():
            out[pre + k] = v
        print('G16 seed', sd, 'near-zero pre-activations per step', out[pre + 'gate_counts'].sum(1).tolist(),
              'of', int(out[pre + 'numel'].sum()))
    np.savez_compressed(os.path.join(OUT, 'e2e_preresnet8_seeds.npz'), **out)


def gen_e2e_cyclic():
    """G12: the reference's cSGHMC and cSGLD end to end on a tiny MLP: per-iteration cyclical lr, noise only in the
    tail of each cycle, samples only from the last epochs of a cycle — parameters of every emitted sample with the
    noise the reference drew."""
    out = {}
    loader = tiny_loader()
    for name, cls, alpha in (('cSGHMC', inference.cSGHMC, 0.5), ('cSGLD', inference.cSGLD, 1.0)):
        hyp = {'lr_0': 0.05, 'prior_std': 1.0, 'num_samples_per_cycle': 2, 'cycle_length': 4, 'burn_in_epochs': 1,
               'num_cycles': 2, 'alpha': alpha}
        util.set_random_seed(3)
        net = tiny_net()
        out[f'{name}/theta0'] = flat(net.parameters())
        s = cls(dict(hyp), net, loader)
        tap = NoiseTap(s.optimizer)
        with quiet():
            ens = s.sample()
        out[f'{name}/eps'] = np.stack([r['eps'] for r in tap.records])
        out[f'{name}/lr'] = np.array([r['lr'] for r in tap.records])
        out[f'{name}/noise'] = np.array([r['noise'] for r in tap.records])
        out[f'{name}/samples'] = np.stack([flat(m.parameters()) for m in ens])
        out[f'{name}/hyper'] = json.dumps(hyp)
        print('G12', name, 'steps', len(tap.records), 'samples', len(ens), 'noisy steps', int(out[f'{name}/noise'].sum()))
    np.savez_compressed(os.path.join(OUT, 'e2e_cyclic.npz'), **out)


def gen_cyclic_update_hyp():
    """G15: the reference's cSGHMC across `update_hyp` (the hyper-optimisation loop's call, hyper_optimization.py:51-73):
    constructor run, then update_hyp with another cycle layout and a second run. Pins the quirk that update_hyp does NOT
    recompute `total_iterations` (csghmc.py:48-62 vs :42-44): the second run's cyclical schedule restarts where the
    constructor's cycle arithmetic says, not where the new hyper-parameters would."""
    loader = tiny_loader()
    hyp = {'lr_0': 0.05, 'prior_std': 1.0, 'num_samples_per_cycle': 2, 'cycle_length': 4, 'burn_in_epochs': 1,
           'num_cycles': 2, 'alpha': 0.5}
    hyp2 = {'lr_0': 0.02, 'prior_std': 2.0, 'num_samples_per_cycle': 2, 'cycle_length': 5, 'burn_in_epochs': 2,
            'num_cycles': 1, 'alpha': 0.3}
    util.set_random_seed(3)
    net = tiny_net()
    out = {'theta0': flat(net.parameters()), 'hyper': json.dumps(hyp), 'hyper2': json.dumps(hyp2)}
    s = inference.cSGHMC(dict(hyp), net, loader)
    tap = NoiseTap(s.optimizer)
    with quiet():
        ens = s.sample()
    out['eps'] = np.stack([r['eps'] for r in tap.records])
    out['lr'] = np.array([r['lr'] for r in tap.records])
    out['samples'] = np.stack([flat(m.parameters()) for m in ens])
    util.set_random_seed(9)
    s.update_hyp(dict(hyp2))
    out['theta1'] = flat(s.model.parameters())
    out['total_iterations_after'] = np.float64(s.total_iterations)
    tap2 = NoiseTap(s.optimizer)
    with quiet():
        ens2 = s.sample()
    out['eps2'] = np.stack([r['eps'] for r in tap2.records])
    out['lr2'] = np.array([r['lr'] for r in tap2.records])
    out['noise2'] = np.array([r['noise'] for r in tap2.records])
    out['samples2'] = np.stack([flat(m.parameters()) for m in ens2])
    np.savez_compressed(os.path.join(OUT, 'e2e_cyclic_update_hyp.npz'), **out)
    print('G15 cyclic update_hyp: steps', len(tap.records), len(tap2.records), 'samples', len(ens), len(ens2),
          'total_iterations', float(out['total_iterations_after']), 'lr2', out['lr2'][:4])


def gen_sgd_sampler():
    """G13: the reference's SGD baseline sampler (inference/sgd.py) on a tiny MLP: constructor run (cosine to lr/100),
    then update_hyp (re-init, cosine to lr/2, loop still uses the constructor's epoch count) and a second run."""
    hyp = {'lr': 0.1, 'epochs': 2, 'momentum': 0.9, 'weight_decay': 1e-3}
    hyp2 = {'lr': 0.05, 'epochs': 5, 'momentum': 0.8, 'weight_decay': 0.0}
    util.set_random_seed(5)
    net = tiny_net()
    out = {'theta0': flat(net.parameters()), 'hyper': json.dumps(hyp), 'hyper2': json.dumps(hyp2)}
    s = inference.SGD(dict(hyp), net, tiny_loader())
    with quiet():
        m = s.sample(num_samples=2)
    out['sample'] = flat(m[0].parameters())
    out['lr_after'] = np.float64(s.optimizer.param_groups[0]['lr'])
    util.set_random_seed(6)
    s.update_hyp(dict(hyp2))
    out['theta1'] = flat(s.model.parameters())
    with quiet():
        m = s.sample()
    out['sample2'] = flat(m[0].parameters())
    out['lr_after2'] = np.float64(s.optimizer.param_groups[0]['lr'])
    np.savez(os.path.join(OUT, 'sgd_sampler.npz'), **out)
    print('G13 sgd_sampler lr after', float(out['lr_after']), float(out['lr_after2']))


def gen_hmc_wrapper():
    """G14: the reference's HMC WRAPPER (inference/hmc.py:62-85 + util.convert_sample_to_net) around a stand-in
    hamiltorch: `sample_model` is replaced by a function that records its keyword arguments and returns a marked
    trajectory (position k is params_init + k), so the fixture pins what the wrapper — not hamiltorch — decides:
    the arguments it derives from the hyper-parameters, which trajectory positions become ensemble members for
    each (num_samples, L, burn), and that members are independent copies. hamiltorch's arithmetic stays unpinned."""
    import hamiltorch
    calls = []

    def flatten(model):
        return torch.cat([p.detach().reshape(-1) for p in model.parameters()])

    def unflatten(model, flat):
        out, off = [], 0
        for p in model.parameters():
            out.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        return out

    def sample_model(model, x, y, params_init=None, model_loss=None, num_samples=None, burn=None, inv_mass=None,
                     step_size=None, num_steps_per_sample=None, tau_out=None, tau_list=None, debug=None):
        calls.append(dict(model_loss=model_loss, num_samples=num_samples, burn=burn, step_size=step_size,
                          num_steps_per_sample=num_steps_per_sample, tau_out=tau_out, tau_list=tau_list.tolist(),
                          inv_mass_unique=sorted(set(inv_mass.tolist())), x_rows=len(x), n_params=len(params_init)))
        return [params_init + float(k) for k in range(num_steps_per_sample * num_samples + 1)]
    hamiltorch.util.flatten, hamiltorch.util.unflatten, hamiltorch.sample_model = flatten, unflatten, sample_model
    out = {'cases': []}
    for num_samples, L, burn in ((4, 1, -1), (4, 1, 0), (4, 3, 0), (5, 2, 1), (6, 3, 2), (3, 4, -1), (2, 5, 0), (7, 1, 3)):
        util.set_random_seed(1)
        net = tiny_net()
        theta0 = flat(net.parameters())
        s = inference.HMC({'step_size': 0.01, 'num_samples': num_samples, 'L': L, 'tau': 2.5, 'burn': burn, 'mass': 4.0},
                          net, tiny_loader())
        with quiet():
            ens = s.sample()
        idx = [int(round(float(flat(m.parameters())[0] - theta0[0]))) for m in ens]
        independent = all(a is not b for a in ens for b in ens if a is not b) and all(m is not net for m in ens)
        out['cases'].append(dict(num_samples=num_samples, L=L, burn=burn, selected=idx, n_members=len(ens),
                                 independent=bool(independent)))
    out['call'] = calls[0]
    json.dump(out, open(os.path.join(OUT, 'hmc_wrapper.json'), 'w'), indent=1)
    print('G14 hmc_wrapper', [(c['num_samples'], c['L'], c['burn'], c['selected']) for c in out['cases']][:4], out['call'])


def gen_mcdropout():
    """G10: the reference's MCdropout (vi_dropout.py) on its own MLP -> MLP_dropout swap: per-minibatch OneCycleLR
    (lr, momentum) pairs, parameters after each sample_iterative, then update_hyp (CosineAnnealingLR per minibatch)
    and one more sample; and 3 stochastic eval forwards (masks stay on). Everything is a function of torch's
    global CPU generator: seed, then construct, then run."""
    hyp = {'lr': 0.05, 'epochs': 1, 'dropout': 0.2, 'lengthscale': 0.01, 'num_samples': 2, 'momentum': 0.9,
           'weight_decay': 0}
    hyp2 = dict(hyp, lr=0.02, momentum=0.8, weight_decay=1e-3)
    loader = tiny_loader()
    torch.manual_seed(21)
    s = inference.MCdropout(dict(hyp), models.mlp.MLP(16, 12, 4), loader)
    theta0 = flat(s.model.parameters())
    log = []
    orig = s.optimizer.step

    def tap(*a, **k):
        g = s.optimizer.param_groups[0]
        log.append((g['lr'], g['momentum']))
        return orig(*a, **k)
    s.optimizer.step = tap
    samples = []
    with quiet():
        for _ in range(2):
            m = s.sample_iterative()
            samples.append(flat(m.parameters()))
    xt = torch.randn(5, 12, generator=torch.Generator().manual_seed(4))
    s.model.eval()
    with torch.no_grad():
        mc = np.stack([s.model(xt).numpy() for _ in range(3)])
    # update_hyp: reset_model re-initialises from the global generator, new optimizer, cosine per minibatch
    s.update_hyp(dict(hyp2))
    theta1 = flat(s.model.parameters())
    log2 = []
    orig2 = s.optimizer.step

    def tap2(*a, **k):
        g = s.optimizer.param_groups[0]
        log2.append((g['lr'], g['momentum']))
        return orig2(*a, **k)
    s.optimizer.step = tap2
    with quiet():
        m = s.sample_iterative()
    np.savez(os.path.join(OUT, 'mcdropout.npz'), hyper=json.dumps(hyp), hyper2=json.dumps(hyp2), theta0=theta0,
             lr_mom=np.array(log), samples=np.stack(samples), mc_logits=mc, x_test=xt.numpy(), theta1=theta1,
             lr_mom2=np.array(log2), sample2=flat(m.parameters()), weight_decay=np.float64(s.weight_decay))
    print('mcdropout: steps', len(log), len(log2), 'lr range', min(l for l, _ in log), max(l for l, _ in log),
          'mom range', min(m_ for _, m_ in log), max(m_ for _, m_ in log), 'mc spread', float(np.abs(mc[0] - mc[1]).max()))


def gen_experiment_columns():
    """G11: the on-disk row format of URSABench/experiment.py:249-266 — six fixed columns, the hyper-parameter
    values in sorted-key order, then results_dic's values in sorted-key order. The key NAMES come from the
    reference's own task objects run here (Prediction.required_metric_list with metric_list='ALL',
    OODDetection.update_statistics(...).keys()), combined by the rules of experiment.py:203-216 and :249-250;
    the OOD data-set names are the literals of experiment.py:115,137. Also time_script.py:114-125's JSON keys."""
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(8, 12, generator=g), torch.randint(0, 4, (8,), generator=g)
    ds = torchvision.datasets.cifar.CIFAR10
    l_in = DataLoader(ds(x, y), batch_size=4)
    l_out = DataLoader(ds(x * 2, y), batch_size=4)
    ms = [member(12, 4, 1, 1.0), member(12, 4, 2, 1.0)]
    pred = tasks.Prediction({'in_distribution_test': l_in}, 4, torch.device('cpu'), 'ALL')
    ood = tasks.OODDetection({'in_distribution_test': l_in, 'out_distribution_test': l_out}, 4, torch.device('cpu'))
    with quiet():
        ood_keys = list(ood.update_statistics(ms, output_performance=True).keys())
    out = {'fixed_columns': ['dataset', 'model', 'seed', 'inference_method', 'task', 'batch_size'], 'datasets': {}}
    for name, oods in (('MNIST', ['FashionMNIST', 'KMNIST']), ('CIFAR10', ['STL10', 'SVHN']), ('CIFAR100', ['STL10', 'SVHN'])):
        keys = [k + '_' + d + sfx for d in oods for k in ood_keys for sfx in ('_mean', '_std')]
        keys += [k + sfx for k in pred.required_metric_list for sfx in ('_mean', '_std')]
        keys += ['cost_mean', 'cost_std']
        out['datasets'][name] = sorted(keys)
    out['time_script_keys_rule'] = ['<method>_mean', '<method>_std']
    out['time_script_methods'] = ['HMC', 'SGLD', 'SGHMC', 'cSGLD', 'cSGHMC', 'SWAG', 'PCA', 'MCdropout', 'SGD', 'PCASubspaceSampler']
    json.dump(out, open(os.path.join(OUT, 'experiment_columns.json'), 'w'), indent=1)
    print('experiment_columns', {k: len(v) for k, v in out['datasets'].items()}, ood_keys)


def gen_model_keys():
    res = {}
    for name, ref, ours in (
        ('PreResNet20_c10', lambda: models.PreResNet8.base(num_classes=10, depth=20), lambda: our_models.PreResNet(10, 20)),
        ('PreResNet164_c100', lambda: models.PreResNet164.base(num_classes=100, depth=164), lambda: our_models.PreResNet(100, 164)),
        ('WideResNet28x10_c100', lambda: models.WideResNet28x10.base(num_classes=100, depth=28, widen_factor=10),
         lambda: our_models.WideResNet(100, 28, 10)),
        ('MLP200MNIST', lambda: models.MLP200MNIST.base(num_classes=10, **models.MLP200MNIST.kwargs),
         lambda: our_models.MLP(200, 784, 10)),
    ):
        r = ref()
        sd = r.state_dict()
        res[name] = dict(keys=list(sd.keys()), shapes=[list(v.shape) for v in sd.values()],
                         n_params=sum(p.numel() for p in r.parameters()), n_param_tensors=len(list(r.parameters())),
                         param_names=[k for k, _ in r.named_parameters()])
        o = ours()
        assert list(o.state_dict().keys()) == res[name]['keys'], name
        assert [list(v.shape) for v in o.state_dict().values()] == res[name]['shapes'], name
        assert [k for k, _ in o.named_parameters()] == res[name]['param_names'], name
    # forward equivalence of our PreResNet-20 with the reference class on the same weights
    r = models.PreResNet8.base(num_classes=10, depth=20)
    o = our_models.PreResNet(10, 20)
    o.load_state_dict(r.state_dict())
    x = torch.randn(4, 3, 32, 32, generator=torch.Generator().manual_seed(3))
    r.eval(); o.eval()
    with torch.no_grad():
        res['PreResNet20_c10']['forward_equal'] = bool(torch.equal(r(x), o(x)))
    json.dump(res, open(os.path.join(OUT, 'model_keys.json'), 'w'))
    print('model_keys', {k: (v['n_params'], v['n_param_tensors']) for k, v in res.items()},
          'fwd equal:', res['PreResNet20_c10']['forward_equal'])


if __name__ == '__main__':
    which = sys.argv[1:] or ['k1', 'sgd', 'sched', 'csghmc', 'tasks', 'tasks_distilled', 'swag', 'swag_e2e', 'e2e', 'e2e_preresnet', 'e2e_preresnet_seeds', 'mcdropout', 'columns', 'cyclic', 'cyclic_update_hyp', 'sgd_sampler', 'hmc_wrapper', 'keys']
    fns = dict(k1=gen_k1, sgd=gen_sgd, swag_e2e=gen_swag_e2e, e2e_preresnet=gen_e2e_preresnet, e2e_preresnet_seeds=gen_e2e_preresnet_seeds, sched=gen_schedules, csghmc=gen_csghmc, tasks=gen_tasks, tasks_distilled=gen_tasks_distilled, swag=gen_swag, e2e=gen_e2e,
               keys=gen_model_keys, mcdropout=gen_mcdropout, columns=gen_experiment_columns, cyclic=gen_e2e_cyclic, cyclic_update_hyp=gen_cyclic_update_hyp, sgd_sampler=gen_sgd_sampler, hmc_wrapper=gen_hmc_wrapper)
    for w in which:
        fns[w]()
