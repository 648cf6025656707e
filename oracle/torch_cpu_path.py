"""Port of the reference's CPU execution path for the hot loop, in stock torch-CPU ops.

TEST INFRASTRUCTURE / CPU BASELINE ONLY (kind "port"): tests/ and bench.py's cpu_baseline leg
may import this; ursabench_amd/ never does. The reference source cannot travel to the GPU box,
so this restates, op for op, what it executes there:
  * sgmcmc_step_per_tensor  — URSABench/inference/optim_sghmc.py:43-67 (per-tensor loop of
    add / mul_ / add_ / randn_like / mul / div / add / add_)
  * sghmc_epoch             — URSABench/inference/sghmc.py:72-86 (forward, zero_grad, CE,
    backward, loss.item() sync, step) for one pass over a loader
  * prediction_accumulate   — URSABench/tasks/prediction.py:52-64 (per batch, per member:
    softmax twice, CPU accumulate of raw p and of the smoothed entropy)
Pinned in tests/test_cpu_port.py: bitwise equal to the golden trajectories captured from the
imported reference (same torch build, same global-generator seeds).
"""
import math
import time

import torch
import torch.nn.functional as F


@torch.no_grad()
def sgmcmc_step_per_tensor(params, state, *, lr, momentum, weight_decay, num_training_samples,
                           add_langevin_noise=True):
    for p in params:
        if p.grad is None:
            continue
        d_p = p.grad
        if weight_decay != 0:
            d_p = d_p.add(p, alpha=weight_decay / num_training_samples)
        if momentum != 0:
            buf = state.get(p)
            if buf is None:
                buf = torch.clone(d_p).detach()
            buf.mul_(momentum).add_(d_p, alpha=-lr)
            d_p = buf
        else:
            d_p = d_p.mul(-lr)
        if add_langevin_noise:
            d_p = d_p.add(torch.randn_like(d_p) * math.sqrt(2 * (1 - momentum) * lr) / num_training_samples)
        p.add_(d_p)
        if momentum != 0:
            state[p] = d_p


def sghmc_epoch(model, batches, state, *, lr, momentum, weight_decay, num_training_samples, max_steps=None):
    """Returns (steps done, seconds). `batches` yields CPU (x, y)."""
    crit = torch.nn.CrossEntropyLoss()
    params = list(model.parameters())
    model.train()
    total, steps = 0.0, 0
    t0 = time.perf_counter()
    for x, y in batches:
        logits = model(x)
        for p in params:
            p.grad = None
        loss = crit(logits, y)
        loss.backward()
        total += loss.item() * len(x)
        sgmcmc_step_per_tensor(params, state, lr=lr, momentum=momentum, weight_decay=weight_decay,
                               num_training_samples=num_training_samples, add_langevin_noise=True)
        steps += 1
        if max_steps is not None and steps >= max_steps:
            break
    return steps, time.perf_counter() - t0


@torch.no_grad()
def prediction_accumulate(models, batches, num_classes, n_rows, gamma=1e-4, max_batches=None):
    """Returns (ensemble_proba, expected_data_uncertainty, rows done, seconds)."""
    proba = torch.zeros(n_rows, num_classes)
    ent = torch.zeros(n_rows)
    start, nb = 0, 0
    t0 = time.perf_counter()
    for x, _ in batches:
        end = start + len(x)
        for m in models:
            m.eval()
            z = m(x)
            proba[start:end] += F.log_softmax(z, dim=-1).exp_()
            q = (1 - gamma) * F.log_softmax(z, dim=-1).exp_() + gamma * 1 / num_classes
            ent[start:end] += -(q * torch.log(q)).sum(dim=-1)
        start = end
        nb += 1
        if max_batches is not None and nb >= max_batches:
            break
    return proba, ent, start, time.perf_counter() - t0
