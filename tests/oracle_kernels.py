"""A kernel set with the interface of ursabench_amd._native.HipKernels, backed by the CPU oracle.

TESTS ONLY: it lets the `-m "not gpu"` suite drive the host logic (arena, optimizer state
machine, schedules, samplers, tasks, gloo sharding) on CPU torch tensors. The product never
selects it — ursabench_amd's default kernel set is the HIP library and refuses CPU tensors."""
import ctypes

import numpy as np
import torch

import oracle_lib as O
from ursabench_amd._native import CTL_BYTES, STEP_ADVANCE, StepCtl


def _np(t):
    if t is None:
        return None
    assert t.device.type == 'cpu' and t.dtype == torch.float32 and t.is_contiguous()
    return t.detach().numpy()


class OracleKernels:
    name = 'cpu-oracle (tests only)'

    def __init__(self):
        self.step_log = []          # (lr, mu, flags, step) of every K1 call, for schedule tests

    def sgmcmc_step(self, theta, grad, mom, *, lr, mu, c_wd, c_noise, n_train, flags, seed=0, step=0, eps=None,
                    snapshot=None):
        f = np.float32
        self.step_log.append((lr, mu, flags, step))
        O.sgmcmc_step(_np(theta), _np(grad), _np(mom), lr=f(lr), mu=f(mu), c_wd=f(c_wd), c_noise=f(c_noise),
                      n_train=f(n_train), flags=flags, seed=seed, step=step, eps=_np(eps), snapshot=_np(snapshot))

    def sgmcmc_step_ctl(self, theta, grad, mom, ctl, *, eps=None, snapshot=None):
        c = StepCtl.from_buffer_copy(bytes(ctl.numpy()))
        self.step_log.append((c.lr, c.mu, c.flags, c.step))
        O.sgmcmc_step(_np(theta), _np(grad), _np(mom) if c.mu != 0 else None, lr=c.lr, mu=c.mu, c_wd=c.c_wd,
                      c_noise=c.c_noise, n_train=c.n_train, flags=c.flags & 0x1F, seed=c.seed, step=c.step, eps=_np(eps),
                      snapshot=_np(snapshot))
        if c.flags & STEP_ADVANCE:             # the launch advances its own control block (last workgroup's job)
            self.step_ctl_advance(ctl)

    def sgmcmc_step_multi(self, theta, grad, mom, ctl, *, n_per_chain=None, eps=None, snapshot=None):
        K, stride = theta.shape
        n = stride if n_per_chain is None else n_per_chain
        for k in range(K):
            self.sgmcmc_step_ctl(theta[k, :n], grad[k, :n], mom[k, :n], ctl[k * CTL_BYTES:(k + 1) * CTL_BYTES],
                                 eps=None if eps is None else eps[k, :n],
                                 snapshot=None if snapshot is None else snapshot[k, :n])

    def step_ctl_advance(self, ctl):
        if ctl.numel() > CTL_BYTES:
            for k in range(ctl.numel() // CTL_BYTES):
                self.step_ctl_advance(ctl[k * CTL_BYTES:(k + 1) * CTL_BYTES])
            return
        c = StepCtl.from_buffer_copy(bytes(ctl.numpy()))
        c.step += 1
        c.flags &= ~O.STEP_FIRST
        if c.sched and c.sched_len:            # the table's (host, in this kernel set) address rides in the block
            sched = np.ctypeslib.as_array((ctypes.c_float * (2 * c.sched_len)).from_address(c.sched)).reshape(-1, 2)
            k = (c.step - c.sched_base) % c.sched_len
            c.lr = float(sched[k, 0])
            if c.flags & O.STEP_SGD:
                c.mu = float(sched[k, 1])
            else:
                c.c_noise = float(sched[k, 1])
        ctl.copy_(torch.frombuffer(bytearray(bytes(c)), dtype=torch.uint8))

    def philox_normal(self, out, *, seed, step):
        out.copy_(torch.from_numpy(O.philox_normal(out.numel(), seed, step)).view_as(out))

    def swag_collect(self, mean, sq, w, *, decay, denom):
        O.swag_collect(_np(mean), _np(sq), _np(w), decay=decay, denom=denom)

    def swag_draw(self, out, mean, sq, *, var_clamp, scale=1.0, seed=0, draw=0, eps=None):
        O.swag_draw(_np(out), _np(mean), _np(sq), var_clamp=var_clamp, scale=scale, seed=seed, draw=draw, eps=_np(eps))

    def swag_std(self, out, mean, sq, *, var_clamp, scale=1.0):
        m, q = _np(mean), _np(sq)
        _np(out)[:] = np.sqrt(np.maximum(q - m * m, np.float32(var_clamp))) * np.float32(scale)

    def swag_draw_std(self, out, mean, std, *, seed=0, draw=0, eps=None):
        e = _np(eps) if eps is not None else O.philox_normal(out.numel(), seed, draw)
        _np(out)[:] = e * _np(std) + _np(mean)

    def bma_accumulate(self, logits, proba_sum, ent_sum=None, *, one_minus_gamma, gamma_over_c, smoothed,
                       risk_sum=None, cost=None):
        if logits.shape[0] == 0 or logits.shape[1] == 0:
            return
        O.bma_accumulate(_np(logits), _np(proba_sum), _np(ent_sum), one_minus_gamma=one_minus_gamma,
                         gamma_over_c=gamma_over_c, smoothed=smoothed, risk_sum=_np(risk_sum), cost=_np(cost))

    def leapfrog(self, theta, mom, grad, *, kick_coef, step_size, inv_mass, flags, kinetic_out=None, ws=None):
        ke = O.leapfrog(_np(theta), _np(mom), _np(grad), kick_coef=kick_coef, step_size=step_size, inv_mass=inv_mass,
                        flags=flags, want_kinetic=kinetic_out is not None)
        if kinetic_out is not None:
            kinetic_out += ke

    def sumsq(self, x, out, ws):
        out += O.sumsq(_np(x))
