"""HMC — drop-in for URSABench/inference/hmc.py:23-85.

The reference delegates all arithmetic to `hamiltorch.sample_model` (hmc.py:71-75), an
un-vendored, un-pinned dependency that is absent from /root/reference: PARITY UNPINNED. This file
restates hamiltorch's published sampler: full-batch log posterior
log p = -sum CE - tau/2 ||theta||^2 (the Normal prior's constants cancel in the accept test and
are dropped), momentum ~ N(0, mass), leapfrog with a half kick, L x (drift, kick) and a final
half-kick correction, Metropolis accept on H = -log p + 1/2 p^T M^-1 p. Forward/backward are stock
PyTorch-ROCm over the whole training set held on the device (hmc.py:44-50); the leapfrog
sub-steps and both energy reductions are the K4 kernels on the flat arena.

Launches per proposal (the leapfrog of hamiltorch, call site hmc.py:71-75): one K4 launch for the
kinetic energy of the fresh momentum, one fused half-kick + first drift, one fused kick + drift
between consecutive gradient evaluations (L - 1 of them, 20 B/param each instead of 24 for two
launches), and after the last evaluation the full kick and the half-kick correction — kept as two
roundings like hamiltorch's two statements — the second with the kinetic-energy reduction fused:
L + 3 launches (round 2: 2L + 4). The Metropolis uniform is Philox4x32-10 under the chain's own
key at (proposal index, lane 2^64 - 1), so a chain's accept sequence does not depend on what else
draws random numbers in the process.

The returned trajectory follows hamiltorch's layout: the initial position, then the L positions of
every proposal (a rejected proposal repeats the previous L), thinned exactly like hmc.py:80 —
`samples[burn*L::L]`. Only the positions that thinning selects are kept in HBM.
"""
import math

import torch

from .. import _native, fused_bn, fused_conv
from .._capture import capture, side_streams
from ..arena import FlatArena, MemberBank
from ..util import deferred_bn_counters, reset_model
from .inference_base import _Inference


def philox4x32_10(ctr, key):
    """Philox4x32-10 on the host (same function as csrc/ursa_rng.h; pinned against the oracle and Random123's
    known-answer vectors in tests/test_hmc_cpu.py). ctr: 4 x u32, key: 2 x u32 -> 4 x u32."""
    c0, c1, c2, c3 = ctr
    k0, k1 = key
    for _ in range(10):
        p0, p1 = 0xD2511F53 * c0, 0xCD9E8D57 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c3 ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF
        k0, k1 = (k0 + 0x9E3779B9) & 0xFFFFFFFF, (k1 + 0xBB67AE85) & 0xFFFFFFFF
    return c0, c1, c2, c3


def mh_uniform(seed, proposal):
    """u in (0, 1) for the accept test of proposal `proposal` of the chain keyed `seed`: Philox lane 2^64 - 1 (no
    arena reaches it: the kernels stop at 2^40 elements) of call `proposal`, first word, as (x + 0.5) / 2^32."""
    x = philox4x32_10((0xFFFFFFFF, 0xFFFFFFFF, proposal & 0xFFFFFFFF, (proposal >> 32) & 0xFFFFFFFF),
                      (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))[0]
    return (x + 0.5) / 4294967296.0


class HMC(_Inference):
    def __init__(self, hyperparameters, model=None, train_loader=None, model_loss='multi_class_linear_output',
                 device=torch.device('cpu'), *, kernels=None, seed=None, use_graph=None):
        super().__init__(hyperparameters, model, train_loader, device)
        if hyperparameters == None:  # noqa: E711  (hmc.py:32-34)
            hyperparameters = {'step_size': 0.001, 'num_samples': 10, 'L': 1, 'tau': 0.1, 'burn': -1, 'mass': 1.0}
        self._read_hyp(hyperparameters)
        self.model_loss = model_loss
        if model_loss != 'multi_class_linear_output':
            raise NotImplementedError
        xs, ys = [], []
        for data, target in train_loader:                       # hmc.py:44-50: the whole set, on the device
            xs.append(data.clone().to(device))
            ys.append(target.clone().to(device))
        self.x, self.y = torch.cat(xs), torch.cat(ys)
        self.model = model.to(device) if isinstance(model, torch.nn.Module) else model
        self.kernels = kernels if kernels is not None else _native.default_kernels()
        self.seed = int(torch.initial_seed() if seed is None else seed) & 0xFFFFFFFFFFFFFFFF
        self.arena = None
        self.use_graph = (torch.device(device).type == 'cuda') if use_graph is None else use_graph
        self._graph, self._evals, self._u = None, 0, None
        self._proposals = 0
        self.accepted = 0

    def _read_hyp(self, h):
        self.step_size = h['step_size']
        self.num_samples = h['num_samples']
        self.L = h['L']
        self.tau = h['tau']
        self.burn = h['burn']
        self.mass = h['mass']

    def update_hyp(self, hyperparameters):
        self._read_hyp(hyperparameters)
        self.model = reset_model(self.model)
        self._graph, self._evals = None, 0          # tau is baked into the captured graph

    # ---- pieces --------------------------------------------------------------------------------
    def _bind(self):
        if self.arena is None:
            a = self.arena = FlatArena(self.model.parameters(), module=self.model)
            dev = a.device
            self.bank = MemberBank(a)
            self._p = torch.zeros(a.n, device=dev)
            self._glogp = torch.zeros(a.n, device=dev)
            self._mask = torch.zeros(a.n, device=dev)
            self._mask[a.layout.gather_index(dev)] = 1.0
            self._ws = torch.zeros(_native.REDUCE_WS_FLOATS, device=dev)
            self._acc = torch.zeros(1, device=dev)
            self._crit = torch.nn.CrossEntropyLoss(reduction='sum')
            self._slot = {id(p): i for i, p in enumerate(a.params)}

    def _eval_potential(self):
        """U(theta) = sum CE + tau/2 ||theta||^2 into self._u and grad log p into self._glogp."""
        a = self.arena
        params, views = a.params, a.grad_views
        # As in ChainEngine.forward_backward: the BatchNorm counters bumped by ONE multi-tensor add, K7's second launches for all
        # convolutions at once, and autograd handing over fresh gradient tensors (p.grad = None) that ONE multi-tensor copy packs
        # into the arena - with the arena views left in p.grad it runs a `grad += new` kernel per parameter tensor (488 for
        # PreResNet-164: 4.5 % of the C5 configuration's kernel time, + 1.4 % for the counters, profiles/r06_c5_kernel_stats*.csv).
        with deferred_bn_counters(self.model), fused_conv.deferred() as pend:
            nll = self._crit(self.model(self.x), self.y.long().view(-1))
        for p in params:
            p.grad = None
        nll.backward()
        grads = [p.grad for p in params]
        with torch.no_grad():
            done = set()
            if pend:
                slot = self._slot
                stray = [tuple(w.shape) for _, w in pend if id(w) not in slot]
                if stray:
                    raise RuntimeError(f'convolution weights {stray} took K7\'s deferred launch but are not in the sampler\'s arena')
                fused_conv.flush(pend, lambda w: views[slot[id(w)]])
                done = {slot[id(w)] for _, w in pend}
                for i in done:
                    if grads[i] is not None:        # the same weight also went through a call K7 does not cover
                        views[i].add_(grads[i])
            have = [i for i in range(len(grads)) if i not in done and grads[i] is not None]
            if have:
                torch._foreach_copy_([views[i] for i in have], [grads[i] for i in have])
            for i in range(len(grads)):
                if i not in done and grads[i] is None:
                    views[i].zero_()                # a parameter the loss does not reach: gradient 0, as autograd leaves it
        for p, gv in zip(params, views):
            p.grad = gv                             # the arena's contract: p.grad is its view
        with torch.no_grad():
            torch.add(a.grad, a.theta, alpha=self.tau, out=self._glogp)
            self._glogp.neg_()
            self._acc.zero_()
            self.kernels.sumsq(a.theta, self._acc, self._ws)
            self._u.copy_(nll.detach() + 0.5 * self.tau * self._acc[0])

    def _neg_logp_and_grad(self):
        """One full-batch forward/backward + prior. The inputs never change (hmc.py:44-50 holds the whole
        training set on the device), so after two eager evaluations the whole thing — ~3,000 launches for
        PreResNet-164 — is captured once into a hipGraph and replayed for every leapfrog step."""
        if self._u is None:
            self._u = torch.zeros((), device=self.arena.device)
        if self.use_graph and self._graph is None and self._evals >= 2:
            g = torch.cuda.CUDAGraph()
            with capture(g):
                self._eval_potential()
            self._graph = g
        if self._graph is not None:
            self._graph.replay()
        elif self.use_graph:
            side = side_streams(self.arena.device, 1)[0]
            side.wait_stream(torch.cuda.current_stream(self.arena.device))
            with torch.cuda.stream(side):
                self._eval_potential()
            torch.cuda.current_stream(self.arena.device).wait_stream(side)
        else:
            self._eval_potential()
        self._evals += 1
        return self._u.clone()

    def _kinetic(self):
        self._acc.zero_()
        self.kernels.leapfrog(None, self._p, None, kick_coef=0.0, step_size=0.0, inv_mass=1.0 / self.mass, flags=0,
                              kinetic_out=self._acc, ws=self._ws)
        return self._acc[0].clone()

    def _wanted_indices(self):
        """Trajectory positions (0 = the initial one, then L per proposal) that hmc.py:77-82 turns into ensemble
        members: `samples[burn*L::L]` of the L*num_samples+1 positions hamiltorch returns (pinned against the
        reference's wrapper by tests/golden/hmc_wrapper.json)."""
        return list(range(self.L * self.num_samples + 1)[self.burn * self.L::self.L])

    def sample(self, debug=False):
        if not isinstance(self.model, torch.nn.Module):
            raise NotImplementedError
        self._bind()
        a, K, L, eps, inv_mass = self.arena, self.kernels, self.L, self.step_size, 1.0 / self.mass
        wanted = set(self._wanted_indices())                            # what hmc.py:80 will select
        kept = {}

        def keep(idx):
            if idx in wanted:
                kept[idx] = self.bank.snapshot(self.model)

        self.model.train()      # hamiltorch runs the functional model in its current (training) mode
        keep(0)
        prev_positions = None
        KD = _native.LEAP_KICK | _native.LEAP_DRIFT
        for n in range(self.num_samples):
            theta0 = a.theta.clone()
            fb0 = None if a.fbuf is None else a.fbuf.clone()
            proposal = self._proposals
            K.philox_normal(self._p, seed=self.seed, step=proposal)
            self._p.mul_(self._mask).mul_(math.sqrt(self.mass))
            self._proposals += 1
            U0 = self._neg_logp_and_grad()
            H0 = U0 + self._kinetic()
            # opening half kick + first drift: one launch
            K.leapfrog(a.theta, self._p, self._glogp, kick_coef=0.5 * eps, step_size=eps, inv_mass=inv_mass, flags=KD)
            positions = []
            for l in range(L):
                U1 = self._neg_logp_and_grad()                          # at the position the last drift reached
                idx = n * L + l + 1
                if idx in wanted:
                    positions.append((idx, self.bank.snapshot(self.model)))
                if l < L - 1:                                           # kick of step l + drift of step l+1: one launch
                    K.leapfrog(a.theta, self._p, self._glogp, kick_coef=eps, step_size=eps, inv_mass=inv_mass, flags=KD)
            # closing: full kick, then the half-kick correction (two roundings, like hamiltorch's two statements) with
            # the kinetic-energy reduction fused into the second launch
            K.leapfrog(None, self._p, self._glogp, kick_coef=eps, step_size=eps, inv_mass=inv_mass, flags=_native.LEAP_KICK)
            self._acc.zero_()
            K.leapfrog(None, self._p, self._glogp, kick_coef=-0.5 * eps, step_size=eps, inv_mass=inv_mass,
                       flags=_native.LEAP_KICK, kinetic_out=self._acc, ws=self._ws)
            H1 = U1 + self._acc[0]
            rho = min(0.0, float(H0 - H1))                  # the one host sync per proposal (MH test)
            fused_bn.check_held(self.device)                # (no-op unless K6's held form was opted into: a starved launch raises here)
            if debug:
                print({'proposal': n, 'H0': float(H0), 'H1': float(H1), 'rho': rho})
            if math.isfinite(rho) and rho >= math.log(mh_uniform(self.seed, proposal)):
                self.accepted += 1
                for idx, m in positions:
                    kept[idx] = m
                prev_positions = {idx - n * L: m for idx, m in positions}
            else:                                           # reject: restore, repeat the previous L positions
                a.theta.copy_(theta0)
                if fb0 is not None:
                    a.fbuf.copy_(fb0)
                for l in range(1, L + 1):
                    idx = n * L + l
                    if idx in wanted:
                        if prev_positions is not None and l in prev_positions:
                            kept[idx] = prev_positions[l]
                        else:
                            kept[idx] = self.bank.snapshot(self.model)
        if len(kept) != len(wanted):
            print('Warning, thinning of sampling not aligned as reject occured in first sample.')
        return [kept[i] for i in sorted(kept)]
