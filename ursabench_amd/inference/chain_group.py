"""ChainGroup — K independent chains on ONE GPU, stepped in lock-step.

A single PreResNet-20 chain at batch 128 cannot fill 256 CUs: its ~400 small kernels per step run one
after another. Independent chains never interact, so K of them can share a GPU: the K
minibatch steps are captured as K PARALLEL BRANCHES of one hipGraph (fork/join over K side streams inside
the capture) and the runtime overlaps the branches. Measured on MI355X, PreResNet-20 SGHMC
(tools/exp/multichain_onegraph.py): 422 -> 648 -> 859 -> 870 aggregate minibatch steps/s for K = 1, 2, 4,
8. (K separate graphs launched on K streams do NOT overlap: 410-445 steps/s for every K.)

Each chain keeps its own sampler object (model, control block, Philox seed, member bank) and its own state
machine; the group only drives their epoch generators together. Every chain computes exactly what it would
compute alone on the same minibatch sequence.

The chains' vectors live in ONE slab per kind — theta[K, n], grad[K, n], mom[K, n] — and their control blocks
in one array ctl[K] (SURVEY.md 8b `n_chains`, 8f-1): the K forward/backward branches join, then ONE launch of
`ursa_sgmcmc_step_multi_f32` (blockIdx.y = chain) updates all K chains and advances their K control blocks.
Round 2 issued K x (update + 1-thread advance) launches per lock-step round.
"""
import torch

from .. import _native
from .._capture import capture, side_streams
from .sghmc import _ChainSampler


class ChainGroup:
    WARMUP_STEPS = 3

    def __init__(self, samplers, use_graph=None):
        samplers = list(samplers)
        if not samplers or not all(isinstance(s, _ChainSampler) for s in samplers):
            raise TypeError('ChainGroup takes SGLD/SGHMC/cSGLD/cSGHMC sampler objects')
        if len({type(s) for s in samplers}) != 1 or len({id(s.train_loader) for s in samplers}) != 1:
            raise ValueError('the chains of a group must be of one sampler class and share one train loader')
        seeds = [s.optimizer.seed for s in samplers]
        if len(set(seeds)) != len(seeds):
            raise ValueError(f'chains of a group share a Philox key {seeds}: they would draw bit-identical noise on the '
                             'same minibatches. Construct them with distinct seed= (or util.set_random_seed between them)')
        if len({s.arena.n for s in samplers}) != 1:
            raise ValueError('the chains of a group must have equally sized arenas (one [K, n] slab, one launch)')
        self.samplers = samplers
        self.loader = samplers[0].train_loader
        self.device = torch.device(samplers[0].device)
        self.use_graph = (self.device.type == 'cuda') if use_graph is None else use_graph
        self.kernels = samplers[0].optimizer.kernels
        self._graph, self._static, self._warm, self._captured_with = None, None, 0, None
        self._graph_eps = False
        self.stats = dict(graph_replays=0, eager_rounds=0, captures=0, update_launches=0)
        # one slab per vector kind, one control-block array: chain k lives in row k
        K, n, dev = len(samplers), samplers[0].arena.n, samplers[0].arena.device
        self.theta, self.grad, self.mom = (torch.zeros(K, n, device=dev) for _ in range(3))
        self.ctl = torch.zeros(K * _native.CTL_BYTES, dtype=torch.uint8, device=dev)
        self.eps = None                                   # [K, n] injected noise (parity runs), allocated on demand
        for k, s in enumerate(samplers):
            s.arena.rehome(self.theta[k], self.grad[k], self.mom[k])
            s.optimizer.rehomed()
            s.optimizer.set_ctl_storage(self.ctl[k * _native.CTL_BYTES:(k + 1) * _native.CTL_BYTES])
            s.engine.invalidate()                         # a graph of the chain alone holds the old addresses

    def __len__(self):
        return len(self.samplers)

    # ---- one lock-step minibatch round ------------------------------------------------------
    def _update(self, keeps):
        """ONE launch for the K chains (+ their control-block advances), then the rare put-backs."""
        self.kernels.sgmcmc_step_multi(self.theta, self.grad, self.mom, self.ctl,
                                       eps=self.eps if self._graph_eps else None)
        self.stats['update_launches'] += 1
        if not self.samplers[0].optimizer.self_advance:       # (hosts that opt out of the self-advancing launch)
            self.kernels.step_ctl_advance(self.ctl)
        for s, keep in zip(self.samplers, keeps):
            s.engine.finish(keep)

    def _forward_backward(self, s, x, y):
        """The chains' forward / backward passes overlap on the device (parallel graph branches): K6's held form, which
        is for one such launch in flight at a time (csrc/ursa_bn.hip), is not taken inside a group of several chains."""
        if len(self.samplers) == 1:
            return s.engine.forward_backward(x, y)
        from .. import fused_bn, fused_block
        with fused_bn.several_streams(), fused_block.group_launches():
            return s.engine.forward_backward(x, y, wgrad_side=False)

    def _round_eager(self, x, y):
        keeps = [self._forward_backward(s, x, y) for s in self.samplers]
        with torch.no_grad():
            self._update(keeps)

    def _capture(self, x, y):
        self._static = (torch.empty_like(x), torch.empty_like(y))
        self._static[0].copy_(x)
        self._static[1].copy_(y)
        side = side_streams(self.device, len(self.samplers))
        g = torch.cuda.CUDAGraph()
        launches = self.stats['update_launches']
        with capture(g):
            cap = torch.cuda.current_stream(self.device)
            keeps = []
            for s, st in zip(self.samplers, side):          # fork: one forward/backward branch per chain
                st.wait_stream(cap)
                with torch.cuda.stream(st):
                    keeps.append(self._forward_backward(s, *self._static))
            for st in side:                                  # join
                cap.wait_stream(st)
            with torch.no_grad():
                self._update(keeps)                          # one update launch for all chains
        self.stats['update_launches'] = launches             # a capture records, it does not execute
        self._graph = g
        self._captured_with = self._device_state()
        self.stats['captures'] += 1

    def _device_state(self):
        """What a captured round bakes in per chain: the optimizer object and the addresses of its control
        block, schedule table and vectors."""
        ptr = lambda t: None if t is None else t.data_ptr()
        return [(id(s.optimizer), ptr(s.optimizer._ctl), ptr(s.optimizer._sched), ptr(s.arena.mom), ptr(s.arena.theta),
                 id(s.engine.gate_probe)) for s in self.samplers]

    def _run_epoch(self, plans):
        for s, (noise, sched) in zip(self.samplers, plans):
            s.model.train()
            s.engine.loss_acc.zero_()
            s.optimizer.ctl_begin(noise, sched)
        for k, s in enumerate(self.samplers):
            if s.optimizer._ctl.data_ptr() != self.ctl.data_ptr() + k * _native.CTL_BYTES:
                raise RuntimeError(f'chain {k}: its optimizer was replaced without adopting the group\'s control block '
                                   '(use the sampler\'s update_hyp)')
            if s.arena.theta.data_ptr() != self.theta[k].data_ptr() or s.arena.mom.data_ptr() != self.mom[k].data_ptr():
                raise RuntimeError(f'chain {k} no longer lives in this group\'s slabs (was its sampler put into another '
                                   'ChainGroup?): one launch over the slabs would step stale vectors')
        providers = [s._eps_for_epoch() for s in self.samplers]
        inject = any(p is not None for p in providers)
        if inject and not all(p is not None for p in providers):
            raise ValueError('either every chain of a group injects noise (eps_provider) or none does')
        if inject and self.eps is None:
            self.eps = torch.zeros_like(self.theta)
        if self._graph is not None and (self._device_state() != self._captured_with or self._graph_eps != inject):
            # update_hyp on a member rebuilt its optimizer (the captured round would step the old one), or the
            # captured update launch reads / does not read the injected-noise slab
            self._graph, self._captured_with = None, None
        self._graph_eps = inject
        gated = [(s.engine.gate_probe, s._gates_for_epoch()) for s in self.samplers if s.gate_provider is not None]
        full = getattr(self.loader, 'batch_size', None)
        seen = steps = 0
        for x, y in self.loader:
            x = x.to(self.device, non_blocking=True)
            y = y.to(self.device, non_blocking=True)
            if inject:
                for k, p in enumerate(providers):
                    self.eps[k].copy_(p(steps))
            for probe, gp in gated:                          # parity runs: the reference run's near-zero gate lists
                probe.load(gp(steps))
            if self.use_graph and x.shape[0] == full:
                if self._graph is None and self._warm >= self.WARMUP_STEPS:
                    self._capture(x, y)
                if self._graph is not None:
                    self._static[0].copy_(x)
                    self._static[1].copy_(y)
                    self._graph.replay()
                    self.stats['graph_replays'] += 1
                    self.stats['update_launches'] += 1
                else:
                    side = side_streams(self.device, 1)[0]
                    side.wait_stream(torch.cuda.current_stream(self.device))
                    with torch.cuda.stream(side):
                        self._round_eager(x, y)
                    torch.cuda.current_stream(self.device).wait_stream(side)
                    self._warm += 1
                    self.stats['eager_rounds'] += 1
            else:
                self._round_eager(x, y)
                self.stats['eager_rounds'] += 1
            for probe, _ in gated:
                probe.collect()
            seen += x.shape[0]
            steps += 1
        for s in self.samplers:
            s.optimizer.ctl_end(steps)
        return seen

    # ---- sampler-like surface -----------------------------------------------------------------
    def sample_iterative(self, val_loader=None, debug_val_loss=False, wandb_debug=False):
        """One posterior sample from EVERY chain (a list of K modules)."""
        gens = [s._epochs(val_loader, debug_val_loss, wandb_debug) for s in self.samplers]
        plans, done = [], 0
        for g in gens:
            plans.append(next(g))
        while True:
            seen = self._run_epoch(plans)
            plans, done = [], 0
            for g in gens:
                try:
                    plans.append(g.send(seen))
                except StopIteration:
                    done += 1
            if done == len(gens):
                break
            if done:
                raise RuntimeError('chains of a group fell out of step (different hyper-parameters?)')
        return [s._snapshot() for s in self.samplers]

    def sample(self, num_samples=None, **kw):
        """list over chains of list over samples, like K separate `sampler.sample()` calls."""
        if num_samples is None:
            s0 = self.samplers[0]
            num_samples = getattr(s0, 'num_samples', None) or s0.num_samples_per_cycle * s0.num_cycles
        per_chain = [[] for _ in self.samplers]
        for _ in range(num_samples):
            for k, m in enumerate(self.sample_iterative(**kw)):
                per_chain[k].append(m)
        return per_chain
