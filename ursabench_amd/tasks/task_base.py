"""_Task base (URSABench/tasks/task_base.py:4-20) + the device-resident ensemble evaluator all
three tasks share."""
import torch
import torch.distributed as dist

from .. import _native

GAMMA = 1e-4   # util.central_smoothing default (util.py:126)


def _prefer_aten_batchnorm_in_eval(module):
    """Inference-mode BatchNorm through MIOpen (`MIOpenBatchNormFwdInferSpatialEst`) takes 39 us per
    layer on a [128,16,32,32] fp32 activation — 58 % of a PreResNet-20 eval forward — against 6 us for
    ATen's own `batch_norm_transform_input_kernel` (tools/exp/bn_probe.py; training-mode BN is the
    other way round, so only eval is redirected). ATen picks the backend from the process-global
    cudnn/MIOpen switch, so it is flipped around each BatchNorm call of OUR twin module only (host
    side, i.e. at graph-capture time) and always restored."""
    from torch.nn.modules.batchnorm import _BatchNorm
    for m in module.modules():
        if isinstance(m, _BatchNorm) and not hasattr(m, '_ursa_bn_wrapped'):
            inner = m.forward

            def forward(x, _inner=inner, _m=m):
                if _m.training:
                    return _inner(x)
                # what torch.backends.cudnn.flags(enabled=False) does, minus its MIOpen "benchmark limit" warning
                prev = torch.backends.cudnn.enabled
                torch._C._set_cudnn_enabled(False)
                try:
                    return _inner(x)
                finally:
                    torch._C._set_cudnn_enabled(prev)
            m.forward = forward
            m._ursa_bn_wrapped = True


class _Task:
    def __init__(self, data_loader=None, num_classes=None, device=torch.device('cpu')):
        self.data_loader = data_loader
        self.num_classes = num_classes
        self.device = device

    def reset(self):
        raise NotImplementedError

    def update_statistics(self, model, output_performance=False):
        raise NotImplementedError

    def ensemble_update_statistics(self, model_list, output_performance=False):
        raise NotImplementedError

    def get_performance_metrics(self):
        raise NotImplementedError


def as_member_list(models):
    """prediction.py:38-48 — a module or a list of modules; anything else is NotImplementedError."""
    if isinstance(models, list):
        if not all(isinstance(m, torch.nn.Module) for m in models):
            raise NotImplementedError
        return models
    if isinstance(models, torch.nn.Module):
        return [models]
    raise NotImplementedError


class EnsembleAccumulator:
    """sum_s softmax / entropy / risk over ensemble members for one data loader.

    Reference loop (prediction.py:52-64): for every (batch, member): move ALL weights host->device,
    forward, softmax twice, two D2H copies, CPU `+=`, move the weights back. Here: members stay in
    HBM, the S logit blocks of a batch land in one [S, B, C] slab and ONE launch of
    `ursa_bma_accumulate_f32` folds them into device accumulators; the host sees the result once,
    after an optional all-reduce across ranks (one process per GPU, members sharded over ranks).
    """

    def __init__(self, loader, num_classes, device, kernels, smoothed, with_entropy=True, cost=None, use_graph=True):
        self.loader, self.C, self.device = loader, int(num_classes), torch.device(device)
        self.use_graph = use_graph
        self._twins = {}
        self.K = kernels if kernels is not None else _native.default_kernels()
        self.smoothed, self.cost = smoothed, cost
        self.N = len(loader.dataset)
        self.with_entropy = with_entropy
        self._slabs = {}
        self.reset(entropy_too=True)

    def reset(self, entropy_too=True):
        self.proba = torch.zeros(self.N, self.C, device=self.device)
        if entropy_too or not hasattr(self, 'ent'):
            self.ent = torch.zeros(self.N, device=self.device) if self.with_entropy else None
        self.risk = torch.zeros(self.N, self.C, device=self.device) if self.cost is not None else None

    # -- member forwards ---------------------------------------------------------------------
    LANES = 4      # member forwards evaluated concurrently (parallel branches of one hipGraph)

    def _resident_twin(self, member):
        """A member that came out of a MemberBank is evaluated through that bank's *twin*: LANES modules
        whose tensors view persistent rows (copy member row -> lane row, 1 MB device-to-device), so the
        forwards are captured once per batch shape into ONE hipGraph with LANES parallel branches and
        replayed for every (batch, group of LANES members): small-network forwards leave most CUs idle,
        independent branches overlap (same effect as inference/chain_group.py). Members of several chains
        (several banks) get one twin per bank. Returns None for foreign modules (eager forward then)."""
        if self.device.type != 'cuda' or not self.use_graph:
            return None
        bank = getattr(member, '_ursa_bank', None)
        if bank is None or getattr(member, '_ursa_row', None) is None:
            return None
        twin = self._twins.get(id(bank))
        if twin is None:
            rows, mods = [], []
            for _ in range(self.LANES):
                row, irow = bank.new_row()
                row.copy_(member._ursa_row)
                for dst, (k, _) in zip(irow, bank.arena.ibufs):
                    dst.copy_(dict(member.named_buffers())[k])
                mod = bank.materialise(row, irow, member)
                mod.eval()
                _prefer_aten_batchnorm_in_eval(mod)
                rows.append(row)
                mods.append(mod)
            twin = self._twins[id(bank)] = dict(rows=rows, mods=mods, graphs={}, bank=bank)
        return twin

    def _twin_graph(self, twin, x):
        """(graph, static input, per-lane outputs) for this batch shape; captured on first use."""
        key = tuple(x.shape)
        g = twin['graphs'].get(key)
        if g is None:
            sx = torch.empty_like(x)
            sx.copy_(x)
            cur = torch.cuda.current_stream(self.device)
            side = [torch.cuda.Stream(self.device) for _ in twin['mods']]
            side[0].wait_stream(cur)
            with torch.cuda.stream(side[0]):
                for m in twin['mods']:
                    m(sx)                                        # warm-up outside capture (MIOpen search)
            cur.wait_stream(side[0])
            graph = torch.cuda.CUDAGraph()
            outs = []
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                cap = torch.cuda.current_stream(self.device)
                for m, st in zip(twin['mods'], side):            # fork: one branch per lane
                    st.wait_stream(cap)
                    with torch.cuda.stream(st):
                        outs.append(m(sx))
                for st in side:                                  # join
                    cap.wait_stream(st)
            g = twin['graphs'][key] = (graph, sx, outs)
        return g

    @torch.no_grad()
    def accumulate(self, members):
        S = len(members)
        for m in members:
            m.to(self.device)          # no-op for bank-resident members; moves foreign CPU models once
            m.eval()
        # plan: per twin, the member indices it serves in groups of LANES; the rest run eagerly
        by_twin, eager = {}, []
        for s, m in enumerate(members):
            twin = self._resident_twin(m)
            if twin is None:
                eager.append(s)
            else:
                by_twin.setdefault(id(twin), (twin, []))[1].append(s)
        start = 0
        for x, _ in self.loader:
            b = len(x)
            x = x.to(self.device, non_blocking=True)
            slab = self._slabs.get((S, b))
            if slab is None:
                slab = self._slabs[(S, b)] = torch.empty(S, b, self.C, device=self.device)
            for twin, idxs in by_twin.values():
                graph, sx, outs = self._twin_graph(twin, x)
                if outs[0].shape != (b, self.C):
                    raise ValueError(f'members return logits {tuple(outs[0].shape)}, expected {(b, self.C)}')
                sx.copy_(x)
                for g0 in range(0, len(idxs), self.LANES):
                    group = idxs[g0:g0 + self.LANES]
                    for lane, s in enumerate(group):
                        twin['rows'][lane].copy_(members[s]._ursa_row)
                    graph.replay()                               # unused lanes recompute a stale member: ignored
                    for lane, s in enumerate(group):
                        slab[s].copy_(outs[lane])
            for s in eager:
                z = members[s](x)
                if z.shape != (b, self.C):
                    raise ValueError(f'member {s} returned logits {tuple(z.shape)}, expected {(b, self.C)}')
                slab[s].copy_(z)
            self.K.bma_accumulate(slab, self.proba[start:start + b],
                                  None if self.ent is None else self.ent[start:start + b],
                                  one_minus_gamma=1 - GAMMA, gamma_over_c=GAMMA * 1 / self.C, smoothed=self.smoothed,
                                  risk_sum=None if self.risk is None else self.risk[start:start + b], cost=self.cost)
            start += b

    def reduced(self, count, group=None):
        """(proba_sum, ent_sum, risk_sum, total member count) summed over ranks: ONE all-reduce of
        the concatenated fp32 accumulators over RCCL/xGMI (gloo on CPU), + the count."""
        parts = [self.proba.reshape(-1)]
        if self.ent is not None:
            parts.append(self.ent)
        if self.risk is not None:
            parts.append(self.risk.reshape(-1))
        if dist.is_available() and dist.is_initialized():       # also at world size 1: the collective is then a no-op
            buf = torch.cat(parts + [torch.tensor([float(count)], device=self.device)])
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            count = int(round(buf[-1].item()))
            buf = buf[:-1]
        else:
            buf = torch.cat(parts)
        nc = self.N * self.C
        proba = buf[:nc].view(self.N, self.C)
        off = nc
        ent = risk = None
        if self.ent is not None:
            ent = buf[off:off + self.N]
            off += self.N
        if self.risk is not None:
            risk = buf[off:off + nc].view(self.N, self.C)
        return proba, ent, risk, count
