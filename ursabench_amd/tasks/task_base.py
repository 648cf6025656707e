"""_Task base (URSABench/tasks/task_base.py:4-20) + the device-resident ensemble evaluator all
three tasks share."""
import torch
import torch.distributed as dist

from .. import _native
from .._capture import capture, side_streams

GAMMA = 1e-4   # util.central_smoothing default (util.py:126)


def _prefer_aten_batchnorm_in_eval(module):
    """Inference-mode BatchNorm through MIOpen (`MIOpenBatchNormFwdInferSpatialEst`) takes 39 us per
    layer on a [128,16,32,32] fp32 activation — 58 % of a PreResNet-20 eval forward — against 6 us for
    ATen's own `batch_norm_transform_input_kernel` (tools/exp/bn_probe.py; training-mode BN is the
    other way round, so only eval is redirected). `torch.batch_norm` picks its backend from the
    process-global cudnn/MIOpen switch; `torch.native_batch_norm` IS the ATen kernel, so the eval
    forward of OUR twin's BatchNorm layers calls that directly: no global state is touched (RCCL
    watchdog / ChainGroup threads may be alive) and no private API is used.

    Only for networks whose BatchNorm layers run their own `forward`: the twin of a FOREIGN module, or of ours with
    K6 switched off (URSA_FUSED_BN=0 A/B runs). The networks of ursabench_amd.models call `fused_bn.bn_relu`, whose
    evaluation path is one K6 launch that never enters `BatchNorm.forward`: they are left untouched."""
    from torch.nn.modules.batchnorm import _BatchNorm
    from .. import fused_bn
    if fused_bn.enabled() and type(module).__module__ == 'ursabench_amd.models':
        return
    for m in module.modules():
        if isinstance(m, _BatchNorm) and not hasattr(m, '_ursa_bn_wrapped'):
            inner = m.forward

            def forward(x, _inner=inner, _m=m):
                if _m.training or _m.running_mean is None or not x.is_cuda:
                    return _inner(x)
                return torch.native_batch_norm(x, _m.weight, _m.bias, _m.running_mean, _m.running_var, False, 0.0,
                                               _m.eps)[0]
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
    HBM, the logits of all S members for the whole test set land in one [S, N, C] slab and ONE launch
    of `ursa_bma_accumulate_f32` folds them into device accumulators; the host sees the result once,
    after an optional all-reduce across ranks (one process per GPU, members sharded over ranks).
    """

    def __init__(self, loader, num_classes, device, kernels, smoothed, with_entropy=True, cost=None, use_graph=None,
                 use_twin=None):
        self.loader, self.C, self.device = loader, int(num_classes), torch.device(device)
        on_hip = self.device.type == 'cuda'
        self.use_graph = on_hip if use_graph is None else bool(use_graph)
        if self.use_graph and not on_hip:
            raise ValueError('hipGraph capture needs a HIP device')
        self.use_twin = on_hip if use_twin is None else bool(use_twin)
        self._twins = {}
        self.stats = dict(captures=0, twin_forwards=0, eager_forwards=0, bma_launches=0)
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
        forwards are captured once per (batch shape, lanes in use) into ONE hipGraph with parallel
        branches and replayed for every (batch, group of LANES members): small-network forwards leave
        most CUs idle, independent branches overlap (same effect as inference/chain_group.py). Members
        of several chains (several banks) get one twin per bank. Returns None for foreign modules
        (eager forward then). `use_twin` None: only on a HIP device (on a CPU tensor the row copy buys
        nothing); the CPU test-suite forces it on with use_graph=False to cover this logic."""
        if not self.use_twin:
            return None
        bank = getattr(member, '_ursa_bank', None)
        row = getattr(member, '_ursa_row', None)
        if bank is None or row is None or row.device != self.device:
            return None
        twin = self._twins.get(id(bank))
        if twin is None:
            rows, irows, mods = [], [], []
            for _ in range(self.LANES):
                lane_row, lane_irow = bank.new_row()
                lane_row.copy_(row)
                mod = bank.materialise(lane_row, lane_irow, member)
                mod.eval()
                _prefer_aten_batchnorm_in_eval(mod)
                rows.append(lane_row)
                irows.append(lane_irow)
                mods.append(mod)
            twin = self._twins[id(bank)] = dict(rows=rows, irows=irows, mods=mods, runners={}, inputs={}, bank=bank)
        return twin

    @staticmethod
    def _load_lane(twin, lane, member):
        """member -> lane: the flat row and the integer buffers (BatchNorm num_batches_tracked: unused
        by an eval forward, copied so that the lane IS the member)."""
        twin['rows'][lane].copy_(member._ursa_row)
        src = getattr(member, '_ursa_irow', None)
        if src is not None:
            for dst, s in zip(twin['irows'][lane], src):
                dst.copy_(s)

    def _twin_runner(self, twin, x, lanes):
        """(run, static input, per-lane outputs) for this batch shape and number of lanes in use;
        with use_graph the forwards are captured on first use and `run` replays the graph."""
        key = (tuple(x.shape), lanes)
        r = twin['runners'].get(key)
        if r is not None:
            return r
        mods = twin['mods'][:lanes]
        sx = twin['inputs'].get(tuple(x.shape))                  # one static input per batch shape, shared by
        if sx is None:                                           # the runners of every lane count
            sx = twin['inputs'][tuple(x.shape)] = torch.empty_like(x)
        sx.copy_(x)
        if not self.use_graph:
            outs = [m(sx) for m in mods]

            def run(_mods=mods, _sx=sx, _outs=outs):
                for m, o in zip(_mods, _outs):
                    o.copy_(m(_sx))
        else:
            cur = torch.cuda.current_stream(self.device)
            side = side_streams(self.device, len(mods))
            side[0].wait_stream(cur)
            with torch.cuda.stream(side[0]):
                for m in mods:
                    m(sx)                                        # warm-up outside capture (MIOpen search)
            cur.wait_stream(side[0])
            graph = torch.cuda.CUDAGraph()
            outs = []
            with capture(graph):
                cap = torch.cuda.current_stream(self.device)
                for m, st in zip(mods, side):                    # fork: one branch per lane
                    st.wait_stream(cap)
                    with torch.cuda.stream(st):
                        outs.append(m(sx))
                for st in side:                                  # join
                    cap.wait_stream(st)
            run = graph.replay
            self.stats['captures'] += 1
        r = twin['runners'][key] = (run, sx, outs)
        return r

    SLAB_BYTES = 1 << 30       # logits of one chunk of test rows, all members: [S, rows, C] fp32
    INPUT_BYTES = 2 << 30      # device copies of the chunk's input batches (host-resident loaders)

    EVAL_ROWS = 1024           # rows per member forward: consecutive loader batches are merged up to this many ...
    MERGE_MAX_PARAMS = 4_000_000   # ... for networks up to this many parameters
    EVAL_ROWS_SMALL = 16384    # ... and up to this many for networks of at most SMALL_PARAMS parameters (round 6, with the fused evaluation
    #                            units: the whole 10,000-row test set in one forward 36.5 k vs 34.7 k BMA-preds/s at 4,096 rows per forward,
    #                            profiles/r06_bma_lanes_rows_ab.json; rounds 2-5: 4,096 with MIOpen's convolutions - PreResNet-20: 660 k ->
    SMALL_PARAMS = 500_000     # 698 k member-forwards/s at 20 members, tools/exp/bma_probe.py; deeper / wider networks keep
                               # 1,024: their activations per row are what bounds the captured forward's memory)

    def _chunks(self, S, eval_rows=None):
        """The loader's batches grouped into chunks of consecutive rows: (first row, rows, [(offset, x on device), ...]).
        A chunk is as many batches as fit the slab / input budgets — the whole 10,000-row test set for every
        configuration in BASELINE.json — so the loader is walked ONCE per call, like the reference's loop
        (prediction.py:52), and a host-resident batch crosses PCIe once. Inside a chunk, consecutive loader
        batches are concatenated into evaluation batches of up to EVAL_ROWS rows: an eval-mode forward is
        row-independent, and a 128-row PreResNet-20 forward is ~100 kernels of a few microseconds each — 8x the
        rows per launch is 8x fewer launches for the same arithmetic. Measured (profiles/r02_bench_line.json,
        r02_c4_bench_line.json): PreResNet-20 280k -> 379k member-forwards/s; WideResNet-28-10 (36.5 M parameters,
        a 128-row forward already fills the GPU) 10.4k -> 9.1k — so `accumulate` merges only for networks of at
        most MERGE_MAX_PARAMS parameters (eval_rows=0: one forward per loader batch)."""
        eval_rows = self.EVAL_ROWS if eval_rows is None else eval_rows
        start, rows, nbytes, batches = 0, 0, 0, []
        per_row = 4 * S * self.C
        pend, pend_rows = [], 0

        def flush():
            nonlocal pend, pend_rows, rows
            if pend:
                x = pend[0] if len(pend) == 1 else torch.cat(pend)
                batches.append((rows, x))
                rows += pend_rows
                pend, pend_rows = [], 0

        for x, _ in self.loader:
            x = x.to(self.device, non_blocking=True)
            b, xb = len(x), x.numel() * x.element_size()
            if (batches or pend) and ((rows + pend_rows + b) * per_row > self.SLAB_BYTES or nbytes + xb > self.INPUT_BYTES):
                flush()
                yield start, rows, batches
                start, rows, nbytes, batches = start + rows, 0, 0, []
            if pend and pend_rows + b > max(eval_rows, 1):
                flush()
            pend.append(x)
            pend_rows += b
            nbytes += xb
            if pend_rows >= eval_rows:
                flush()
        flush()
        if batches:
            yield start, rows, batches

    def _param_count(self, member):
        bank = getattr(member, '_ursa_bank', None)
        if bank is not None:
            return bank.arena.num_parameters
        return sum(p.numel() for p in member.parameters())

    @torch.no_grad()
    def accumulate(self, members):
        """Fold `members` into the accumulators. Order of work: for every chunk of test rows (normally ONE: the
        whole test set), for every group of LANES bank-resident members: load the group into the twin's lanes
        once, then replay the forward graph over all the chunk's batches, the logits landing in the
        [S, rows, C] slab; then ONE launch of the BMA kernel folds the slab into the accumulators."""
        S = len(members)
        if S == 0:
            return
        for m in members:
            if getattr(m, '_ursa_row', None) is None:
                m.to(self.device)      # foreign (e.g. CPU deep copies from the reference's samplers): moved once
            m.eval()
        # plan: per twin, the member indices it serves in groups of LANES; the rest run eagerly
        by_twin, eager = {}, []
        for s, m in enumerate(members):
            twin = self._resident_twin(m)
            if twin is None:
                eager.append(s)
            else:
                by_twin.setdefault(id(twin), (twin, []))[1].append(s)
        biggest = max(self._param_count(m) for m in members)
        eval_rows = self.EVAL_ROWS if biggest <= self.MERGE_MAX_PARAMS else 0
        if eval_rows and biggest <= self.SMALL_PARAMS:
            eval_rows = self.EVAL_ROWS_SMALL
        for start, rows, batches in self._chunks(S, eval_rows):
            slab = self._slabs.get((S, rows))
            if slab is None:
                if len(self._slabs) >= 4:
                    self._slabs.clear()
                slab = self._slabs[(S, rows)] = torch.empty(S, rows, self.C, device=self.device)
            for twin, idxs in by_twin.values():
                for g0 in range(0, len(idxs), self.LANES):
                    group = idxs[g0:g0 + self.LANES]             # a partial group runs only its lanes
                    for lane, s in enumerate(group):
                        self._load_lane(twin, lane, members[s])
                    for off, x in batches:
                        b = len(x)
                        run, sx, outs = self._twin_runner(twin, x, len(group))
                        if outs[0].shape != (b, self.C):
                            raise ValueError(f'members return logits {tuple(outs[0].shape)}, expected {(b, self.C)}')
                        sx.copy_(x)
                        run()
                        for lane, s in enumerate(group):
                            slab[s, off:off + b].copy_(outs[lane])
                    self.stats['twin_forwards'] += len(group) * len(batches)
            for s in eager:
                for off, x in batches:
                    b = len(x)
                    z = members[s](x)
                    if z.shape != (b, self.C):
                        raise ValueError(f'member {s} returned logits {tuple(z.shape)}, expected {(b, self.C)}')
                    slab[s, off:off + b].copy_(z)
                self.stats['eager_forwards'] += len(batches)
            self.K.bma_accumulate(slab, self.proba[start:start + rows],
                                  None if self.ent is None else self.ent[start:start + rows],
                                  one_minus_gamma=1 - GAMMA, gamma_over_c=GAMMA * 1 / self.C, smoothed=self.smoothed,
                                  risk_sum=None if self.risk is None else self.risk[start:start + rows], cost=self.cost)
            self.stats['bma_launches'] += 1

    def local(self):
        """Host copies of this rank's accumulators (zeros right after construction / reset): no collective."""
        cpu = lambda t: None if t is None else t.cpu()
        return cpu(self.proba), cpu(self.ent), cpu(self.risk)

    def reduced(self, count, group=None):
        """(proba_sum, ent_sum, risk_sum, total member count) summed over ranks, as HOST tensors: ONE
        all-reduce of the concatenated fp32 accumulators (+ the member count in the last slot) over
        RCCL/xGMI (gloo on CPU), then ONE device-to-host copy. This is the path's only collective and the
        only place the host waits for the device; it is entered from update_statistics alone (never from
        a constructor or reset()), so every rank must call update_statistics the same number of times —
        with an empty member list if it holds none. `group=False`: this rank's own sums, no collective (a task
        built with process_group=False inside a job: tools/c3_partition_check.py compares the two)."""
        parts = [self.proba.reshape(-1)]
        if self.ent is not None:
            parts.append(self.ent)
        if self.risk is not None:
            parts.append(self.risk.reshape(-1))
        buf = torch.cat(parts + [torch.tensor([float(count)], device=self.device)])
        if group is not False and dist.is_available() and dist.is_initialized():   # also at world size 1: a no-op collective
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
        buf = buf.cpu()
        count = int(round(float(buf[-1])))
        nc = self.N * self.C
        proba = buf[:nc].view(self.N, self.C).clone()
        off = nc
        ent = risk = None
        if self.ent is not None:
            ent = buf[off:off + self.N].clone()
            off += self.N
        if self.risk is not None:
            risk = buf[off:off + nc].view(self.N, self.C).clone()
        return proba, ent, risk, count
