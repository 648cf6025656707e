"""Round 5 chase of the GateProbe.n_open overwrite (DESIGN.md §10 of round 4, profiles/r04_gate_probe_overwrite.txt): with a
small-pool allocation at the end of every epoch, one int64 open-gate counter of the last G16 replay came back holding two
float bit patterns. This tool puts the counters in the middle of a canary buffer and bisects WRITE side vs READ side:

  plain      the round-4 code and layout untouched (reproduction check)
  base       the round-4 code, counters inside a canary buffer: n_open[k].copy_((flat > 0).sum())  - inside a capture that is a D2D MEMCPY NODE from a
             graph-pool temporary
  sum_out    torch.sum(flat > 0, dim=0, keepdim=True, out=n_open[k:k+1]) - the reduction kernel writes the counter itself:
             no temporary, no memcpy node
  sync_read  base, with torch.cuda.synchronize() before collect() reads
  clone_read base, collect() reads a device-side clone made on the current stream

    python tools/exp/gate_probe_overwrite2.py [repeats]      -> gpurun_out/gate_probe_overwrite2.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import test_gate_parity_gpu as T
from ursabench_amd import fused_bn as FB
from ursabench_amd.inference import engine as E

GOLD = os.path.join(ROOT, 'tests', 'golden')
CANARY = 0x5A5A5A5A5A5A5A5A
GUARD = 64
REPEATS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
VARIANT = {'name': 'base'}
LOG = []

orig_init, orig_collect, orig_run_epoch = FB.GateProbe.__init__, FB.GateProbe.collect, E.ChainEngine.run_epoch
orig_observe = FB.GateProbe.observe


def init(self, n_calls, capacity, device, force):
    orig_init(self, n_calls, capacity, device, force)
    self._reads = []
    self._tmps = []
    self._ptmp = torch.zeros(n_calls, dtype=torch.int64, device=device) if VARIANT['name'] == 'plain_persist_tmp' else None
    if VARIANT['name'].startswith('plain'):     # the round-4 layout exactly: no canary buffer
        self._buf = None
        return
    self._buf = torch.full((n_calls + 2 * GUARD,), CANARY, dtype=torch.int64, device=device)
    self.n_open = self._buf[GUARD:GUARD + n_calls]
    self.n_open.zero_()


def observe(self, k, y):
    v = VARIANT['name']
    if v == 'plain_product':
        return orig_observe(self, k, y)
    flat = y.detach().reshape(-1)
    at = self.idx[k].clamp(max=flat.numel() - 1).long()
    self.seen[k].copy_(flat[at] > 0)
    if v in ('sum_out', 'plain_sum_out'):       # no temporary, no memcpy node
        torch.sum(flat > 0, dim=0, keepdim=True, out=self.n_open[k:k + 1])
    elif v == 'plain_kernel_copy':              # the same temporary, copied by a KERNEL instead of a memcpy node
        torch.add((flat > 0).sum().view(1), 0, out=self.n_open[k:k + 1])
    elif v == 'plain_product':                  # the product's observe() as it is now (rows of 4096 reduced per workgroup, then the row sums)
        return orig_observe(self, k, y)
    elif v == 'plain_two_stage_tmp':            # no multi-workgroup reduction, but still a graph-pool temporary + memcpy node
        part = (flat > 0).view(-1, 4096).sum(1)
        self.n_open[k].copy_(part.sum())
    elif v == 'plain_persist_tmp':              # the reduction writes a PERSISTENT temporary (default pool, allocated before the
        torch.sum(flat > 0, dim=0, keepdim=True, out=self._ptmp[k:k + 1])     # capture); the memcpy node copies persistent -> persistent
        self.n_open[k].copy_(self._ptmp[k])
    elif v == 'plain_hold_tmp':                 # the memcpy node stays, but its source is never freed inside the capture
        t = (flat > 0).sum()
        self._tmps.append(t)
        self.n_open[k].copy_(t)
    else:
        self.n_open[k].copy_((flat > 0).sum())


def collect(self):
    v = VARIANT['name']
    if self._buf is None:
        rec = orig_collect(self)                 # no read of our own: an extra synchronize + re-read made the effect go away in round 4
        raw = np.asarray(rec['n_open_as_reference'], np.int64)
        extra = {}
        if len(self._reads) == 3 and True:                # the LAST step (nothing follows that a further read could perturb): read again,
            a = self.n_open.cpu().numpy().copy()  # synchronize, read once more - does the device still hold the pattern?
            torch.cuda.synchronize()
            b = self.n_open.cpu().numpy().copy()
            extra = dict(second_read=[int(x) for x in a], third_read_after_sync=[int(x) for x in b])
            raw = a
            if self._tmps:
                held = [int(t) for t in self._tmps[-self.n_calls:]]          # the capture's own temporaries (graph pool), still alive
                extra['held_tmps_of_the_capture'] = held
                extra['tmp_ptrs'] = [hex(t.data_ptr()) for t in self._tmps[-self.n_calls:]]
                extra['n_open_ptr'] = hex(self.n_open.data_ptr())
                print('   held temporaries of the capture:', held, '| n_open now', [int(x) for x in b], flush=True)
                print('   tmp ptrs', extra['tmp_ptrs'], 'n_open ptr', extra['n_open_ptr'], flush=True)
                print('   pools:', [where_ptr(t.data_ptr()) for t in self._tmps[-self.n_calls:]], 'n_open:', where_ptr(self.n_open.data_ptr()), flush=True)
                for nm, ptr, th, cap in BWD_ALLOCS[-6:]:
                    print('   allocation made by', nm, 'thread', th, 'capturing', cap, '->', where_ptr(ptr), flush=True)
        bad = [int(i) for i in np.nonzero((raw < 0) | (raw > 10 ** 9))[0]]
        self._reads.append(dict(guards_ok=True, bad_entries=bad, raw=[int(x) for x in raw], bad_guard_words=[],
                                bad_as_floats=[[float(f) for f in np.array([raw[i]], np.int64).view(np.float32)] for i in bad],
                                rec_outside=rec['n_open_as_reference'], **extra))
        return rec
    if v == 'sync_read':
        torch.cuda.synchronize()
    if v == 'clone_read':
        whole = self._buf.clone().cpu().numpy()
    else:
        whole = self._buf.cpu().numpy().copy()
    rec = orig_collect(self)
    guards_ok = bool((whole[:GUARD] == CANARY).all() and (whole[-GUARD:] == CANARY).all())
    mid = whole[GUARD:-GUARD]
    bad = [int(i) for i in np.nonzero((mid < 0) | (mid > 10 ** 9))[0]]
    self._reads.append(dict(guards_ok=guards_ok, bad_entries=bad, raw=[int(x) for x in mid],
                            bad_as_floats=[[float(f) for f in np.array([mid[i]], np.int64).view(np.float32)] for i in bad],
                            bad_guard_words=[int(i) for i in np.nonzero(np.concatenate([whole[:GUARD], whole[-GUARD:]]) != CANARY)[0]]))
    return rec


def run_epoch(self, *a, **k):
    r = orig_run_epoch(self, *a, **k)
    torch.zeros(8193, dtype=torch.int32, device=self.device)        # the epoch-end small-pool allocation of the round-4 record
    return r


_cb = torch.cuda.CUDAGraph.capture_begin


def capture_begin(self, *a, **k):
    try:
        self.enable_debug_mode()
    except Exception as e:      # noqa: BLE001
        print('enable_debug_mode failed', repr(e))
    return _cb(self, *a, **k)


torch.cuda.CUDAGraph.capture_begin = capture_begin
_ecap = E.ChainEngine._capture
DOTS = []


def ecap(self, x, y):
    _ecap(self, x, y)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    path = os.path.join(ROOT, 'gpurun_out', f'step_graph_{len(DOTS)}.dot')
    try:
        self._graph.debug_dump(path)
        txt = open(path).read()
        import re
        edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
        outdeg, indeg = {}, {}
        for a_, b_ in edges:
            outdeg[a_] = outdeg.get(a_, 0) + 1
            indeg[b_] = indeg.get(b_, 0) + 1
        kinds = {kw: txt.count(kw) for kw in ('KERNEL', 'MEMCPY', 'MEMSET', 'EMPTY', 'EVENT')}
        print('   captured step graph:', len(edges), 'edges; nodes with >1 successors', sum(1 for v in outdeg.values() if v > 1),
              '; nodes with >1 predecessors', sum(1 for v in indeg.values() if v > 1), '; node kinds', kinds, flush=True)
        DOTS.append(path)
    except Exception as e:      # noqa: BLE001
        print('   debug_dump failed:', repr(e), flush=True)


E.ChainEngine._capture = ecap
FB.GateProbe.__init__, FB.GateProbe.observe, FB.GateProbe.collect = init, observe, collect
E.ChainEngine.run_epoch = run_epoch

import threading
BWD_ALLOCS = []


def where_ptr(ptr):
    """(pool id, segment address, block state) of the caching-allocator block that contains `ptr`."""
    for seg in torch.cuda.memory_snapshot():
        a = seg['address']
        if a <= ptr < a + seg['total_size']:
            off = a
            for blk in seg['blocks']:
                if off <= ptr < off + blk['size']:
                    return dict(pool=tuple(seg.get('segment_pool_id', ())), seg=hex(a), seg_kind=seg['segment_type'], block_state=blk['state'],
                                block_off=off - a, block_size=blk['size'])
                off += blk['size']
    return None


class _Spy(torch.autograd.Function):
    """Identity whose backward runs on the autograd engine's thread: allocates a small tensor there and records where it came from."""

    @staticmethod
    def forward(ctx, x):
        t = x.new_empty(8)
        BWD_ALLOCS.append(('forward', t.data_ptr(), threading.current_thread().name, torch.cuda.is_current_stream_capturing()))
        ctx._keep = t
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        t = g.new_empty(8)
        BWD_ALLOCS.append(('backward', t.data_ptr(), threading.current_thread().name, torch.cuda.is_current_stream_capturing()))
        ctx._keep2 = t
        return g


_fb = E.ChainEngine.forward_backward


import ursabench_amd.inference as INF
SAMPLERS = []
_sinit = INF.SGHMC.__init__


def sinit(self, *a, **k):
    _sinit(self, *a, **k)
    SAMPLERS.append(self)


INF.SGHMC.__init__ = sinit
_ecrit = E.ChainEngine.__init__


def einit(self, model, optimizer, loss_criterion, device, use_graph=None):
    crit = loss_criterion
    _ecrit(self, model, optimizer, (lambda logits, y: crit(_Spy.apply(logits), y)) if 'spy' in VARIANT.get('mods', '') else crit, device, use_graph)


E.ChainEngine.__init__ = einit


def where_is(f):
    """Exact matches of float f in the last sampler's flat buffers (theta / grad / momentum / BatchNorm statistics)."""
    s = SAMPLERS[-1]
    hits = {}
    for name in ('theta', 'grad', 'mom', 'fbuf'):
        t = getattr(s.arena, name, None)
        if t is None or f == 0.0:
            continue
        at = torch.nonzero(t == f).flatten()[:4].tolist()
        if at:
            hits[name] = at
    return hits


out = {}
for variant in (sys.argv[2].split(',') if len(sys.argv) > 2 else ('plain', 'base', 'sum_out', 'sync_read', 'clone_read', 'plain')):
    VARIANT['name'], _, mods = variant.partition('@')
    VARIANT['mods'] = mods
    FUSED, GRAPH = 'stock' not in mods, 'eager' not in mods          # name@stock: MIOpen's BatchNorm launches (observation only); name@eager: no hipGraph
    runs = []
    for rep in range(REPEATS):
        for sd in (0, 3):
            T._cache.pop((sd, FUSED, FUSED, GRAPH), None)
            probes = []
            old = FB.GateProbe.__init__

            def spy(self, *a, **k):
                init(self, *a, **k)
                probes.append(self)
            FB.GateProbe.__init__ = spy
            try:
                r = T.replay(GOLD, sd, fused=FUSED, force=FUSED, use_graph=GRAPH)
            finally:
                FB.GateProbe.__init__ = old
            reads = probes[-1]._reads
            runs.append(dict(seed=sd, rep=rep, engine=r['engine'],
                             steps=[dict(step=st['step'], outside_band_changed=st['outside_band_changed'], err=st['err_proba']) for st in r['steps']],
                             reads=[dict(guards_ok=x['guards_ok'], bad_entries=x['bad_entries'], bad_as_floats=x['bad_as_floats'],
                                         bad_guard_words=x['bad_guard_words']) for x in reads]))
            hit = [(i + 1, x['bad_entries'], x['bad_as_floats']) for i, x in enumerate(reads) if x['bad_entries'] or not x['guards_ok']]
            for x in reads:
                if x.get('second_read'):
                    print('   step-4 reads: collect', x.get('rec_outside'), '| again', x['second_read'], '| after sync', x['third_read_after_sync'], flush=True)
                for pair in x['bad_as_floats']:
                    print('   float pair', pair, 'found in', [where_is(f) for f in pair], flush=True)
            print(variant, 'seed', sd, 'rep', rep, 'hits (step, entries, floats):', hit, flush=True)
    key = variant if variant not in out else variant + '_again'
    out[key] = runs
os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, 'gpurun_out', 'gate_probe_overwrite2.json'), 'w'), indent=1)
print('written')
