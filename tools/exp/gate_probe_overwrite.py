"""Chase: GateProbe.n_open came back holding float bit patterns when every epoch of the G16 replay ended with a small
zero-filled allocation + .item(). Replays seed 0 of G16 with one of several epoch-end actions and prints
the probe's counters per step.   python tools/exp/gate_probe_overwrite.py
Found so far (profiles/r04_gate_probe_overwrite.txt): any SMALL-pool allocation (32 KB; zero-filled or not, read or not) at
the end of every epoch makes one n_open entry of the LAST step come back as two float bit patterns (~4e-6, ~6e-5); a 4 MB
allocation does not; neither does the same run with an extra synchronize + re-read after every collect(), nor eager
launches with that re-read. The predictive stays within 1e-5 every time: the instrument's counter is hit, not the step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch
import test_gate_parity_gpu as T
from ursabench_amd.inference import engine as E

GOLD = os.path.join(ROOT, 'tests', 'golden')
orig = E.ChainEngine.run_epoch
ACTIONS = {
    'none': lambda dev: None,
    'zeros_item': lambda dev: torch.zeros(8193, dtype=torch.int32, device=dev)[33].item(),
    'zeros_only': lambda dev: torch.zeros(8193, dtype=torch.int32, device=dev),
    'empty_item': lambda dev: torch.empty(8193, dtype=torch.int32, device=dev)[33].item(),
    'zeros_sync': lambda dev: (torch.zeros(8193, dtype=torch.int32, device=dev), torch.cuda.synchronize()),
    'big_zeros_item': lambda dev: torch.zeros(1 << 20, dtype=torch.int32, device=dev)[33].item(),
}
MODES = [('graph', True), ('eager', False)]
ACTIONS = {k: ACTIONS[k] for k in ('none', 'zeros_only')}
orig_collect = None
from ursabench_amd import fused_bn as FB
_oc = FB.GateProbe.collect
def collect2(self):
    rec = _oc(self)
    a = self.n_open.cpu().numpy().copy()
    torch.cuda.synchronize()
    b = self.n_open.cpu().numpy().copy()
    if (np.abs(a) > 10 ** 9).any() or (a != b).any():
        print('   n_open read twice:', a.tolist(), b.tolist(), 'data_ptr', hex(self.n_open.data_ptr()), 'seen ptr', hex(self.seen.data_ptr()), flush=True)
    return rec
FB.GateProbe.collect = collect2
for name, act in [(n + '/' + m, a) for n, a in ACTIONS.items() for m, _ in MODES]:
    def patched(self, *a, _act=act, **k):
        r = orig(self, *a, **k)
        _act(self.device)
        return r
    E.ChainEngine.run_epoch = patched
    T._cache.pop((0, True, True, True), None)
    try:
        ug = name.endswith('graph')
        T._cache.pop((0, True, True, ug), None)
        r = T.replay(GOLD, 0, fused=True, force=True, use_graph=ug)
        print(name, [(st['step'], st['outside_band_changed'], st['flips'], round(st['err_proba'], 9)) for st in r['steps']], flush=True)
    except Exception as e:
        print(name, 'EXC', repr(e)[:200], flush=True)
E.ChainEngine.run_epoch = orig
