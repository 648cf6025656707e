"""K4's four launch forms at 2^26 elements (HBM-bound) and at PreResNet-164's 1,726,388 (cache-resident): us per launch and
fraction of 8 TB/s on the algorithmic bytes, the way bench.py's roofline_kernels leg times them (HIP events, median of 5 batches).
    python tools/exp/k4_forms_ab.py   -> one JSON line"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from bench import event_time_ms, HBM_PEAK_GBPS
from ursabench_amd import _native
import glob
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream()
LIBS = [('shipped', None)] + [(os.path.basename(p)[len('libursa_hip_'):-3], p) for p in sorted(glob.glob(os.path.join(ROOT, 'tools/exp/_libs/libursa_hip_leapu*.so')))]
res = {}
for tag, path in LIBS:
  K = _native.default_kernels() if path is None else _native.HipKernels(_native.load_library(path))
  out = res[tag] = {}
  ws, acc = torch.zeros(_native.REDUCE_WS_FLOATS, device=dev), torch.zeros(1, device=dev)
  KD = _native.LEAP_KICK | _native.LEAP_DRIFT
  for label, n, resident in (('1.73M', 1726388 + (-1726388) % 64, True), ('2^26', 1 << 26, False)):
      th, p, g = (torch.randn(n, device=dev) for _ in range(3))
      forms = {'kick_drift': (20, lambda: K.leapfrog(th, p, g, kick_coef=1e-4, step_size=2e-4, inv_mass=1.0, flags=KD)),
               'kick': (12, lambda: K.leapfrog(None, p, g, kick_coef=1e-4, step_size=2e-4, inv_mass=1.0, flags=_native.LEAP_KICK)),
               'kick_kinetic': (12, lambda: K.leapfrog(None, p, g, kick_coef=-1e-4, step_size=2e-4, inv_mass=1.0, flags=_native.LEAP_KICK, kinetic_out=acc, ws=ws)),
               'kinetic_only': (4, lambda: K.leapfrog(None, p, None, kick_coef=0.0, step_size=0.0, inv_mass=1.0, flags=0, kinetic_out=acc, ws=ws)),
               'sumsq': (4, lambda: K.sumsq(th, acc, ws))}
      for form, (bpe, fn) in forms.items():
          if resident:
              b = sorted(event_time_ms(fn, 1024, stream, graph_batch=128) for _ in range(5))
          else:
              b = sorted(event_time_ms(fn, 10, stream) for _ in range(5))
          out[f'{form}_{label}'] = dict(us=round(b[2] * 1e3, 3), frac=round(bpe * n / (b[2] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), us_batches=[round(x * 1e3, 2) for x in b])
print(json.dumps(res))
