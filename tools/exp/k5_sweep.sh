#!/bin/bash
for cfg in "50 10000 10|4,4 8,2 16,1" "30 10000 100|16,8 8,16 32,4 64,2" "30 10000 20|4,8 8,4 16,2 32,1" "30 10000 50|8,8 16,4 32,2 64,1" "8 10000 200|16,16 32,8 64,4" "8 10000 1000|64,16"; do
  shape=${cfg%%|*}; opts=${cfg##*|}
  for o in $opts; do
    echo -n "S,B,C=$shape force=$o: "
    URSA_BMA_FORCE=$o python - $shape <<'PY'
import sys, os, torch
sys.path.insert(0, os.getcwd())
from ursabench_amd import _native
K = _native.default_kernels()
S, B, C = (int(a) for a in sys.argv[1:4])
z = torch.randn(S, B, C, device='cuda') * 3
p, e = torch.zeros(B, C, device='cuda'), torch.zeros(B, device='cuda')
f = lambda: K.bma_accumulate(z, p, e, one_minus_gamma=0.9999, gamma_over_c=1e-4 / C, smoothed=False)
for _ in range(5): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    a.record(); f(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
byt = 4 * S * B * C + 8 * B * (C + 1)
print(f'{ts[len(ts)//2]:8.1f} us  {byt / ts[len(ts)//2] / 1e3:8.1f} GB/s')
PY
  done
done
