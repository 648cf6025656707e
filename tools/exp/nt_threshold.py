"""Where should the streaming kernels switch to non-temporal loads/stores? K1 / K2 / K3 timed at the
WideResNet-28-10 arena size (36.5 M elements: 438-731 MB per launch, around the 256 MiB Infinity Cache) and at
2^26, with the threshold forced low (always NT) and high (never NT). One child process per setting."""
import json, os, subprocess, sys
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from ursabench_amd import _native
    from tools.kbench import timeit
    K = _native.knobs_kernels()        # the -DURSA_DEBUG_KNOBS build: the shipped library reads no environment
    out = {}
    for n in (36546980 + (-36546980) % 4, 1 << 26, 1 << 24):
        th, g, m, snap = (torch.randn(n, device='cuda') for _ in range(4))
        sc = dict(lr=1e-3, c_wd=8e-5, c_noise=0.03, n_train=50000.0, seed=1)
        for name, bpe, fn in (('k1', 20, lambda: K.sgmcmc_step(th, g, m, mu=0.5, flags=0x1 | 0x8, step=3, **sc)),
                              ('k2', 20, lambda: K.swag_collect(th, m, g, decay=0.75, denom=4.0)),
                              ('k3', 12, lambda: K.swag_draw(snap, th, m, var_clamp=1e-30, seed=1, draw=2))):
            med, best = timeit(fn, 15, batch=8)
            out[f'{name}@{n}'] = round(bpe * n / med / 1e9, 1)
        del th, g, m, snap
    print('RESULT ' + json.dumps(out))
else:
    res = {}
    for name, mib in (('default_512MiB', None), ('always_nt', '1'), ('never_nt', '100000'), ('nt_above_256MiB', '256')):
        e = dict(os.environ)
        if mib:
            e['URSA_NT_MIB'] = mib
        p = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=e, capture_output=True, text=True, timeout=600)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith('RESULT ')]
        res[name] = json.loads(line[0][7:]) if line else {'error': p.stderr[-300:]}
        print(name, json.dumps(res[name]), flush=True)
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(res, open('gpurun_out/nt_threshold.json', 'w'), indent=1)
