"""11.6 % of a PreResNet-20 SGHMC step is `batched_transpose_*` — MIOpen's NCHW<->NHWC round
trips around its `igemm_wrw...nhwc` weight-gradient solver. A/B of MIOpen solver switches (environment variables,
so one child process per setting): graph-replayed engine step time + the transposes' share of kernel time.
    python tools/exp/wrw_solver_ab.py            # parent: runs every setting
"""
import json, os, subprocess, sys, tempfile
SETTINGS = {
    'baseline': {},
    'no_asm_wrw_nhwc': {'MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC': '0'},
    'no_asm_nhwc_all': {'MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_WRW_GTC_XDLOPS_NHWC': '0', 'MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC': '0',
                        'MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC': '0'},
    'no_implicit_gemm': {'MIOPEN_DEBUG_CONV_IMPLICIT_GEMM': '0'},
    'find_mode_normal': {'MIOPEN_FIND_MODE': '1'},
    'find_enforce_search': {'MIOPEN_FIND_MODE': '1', 'MIOPEN_FIND_ENFORCE': '3'},
}


def child():
    import time
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from ursabench_amd import inference, models, util
    from ursabench_amd.data import synthetic
    dev = torch.device('cuda', 0)
    util.set_random_seed(0)
    train = synthetic(128 * 120, (3, 32, 32), 10, seed=0, device=dev, batch_size=128)
    s = inference.SGHMC({'lr': 0.1, 'prior_std': 0.5, 'num_samples': 3, 'alpha': 0.5, 'burn_in_epochs': 0},
                        models.PreResNet(10, 20).to(dev), train, device=dev)
    s.sample_iterative()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s.sample_iterative()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 120 * 1e3
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        s.sample_iterative()
        torch.cuda.synchronize()
    ev = prof.key_averages()
    tot = sum(e.device_time_total for e in ev)
    share = lambda pat: round(100 * sum(e.device_time_total for e in ev if pat in e.key) / tot, 2)
    top = [(e.key[:60], round(100 * e.device_time_total / tot, 1)) for e in sorted(ev, key=lambda e: -e.device_time_total)[:6]]
    print('RESULT ' + json.dumps({'ms_per_step': round(ms, 4), 'transpose_pct': share('batched_transpose'), 'wrw_pct': share('wrw'),
                                 'kernel_ms_per_step': round(tot / 120 / 1e3, 4), 'top': top}))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child()
    else:
        res = {}
        for name, env in SETTINGS.items():
            e = dict(os.environ, MIOPEN_USER_DB_PATH=tempfile.mkdtemp(prefix='ursa_wrw_'), **env)
            p = subprocess.run([sys.executable, os.path.abspath(__file__), 'child'], env=e, capture_output=True, text=True, timeout=600)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith('RESULT ')]
            res[name] = json.loads(line[0][7:]) if line else {'error': (p.stderr or p.stdout)[-400:]}
            res[name]['env'] = env
            print(name, json.dumps(res[name]), flush=True)
        os.makedirs('gpurun_out', exist_ok=True)
        json.dump(res, open('gpurun_out/wrw_solver_ab.json', 'w'), indent=1)
