"""K1 through the control-block entry points at the workload's size: one PreResNet-20 chain (273,408 elements) and
K chains in one `ursa_sgmcmc_step_multi_f32` launch, per workgroup size (URSA_CTL_BLOCK is read once per process, so
every block size is its own process).
    python tools/k1_ctl_bench.py            # sweeps block sizes in child processes, writes gpurun_out/k1_ctl_bench.json
    python tools/k1_ctl_bench.py --block 128
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(block):
    if block:
        os.environ['URSA_CTL_BLOCK'] = str(block)
    import torch
    from ursabench_amd import _native
    from tools.kbench import timeit
    K = _native.knobs_kernels()        # the -DURSA_DEBUG_KNOBS build: the shipped library reads no environment
    n = 273408
    out = []
    for chains in (1, 2, 4, 8, 16):
        th, g, m = (torch.randn(chains, n, device='cuda') for _ in range(3))
        for mode in ('self_advance', 'no_advance', 'explicit_advance_launch'):
            fl = 0x1 | 0x8 | (0x20 if mode == 'self_advance' else 0)
            blocks = b''.join(bytes(_native.StepCtl(lr=0.1, mu=0.5, c_wd=8e-5, c_noise=0.3, n_train=50000.0, flags=fl,
                                                    seed=1 + k, step=0)) for k in range(chains))
            ctl = torch.frombuffer(bytearray(blocks), dtype=torch.uint8).cuda()

            def fn():
                K.sgmcmc_step_multi(th, g, m, ctl)
                if mode == 'explicit_advance_launch':
                    K.step_ctl_advance(ctl)
            med, best = timeit(fn, 30, batch=64)
            back = _native.StepCtl.from_buffer_copy(bytes(ctl.cpu().numpy())[:_native.CTL_BYTES])
            byt = 20 * n * chains
            out.append(dict(block=block or 'auto', chains=chains, mode=mode, elements=n * chains, median_us=round(med * 1e6, 3),
                            best_us=round(best * 1e6, 3), GBps_median=round(byt / med / 1e9, 1),
                            frac_of_8TBps=round(byt / med / 8e12, 4), ctl_step_after=back.step, tickets_clear=back.tickets_clear()))
            print(json.dumps(out[-1]), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--block', type=int, default=None)
    a = ap.parse_args()
    if a.block is not None:
        one(a.block)
        return
    res = []
    for b in (0, 256, 1024):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), '--block', str(b)], capture_output=True, text=True)
        sys.stderr.write(p.stderr[-2000:])
        for ln in p.stdout.splitlines():
            if ln.startswith('{'):
                res.append(json.loads(ln))
                print(ln, flush=True)
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, 'gpurun_out', 'k1_ctl_bench.json'), 'w'), indent=1)


if __name__ == '__main__':
    main()
