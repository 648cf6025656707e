"""CPU: bench.py's control flow — the JSON line survives a failing leg, and the N > 1 assembly
(value = world x K / max-over-ranks time, scaling weak, one all-reduce for the predictive) runs under
gloo at world size 2. `--dry-run-cpu` is the toy-size walk of the c2 flow on CPU tensors with the
tests' oracle kernel set; its line says it is not a measurement."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
        'vs_baseline', 'dtype', 'data', 'config', 'errors'}


def run_bench(args, world=1, launcher=True, tmp=None):
    """Runs bench.py (under torch.distributed.run for world > 1 unless launcher=False: bench.py then starts it itself) and
    returns (exit code, the ONE compact stdout line). The line must be the LAST stdout line, parse by itself and stay under
    8,000 bytes (the driver keeps a tail of stdout; round 4's 27.7 KB line went unparsed); the full record sits
    in the file the line names (`detail(line)`)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop('WORLD_SIZE', None), env.pop('RANK', None), env.pop('LOCAL_RANK', None)
    import tempfile
    args = args + ['--detail-out', os.path.join(tmp or tempfile.mkdtemp(prefix='ursa_bench_test_'), 'detail.json')]
    if world == 1 or not launcher:
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py')] + args
    else:
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr',
               '127.0.0.1', '--master-port', str(port), os.path.join(ROOT, 'bench.py')] + args
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, (p.stdout[-2000:], p.stderr[-2000:])
    last = p.stdout.rstrip('\n').splitlines()[-1]
    assert last == lines[0] and len(last.encode()) < 8000, len(last)
    return p.returncode, json.loads(last)


def detail(line):
    path = line['detail']
    with open(path if os.path.isabs(path) else os.path.join(ROOT, path)) as f:
        return json.load(f)


def test_dry_run_single_process():
    rc, line = run_bench(['--dry-run-cpu', '--steps', '2', '--warmup', '1'])
    assert rc == 0 and line['errors'] == {}
    assert KEYS <= set(line) and 'NOT a measurement' in line['data']
    assert line['n_gpus'] == 1 and line['steps'] == 2 and line['scaling'] == 'weak' and line['vs_baseline'] is None
    assert line['value'] > 0 and line['ms_per_step'] > 0 and line['bma_members'] == 2
    assert line['value'] == pytest.approx(2 / (line['ms_per_step'] * 2 / 1e3), rel=1e-3)
    assert line['config']['workload'].startswith('PreResNet-20')
    assert line['rccl']['world'] == 1 and line['rccl']['backend'] is None and line['rccl']['all_reduce_us'] is None


def test_dry_run_world_size_2_gloo():
    rc, line = run_bench(['--dry-run-cpu', '--gpus', '2', '--steps', '2', '--warmup', '0'], world=2)
    assert rc == 0 and line['errors'] == {}
    assert line['n_gpus'] == 2 and line['config']['chains'] == 2
    assert line['bma_members'] == 4                       # 2 members per rank, summed by the all-reduce
    # the line proves by itself what the process group was: backend, ranks that answered, one device record per rank
    r = line['rccl']
    assert r['backend'] == 'gloo' and r['world'] == 2 and r['ranks_seen'] == [0, 1]
    full = detail(line)
    assert KEYS <= set(full) and full['value'] == line['value'] and full['rccl']['ranks_seen'] == [0, 1]
    devs = full['rccl']['devices']
    assert len(devs) == 2 and {d['rank'] for d in devs} == {0, 1} and len({d['pid'] for d in devs}) == 2
    assert r['all_reduce_bytes'] == 4 * (96 * 10 + 96 + 1) and r['all_reduce_us'] > 0
    # whole-job aggregate: world x K samples over the max-over-ranks time
    assert line['value'] == pytest.approx(2 * 2 / (line['ms_per_step'] * 2 / 1e3), rel=1e-3)


@pytest.mark.parametrize('leg', ['bma', 'sampling'])
def test_a_failing_leg_keeps_the_line(leg):
    rc, line = run_bench(['--dry-run-cpu', '--steps', '1', '--warmup', '0', '--inject-failure', leg])
    assert rc != 0 and leg in line['errors'] and 'injected failure' in line['errors'][leg]
    assert KEYS <= set(line)
    if leg == 'bma':                                      # the timed sampling result survives a late failure
        assert line['value'] > 0 and line['ms_per_step'] > 0 and 'bma_preds_per_s' not in line
    else:
        assert line['value'] is None


def test_dry_run_c4_world_size_2_gloo():
    """BASELINE configs[3]'s N > 1 flow: rank 0 runs the SWAG trajectory, the moments are broadcast, every rank draws
    its own members and evaluates them, one all-reduce; toy WideResNet on CPU tensors."""
    rc, line = run_bench(['--dry-run-cpu', '--config', 'c4', '--gpus', '2', '--steps', '2', '--warmup', '1', '--c4-epochs', '2', '--c4-weak'], world=2)
    assert rc == 0 and line['errors'] == {}, line['errors']
    assert line['n_gpus'] == 2 and line['config']['members'] == 4 and line['bma_members'] == 4 and line['scaling'] == 'weak'
    assert line['unit'] == 'SWAG members/s' and line['value'] > 0 and 'NOT a measurement' in line['data']
    assert line['value'] == pytest.approx(2 * 2 / (line['ms_per_step'] * 2 / 1e3), rel=1e-3)


def test_dry_run_c4_shards_the_ensemble_world_size_4_gloo():
    """the config's members are SHARDED over the ranks (30 -> 8, 8, 7, 7 at world size 4), scaling strong,
    value = members / max-over-ranks time; the warm-up member of every rank is not part of the ensemble. Also 2 members on
    4 ranks: two ranks hold none and still take part in the one all-reduce."""
    rc, line = run_bench(['--dry-run-cpu', '--config', 'c4', '--gpus', '4', '--steps', '30', '--warmup', '1', '--c4-epochs', '2'], world=4)
    assert rc == 0 and line['errors'] == {}, line['errors']
    assert line['scaling'] == 'strong' and line['n_gpus'] == 4 and line['steps'] == 30
    assert line['config']['members'] == 30 and line['config']['members_on_rank0'] == 8 and line['bma_members'] == 30
    assert line['value'] == pytest.approx(30 / (line['ms_per_step'] * 30 / 1e3), rel=1e-3)
    rc, line = run_bench(['--dry-run-cpu', '--config', 'c4', '--gpus', '4', '--steps', '2', '--warmup', '0', '--c4-epochs', '2'], world=4)
    assert rc == 0 and line['errors'] == {}, line['errors']
    assert line['bma_members'] == 2 and line['config']['members_on_rank0'] == 1


@pytest.mark.parametrize('chains', [4, 2])
def test_dry_run_c5_world_size_4_gloo(chains):
    """BASELINE configs[4] on 4 ranks: chain c on rank c mod 4, no communication; with 2 chains two ranks hold none, skip
    their legs cleanly and only join the barriers (round 2: IndexError on those ranks). value counts every chain's proposals."""
    rc, line = run_bench(['--dry-run-cpu', '--config', 'c5', '--gpus', '4', '--steps', '2', '--warmup', '0', '--c5-chains', str(chains),
                          '--c5-L', '2'], world=4)
    assert rc == 0 and line['errors'] == {}, line['errors']
    assert line['n_gpus'] == 4 and line['config']['chains'] == chains and line['config']['ranks_without_a_chain'] == 4 - chains
    assert line['value'] == pytest.approx(chains * 2 / (line['ms_per_step'] * 2 / 1e3), rel=1e-3)
    assert 0 <= line['acceptance_rate_rank0'] <= 1


def test_a_leg_failing_on_another_rank_reaches_the_line():
    rc, line = run_bench(['--dry-run-cpu', '--gpus', '2', '--steps', '1', '--warmup', '0', '--inject-failure', 'bma@1:after'], world=2)
    assert rc != 0 and 'rank1:bma' in line['errors'] and 'injected failure' in line['errors']['rank1:bma']


def test_gpus_2_without_a_launcher_starts_one_itself():
    """the driver's N > 1 command may be plain `python bench.py --gpus N`. The parent must not touch the GPU; it
    starts torch.distributed.run as a child and relays rank 0's line and the exit code."""
    rc, line = run_bench(['--dry-run-cpu', '--gpus', '2', '--steps', '1', '--warmup', '0'], world=2, launcher=False)
    assert rc == 0 and line['errors'] == {}
    assert line['n_gpus'] == 2 and line['rccl']['world'] == 2 and line['rccl']['ranks_seen'] == [0, 1] and line['bma_members'] == 2
    rc, line = run_bench(['--dry-run-cpu', '--gpus', '2', '--steps', '1', '--warmup', '0', '--inject-failure', 'bma@1:after'], world=2, launcher=False)
    assert rc != 0 and 'rank1:bma' in line['errors']


def test_the_compact_line_is_bounded_whatever_the_legs_return():
    """compact_line() on a record as large as round 4's (27.7 KB: six parity trials, 40 kernel entries ...) and on one with huge
    error strings stays under the bound and keeps every contract key."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module('bench')
    big = json.load(open(os.path.join(ROOT, 'profiles', 'r04_bench_line_driver_cmd.json')))
    assert len(json.dumps(big)) > 20000
    out = bench.compact_line(big, 'gpurun_out/bench_detail_c2.json')
    assert len(json.dumps(out)) < 8000 and KEYS <= set(out) and out['value'] == big['value'] and out['detail']
    assert out['roofline']['frac'] == big['roofline']['frac'] and out['cpu_baseline']['cores'] == big['cpu_baseline']['cores']
    assert out['parity']['pass'] is True and 'trials' not in out['parity']
    big['errors'] = {f'rank{r}:leg': 'x' * 5000 for r in range(8)}
    big['roofline_kernels'] = {f'k{i}': {'frac': 0.5, 'us': 1.0} for i in range(2000)}
    out = bench.compact_line(big, 'd.json')
    assert len(json.dumps(out)) < 8000 and KEYS <= set(out) and len(out['errors']) == 8 and 'roofline' in out and 'cpu_baseline' in out
