"""Many samplers / tasks in one process (what a hyper-optimisation loop does): does anything leak or crash as
hipGraphs accumulate and old ones are destroyed?  python tools/exp/graph_stress.py            (runs every mode in a child)
                                              python tools/exp/graph_stress.py <mode> [iters]"""
import gc, os, subprocess, sys, tempfile
MODES = ["full", "full_fresh_streams", "tasks", "engine", "update_hyp_loop", "group_update_hyp_loop"]
if len(sys.argv) == 1:
    for m in MODES:
        env = dict(os.environ, URSA_SIDE_STREAMS='fresh') if m == 'full_fresh_streams' else dict(os.environ)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), m, '30'], capture_output=True, text=True, timeout=900, env=env)
        last = [ln for ln in p.stdout.splitlines() if ln.startswith('it ')]
        print(f'{m:22s} rc={p.returncode:4d} last: {last[-1] if last else None}', flush=True)
    sys.exit(0)
os.environ.setdefault('MIOPEN_USER_DB_PATH', tempfile.mkdtemp(prefix='ursa_stress_'))
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ursabench_amd import inference, models, tasks, util
from ursabench_amd.data import synthetic
from ursabench_amd.tasks import task_base
mode, iters = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device('cuda', 0)
train = synthetic(512, (3, 32, 32), 10, seed=0, device=dev, batch_size=128)
test = synthetic(300, (3, 32, 32), 10, seed=1, device=dev, batch_size=128)
hyp = {'lr': 0.05, 'prior_std': 0.5, 'num_samples': 3, 'alpha': 0.5, 'burn_in_epochs': 0}
task_base.EnsembleAccumulator.EVAL_ROWS = 0      # one forward (and one captured graph) per loader-batch shape, as when the crash was found
if mode == 'tasks_lanes1':
    task_base.EnsembleAccumulator.LANES = 1
if mode == 'tasks_nobn':
    task_base._prefer_aten_batchnorm_in_eval = lambda m: None
keep = []


def make(it, graph=True):
    util.set_random_seed(it)
    s = inference.SGHMC(dict(hyp), models.PreResNet(10, 8).to(dev), train, device=dev, use_graph=graph and mode != 'full_sampler_nograph')
    return s, s.sample()


if mode.startswith('tasks'):
    s0, ens0 = make(0, graph=False)
    if mode == 'tasks_eager_members':
        import copy
        ens0 = [copy.deepcopy(m) for m in ens0]
        for m in ens0:
            del m._ursa_bank
if mode == 'full_gcdisabled':
    gc.disable()
if mode.endswith('update_hyp_loop'):
    # the reference's hyper-optimisation flow (hyper_optimization.py:51-73): ONE sampler and ONE task object,
    # update_hyp -> reset -> sample -> update_statistics per trial; every trial re-captures the step graph
    import random
    util.set_random_seed(0)
    if mode.startswith('group'):
        chains = [inference.SGHMC(dict(hyp), models.PreResNet(10, 8).to(dev), train, device=dev, seed=c) for c in range(3)]
        group = inference.ChainGroup(chains)
    else:
        s = inference.SGHMC(dict(hyp), models.PreResNet(10, 8).to(dev), train, device=dev)
    p = tasks.Prediction({'in_distribution_test': test}, 10, dev, ['nll'])
    for it in range(iters):
        h = dict(hyp, lr=random.uniform(0.01, 0.1), alpha=random.uniform(0.1, 0.9))
        if mode.startswith('group'):
            for c in chains:
                c.update_hyp(dict(h))
            ens = [m for ch in group.sample() for m in ch]
        else:
            s.update_hyp(dict(h))
            ens = s.sample()
        p.reset()
        v = p.update_statistics(ens, output_performance=True)
        torch.cuda.synchronize()
        print('it', it, 'ok nll %.4f' % v, 'alloc MB', torch.cuda.memory_allocated() >> 20, flush=True)
    sys.exit(0)
for it in range(iters):
    if mode == 'full_del_first' and it:
        torch.cuda.synchronize()
        del s, ens, p
    if mode.startswith('full') or mode == 'engine':
        s, ens = make(it)
    else:
        ens = ens0
    if mode != 'engine':
        for _ in range(1 if mode == 'full_one_pred' else 2):
            p = tasks.Prediction({'in_distribution_test': test}, 10, dev, 'ALL', acc_kw=dict(use_graph=False) if mode == 'full_tasks_nograph' else None)
            p.update_statistics(ens, output_performance=False)
    torch.cuda.synchronize()
    print('it', it, 'ok alloc MB', torch.cuda.memory_allocated() >> 20, flush=True)
    if mode == 'full_keepalive':
        keep.append((s, ens, p))
    if mode == 'full_keep_samplers':
        keep.append(s)
    if mode == 'full_keep_tasks':
        keep.append(p)
    if mode == 'full_gc':
        del s, ens, p
        gc.collect()
        torch.cuda.empty_cache()
