"""Member-bank checkpoints: an ensemble is S flat rows `[theta | float buffers]` plus a manifest of
tensor names/shapes/offsets, so saving S posterior samples is one stacked tensor instead of S pickled
modules, and a saved ensemble can be re-materialised on any model with the same state_dict layout.
`to_state_dicts` bridges to the reference's intended per-sample `.pt` files
(URSABench/experiment.py:77-80).

Chain checkpoints (`save_chain` / `load_chain`): the reference has no resume for samplers — a long
SG-MCMC run that dies starts over. Here a chain's whole state is a few flat vectors and counters: theta / momentum /
BatchNorm buffers, the update counter (which IS the Philox call index: noise is counter-based, so the resumed chain draws
exactly the noise the uninterrupted one would have drawn), the optimizer's scalars, the LR scheduler and the sampler's
epoch bookkeeping. A chain resumed from a checkpoint continues bit-identically given the same minibatches and gradients
(asserted on CPU, where gradients are deterministic)."""
import torch

from .arena import FlatArena, MemberBank

FORMAT = 'ursabench-amd-member-bank-v1'


def _bank_of(members):
    bank = getattr(members[0], '_ursa_bank', None)
    if bank is None or any(getattr(m, '_ursa_bank', None) is not bank for m in members):
        raise ValueError('members must come from one sampler (one MemberBank)')
    return bank


def save_ensemble(members, path):
    bank = _bank_of(members)
    a = bank.arena
    rows = torch.stack([m._ursa_row for m in members]).cpu()
    ibufs = {k: torch.stack([dict(m.named_buffers())[k].cpu() for m in members]) for k, _ in a.ibufs}
    manifest = {'format': FORMAT,
                'params': [(n, list(s), o) for n, s, o in zip(a.param_names, a.layout.shapes, a.layout.offsets)],
                'param_width': a.layout.padded,
                'float_buffers': [(n, list(s), o) for n, s, o in zip(a.fbuf_layout.names, a.fbuf_layout.shapes,
                                                                     a.fbuf_layout.offsets)],
                'int_buffers': [k for k, _ in a.ibufs]}
    torch.save({'manifest': manifest, 'rows': rows, 'int_buffers': ibufs}, path)


def load_ensemble(path, like, device=None):
    """Re-materialise a saved ensemble as modules shaped like `like` (left untouched)."""
    import copy
    ck = torch.load(path)
    if ck['manifest']['format'] != FORMAT:
        raise ValueError(f'unknown checkpoint format {ck["manifest"]["format"]}')
    proto = copy.deepcopy(like)
    if device is not None:
        proto = proto.to(device)
    arena = FlatArena(proto.parameters(), module=proto)
    names = [n for n, _, _ in ck['manifest']['params']]
    if names != arena.param_names or ck['manifest']['param_width'] != arena.layout.padded:
        raise ValueError('checkpoint layout does not match the model')
    bank = MemberBank(arena)
    out = []
    for s in range(ck['rows'].shape[0]):
        row, irow = bank.new_row()
        row.copy_(ck['rows'][s])
        for dst, k in zip(irow, ck['manifest']['int_buffers']):
            dst.copy_(ck['int_buffers'][k][s])
        out.append(bank.materialise(row, irow, proto))
    return out


def to_state_dicts(members):
    """One ordinary `state_dict` (CPU clones) per member."""
    return [{k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for m in members]


CHAIN_FORMAT = 'ursabench-amd-chain-v1'
_SAMPLER_FIELDS = ('burnt_in', 'epochs_run', 'lr', 'lr_0', 'lr_final', '_draws')      # '_draws': SWAG's Philox draw index


def save_chain(sampler, path):
    """Everything SGLD / SGHMC / cSGLD / cSGHMC / SWA / SWAG needs to continue where it is (between two
    `sample_iterative` calls); for SWA / SWAG that includes the two moment vectors and the collected count."""
    a, opt = sampler.arena, sampler.optimizer
    cpu = lambda t: None if t is None else t.detach().cpu().clone()
    sched = getattr(sampler, 'optimizer_scheduler', None)
    torch.save({'format': CHAIN_FORMAT, 'kind': type(sampler).__name__, 'n': a.n, 'param_names': list(a.param_names),
                'theta': cpu(a.theta), 'mom': cpu(a.mom), 'fbuf': cpu(a.fbuf), 'ibufs': [cpu(b) for _, b in a.ibufs],
                'step': opt._step, 'has_mom': list(opt._has_mom), 'seed': opt.seed,
                'param_groups': [{k: v for k, v in g.items() if k != 'params'} for g in opt.param_groups],
                'scheduler': None if sched is None else sched.state_dict(),
                'swag': None if not hasattr(sampler, '_mean') else {'mean': cpu(sampler._mean), 'sq': cpu(sampler._sq),
                                                                   'collected': sampler.num_models_collected.clone(),
                                                                   # swag_model's BatchNorm step counters keep counting across members (util.py:195-199)
                                                                   'swag_ibufs': [cpu(b) for _, b in sampler.swag_arena.ibufs]},
                'fields': {k: getattr(sampler, k) for k in _SAMPLER_FIELDS if hasattr(sampler, k)}}, path)


def load_chain(sampler, path):
    """Put a freshly constructed sampler (same class, model architecture, loader, hyper-parameters) into the saved state."""
    ck = torch.load(path, weights_only=True)          # tensors, lists, dicts and scalars only: no pickled code runs
    if ck.get('format') != CHAIN_FORMAT:
        raise ValueError(f'unknown chain checkpoint format {ck.get("format")}')
    a, opt = sampler.arena, sampler.optimizer
    if ck['kind'] != type(sampler).__name__ or ck['n'] != a.n or ck['param_names'] != list(a.param_names):
        raise ValueError('chain checkpoint does not match this sampler (class / model layout)')
    # buffer layout: a silent zip() truncation would resume with partly stale BatchNorm counters / statistics
    if len(ck['ibufs']) != len(a.ibufs) or any(v.shape != b.shape for (_, b), v in zip(a.ibufs, ck['ibufs'])):
        raise ValueError(f'chain checkpoint holds {len(ck["ibufs"])} integer buffers, the model has {len(a.ibufs)} (or shapes differ)')
    if (ck['fbuf'] is None) != (a.fbuf is None) or (a.fbuf is not None and ck['fbuf'].numel() != a.fbuf.numel()):
        raise ValueError('chain checkpoint and model disagree on the floating-point buffers (BatchNorm running statistics)')
    if ck.get('swag') is not None:
        if not hasattr(sampler, 'swag_arena') or len(ck['swag']['swag_ibufs']) != len(sampler.swag_arena.ibufs):
            raise ValueError('chain checkpoint and sampler disagree on the SWAG model\'s integer buffers')
    with torch.no_grad():
        a.theta.copy_(ck['theta'])
        if ck['mom'] is not None:
            a.ensure_mom().copy_(ck['mom'])
        if ck['fbuf'] is not None and a.fbuf is not None:
            a.fbuf.copy_(ck['fbuf'])
        for (_, b), v in zip(a.ibufs, ck['ibufs']):
            b.copy_(v)
    opt.seed, opt._step = ck['seed'], ck['step']
    opt._has_mom = list(ck['has_mom'])
    for gi, has in enumerate(opt._has_mom):
        if has:
            opt._register_momentum_views(gi)
    for g, saved in zip(opt.param_groups, ck['param_groups']):
        g.update(saved)
    sched = getattr(sampler, 'optimizer_scheduler', None)
    if sched is not None and ck['scheduler'] is not None:
        sched.load_state_dict(ck['scheduler'])
    for k, v in ck['fields'].items():
        setattr(sampler, k, v)
    if ck.get('swag') is not None:
        with torch.no_grad():
            sampler._mean.copy_(ck['swag']['mean'])
            sampler._sq.copy_(ck['swag']['sq'])
        sampler.num_models_collected = ck['swag']['collected'].clone()
        with torch.no_grad():
            for (_, b), v in zip(sampler.swag_arena.ibufs, ck['swag']['swag_ibufs']):
                b.copy_(v)
        sampler._std = None
        if ck['fields'].get('burnt_in') and hasattr(sampler, 'adopt_moments'):
            sampler.adopt_moments()
    if hasattr(sampler, 'seed'):
        sampler.seed = ck['seed']
    sampler.engine.invalidate()            # a captured graph is still valid address-wise, but keep the contract simple
    return sampler
