"""Member-bank checkpoints: an ensemble is S flat rows `[theta | float buffers]` plus a manifest of
tensor names/shapes/offsets, so saving S posterior samples is one stacked tensor instead of S pickled
modules, and a saved ensemble can be re-materialised on any model with the same state_dict layout.
`to_state_dicts` bridges to the reference's intended per-sample `.pt` files
(URSABench/experiment.py:77-80)."""
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
