"""Flat device-resident storage for a chain's parameters, gradients, momentum and BN buffers.

The reference updates every parameter tensor with ~8 tiny torch ops in a Python loop
(URSABench/inference/optim_sghmc.py:43-67), flattens to a CPU vector for SWAG
(URSABench/util.py:163-169, swa.py:81) and snapshots a posterior sample with
deepcopy(model.cpu()) (sghmc.py:99). Here the model's tensors are *views* into a few flat
fp32 buffers in HBM, so one kernel launch covers the whole chain, "flatten" is free, and a
snapshot is a device-to-device copy into a member-bank row.

Layout: tensor k of `model.parameters()` order starts at a multiple of ALIGN elements
(256 B); pads are zero-initialised and are never read by anyone (the flat kernels do update
them — they hold noise-only random walks — but no view covers them).
"""
import copy

import torch
import torch.nn as nn

ALIGN = 64   # elements (256 B): every tensor view is float4-aligned and cache-line aligned


def _round_up(x, a=ALIGN):
    return (x + a - 1) // a * a


class Layout:
    """Offsets of a list of named tensors inside a padded flat vector."""

    def __init__(self, named_shapes):
        self.names, self.shapes, self.offsets, self.numels = [], [], [], []
        off = 0
        for name, shape in named_shapes:
            n = 1
            for s in shape:
                n *= int(s)
            self.names.append(name)
            self.shapes.append(tuple(shape))
            self.offsets.append(off)
            self.numels.append(n)
            off = _round_up(off + n)
        self.padded = off                      # multiple of ALIGN (0 if empty)
        self.total = sum(self.numels)          # unpadded element count (the reference's P)
        self._index = None

    def views(self, flat):
        return [flat[o:o + n].view(s) for o, n, s in zip(self.offsets, self.numels, self.shapes)]

    def gather_index(self, device):
        """int64 index such that flat[index] is the reference's unpadded flatten() order."""
        if self._index is None or self._index.device != torch.device(device):
            parts = [torch.arange(o, o + n, device=device) for o, n in zip(self.offsets, self.numels)]
            self._index = torch.cat(parts) if parts else torch.zeros(0, dtype=torch.long, device=device)
        return self._index


class FlatArena:
    """theta / grad / (lazily) mom flat buffers for a list of parameters, plus the float and
    integer buffers of the owning module when one is given."""

    def __init__(self, params, module=None, device=None):
        params = list(params)
        if not params:
            raise ValueError('FlatArena needs at least one parameter')
        if any(p.dtype != torch.float32 for p in params):
            raise TypeError('the gfx950 kernels are fp32: every parameter must be float32')
        self.device = torch.device(device) if device is not None else params[0].device
        self.params = params
        self.layout = Layout((str(i), p.shape) for i, p in enumerate(params))
        n = self.layout.padded
        self.theta = torch.zeros(n, device=self.device)
        self.grad = torch.zeros(n, device=self.device)
        self.mom = None                                        # created by the optimizer when mu != 0
        with torch.no_grad():
            for p, v in zip(params, self.layout.views(self.theta)):
                v.copy_(p.detach())
                p.data = v                                     # the parameter now lives in the arena
        self.grad_views = self.layout.views(self.grad)
        for p, gv in zip(params, self.grad_views):
            if p.grad is not None:
                gv.copy_(p.grad)
            p.grad = gv                                        # autograd accumulates in place into the arena
        # buffers (BN running stats are part of a posterior sample: sghmc.py:99 deep-copies them too)
        self.module = module
        self.fbuf_layout, self.fbuf, self.ibufs = Layout([]), None, []
        if module is not None:
            fl = [(k, b) for k, b in module.named_buffers() if b is not None and b.dtype == torch.float32]
            self.ibufs = [(k, b) for k, b in module.named_buffers() if b is not None and b.dtype != torch.float32]
            self.fbuf_layout = Layout((k, b.shape) for k, b in fl)
            self.fbuf = torch.zeros(self.fbuf_layout.padded, device=self.device)
            self._fbuf_tensors = [b for _, b in fl]
            with torch.no_grad():
                for (k, b), v in zip(fl, self.fbuf_layout.views(self.fbuf)):
                    v.copy_(b)
                    b.data = v
            # names by parameter identity (the optimizer may have been given any subset / order of
            # module.parameters()); a tied parameter has several names: all of them -> its one slot
            names_of = {}
            for k, p in module.named_parameters(remove_duplicate=False):
                names_of.setdefault(id(p), []).append(k)
            missing = [i for i, p in enumerate(params) if id(p) not in names_of]
            if missing:
                raise ValueError(f'parameters {missing} of the arena do not belong to the module given as module=')
            self.param_names = [names_of[id(p)][0] for p in params]
            self.param_aliases = {alias: names_of[id(p)][0] for p in params for alias in names_of[id(p)][1:]}
        else:
            self.param_names = [str(i) for i in range(len(params))]
            self.param_aliases = {}

    # -- reference-compatible flat vectors -------------------------------------------------
    @property
    def n(self):
        return self.layout.padded

    @property
    def num_parameters(self):
        return self.layout.total

    def ensure_mom(self):
        if self.mom is None:
            self.mom = torch.zeros_like(self.theta)
        return self.mom

    def rehome(self, theta, grad, mom):
        """Move the chain's vectors into caller-provided storage (rows of a [K, n] slab that several chains of one
        GPU share, so ONE update launch covers them all): contents are copied, parameters / gradients become views
        of the new rows. Anything that cached views of the old buffers (captured graphs, momentum_buffer state
        entries) must be rebuilt by the caller."""
        n = self.layout.padded
        for name, t in (('theta', theta), ('grad', grad), ('mom', mom)):
            if not (t.dim() == 1 and t.numel() == n and t.is_contiguous() and t.dtype == torch.float32
                    and t.device == self.theta.device):
                raise ValueError(f'rehome: {name} must be a contiguous float32 [{n}] tensor on {self.theta.device}')
        with torch.no_grad():
            theta.copy_(self.theta)
            grad.copy_(self.grad)
            if self.mom is not None:
                mom.copy_(self.mom)
            else:
                mom.zero_()
            bound = [p.grad is not None and p.grad.data_ptr() == gv.data_ptr() for p, gv in zip(self.params, self.grad_views)]
            self.theta, self.grad, self.mom = theta, grad, mom
            for p, v in zip(self.params, self.layout.views(theta)):
                p.data = v
            self.grad_views = self.layout.views(grad)
            for p, gv, b in zip(self.params, self.grad_views, bound):
                if b:
                    p.grad = gv

    def flatten(self, which='theta'):
        """Unpadded flat copy in model.parameters() order == URSABench/util.py:163-169 flatten()."""
        src = getattr(self, which)
        return src[self.layout.gather_index(self.device)]

    def load_flat(self, vec, which='theta'):
        """Inverse of flatten(): util.set_weights (util.py:172-176) in one indexed store."""
        getattr(self, which)[self.layout.gather_index(self.device)] = vec.to(self.device, torch.float32)

    def rebind(self):
        """Re-point parameters / grads / buffers at the arena if something replaced them
        (optimizer.zero_grad() sets .grad to None and autograd then hands out fresh tensors; module.to(),
        load_state_dict keep data in place but a user may assign p.grad). Values found outside the arena are copied
        in — the gradients with ONE multi-tensor copy."""
        with torch.no_grad():
            src, dst = [], []
            for p, v, gv in zip(self.params, self.layout.views(self.theta), self.grad_views):
                if p.data_ptr() != v.data_ptr():
                    v.copy_(p.detach())
                    p.data = v
                if p.grad is None:
                    gv.zero_()               # "no gradient": never re-apply the previous step's (optim_sghmc.py:44-45)
                    p.grad = gv
                elif p.grad.data_ptr() != gv.data_ptr():
                    src.append(p.grad)
                    dst.append(gv)
                    p.grad = gv
            if src:
                torch._foreach_copy_(dst, src)

    def stash(self, indices):
        """Copies of the theta (and momentum) slices of the given tensors, to undo a flat update on them."""
        L = self.layout
        spans = [(L.offsets[i], L.offsets[i] + L.numels[i]) for i in indices]
        return [(lo, hi, self.theta[lo:hi].clone(), None if self.mom is None else self.mom[lo:hi].clone())
                for lo, hi in spans]

    def unstash(self, kept, snapshot=None):
        for lo, hi, th, mo in kept:
            self.theta[lo:hi].copy_(th)
            if mo is not None:
                self.mom[lo:hi].copy_(mo)
            if snapshot is not None:
                snapshot[lo:hi].copy_(th)

    def grads_bound(self):
        return all(p.grad is not None and p.grad.data_ptr() == gv.data_ptr()
                   for p, gv in zip(self.params, self.grad_views))


class MemberBank:
    """Device-resident posterior samples. A member is one flat row
    [theta (padded) | float buffers (padded)] plus copies of the integer buffers (BatchNorm
    num_batches_tracked), exposed as an nn.Module whose tensors are views of that row.
    Replaces the reference's CPU deep copies (sghmc.py:99, csghmc.py:109, swag.py:125).
    The bank keeps no reference to the rows: a member lives as long as its module does."""

    def __init__(self, arena):
        self.arena = arena
        self.width = arena.layout.padded + arena.fbuf_layout.padded
        self._skeleton = None

    def new_row(self):
        a = self.arena
        return torch.empty(self.width, device=a.device), [torch.empty_like(b) for _, b in a.ibufs]

    def theta_of(self, row):
        return row[:self.arena.layout.padded]

    def store(self, row, irow, theta_already_written=False):
        """Snapshot the live chain into `row` (device-to-device; no host sync)."""
        a = self.arena
        with torch.no_grad():
            if not theta_already_written:
                self.theta_of(row).copy_(a.theta)
            if a.fbuf is not None and a.fbuf.numel():
                row[a.layout.padded:].copy_(a.fbuf)
            for dst, (_, b) in zip(irow, a.ibufs):
                dst.copy_(b)

    def materialise(self, row, irow, like):
        """An nn.Module whose parameters and buffers are views of `row` (no copy). `like` is the
        live module; its structure is cloned on the meta device once."""
        if self._skeleton is None:
            self._skeleton = copy.deepcopy(like).to('meta')
        m = copy.deepcopy(self._skeleton)
        a = self.arena
        pviews = dict(zip(a.param_names, a.layout.views(self.theta_of(row))))
        for alias, canonical in a.param_aliases.items():                 # tied weights: every alias -> one view
            pviews[alias] = pviews[canonical]
        fviews = dict(zip(a.fbuf_layout.names, a.fbuf_layout.views(row[a.layout.padded:])))
        iviews = {k: t for t, (k, _) in zip(irow, a.ibufs)}
        req = {k: p.requires_grad for k, p in like.named_parameters()}
        for prefix, mod in m.named_modules():
            for name in list(mod._parameters):
                if mod._parameters[name] is None:
                    continue
                full = f'{prefix}.{name}' if prefix else name
                mod._parameters[name] = nn.Parameter(pviews[full], requires_grad=req.get(full, True))
            for name in list(mod._buffers):
                if mod._buffers[name] is None:
                    continue
                full = f'{prefix}.{name}' if prefix else name
                mod._buffers[name] = fviews[full] if full in fviews else iviews[full]
        m.train(like.training)
        m._ursa_row = row            # flat handle for hosts that want the member without the module
        m._ursa_irow = irow
        m._ursa_bank = self
        return m

    def snapshot(self, like):
        row, irow = self.new_row()
        self.store(row, irow)
        return self.materialise(row, irow, like)
