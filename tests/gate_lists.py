"""Test infrastructure: the REFERENCE side of the ReLU-gate comparison (ursabench_amd.fused_bn.GateProbe is the
device side). Forward hooks on every BatchNorm module of a CPU model (the reference's own classes in
tools/gen_golden.py, our classes on host tensors in bench.py's parity leg - both run torch's CPU BatchNorm) record, per
training-mode call in call order, the pre-activations torch computed within `tau` of zero and the gate it took there.

Only tests/, tools/gen_golden.py and bench.py's `parity` leg import this; nothing under ursabench_amd/ does."""
import numpy as np
import torch

TAU = 1e-4      # band around zero: ~100x the MIOpen-vs-oneDNN convolution differences, ~1e-4 of the gates


class NearZeroGates:
    def __init__(self, model, tau=TAU):
        self.tau = tau
        self.calls = []
        self._handles = [m.register_forward_hook(self._hook) for m in model.modules()
                         if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]

    def _hook(self, mod, inputs, out):
        # fires on bn(x)'s result, before the in-place ReLU that follows it (preresnet.py:40-41)
        if not mod.training:
            return
        t = out.detach().reshape(-1)
        near = (t.abs() < self.tau).nonzero().flatten()
        self.calls.append(dict(idx=near.to(torch.int32).numpy().copy(), open=(t[near] > 0).to(torch.uint8).numpy().copy(),
                               n_open=int((t > 0).sum()), numel=t.numel()))

    def take(self):
        """The calls recorded since the last take(): one forward pass = one minibatch step."""
        calls, self.calls = self.calls, []
        return calls

    def remove(self):
        for h in self._handles:
            h.remove()

    def __deepcopy__(self, memo):
        # a deep copy of the hooked model (the samplers snapshot that way) must not drag a copy of the log along
        return self


def pack(steps):
    """steps: list (minibatch steps) of list (calls) of dicts -> flat arrays for an .npz fixture."""
    counts = np.array([[len(c['idx']) for c in calls] for calls in steps], np.int64)
    return dict(gate_idx=np.concatenate([c['idx'] for calls in steps for c in calls]).astype(np.int32),
                gate_open=np.concatenate([c['open'] for calls in steps for c in calls]).astype(np.uint8),
                gate_counts=counts,
                n_open=np.array([[c['n_open'] for c in calls] for calls in steps], np.int64),
                numel=np.array([c['numel'] for c in steps[0]], np.int64))


def unpack(g, prefix=''):
    """Inverse of pack(): list (steps) of list (calls) of (idx, open)."""
    counts = g[prefix + 'gate_counts']
    gi, go = g[prefix + 'gate_idx'], g[prefix + 'gate_open']
    out, off = [], 0
    for row in counts:
        calls = []
        for n in row:
            calls.append((gi[off:off + n], go[off:off + n]))
            off += int(n)
        out.append(calls)
    return out
