"""hipGraph capture with the Python garbage collector held off.

Every capture in this package uses capture_error_mode='thread_local' (RCCL's watchdog thread may call into the
HIP runtime while this thread captures). That mode also stops the runtime from rejecting *this* thread's
capture-unsafe calls — and the cyclic garbage collector can run at any allocation: if it fires inside a capture
and finalises an old sampler or task, that object's hipGraphExec and its private memory pool are destroyed in
the middle of the new capture. Measured on MI355X / ROCm 7.2 (tools/exp/graph_stress.py): a loop that builds a
sampler and two Prediction tasks per iteration — what a hyper-optimisation loop does — segfaulted in
hipGraphLaunch on the third iteration; with the collector paused during captures it runs indefinitely.
torch.cuda.graph() itself collects once on entry; this keeps it from collecting again until the capture ends."""
import contextlib
import gc

import torch


@contextlib.contextmanager
def capture(graph, **kw):
    """`with capture(g): ...` == `with torch.cuda.graph(g, capture_error_mode='thread_local'): ...` with the
    cyclic collector disabled for the duration (and restored to its previous state afterwards)."""
    kw.setdefault('capture_error_mode', 'thread_local')
    was_enabled = gc.isenabled()
    ctx = torch.cuda.graph(graph, **kw)
    ctx.__enter__()                      # synchronizes, gc.collect()s and empties the cache, then begins the capture
    gc.disable()
    try:
        yield graph
    except BaseException as e:           # noqa: BLE001
        if was_enabled:
            gc.enable()
        if not ctx.__exit__(type(e), e, e.__traceback__):
            raise
    else:
        try:
            ctx.__exit__(None, None, None)
        finally:
            if was_enabled:
                gc.enable()


_side = {}


def side_streams(device, n):
    """`n` side streams for forking a capture into parallel branches (or for eager warm-up off the default
    stream). A FIXED per-device set, created once and shared by every capture of the process, instead of
    `torch.cuda.Stream()` per capture: PyTorch hands those out round-robin from a pool of 32, so after a few
    captures a new graph forks onto streams that an older, still-live multi-branch graph was captured on.
    URSA_SIDE_STREAMS=fresh restores the per-capture behaviour (tools/exp/graph_stress.py)."""
    import os
    device = torch.device(device)
    if os.environ.get('URSA_SIDE_STREAMS') == 'fresh':
        return [torch.cuda.Stream(device) for _ in range(n)]
    have = _side.setdefault(device, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device))
    return have[:n]
