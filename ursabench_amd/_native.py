"""ctypes binding of the gfx950 kernel library (include/ursa_hip.h).

This is the only way the package computes an MCMC update or a BMA reduction. There is no CPU
or eager-PyTorch fallback: if ``csrc/libursa_hip.so`` is missing, or a tensor is not a
contiguous fp32 CUDA(HIP) tensor, the call raises.

torch is imported *before* the library is loaded on purpose: PyTorch-ROCm bundles its own
``libamdhip64.so`` (SONAME ``libamdhip64.so.7``); loading ours afterwards makes the dynamic
loader resolve our NEEDED entry to the runtime torch already initialised, so the stream
handles torch hands us are valid in the runtime our launches go through.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'csrc', 'libursa_hip.so')
KNOBS_LIB_PATH = os.path.join(_HERE, 'csrc', 'libursa_hip_knobs.so')   # -DURSA_DEBUG_KNOBS build: tests / tools only

ABI_VERSION = 8

# flags (mirror include/ursa_hip.h)
STEP_NOISE, STEP_FIRST, STEP_ZERO_GRAD, STEP_WD, STEP_SGD, STEP_ADVANCE = 0x1, 0x2, 0x4, 0x8, 0x10, 0x20
BMA_SMOOTHED = 0x1
LEAP_KICK, LEAP_DRIFT = 0x1, 0x2
REDUCE_WS_FLOATS = 2048
BN_RELU, BN_TWO_LAUNCH, BN_HELD = 0x1, 0x2, 0x4
CONV_FLIP, CONV_STRIDE2 = 0x1, 0x2
PREACT_BN, PREACT_STATS, PREACT_ADD, PREACT_BNBWD, PREACT_EVAL = 0x10, 0x20, 0x40, 0x80, 0x100


def bn_ws_floats(channels):
    """URSA_BN_WS_FLOATS(C): scratch of one BatchNorm call (partial sums, then the held form's sync words)."""
    return int(channels) * 64 * 8 + (int(channels) + 2) * 32


BN_HELD_MIN_BYTES = 24 << 20      # activations from this size on may take the held form (ursa_bn.hip kHeldMinFloat4Bwd) ...
BN_HELD_MIN_BYTES_FWD = 48 << 20  # ... the forward from this size on (kHeldMinFloat4Fwd)
BMA_MAX_CLASSES = 1024

_vp, _i64, _i32, _u64, _u32, _f = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint64,
                                   ctypes.c_uint32, ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)

#: every symbol include/ursa_hip.h declares, with its ctypes signature
SIGNATURES = {
    'ursa_abi_version': (ctypes.c_int, []),
    'ursa_strerror': (ctypes.c_char_p, [ctypes.c_int]),
    'ursa_sgmcmc_step_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _u64, _u64, _u32, _vp]),
    'ursa_sgmcmc_step_ctl_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp]),
    'ursa_step_ctl_advance': (ctypes.c_int, [_vp, _i32, _vp]),
    'ursa_sgmcmc_step_multi_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp]),
    'ursa_philox_normal_f32': (ctypes.c_int, [_vp, _i64, _u64, _u64, _vp]),
    'ursa_selftest_rng_f32': (ctypes.c_int, [_vp, _vp]),
    'ursa_swag_collect_f32': (ctypes.c_int, [_vp, _vp, _vp, _i64, _f, _f, _vp]),
    'ursa_swag_draw_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _f, _f, _u64, _u64, _vp]),
    'ursa_swag_std_f32': (ctypes.c_int, [_vp, _vp, _vp, _i64, _f, _f, _vp]),
    'ursa_swag_draw_std_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _u64, _u64, _vp]),
    'ursa_bma_accumulate_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _f, _f, _u32, _vp]),
    'ursa_leapfrog_f32': (ctypes.c_int, [_vp, _vp, _vp, _i64, _f, _f, _f, _u32, _vp, _vp, _vp]),
    'ursa_sumsq_f32': (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp]),
    'ursa_bn_relu_fwd_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f, _f, _u32, _vp]),
    'ursa_bn_relu_eval_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f, _u32, _vp]),
    'ursa_bn_relu_bwd_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _u32, _vp]),
    'ursa_bn_relu_bwd_gated_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _u32,
                                                  _vp, _vp, _i64, _vp]),
    'ursa_conv_wgrad_ws_floats': (_i64, [_i64, _i64, _i64, _i64, _i64, _i32, _i32]),
    'ursa_conv_wgrad_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _vp]),
    'ursa_conv_wgrad_partial_f32': (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _i32, _vp]),
    'ursa_conv_wgrad_reduce_f32': (ctypes.c_int, [_vp, _i32, _vp]),
    'ursa_conv3x3_supported': (ctypes.c_int, [_i64, _i64, _i64, _i64, _i64, _u32]),
    'ursa_conv3x3_f32': (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _u32, _vp]),
    'ursa_conv1x1s2_supported': (ctypes.c_int, [_i64, _i64, _i64, _i64, _i64, _u32]),
    'ursa_conv1x1s2_f32': (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _u32, _vp]),
    'ursa_conv1x1_supported': (ctypes.c_int, [_i64, _i64, _i64, _i64, _i64, _u32]),
    'ursa_conv1x1_f32': (ctypes.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _u32, _vp]),
    'ursa_bn_stats_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f, _f, _vp]),
    'ursa_preact_conv1x1_supported': (ctypes.c_int, [_i64, _i64, _i64, _i64, _i64]),
    'ursa_preact_conv1x1_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp]),
    'ursa_preact_wgrad1x1_partial_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp]),
    'ursa_preact_conv1x1_bwd_nl': (ctypes.c_int64, [_i64, _i64, _i64, _i64, _i64]),
    'ursa_preact_conv1x1_bwd_sums_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp]),
    'ursa_bn_bwd_coef_f32': (ctypes.c_int, [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    'ursa_preact_conv1x1_bwd_dx_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp]),
    'ursa_preact_geometry': (ctypes.c_int, [_i64, _i64, _i64, _i64, _i64, _u32, _i64p]),
    'ursa_preact_conv3x3_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _i64,
                                               _i64, _i64, _i64, _i64, _i64, _u32, _vp]),
    'ursa_preact_wgrad_partial_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i32, _vp]),
    'ursa_preact_bwd_pair_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _u32, _vp]),
    'ursa_bn_relu_pool_f32': (ctypes.c_int, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f, _f, _vp]),
    'ursa_fc_ce_supported': (ctypes.c_int, [_i64, _i64, _i64]),
    'ursa_fc_ce_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp]),
    'ursa_bn_relu_pool_bwd_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _vp]),
    'ursa_bn_apply_f32': (ctypes.c_int, [_vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f, _f, _u32, _vp]),
    'ursa_bn_bwd_dx_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i64, _i64, _i64, _vp]),
}


#: declared under `#ifdef URSA_DEBUG_KNOBS` in the header: exported by csrc/libursa_hip_knobs.so ONLY (the parked NHWC-twin
#: experiment; tools/exp and its kernel-level test) - never by the shipped library
KNOBS_SIGNATURES = {
    'ursa_bn_relu_fwd_nhwc_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f, _f, _u32, _vp]),
    'ursa_bn_relu_bwd_nhwc_f32': (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _u32, _vp]),
}

class ConvPending(ctypes.Structure):
    """struct ursa_conv_pending: one layer whose second K7 launch is still to be taken."""
    _fields_ = [('ws', _vp), ('dw', _vp), ('N', _i64), ('Cin', _i64), ('Cout', _i64), ('H', _i64), ('W', _i64),
                ('ksize', _i32), ('stride', _i32)]


CTL_TICKET_LINES = 16


class StepCtl(ctypes.Structure):
    """struct ursa_step_ctl (2,304 bytes = 18 lines of 128). `sched` is a DEVICE address; `tickets` is device scratch
    (upload zeros): the two-level ticket tree of the self-advancing launch, one counter per 128-byte line."""
    _fields_ = [('lr', _f), ('mu', _f), ('c_wd', _f), ('c_noise', _f), ('n_train', _f), ('flags', _u32),
                ('seed', _u64), ('step', _u64), ('sched_base', _u64), ('sched', _u64), ('sched_len', _u32),
                ('reserved', _u32), ('pad', _u32 * 16), ('tickets', _u32 * ((CTL_TICKET_LINES + 1) * 32))]

    def tickets_clear(self):
        return not any(self.tickets)


CTL_BYTES = ctypes.sizeof(StepCtl)
assert CTL_BYTES == 2304


def _ctl_ptr(ctl, n_chains=1):
    if not (isinstance(ctl, torch.Tensor) and ctl.is_cuda and ctl.dtype == torch.uint8
            and ctl.numel() == n_chains * CTL_BYTES and ctl.is_contiguous() and ctl.data_ptr() % 128 == 0):
        raise ValueError(f'ctl must be a contiguous 128-byte aligned uint8 HIP tensor of {n_chains} x sizeof(ursa_step_ctl) bytes')
    return ctl.data_ptr()


class NativeLibraryMissing(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """dlopen the kernel library and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise NativeLibraryMissing(
            f'{p} not found: the HIP kernel library is not built. Run '
            f'`python -c "import __graft_entry__ as g; g.build()"` (or `make -C ursabench_amd/csrc`). '
            f'ursabench_amd has no CPU fallback.')
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    for name, (res, args) in KNOBS_SIGNATURES.items():
        fn = getattr(lib, name, None)    # the knobs build only
        if fn is not None:
            fn.restype, fn.argtypes = res, args
    got = lib.ursa_abi_version()
    if got != ABI_VERSION:
        raise NativeLibraryMissing(f'{p}: ABI version {got}, binding expects {ABI_VERSION}; rebuild the library')
    if path is None:
        _lib = lib
    return lib


def _check(lib, code, what):
    if code == 0:
        return
    msg = lib.ursa_strerror(code).decode()
    if code < 0:
        raise ValueError(f'{what}: {msg} (ursa error {code})')
    raise RuntimeError(f'{what}: HIP error {code}: {msg}')


def _ptr(t, name, n=None, device=None, optional=False):
    """Host-side operand check (shape/dtype/device/contiguity) before any launch."""
    if t is None:
        if optional:
            return None
        raise ValueError(f'{name} is required')
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name} must be a torch.Tensor, got {type(t).__name__}')
    if not t.is_cuda:
        raise RuntimeError(f'{name} must live on a HIP device (got {t.device}); ursabench_amd has no CPU path')
    if t.dtype != torch.float32:
        raise TypeError(f'{name} must be float32, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{name} must be contiguous')
    if n is not None and t.numel() != n:
        raise ValueError(f'{name} has {t.numel()} elements, expected {n}')
    if device is not None and t.device != device:
        raise ValueError(f'{name} is on {t.device}, expected {device}')
    return t.data_ptr()


def nhwc_twin_supported(x):
    """Whether K6 can store a channels-last twin of an output shaped like `x` (ursa_bn_relu_*_nhwc_f32's conditions)."""
    return (x.dim() == 4 and x.is_contiguous() and x.shape[1] % 4 == 0
            and (x.shape[2] * x.shape[3]) % 4 == 0 and x.data_ptr() % 16 == 0)


def nhwc_twin(x):
    """An uninitialised tensor with x's logical shape and channels-last strides (physically [N, H, W, C])."""
    return torch.empty_like(x, memory_format=torch.channels_last)


def _twin_ptr(t, like):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.shape == like.shape
            and t.device == like.device and t.is_contiguous(memory_format=torch.channels_last)):
        raise ValueError('the NHWC twin must be a float32 channels-last tensor of the output\'s shape on its device')
    return t.data_ptr()


def _held_flags(held):
    return BN_HELD if held else 0


def _stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class HipKernels:
    """The product kernel set: thin, checked wrappers that enqueue on torch's current stream."""

    name = 'hip-gfx950'

    def __init__(self, lib=None):
        """lib: a library handle from load_library(path) - tests and tools that need the experiment knobs pass
        csrc/libursa_hip_knobs.so (KNOBS_LIB_PATH); the product uses the shipped library, which reads no environment."""
        self.lib = load_library() if lib is None else lib

    # K1 ------------------------------------------------------------------------------
    def sgmcmc_step(self, theta, grad, mom, *, lr, mu, c_wd, c_noise, n_train, flags, seed=0, step=0,
                    eps=None, snapshot=None):
        n, dev = theta.numel(), theta.device
        args = (_ptr(theta, 'theta'), _ptr(grad, 'grad', n, dev), _ptr(mom, 'mom', n, dev, optional=(mu == 0)),
                _ptr(eps, 'eps', n, dev, optional=True), _ptr(snapshot, 'snapshot', n, dev, optional=True))
        with torch.cuda.device(dev):
            rc = self.lib.ursa_sgmcmc_step_f32(*args, n, lr, mu, c_wd, c_noise, n_train, seed, step, flags,
                                               _stream(dev))
        _check(self.lib, rc, 'ursa_sgmcmc_step_f32')

    def sgmcmc_step_ctl(self, theta, grad, mom, ctl, *, eps=None, snapshot=None):
        """Scalars from the device control block; with STEP_ADVANCE in its flags the launch advances it too."""
        n, dev = theta.numel(), theta.device
        args = (_ptr(theta, 'theta'), _ptr(grad, 'grad', n, dev), _ptr(mom, 'mom', n, dev),
                _ptr(eps, 'eps', n, dev, optional=True), _ptr(snapshot, 'snapshot', n, dev, optional=True))
        cp = _ctl_ptr(ctl)
        with torch.cuda.device(dev):
            rc = self.lib.ursa_sgmcmc_step_ctl_f32(*args, n, cp, _stream(dev))
        _check(self.lib, rc, 'ursa_sgmcmc_step_ctl_f32')

    def sgmcmc_step_multi(self, theta, grad, mom, ctl, *, n_per_chain=None, eps=None, snapshot=None):
        """K chains in one launch: theta / grad / mom (/ eps / snapshot) are [K, chain_stride] slabs, ctl holds K
        control blocks. Host-side shape check before the launch: every slab has the same shape and the grid the
        library derives from (n_per_chain, K) stays inside it."""
        if theta.dim() != 2:
            raise ValueError(f'theta must be a [K, chain_stride] slab, got {tuple(theta.shape)}')
        K, stride = theta.shape
        n = stride if n_per_chain is None else int(n_per_chain)
        if not 0 <= n <= stride:
            raise ValueError(f'n_per_chain {n} outside the slab row of {stride} elements')
        dev, tot = theta.device, K * stride
        for name, t in (('grad', grad), ('mom', mom), ('eps', eps), ('snapshot', snapshot)):
            if t is not None and tuple(t.shape) != (K, stride):
                raise ValueError(f'{name} must have the slab shape {(K, stride)}, got {tuple(t.shape)}')
        args = (_ptr(theta, 'theta'), _ptr(grad, 'grad', tot, dev), _ptr(mom, 'mom', tot, dev),
                _ptr(eps, 'eps', tot, dev, optional=True), _ptr(snapshot, 'snapshot', tot, dev, optional=True))
        cp = _ctl_ptr(ctl, K)
        with torch.cuda.device(dev):
            rc = self.lib.ursa_sgmcmc_step_multi_f32(*args, n, K, stride, cp, _stream(dev))
        _check(self.lib, rc, 'ursa_sgmcmc_step_multi_f32')

    def step_ctl_advance(self, ctl):
        """Advance every control block of `ctl` (one launch, one thread per block)."""
        dev = ctl.device
        n_ctl = ctl.numel() // CTL_BYTES
        cp = _ctl_ptr(ctl, n_ctl)
        with torch.cuda.device(dev):
            rc = self.lib.ursa_step_ctl_advance(cp, n_ctl, _stream(dev))
        _check(self.lib, rc, 'ursa_step_ctl_advance')

    def philox_normal(self, out, *, seed, step):
        dev = out.device
        with torch.cuda.device(dev):
            rc = self.lib.ursa_philox_normal_f32(_ptr(out, 'out'), out.numel(), seed, step, _stream(dev))
        _check(self.lib, rc, 'ursa_philox_normal_f32')

    def selftest_rng(self, device='cuda'):
        """(radius mismatches, logarithm mismatches) of the generator's fast division / square root against the IEEE
        forms over all 2^32 Philox words; both must be 0."""
        dev = torch.device(device)
        counts = torch.zeros(2, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            rc = self.lib.ursa_selftest_rng_f32(counts.data_ptr(), _stream(dev))
        _check(self.lib, rc, 'ursa_selftest_rng_f32')
        return tuple(int(v) for v in counts.cpu())

    # K2 / K3 -------------------------------------------------------------------------
    def swag_collect(self, mean, sq, w, *, decay, denom):
        n, dev = mean.numel(), mean.device
        with torch.cuda.device(dev):
            rc = self.lib.ursa_swag_collect_f32(_ptr(mean, 'mean'), _ptr(sq, 'sq', n, dev), _ptr(w, 'w', n, dev), n,
                                                decay, denom, _stream(dev))
        _check(self.lib, rc, 'ursa_swag_collect_f32')

    def swag_draw(self, out, mean, sq, *, var_clamp, scale=1.0, seed=0, draw=0, eps=None):
        n, dev = out.numel(), out.device
        with torch.cuda.device(dev):
            rc = self.lib.ursa_swag_draw_f32(_ptr(out, 'theta_out'), _ptr(mean, 'mean', n, dev),
                                             _ptr(sq, 'sq', n, dev), _ptr(eps, 'eps', n, dev, optional=True), n,
                                             var_clamp, scale, seed, draw, _stream(dev))
        _check(self.lib, rc, 'ursa_swag_draw_f32')

    def swag_std(self, out, mean, sq, *, var_clamp, scale=1.0):
        """std = sqrt(max(sq - mean^2, var_clamp)) * scale, once per pair of moment vectors."""
        n, dev = out.numel(), out.device
        with torch.cuda.device(dev):
            rc = self.lib.ursa_swag_std_f32(_ptr(out, 'std_out'), _ptr(mean, 'mean', n, dev), _ptr(sq, 'sq', n, dev), n,
                                            var_clamp, scale, _stream(dev))
        _check(self.lib, rc, 'ursa_swag_std_f32')

    def swag_draw_std(self, out, mean, std, *, seed=0, draw=0, eps=None):
        """theta = eps * std + mean: the per-member draw from a stored standard deviation."""
        n, dev = out.numel(), out.device
        with torch.cuda.device(dev):
            rc = self.lib.ursa_swag_draw_std_f32(_ptr(out, 'theta_out'), _ptr(mean, 'mean', n, dev), _ptr(std, 'std', n, dev),
                                                 _ptr(eps, 'eps', n, dev, optional=True), n, seed, draw, _stream(dev))
        _check(self.lib, rc, 'ursa_swag_draw_std_f32')

    # K5 ------------------------------------------------------------------------------
    def bma_accumulate(self, logits, proba_sum, ent_sum=None, *, one_minus_gamma, gamma_over_c, smoothed,
                       risk_sum=None, cost=None):
        if logits.dim() != 3:
            raise ValueError(f'logits must be [S, B, C], got {tuple(logits.shape)}')
        S, B, C = logits.shape
        dev = logits.device
        with torch.cuda.device(dev):
            rc = self.lib.ursa_bma_accumulate_f32(
                _ptr(logits, 'logits'), _ptr(proba_sum, 'proba_sum', B * C, dev),
                _ptr(ent_sum, 'ent_sum', B, dev, optional=True), _ptr(risk_sum, 'risk_sum', B * C, dev, optional=True),
                _ptr(cost, 'cost', C * C, dev, optional=True), S, B, C, one_minus_gamma, gamma_over_c,
                BMA_SMOOTHED if smoothed else 0, _stream(dev))
        _check(self.lib, rc, 'ursa_bma_accumulate_f32')

    # K4 ------------------------------------------------------------------------------
    def leapfrog(self, theta, mom, grad, *, kick_coef, step_size, inv_mass, flags, kinetic_out=None, ws=None):
        n, dev = mom.numel(), mom.device
        if kinetic_out is not None and (ws is None or ws.numel() < REDUCE_WS_FLOATS):
            raise ValueError(f'ws must hold {REDUCE_WS_FLOATS} floats')
        with torch.cuda.device(dev):
            rc = self.lib.ursa_leapfrog_f32(_ptr(theta, 'theta', n, dev, optional=not (flags & LEAP_DRIFT)),
                                            _ptr(mom, 'mom'),
                                            _ptr(grad, 'grad', n, dev, optional=not (flags & LEAP_KICK)), n,
                                            kick_coef, step_size, inv_mass, flags,
                                            _ptr(kinetic_out, 'kinetic_out', 1, dev, optional=True),
                                            _ptr(ws, 'ws', None, dev, optional=True), _stream(dev))
        _check(self.lib, rc, 'ursa_leapfrog_f32')

    def sumsq(self, x, out, ws):
        dev = x.device
        if ws.numel() < REDUCE_WS_FLOATS:
            raise ValueError(f'ws must hold {REDUCE_WS_FLOATS} floats')
        with torch.cuda.device(dev):
            rc = self.lib.ursa_sumsq_f32(_ptr(x, 'x'), x.numel(), _ptr(out, 'out', 1, dev), _ptr(ws, 'ws', None, dev),
                                         _stream(dev))
        _check(self.lib, rc, 'ursa_sumsq_f32')

    # K6 ------------------------------------------------------------------------------
    def _knob(self, name):
        fn = getattr(self.lib, name, None)
        if fn is None:
            raise RuntimeError(f'{name} is not part of the product ABI: only csrc/libursa_hip_knobs.so exports it (_native.knobs_kernels())')
        return fn

    @staticmethod
    def _bn_dims(x):
        if x.dim() < 2:
            raise ValueError(f'BatchNorm input must be [N, C, ...], got {tuple(x.shape)}')
        N, C = x.shape[0], x.shape[1]
        return N, C, x.numel() // max(N * C, 1)

    def bn_relu_forward(self, x, y, gamma, beta, running_mean, running_var, save_mean, save_invstd, ws, *, eps,
                        momentum, relu=True, addend=None, z_out=None, two_launch=False, held=False, y_nhwc=None, save_gate=None):
        """Training-mode BatchNorm (+ ReLU) of a contiguous [N, C, *] tensor: batch statistics, running statistics
        updated in place (skipped when both are None), mean / invstd saved for the backward. With `addend` the
        normalised tensor is z = x + addend, also stored to `z_out` (the residual sum folded into the statistics pass).
        held=True: `ws` is ZEROED (at least its sync words) - the library may then take the one-launch held form.
        save_gate: 2*C floats that receive the scale / shift this forward applied (hand them to bn_relu_backward as `gate`)."""
        if (addend is None) != (z_out is None):
            raise ValueError('addend and z_out go together')
        N, C, HW = self._bn_dims(x)
        dev, n = x.device, x.numel()
        if ws.numel() < bn_ws_floats(C):
            raise ValueError(f'ws must hold {bn_ws_floats(C)} floats')
        head = (_ptr(x, 'x'), _ptr(addend, 'addend', n, dev, optional=True), _ptr(z_out, 'z_out', n, dev, optional=True),
                _ptr(y, 'y', n, dev))
        tail = (_ptr(gamma, 'gamma', C, dev), _ptr(beta, 'beta', C, dev),
                _ptr(running_mean, 'running_mean', C, dev, optional=True),
                _ptr(running_var, 'running_var', C, dev, optional=True), _ptr(save_mean, 'save_mean', C, dev),
                _ptr(save_invstd, 'save_invstd', C, dev), _ptr(save_gate, 'save_gate', 2 * C, dev, optional=True),
                _ptr(ws, 'ws', None, dev), N, C, HW, eps, momentum,
                (BN_RELU if relu else 0) | (BN_TWO_LAUNCH if two_launch else 0) | _held_flags(held and y_nhwc is None), _stream(dev))
        with torch.cuda.device(dev):
            if y_nhwc is not None:              # (knobs build only) second output: the same floats channels-last (nhwc_twin() allocates it)
                if save_gate is not None:
                    raise ValueError('the NHWC-twin experiment has no save_gate')
                rc = self._knob('ursa_bn_relu_fwd_nhwc_f32')(*head, _twin_ptr(y_nhwc, x), *(tail[:6] + tail[7:]))
            else:
                rc = self.lib.ursa_bn_relu_fwd_f32(*head, *tail)
        _check(self.lib, rc, 'ursa_bn_relu_fwd_f32')

    def bn_relu_eval(self, x, y, gamma, beta, running_mean, running_var, *, eps, relu=True, addend=None, z_out=None):
        if (addend is None) != (z_out is None):
            raise ValueError('addend and z_out go together')
        N, C, HW = self._bn_dims(x)
        dev, n = x.device, x.numel()
        with torch.cuda.device(dev):
            rc = self.lib.ursa_bn_relu_eval_f32(
                _ptr(x, 'x'), _ptr(addend, 'addend', n, dev, optional=True), _ptr(z_out, 'z_out', n, dev, optional=True),
                _ptr(y, 'y', n, dev), _ptr(gamma, 'gamma', C, dev), _ptr(beta, 'beta', C, dev),
                _ptr(running_mean, 'running_mean', C, dev), _ptr(running_var, 'running_var', C, dev), N, C, HW, eps,
                BN_RELU if relu else 0, _stream(dev))
        _check(self.lib, rc, 'ursa_bn_relu_eval_f32')

    def bn_relu_backward(self, x, dy, dx, gamma, beta, save_mean, save_invstd, dgamma, dbeta, ws, *, relu=True, dz=None,
                         two_launch=False, gates=None, held=False, dx_nhwc=None, gate=None):
        """`x` is the tensor the forward normalised (z_out in the residual form); with `dz` the result is dx + dz.
        `gates=(idx int32 [n], open uint8 [n])`: the parity instrument ursa_bn_relu_bwd_gated_f32 - the ReLU gates of the
        listed element offsets (ascending; INT32_MAX = padding) are taken from `open` instead of recomputed.
        `gate`: the forward's `save_gate` (2*C floats): the ReLU gate is recomputed from the scale / shift the forward applied
        instead of from the live gamma / beta."""
        N, C, HW = self._bn_dims(x)
        dev, n = x.device, x.numel()
        if ws.numel() < bn_ws_floats(C):
            raise ValueError(f'ws must hold {bn_ws_floats(C)} floats')
        if dx_nhwc is not None:
            if gates is not None or gate is not None:
                raise ValueError('the gated backward has no NHWC twin')
            with torch.cuda.device(dev):
                rc = self._knob('ursa_bn_relu_bwd_nhwc_f32')(
                    _ptr(x, 'x'), _ptr(dy, 'dy', n, dev), _ptr(dz, 'dz', n, dev, optional=True), _ptr(dx, 'dx', n, dev),
                    _twin_ptr(dx_nhwc, x), _ptr(gamma, 'gamma', C, dev), _ptr(beta, 'beta', C, dev),
                    _ptr(save_mean, 'save_mean', C, dev), _ptr(save_invstd, 'save_invstd', C, dev), _ptr(dgamma, 'dgamma', C, dev),
                    _ptr(dbeta, 'dbeta', C, dev), _ptr(ws, 'ws', None, dev), N, C, HW,
                    (BN_RELU if relu else 0) | (BN_TWO_LAUNCH if two_launch else 0), _stream(dev))
            _check(self.lib, rc, 'ursa_bn_relu_bwd_nhwc_f32')
            return
        args = (_ptr(x, 'x'), _ptr(dy, 'dy', n, dev), _ptr(dz, 'dz', n, dev, optional=True), _ptr(dx, 'dx', n, dev),
                _ptr(gamma, 'gamma', C, dev), _ptr(beta, 'beta', C, dev), _ptr(save_mean, 'save_mean', C, dev),
                _ptr(save_invstd, 'save_invstd', C, dev), _ptr(gate, 'gate', 2 * C, dev, optional=True),
                _ptr(dgamma, 'dgamma', C, dev), _ptr(dbeta, 'dbeta', C, dev),
                _ptr(ws, 'ws', None, dev), N, C, HW,
                (BN_RELU if relu else 0) | (BN_TWO_LAUNCH if two_launch else 0) | (_held_flags(held) if gates is None else 0))
        if gates is not None:
            gi, go = gates
            for t, dt, nm in ((gi, torch.int32, 'gate idx'), (go, torch.uint8, 'gate open')):
                if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == dev and t.dtype == dt and t.is_contiguous()):
                    raise ValueError(f'{nm} must be a contiguous {dt} tensor on {dev}')
            if gi.numel() != go.numel():
                raise ValueError('gate idx / open lengths differ')
            with torch.cuda.device(dev):
                rc = self.lib.ursa_bn_relu_bwd_gated_f32(*args, gi.data_ptr(), go.data_ptr(), gi.numel(), _stream(dev))
            _check(self.lib, rc, 'ursa_bn_relu_bwd_gated_f32')
            return
        with torch.cuda.device(dev):
            rc = self.lib.ursa_bn_relu_bwd_f32(*args, _stream(dev))
        _check(self.lib, rc, 'ursa_bn_relu_bwd_f32')

    # K7 ------------------------------------------------------------------------------
    def conv_wgrad_ws_floats(self, x_shape, cout, ksize=3, stride=1):
        """Scratch floats K7 needs for the weight gradient of a ksize x ksize (pad ksize // 2) convolution over an input of
        `x_shape` ([N, Cin, H, W]) with `cout` output channels; 0 = shape not covered (the caller keeps the stock gradient)."""
        n, cin, h, w = (int(v) for v in x_shape)
        return int(self.lib.ursa_conv_wgrad_ws_floats(n, cin, int(cout), h, w, int(ksize), int(stride)))

    @staticmethod
    def _conv_dims(x, dy, dw_shape, stride):
        if x.dim() != 4 or dy.dim() != 4 or len(dw_shape) != 4:
            raise ValueError('x, dy, dw must be 4-d (NCHW / OIHW)')
        N, Cin, H, W = x.shape
        Cout, ksize = int(dw_shape[0]), int(dw_shape[2])
        if tuple(dw_shape) != (Cout, Cin, ksize, ksize) or tuple(dy.shape) != (N, Cout, H // stride, W // stride):
            raise ValueError(f'shapes do not belong to one convolution: x {tuple(x.shape)}, dy {tuple(dy.shape)}, dw {tuple(dw_shape)}')
        return N, Cin, Cout, H, W, ksize

    def conv_wgrad(self, x, dy, dw, ws, stride=1):
        """dw[Cout, Cin, k, k] = the weight gradient of conv2d(x, w, stride=stride, padding=k // 2) for the output gradient dy
        (both launches)."""
        N, Cin, Cout, H, W, ksize = self._conv_dims(x, dy, dw.shape, stride)
        dev = x.device
        with torch.cuda.device(dev):
            rc = self.lib.ursa_conv_wgrad_f32(_ptr(x, 'x'), _ptr(dy, 'dy', None, dev), _ptr(dw, 'dw', None, dev),
                                              _ptr(ws, 'ws', None, dev), ws.numel(), N, Cin, Cout, H, W, ksize, int(stride), _stream(dev))
        _check(self.lib, rc, 'ursa_conv_wgrad_f32')

    def conv_wgrad_partial(self, x, dy, dw_shape, ws, stride=1):
        """The first launch only: `ws` receives the K slices' partial sums. Returns the record `conv_wgrad_reduce` takes."""
        N, Cin, Cout, H, W, ksize = self._conv_dims(x, dy, dw_shape, stride)
        dev = x.device
        with torch.cuda.device(dev):
            rc = self.lib.ursa_conv_wgrad_partial_f32(_ptr(x, 'x'), _ptr(dy, 'dy', None, dev), _ptr(ws, 'ws', None, dev), ws.numel(),
                                                      N, Cin, Cout, H, W, ksize, int(stride), _stream(dev))
        _check(self.lib, rc, 'ursa_conv_wgrad_partial_f32')
        return (ws, N, Cin, Cout, H, W, ksize, int(stride))

    def conv_wgrad_reduce(self, pending):
        """The second launch for many layers at once. pending: [(record from conv_wgrad_partial, dw tensor), ...], one device."""
        if not pending:
            return
        arr = (ConvPending * len(pending))()
        dev = pending[0][1].device
        for a, ((ws, N, Cin, Cout, H, W, ksize, stride), dw) in zip(arr, pending):
            if tuple(dw.shape) != (Cout, Cin, ksize, ksize):
                raise ValueError(f'dw {tuple(dw.shape)} does not belong to the pending record ({Cout}, {Cin}, {ksize}, {ksize})')
            a.ws, a.dw = _ptr(ws, 'ws', None, dev), _ptr(dw, 'dw', None, dev)
            a.N, a.Cin, a.Cout, a.H, a.W, a.ksize, a.stride = N, Cin, Cout, H, W, ksize, stride
        with torch.cuda.device(dev):
            rc = self.lib.ursa_conv_wgrad_reduce_f32(ctypes.cast(arr, ctypes.c_void_p), len(pending), _stream(dev))
        _check(self.lib, rc, 'ursa_conv_wgrad_reduce_f32')


    # K8 ------------------------------------------------------------------------------
    def conv3x3_supported(self, x_shape, cout, flip=False, stride=1):
        """Whether K8 takes a 3x3 / pad 1 convolution of an input of `x_shape` ([N, Cin, H, W]) to `cout` channels at `stride`
        (flip: `x_shape` is the output gradient's shape and `cout` the layer's input channels)."""
        n, cin, h, w = (int(v) for v in x_shape)
        return bool(self.lib.ursa_conv3x3_supported(n, cin, int(cout), h, w, self._conv_flags(flip, stride)))

    @staticmethod
    def _conv_flags(flip, stride):
        if stride not in (1, 2):
            raise ValueError('stride must be 1 or 2')
        return (CONV_FLIP if flip else 0) | (CONV_STRIDE2 if stride == 2 else 0)

    def conv3x3(self, x, w, y=None, flip=False, stride=1):
        """y = conv2d(x, w, stride=stride, padding=1) (w: [Cout, Cin, 3, 3]); flip=True: the input gradient of that layer - `x`
        is dy [N, Cout, H/stride, W/stride], the result dx [N, Cin, H, W], `w` the layer's own weight."""
        if x.dim() != 4 or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or w.shape[0 if flip else 1] != x.shape[1]:
            raise ValueError(f'not a 3x3 convolution: x {tuple(x.shape)}, w {tuple(w.shape)}, flip={flip}')
        N, Cin, H, W = x.shape
        Cout = w.shape[1 if flip else 0]
        dev = x.device
        shape = (N, Cout, H * stride, W * stride) if flip else (N, Cout, H // stride, W // stride)
        if y is None:
            y = x.new_empty(shape)
        elif tuple(y.shape) != shape:
            raise ValueError(f'y {tuple(y.shape)} should be {shape}')
        with torch.cuda.device(dev):
            rc = self.lib.ursa_conv3x3_f32(_ptr(x, 'x'), _ptr(w, 'w', None, dev), _ptr(y, 'y', None, dev), N, Cin, Cout, H, W,
                                           self._conv_flags(flip, stride), _stream(dev))
        _check(self.lib, rc, 'ursa_conv3x3_f32')
        return y

    # K9 ------------------------------------------------------------------------------
    def conv1x1s2_supported(self, x_shape, cout, flip=False):
        """Whether K9 takes the 1x1 / stride 2 convolution of an input of `x_shape` to `cout` channels (flip: `x_shape` is the
        output gradient's shape and `cout` the layer's input channels)."""
        n, cin, h, w = (int(v) for v in x_shape)
        return bool(self.lib.ursa_conv1x1s2_supported(n, cin, int(cout), h, w, CONV_FLIP if flip else 0))

    def conv1x1s2(self, x, w, y=None, flip=False):
        """y = conv2d(x, w, stride=2) for a 1x1 weight [Cout, Cin, 1, 1]; flip=True: the input gradient of that layer - `x` is
        dy [N, Cout, H/2, W/2], the result dx [N, Cin, H, W]."""
        if x.dim() != 4 or w.dim() != 4 or tuple(w.shape[2:]) != (1, 1) or w.shape[0 if flip else 1] != x.shape[1]:
            raise ValueError(f'not a 1x1 convolution: x {tuple(x.shape)}, w {tuple(w.shape)}, flip={flip}')
        N, Cin, H, W = x.shape
        Cout = w.shape[1 if flip else 0]
        dev = x.device
        shape = (N, Cout, 2 * H, 2 * W) if flip else (N, Cout, H // 2, W // 2)
        if y is None:
            y = x.new_empty(shape)
        elif tuple(y.shape) != shape:
            raise ValueError(f'y {tuple(y.shape)} should be {shape}')
        with torch.cuda.device(dev):
            rc = self.lib.ursa_conv1x1s2_f32(_ptr(x, 'x'), _ptr(w, 'w', None, dev), _ptr(y, 'y', None, dev), N, Cin, Cout, H, W,
                                             CONV_FLIP if flip else 0, _stream(dev))
        _check(self.lib, rc, 'ursa_conv1x1s2_f32')
        return y


    # K12 -----------------------------------------------------------------------------
    def conv1x1_supported(self, x_shape, cout, flip=False):
        """Whether K12 takes the 1x1 / stride 1 convolution of an input of `x_shape` to `cout` channels (flip: `x_shape` is the
        output gradient's shape and `cout` the layer's input channels)."""
        n, cin, h, w = (int(v) for v in x_shape)
        return bool(self.lib.ursa_conv1x1_supported(n, cin, int(cout), h, w, CONV_FLIP if flip else 0))

    def conv1x1(self, x, w, y=None, flip=False):
        """y = conv2d(x, w) for a 1x1 weight [Cout, Cin, 1, 1]; flip=True: the input gradient of that layer - `x` is dy
        [N, Cout, H, W], the result dx [N, Cin, H, W]."""
        if x.dim() != 4 or w.dim() != 4 or tuple(w.shape[2:]) != (1, 1) or w.shape[0 if flip else 1] != x.shape[1]:
            raise ValueError(f'not a 1x1 convolution: x {tuple(x.shape)}, w {tuple(w.shape)}, flip={flip}')
        N, Cin, H, W = x.shape
        Cout = w.shape[1 if flip else 0]
        dev = x.device
        shape = (N, Cout, H, W)
        if y is None:
            y = x.new_empty(shape)
        elif tuple(y.shape) != shape:
            raise ValueError(f'y {tuple(y.shape)} should be {shape}')
        with torch.cuda.device(dev):
            rc = self.lib.ursa_conv1x1_f32(_ptr(x, 'x'), _ptr(w, 'w', None, dev), _ptr(y, 'y', None, dev), N, Cin, Cout, H, W,
                                           CONV_FLIP if flip else 0, _stream(dev))
        _check(self.lib, rc, 'ursa_conv1x1_f32')
        return y

    # K13 -----------------------------------------------------------------------------
    def bn_stats(self, x, gamma, beta, running_mean, running_var, save, ws, *, eps, momentum, addend=None, z_out=None):
        """K6's statistics launch + the merge of its partial sums, alone: `save` [4, C] receives (mean, invstd, scale, shift) of the
        training-mode BatchNorm of x (addend: of z = x + addend, stored to z_out), the running statistics are updated; nothing is
        normalised - `preact_conv1x1` / `preact_wgrad1x1_partial` apply relu(fma(x, scale, shift)) themselves."""
        if (addend is None) != (z_out is None):
            raise ValueError('addend and z_out go together')
        N, C, HW = self._bn_dims(x)
        dev, n = x.device, x.numel()
        if ws.numel() < bn_ws_floats(C):
            raise ValueError(f'ws must hold {bn_ws_floats(C)} floats')
        with torch.cuda.device(dev):
            rc = self.lib.ursa_bn_stats_f32(_ptr(x, 'x'), _ptr(addend, 'addend', n, dev, optional=True), _ptr(z_out, 'z_out', n, dev, optional=True),
                                            _ptr(gamma, 'gamma', C, dev), _ptr(beta, 'beta', C, dev),
                                            _ptr(running_mean, 'running_mean', C, dev, optional=True),
                                            _ptr(running_var, 'running_var', C, dev, optional=True), _ptr(save, 'save', 4 * C, dev),
                                            _ptr(ws, 'ws', None, dev), N, C, HW, eps, momentum, _stream(dev))
        _check(self.lib, rc, 'ursa_bn_stats_f32')

    def preact_conv1x1_supported(self, x_shape, cout):
        N, Cin, H, W = x_shape
        return bool(self.lib.ursa_preact_conv1x1_supported(N, Cin, cout, H, W))

    def preact_conv1x1(self, x, bn_save, w, y=None):
        """y = conv2d(relu(fma(x, scale, shift)), w) for a 1x1 weight [Cout, Cin, 1, 1], scale / shift = rows 2, 3 of bn_save [4, Cin]."""
        if x.dim() != 4 or w.dim() != 4 or tuple(w.shape[2:]) != (1, 1) or w.shape[1] != x.shape[1]:
            raise ValueError(f'not a 1x1 convolution: x {tuple(x.shape)}, w {tuple(w.shape)}')
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        dev = x.device
        if y is None:
            y = x.new_empty((N, Cout, H, W))
        elif tuple(y.shape) != (N, Cout, H, W):
            raise ValueError(f'y {tuple(y.shape)} should be {(N, Cout, H, W)}')
        with torch.cuda.device(dev):
            rc = self.lib.ursa_preact_conv1x1_f32(_ptr(x, 'x'), _ptr(bn_save, 'bn_save', 4 * Cin, dev), _ptr(w, 'w', None, dev),
                                                  _ptr(y, 'y', None, dev), N, Cin, Cout, H, W, _stream(dev))
        _check(self.lib, rc, 'ursa_preact_conv1x1_f32')
        return y

    def preact_wgrad1x1_partial(self, x, bn_save, dy, dw_shape, ws):
        """K12's weight-gradient launch with the x operand = relu(bn(x)) rebuilt from bn_save while staged; returns conv_wgrad_reduce's record."""
        N, Cin, Cout, H, W, ksize = self._conv_dims(x, dy, dw_shape, 1)
        if ksize != 1:
            raise ValueError('this form is for the 1x1 layers')
        dev = x.device
        with torch.cuda.device(dev):
            rc = self.lib.ursa_preact_wgrad1x1_partial_f32(_ptr(x, 'x'), _ptr(bn_save, 'bn_save', 4 * Cin, dev), _ptr(dy, 'dy', None, dev),
                                                           _ptr(ws, 'ws', None, dev), ws.numel(), N, Cin, Cout, H, W, _stream(dev))
        _check(self.lib, rc, 'ursa_preact_wgrad1x1_partial_f32')
        return (ws, N, Cin, Cout, H, W, ksize, 1)

    # K14 -----------------------------------------------------------------------------
    def preact_conv1x1_bwd_nl(self, dy_shape, cx):
        """Partial sums per channel the K14 launches leave for an output gradient of `dy_shape` and a layer with `cx` input channels; 0: not covered."""
        N, Cd, H, W = dy_shape
        return int(self.lib.ursa_preact_conv1x1_bwd_nl(N, Cd, int(cx), H, W))

    def preact_conv1x1_bwd(self, dy, w, x, bn_save, gamma, dx, dgamma, dbeta, dz=None):
        """Backward of conv1x1(relu(bn(x))) w.r.t. x (and the BatchNorm's gamma / beta) without storing the convolution's input
        gradient: sums launch, merge, dx launch. dy [N, Cd, H, W], w [Cd, Cx, 1, 1], x / dx / dz [N, Cx, H, W], bn_save [4, Cx]."""
        if dy.dim() != 4 or w.dim() != 4 or tuple(w.shape[2:]) != (1, 1) or w.shape[0] != dy.shape[1] or tuple(x.shape) != (dy.shape[0], w.shape[1]) + tuple(dy.shape[2:]):
            raise ValueError(f'shapes: dy {tuple(dy.shape)}, w {tuple(w.shape)}, x {tuple(x.shape)}')
        N, Cd, H, W = dy.shape
        Cx = w.shape[1]
        dev, n = dy.device, x.numel()
        nl = self.preact_conv1x1_bwd_nl(dy.shape, Cx)
        if nl <= 0:
            raise ValueError(f'K14 does not cover dy {tuple(dy.shape)} -> {Cx} channels')
        pb = torch.empty(Cx, nl, 2, dtype=torch.float64, device=dev)
        coef = torch.empty(3, Cx, device=dev)
        with torch.cuda.device(dev):
            st = _stream(dev)
            rc = self.lib.ursa_preact_conv1x1_bwd_sums_f32(_ptr(dy, 'dy'), _ptr(w, 'w', None, dev), _ptr(x, 'x', None, dev),
                                                           _ptr(bn_save, 'bn_save', 4 * Cx, dev), self._f64ptr(pb, 'partial', 2 * Cx * nl, dev), N, Cd, Cx, H, W, st)
            _check(self.lib, rc, 'ursa_preact_conv1x1_bwd_sums_f32')
            rc = self.lib.ursa_bn_bwd_coef_f32(self._f64ptr(pb, 'partial', 2 * Cx * nl, dev), nl, _ptr(bn_save, 'bn_save', 4 * Cx, dev),
                                               _ptr(gamma, 'gamma', Cx, dev), _ptr(coef, 'coef', 3 * Cx, dev), _ptr(dgamma, 'dgamma', Cx, dev),
                                               _ptr(dbeta, 'dbeta', Cx, dev), N * H * W, Cx, st)
            _check(self.lib, rc, 'ursa_bn_bwd_coef_f32')
            rc = self.lib.ursa_preact_conv1x1_bwd_dx_f32(_ptr(dy, 'dy'), _ptr(w, 'w', None, dev), _ptr(x, 'x', None, dev),
                                                         _ptr(bn_save, 'bn_save', 4 * Cx, dev), _ptr(coef, 'coef', 3 * Cx, dev),
                                                         _ptr(dz, 'dz', n, dev, optional=True), _ptr(dx, 'dx', n, dev), N, Cd, Cx, H, W, st)
            _check(self.lib, rc, 'ursa_preact_conv1x1_bwd_dx_f32')
        return pb, coef

    # K10 -----------------------------------------------------------------------------
    def preact_geometry(self, x_shape, cout, *, flip=False, stride=1, bn=False, add=False):
        """(nl, scratch bytes, workgroups per channel, byte offset of the error word) of the K10 launch over an input of `x_shape` ([N, Cin, H, W]) to `cout`
        channels, or None when the library does not cover it. Forward forms leave the statistics of their output (bn: the
        BatchNorm + ReLU in front of the convolution folded in; add: `out += residual` folded in); flip: the input gradient
        with the BatchNorm backward's sums (x_shape = the output gradient's shape, cout = the layer's input channels)."""
        n, cin, h, w = (int(v) for v in x_shape)
        out = (ctypes.c_int64 * 4)()
        rc = self.lib.ursa_preact_geometry(n, cin, int(cout), h, w, self._preact_flags(flip, stride, bn, add), out)
        if rc == -5:
            return None
        _check(self.lib, rc, 'ursa_preact_geometry')
        return int(out[0]), int(out[1]), int(out[2]), int(out[3])

    def _preact_flags(self, flip, stride, bn, add):
        fl = self._conv_flags(flip, stride)
        if flip:
            return fl | PREACT_BNBWD
        return fl | PREACT_STATS | (PREACT_BN if bn else 0) | (PREACT_ADD if add else 0)

    @staticmethod
    def _f64ptr(t, name, n, dev):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == dev and t.dtype == torch.float64 and t.is_contiguous()
                and t.numel() == n):
            raise ValueError(f'{name} must be a contiguous float64 tensor of {n} elements on {dev}')
        return t.data_ptr()

    @staticmethod
    def _scratch_ptr(t, need, dev):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.device == dev and t.dtype == torch.uint8 and t.is_contiguous()
                and t.numel() >= need and t.data_ptr() % 128 == 0):
            raise ValueError(f'scratch must be a contiguous 128-byte aligned uint8 tensor of >= {need} bytes on {dev} (zeroed once)')
        return t.data_ptr()

    def preact_conv3x3(self, x, w, y, out_partial, scratch, *, stride=1, flip=False, bn=None, add=None, bwd=None):
        """One K10 launch (include/ursa_hip.h). bn = (in_partial [Cin, in_nl, 2] float64, gamma, beta, running_mean | None,
        running_var | None, bn_save [4, Cin], eps, momentum): convolve relu(bn(x)); add = the tensor added to the result before it is
        stored and summed; bwd = (bn_input, bn_save [4, C]) with flip=True: the gated input gradient and the BatchNorm backward's
        sums. out_partial: float64 [Cout, nl, 2]; scratch: uint8, zeroed once, private to this layer (forward forms only: the
        input-gradient forms need none - pass None)"""
        if x.dim() != 4 or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or w.shape[0 if flip else 1] != x.shape[1]:
            raise ValueError(f'not a 3x3 convolution: x {tuple(x.shape)}, w {tuple(w.shape)}, flip={flip}')
        if flip != (bwd is not None) or (flip and (bn is not None or add is not None)):
            raise ValueError('flip goes with bwd and excludes bn / add')
        N, Cin, H, W = x.shape
        Cout = w.shape[1 if flip else 0]
        dev = x.device
        shape = (N, Cout, H * stride, W * stride) if flip else (N, Cout, H // stride, W // stride)
        if tuple(y.shape) != shape:
            raise ValueError(f'y {tuple(y.shape)} should be {shape}')
        geo = self.preact_geometry(x.shape, Cout, flip=flip, stride=stride, bn=bn is not None, add=add is not None)
        if geo is None:
            raise ValueError(f'K10 does not cover x {tuple(x.shape)} -> {Cout} channels (flip={flip}, stride={stride}, bn={bn is not None}, add={add is not None})')
        nl, sbytes = geo[:2]
        ny = y.numel()
        if bn is not None:
            ip, gamma, beta, rm, rv, save, eps, mom = bn
            if ip.dim() != 3 or ip.shape[0] != Cin or ip.shape[2] != 2:
                raise ValueError(f'in_partial {tuple(ip.shape)} should be [{Cin}, nl, 2]')
            bnargs = (self._f64ptr(ip, 'in_partial', ip.numel(), dev), int(ip.shape[1]), _ptr(gamma, 'gamma', Cin, dev), _ptr(beta, 'beta', Cin, dev),
                      _ptr(rm, 'running_mean', Cin, dev, optional=True), _ptr(rv, 'running_var', Cin, dev, optional=True),
                      _ptr(save, 'bn_save', 4 * Cin, dev), float(eps), float(mom))
        else:
            bnargs = (None, 0, None, None, None, None, None, 0.0, 0.0)
        aux = aux_save = None
        if add is not None:
            aux = _ptr(add, 'addend', ny, dev)
        if bwd is not None:
            aux, aux_save = _ptr(bwd[0], 'bn input', ny, dev), _ptr(bwd[1], 'bn_save', 4 * Cout, dev)
        with torch.cuda.device(dev):
            rc = self.lib.ursa_preact_conv3x3_f32(_ptr(x, 'x'), _ptr(w, 'w', None, dev), _ptr(y, 'y', None, dev), *bnargs, aux, aux_save,
                                                  self._f64ptr(out_partial, 'out_partial', Cout * nl * 2, dev),
                                                  self._scratch_ptr(scratch, sbytes, dev) if sbytes else None, scratch.numel() if sbytes else 0,
                                                  N, Cin, Cout, H, W,
                                                  self._preact_flags(flip, stride, bn is not None, add is not None), _stream(dev))
        _check(self.lib, rc, 'ursa_preact_conv3x3_f32')
        return y

    def preact_eval_supported(self, x_shape, cout, stride=1, add=False):
        """Whether the evaluation-mode unit conv(relu(bn_eval(x))) [+ addend] of an input of `x_shape` to `cout` channels is covered."""
        n, cin, h, w = (int(v) for v in x_shape)
        out = (ctypes.c_int64 * 4)()
        fl = self._conv_flags(False, stride) | PREACT_BN | PREACT_EVAL | (PREACT_ADD if add else 0)
        return self.lib.ursa_preact_geometry(n, cin, int(cout), h, w, fl, out) == 0

    def preact_eval(self, x, w, y, gamma, beta, running_mean, running_var, *, eps, stride=1, add=None):
        """y = conv2d(relu(batch_norm_eval(x)), w, stride, padding=1) [+ add] in ONE launch (ursa_preact_conv3x3_f32 with
        URSA_PREACT_EVAL): K6's evaluation expressions applied while the tile is staged, K8's convolution."""
        if x.dim() != 4 or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or w.shape[1] != x.shape[1]:
            raise ValueError(f'not a 3x3 convolution: x {tuple(x.shape)}, w {tuple(w.shape)}')
        N, Cin, H, W = x.shape
        Cout = w.shape[0]
        dev = x.device
        shape = (N, Cout, H // stride, W // stride)
        if tuple(y.shape) != shape:
            raise ValueError(f'y {tuple(y.shape)} should be {shape}')
        fl = self._conv_flags(False, stride) | PREACT_BN | PREACT_EVAL | (PREACT_ADD if add is not None else 0)
        with torch.cuda.device(dev):
            rc = self.lib.ursa_preact_conv3x3_f32(_ptr(x, 'x'), _ptr(w, 'w', None, dev), _ptr(y, 'y', None, dev), None, 0,
                                                  _ptr(gamma, 'gamma', Cin, dev), _ptr(beta, 'beta', Cin, dev),
                                                  _ptr(running_mean, 'running_mean', Cin, dev), _ptr(running_var, 'running_var', Cin, dev),
                                                  None, float(eps), 0.0, _ptr(add, 'addend', y.numel(), dev, optional=True), None, None, None, 0,
                                                  N, Cin, Cout, H, W, fl, _stream(dev))
        _check(self.lib, rc, 'ursa_preact_conv3x3_f32 (evaluation)')
        return y

    def preact_wgrad_partial(self, x, bn_save, dy, dw_shape, ws, stride=1):
        """K7's first launch with the x operand = relu(bn(x)) rebuilt from bn_save while staged; returns conv_wgrad_reduce's record."""
        N, Cin, Cout, H, W, ksize = self._conv_dims(x, dy, dw_shape, stride)
        if ksize != 3:
            raise ValueError('the fused form is for the 3x3 layers')
        dev = x.device
        with torch.cuda.device(dev):
            rc = self.lib.ursa_preact_wgrad_partial_f32(_ptr(x, 'x'), _ptr(bn_save, 'bn_save', 4 * Cin, dev), _ptr(dy, 'dy', None, dev),
                                                        _ptr(ws, 'ws', None, dev), ws.numel(), N, Cin, Cout, H, W, int(stride), _stream(dev))
        _check(self.lib, rc, 'ursa_preact_wgrad_partial_f32')
        return (ws, N, Cin, Cout, H, W, ksize, int(stride))

    def preact_bwd_pair(self, dy, w, g, x, bn_save, out_partial, ws, stride=1):
        """preact_conv3x3(dy, w, g, out_partial, None, flip=True, bwd=(x, bn_save)) and preact_wgrad_partial(x, bn_save, dy, w.shape, ws)
        in ONE launch (their workgroups interleaved); returns conv_wgrad_reduce's record for the weight gradient."""
        if dy.dim() != 4 or w.dim() != 4 or tuple(w.shape[2:]) != (3, 3) or w.shape[0] != dy.shape[1]:
            raise ValueError(f'not a 3x3 convolution: dy {tuple(dy.shape)}, w {tuple(w.shape)}')
        N, Cd, H, W = dy.shape
        Cx = w.shape[1]
        dev = dy.device
        shape = (N, Cx, H * stride, W * stride)
        if tuple(x.shape) != shape or tuple(g.shape) != shape:
            raise ValueError(f'x {tuple(x.shape)} / g {tuple(g.shape)} should be {shape}')
        geo = self.preact_geometry(dy.shape, Cx, flip=True, stride=stride)
        if geo is None:
            raise ValueError(f'K10 does not cover the backward of dy {tuple(dy.shape)} -> {Cx} channels (stride {stride})')
        with torch.cuda.device(dev):
            rc = self.lib.ursa_preact_bwd_pair_f32(_ptr(dy, 'dy'), _ptr(w, 'w', None, dev), _ptr(g, 'g', None, dev), _ptr(x, 'x', None, dev),
                                                   _ptr(bn_save, 'bn_save', 4 * Cx, dev), self._f64ptr(out_partial, 'out_partial', Cx * geo[0] * 2, dev),
                                                   _ptr(ws, 'ws', None, dev), ws.numel(), N, Cd, Cx, H, W, CONV_STRIDE2 if stride == 2 else 0,
                                                   _stream(dev))
        _check(self.lib, rc, 'ursa_preact_bwd_pair_f32')
        return (ws, N, Cx, Cd, H * stride, W * stride, 3, int(stride))

    def bn_apply(self, x, y, partial, gamma, beta, running_mean, running_var, save, *, eps, momentum, relu=True):
        """K6's normalise launch alone, statistics from a convolution launch's partial sums (float64 [C, nl, 2])."""
        N, C, HW = self._bn_dims(x)
        dev = x.device
        if partial.dim() != 3 or partial.shape[0] != C or partial.shape[2] != 2:
            raise ValueError(f'partial {tuple(partial.shape)} should be [{C}, nl, 2]')
        with torch.cuda.device(dev):
            rc = self.lib.ursa_bn_apply_f32(_ptr(x, 'x'), _ptr(y, 'y', x.numel(), dev), self._f64ptr(partial, 'partial', partial.numel(), dev),
                                            int(partial.shape[1]), _ptr(gamma, 'gamma', C, dev), _ptr(beta, 'beta', C, dev),
                                            _ptr(running_mean, 'running_mean', C, dev, optional=True),
                                            _ptr(running_var, 'running_var', C, dev, optional=True), _ptr(save, 'save', 4 * C, dev),
                                            N, C, HW, float(eps), float(momentum), BN_RELU if relu else 0, _stream(dev))
        _check(self.lib, rc, 'ursa_bn_apply_f32')

    def bn_bwd_dx(self, x, g, dx, gamma, save, partial, dgamma, dbeta, *, dz=None):
        """K6's dx launch alone: g is the gated output gradient, partial (float64 [C, nl, 2]) its sums from the K10 launch."""
        N, C, HW = self._bn_dims(x)
        dev, n = x.device, x.numel()
        if partial.dim() != 3 or partial.shape[0] != C or partial.shape[2] != 2:
            raise ValueError(f'partial {tuple(partial.shape)} should be [{C}, nl, 2]')
        with torch.cuda.device(dev):
            rc = self.lib.ursa_bn_bwd_dx_f32(_ptr(x, 'x'), _ptr(g, 'g', n, dev), _ptr(dz, 'dz', n, dev, optional=True), _ptr(dx, 'dx', n, dev),
                                             _ptr(gamma, 'gamma', C, dev), _ptr(save, 'save', 4 * C, dev),
                                             self._f64ptr(partial, 'partial', partial.numel(), dev), int(partial.shape[1]),
                                             _ptr(dgamma, 'dgamma', C, dev), _ptr(dbeta, 'dbeta', C, dev), N, C, HW, _stream(dev))
        _check(self.lib, rc, 'ursa_bn_bwd_dx_f32')


    # K11 -----------------------------------------------------------------------------
    def head_supported(self, z_shape, num_classes):
        """Whether the three K11 launches cover a final activation of `z_shape` ([N, C, 8, 8]) and `num_classes` outputs."""
        n, c, h, w = (int(v) for v in z_shape)
        return h == 8 and w == 8 and n <= 128 and bool(self.lib.ursa_fc_ce_supported(n, c, int(num_classes)))

    def bn_relu_pool(self, z, partial, gamma, beta, running_mean, running_var, save, pooled, *, eps, momentum):
        N, C, HW = self._bn_dims(z)
        dev = z.device
        with torch.cuda.device(dev):
            rc = self.lib.ursa_bn_relu_pool_f32(_ptr(z, 'z'), self._f64ptr(partial, 'partial', partial.numel(), dev), int(partial.shape[1]),
                                                _ptr(gamma, 'gamma', C, dev), _ptr(beta, 'beta', C, dev),
                                                _ptr(running_mean, 'running_mean', C, dev, optional=True),
                                                _ptr(running_var, 'running_var', C, dev, optional=True), _ptr(save, 'save', 4 * C, dev),
                                                _ptr(pooled, 'pooled', N * C, dev), N, C, HW, float(eps), float(momentum), _stream(dev))
        _check(self.lib, rc, 'ursa_bn_relu_pool_f32')

    def fc_ce(self, pooled, W, b, target, loss, dW, db, dpooled, *, logits=None, ignore_index=-100):
        N, C = pooled.shape
        K = W.shape[0]
        dev = pooled.device
        if not (isinstance(target, torch.Tensor) and target.is_cuda and target.device == dev and target.dtype == torch.int64
                and target.is_contiguous() and target.numel() == N):
            raise ValueError(f'target must be a contiguous int64 tensor of {N} labels on {dev}')
        with torch.cuda.device(dev):
            rc = self.lib.ursa_fc_ce_f32(_ptr(pooled, 'pooled'), _ptr(W, 'W', K * C, dev), _ptr(b, 'b', K, dev, optional=True), target.data_ptr(),
                                         _ptr(loss, 'loss', 1, dev), _ptr(logits, 'logits', N * K, dev, optional=True), _ptr(dW, 'dW', K * C, dev),
                                         _ptr(db, 'db', K, dev, optional=True), _ptr(dpooled, 'dpooled', N * C, dev), N, C, K, int(ignore_index),
                                         _stream(dev))
        _check(self.lib, rc, 'ursa_fc_ce_f32')

    def bn_relu_pool_bwd(self, z, dpooled, gamma, save, dz, dgamma, dbeta):
        N, C, HW = self._bn_dims(z)
        dev = z.device
        with torch.cuda.device(dev):
            rc = self.lib.ursa_bn_relu_pool_bwd_f32(_ptr(z, 'z'), _ptr(dpooled, 'dpooled', N * C, dev), _ptr(gamma, 'gamma', C, dev),
                                                    _ptr(save, 'save', 4 * C, dev), _ptr(dz, 'dz', z.numel(), dev), _ptr(dgamma, 'dgamma', C, dev),
                                                    _ptr(dbeta, 'dbeta', C, dev), N, C, HW, _stream(dev))
        _check(self.lib, rc, 'ursa_bn_relu_pool_bwd_f32')


def knobs_kernels():
    """A HipKernels bound to csrc/libursa_hip_knobs.so - the build whose kernel selection can be steered by URSA_*
    environment variables (tools/, A/B tests). Never used by the samplers or tasks."""
    return HipKernels(load_library(KNOBS_LIB_PATH))


_default = None


def default_kernels():
    """The process-wide HipKernels instance (raises NativeLibraryMissing if the library is not built)."""
    global _default
    if _default is None:
        _default = HipKernels()
    return _default
