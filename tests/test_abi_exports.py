"""CPU: the C-ABI library loads and exports every symbol include/ursa_hip.h declares
(no compute calls without a GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(knobs=False):
    """Function names the header declares: the product ABI (knobs=False: everything outside `#ifdef URSA_DEBUG_KNOBS`) or the
    experiment-only entry points inside it (knobs=True)."""
    src = open(os.path.join(ROOT, 'include', 'ursa_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    inside = ''.join(re.findall(r'#ifdef URSA_DEBUG_KNOBS(.*?)#endif', src, flags=re.S))
    outside = re.sub(r'#ifdef URSA_DEBUG_KNOBS.*?#endif', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(ursa_[a-z0-9_]+)\s*\(', inside if knobs else outside)))


def test_library_exports_every_declared_symbol():
    from ursabench_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        pytest.fail(f'{_native.LIB_PATH} is not built: run __graft_entry__.build()')
    lib = _native.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 11
    for sym in declared:
        assert hasattr(lib, sym), f'{sym} declared in include/ursa_hip.h but not exported'
    assert sorted(_native.SIGNATURES) == declared, 'binding table and header disagree'
    assert sorted(_native.KNOBS_SIGNATURES) == _declared_symbols(knobs=True)
    assert lib.ursa_abi_version() == _native.ABI_VERSION
    assert lib.ursa_strerror(-1).decode() == 'required pointer is NULL'


def test_argument_errors_do_not_need_a_gpu():
    """Argument validation happens before any launch, so it can be exercised on CPU."""
    from ursabench_amd import _native
    lib = _native.load_library()
    assert lib.ursa_sgmcmc_step_f32(None, None, None, None, None, -1, 0, 0, 0, 0, 1, 0, 0, 0, None) == -2   # ESIZE
    assert lib.ursa_sgmcmc_step_f32(None, None, None, None, None, 8, 0, 0, 0, 0, 1, 0, 0, 0, None) == -1    # ENULL
    assert lib.ursa_sgmcmc_step_f32(None, None, None, None, None, 8, 0, 0, 0, 0, 1, 0, 0, 0x100, None) == -4  # EFLAGS
    assert lib.ursa_sgmcmc_step_f32(None, None, None, None, None, 0, 0, 0, 0, 0, 1, 0, 0, 0, None) == 0     # n == 0: no-op
    assert lib.ursa_bma_accumulate_f32(None, None, None, None, None, 1, 1, 0, 0, 0, 0, None) == -5           # C out of range
    assert lib.ursa_bma_accumulate_f32(None, None, None, None, None, 1, 1, 2000, 0, 0, 0, None) == -5
    # K6: flags, sizes, NULLs, the addend / z_out pairing and torch's "more than 1 value per channel" rule
    import ctypes
    buf = (ctypes.c_float * 64)()                       # a 16-byte aligned HOST address: never dereferenced before the checks
    p = ctypes.addressof(buf)
    p -= p % 16
    fwd, bwd, ev = lib.ursa_bn_relu_fwd_f32, lib.ursa_bn_relu_bwd_f32, lib.ursa_bn_relu_eval_f32
    assert fwd(None, None, None, None, None, None, None, None, None, None, None, None, 2, 3, 4, 1e-5, 0.1, 0x8, None) == -4   # EFLAGS
    assert fwd(p, None, None, p, p, p, None, None, p, p, None, p, -1, 3, 4, 1e-5, 0.1, 0, None) == -2                         # ESIZE
    assert fwd(None, None, None, None, None, None, None, None, None, None, None, None, 0, 3, 4, 1e-5, 0.1, 0, None) == 0      # empty: no-op
    assert fwd(None, None, None, None, None, None, None, None, None, None, None, None, 2, 3, 4, 1e-5, 0.1, 1, None) == -1     # ENULL
    assert fwd(p, p, None, p, p, p, None, None, p, p, None, p, 2, 3, 4, 1e-5, 0.1, 1, None) == -1          # addend without z_out
    assert fwd(p, None, None, p, p, p, p, None, p, p, None, p, 2, 3, 4, 1e-5, 0.1, 1, None) == -1          # running_mean without running_var
    assert fwd(p, None, None, p, p, p, None, None, p, p, p, p, 1, 3, 1, 1e-5, 0.1, 1, None) == -5       # one value per channel
    assert fwd(p, None, None, p, p, p, None, None, p, p, None, p, 2, 70000, 4, 1e-5, 0.1, 1, None) == -2   # channels beyond the grid
    assert fwd(p + 2, None, None, p, p, p, None, None, p, p, None, p, 2, 3, 4, 1e-5, 0.1, 1, None) == -3   # EALIGN
    assert fwd(p, None, None, p, p, p, None, None, p, p, p + 2, p, 2, 3, 4, 1e-5, 0.1, 1, None) == -3   # save_gate misaligned
    assert bwd(None, None, None, None, None, None, None, None, None, None, None, None, 2, 3, 4, 1, None) == -1
    assert bwd(None, None, None, None, None, None, None, None, None, None, None, None, 2, 3, 4, 0x10, None) == -4
    assert ev(None, None, None, None, None, None, None, None, 2, 3, 4, 1e-5, 1, None) == -1
    assert ev(p, None, p, p, p, p, p, p, 2, 3, 4, 1e-5, 1, None) == -1                              # z_out without addend
    # the gated backward (parity instrument): list pointers, list size, RELU required, 32-bit element offsets
    gb = lib.ursa_bn_relu_bwd_gated_f32
    assert gb(p, p, None, p, p, p, p, p, None, p, p, p, 2, 3, 4, 1, None, None, 5, None) == -1               # list pointers missing
    assert gb(p, p, None, p, p, p, p, p, None, p, p, p, 2, 3, 4, 1, p, p, -1, None) == -2                    # negative list size
    assert gb(p, p, None, p, p, p, p, p, None, p, p, p, 2, 3, 4, 0, p, p, 1, None) == -4                     # without URSA_BN_RELU
    assert gb(p, p, None, p, p, p, p, p, None, p, p, p, 1 << 20, 64, 1 << 10, 1, p, p, 1, None) == -2        # 2^36 elements: offsets do not fit
    assert gb(None, None, None, None, None, None, None, None, None, None, None, None, 0, 3, 4, 1, None, None, 0, None) == 0


def test_the_shipped_library_reads_no_environment():
    """Kernel selection by URSA_* environment variables exists only in the -DURSA_DEBUG_KNOBS build
    (csrc/libursa_hip_knobs.so: tests' A/B cases and tools/): the shipped library does not import getenv at all."""
    import subprocess
    from ursabench_amd import _native
    syms = subprocess.run(['nm', '-D', '--undefined-only', _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert 'getenv' not in syms
    if os.path.exists(_native.KNOBS_LIB_PATH):
        knobs = subprocess.run(['nm', '-D', '--undefined-only', _native.KNOBS_LIB_PATH], capture_output=True, text=True, check=True).stdout
        assert 'getenv' in knobs
        assert _native.load_library(_native.KNOBS_LIB_PATH).ursa_abi_version() == _native.ABI_VERSION


def test_parked_experiments_are_not_in_the_product_abi():
    """the NHWC-twin launches (measured -2 % on the workload) live in the knobs build only: `nm` of
    the shipped library shows no nhwc symbol - neither an entry point nor a kernel - and nothing under ursabench_amd/ names them
    outside the binding's knobs table."""
    import subprocess
    from ursabench_amd import _native
    syms = subprocess.run(['nm', '-D', '--defined-only', _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert 'nhwc' not in syms.lower() and 'k_bn_fwd_apply_4' not in syms and 'k_bn_bwd_dx_4' not in syms
    lib = _native.load_library()
    for sym in _declared_symbols(knobs=True):
        assert not hasattr(lib, sym), f'{sym} is exported by the shipped library'
    if os.path.exists(_native.KNOBS_LIB_PATH):
        knobs = _native.load_library(_native.KNOBS_LIB_PATH)
        assert all(hasattr(knobs, sym) for sym in _declared_symbols(knobs=True))
    pkg = os.path.join(ROOT, 'ursabench_amd')
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py') and f != '_native.py':
                assert 'nhwc' not in open(os.path.join(d, f)).read().lower(), os.path.join(d, f)
    # (fused_conv.py is K7's module since round 5 - the weight-gradient kernel; the twins' module of that name is gone)
    src = open(os.path.join(pkg, 'fused_conv.py')).read()
    assert 'channels_last' not in src and 'twin' not in src.lower()


def test_wrappers_refuse_cpu_tensors():
    import torch
    from ursabench_amd import _native
    k = _native.default_kernels()
    t = torch.zeros(8)
    with pytest.raises(RuntimeError, match='no CPU path'):
        k.sgmcmc_step(t, t.clone(), t.clone(), lr=0.1, mu=0.5, c_wd=0.0, c_noise=0.1, n_train=10.0, flags=0)


def test_missing_library_fails_loudly(tmp_path):
    from ursabench_amd import _native
    with pytest.raises(_native.NativeLibraryMissing, match='no CPU fallback'):
        _native.load_library(str(tmp_path / 'nope.so'))
