"""ctypes front end of oracle/liboracle.so for the tests (numpy in, numpy out).

The oracle is the CHECKER: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg import this module. Nothing under ursabench_amd/ does.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_L = ctypes.CDLL(os.path.join(ROOT, 'oracle', 'liboracle.so'))

STEP_NOISE, STEP_FIRST, STEP_ZERO_GRAD, STEP_WD, STEP_SGD = 1, 2, 4, 8, 16
BMA_SMOOTHED = 1
LEAP_KICK, LEAP_DRIFT = 1, 2

_vp, _i64, _i32, _u64, _u32, _f = (ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint32,
                                   ctypes.c_float)
_L.oracle_sgmcmc_step_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _u64, _u64, _u32]
_L.oracle_philox_normal_f32.argtypes = [_vp, _i64, _u64, _u64]
_L.oracle_philox_normal_f32.restype = None
_L.oracle_philox_normal_range_f32.argtypes = [_vp, _i64, _i64, _u64, _u64]
_L.oracle_philox_normal_range_f32.restype = None
_L.oracle_philox4x32_10.restype = None
_L.oracle_swag_collect_f32.argtypes = [_vp, _vp, _vp, _i64, _f, _f]
_L.oracle_swag_draw_f32.argtypes = [_vp, _vp, _vp, _vp, _i64, _f, _f, _u64, _u64]
_L.oracle_bma_accumulate_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _f, _f, _u32]
_L.oracle_leapfrog_f32.argtypes = [_vp, _vp, _vp, _i64, _f, _f, _f, _u32, _vp]
_L.oracle_sumsq_f32.argtypes = [_vp, _i64, _vp]
_L.oracle_bn_relu_fwd_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f, _f, _i32]
_L.oracle_bn_relu_bwd_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32]
_L.oracle_bn_relu_bwd_gated_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i32, _vp, _vp, _i64]
_L.oracle_bn_relu_eval_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f, _i32]


def _p(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags['C_CONTIGUOUS'], (a.dtype, a.flags)
    return a.ctypes.data


def philox4x32_10(ctr, key):
    c = (ctypes.c_uint32 * 4)(*ctr)
    k = (ctypes.c_uint32 * 2)(*key)
    o = (ctypes.c_uint32 * 4)()
    _L.oracle_philox4x32_10(c, k, o)
    return list(o)


def philox_normal(n, seed, step):
    out = np.empty(n, np.float32)
    _L.oracle_philox_normal_f32(_p(out), n, seed, step)
    return out


def philox_normal_range(start, count, seed, step):
    assert start % 4 == 0
    out = np.empty(count, np.float32)
    _L.oracle_philox_normal_range_f32(_p(out), start, count, seed, step)
    return out


def sgmcmc_step(theta, grad, mom, *, lr, mu, c_wd, c_noise, n_train, flags, seed=0, step=0, eps=None, snapshot=None):
    """In place on numpy arrays, same argument meaning as ursa_sgmcmc_step_f32."""
    rc = _L.oracle_sgmcmc_step_f32(_p(theta), _p(grad), _p(mom), _p(eps), _p(snapshot), theta.size, lr, mu, c_wd,
                                   c_noise, n_train, seed, step, flags)
    assert rc == 0


def swag_collect(mean, sq, w, *, decay, denom):
    assert _L.oracle_swag_collect_f32(_p(mean), _p(sq), _p(w), mean.size, decay, denom) == 0


def swag_draw(out, mean, sq, *, var_clamp, scale=1.0, seed=0, draw=0, eps=None):
    assert _L.oracle_swag_draw_f32(_p(out), _p(mean), _p(sq), _p(eps), out.size, var_clamp, scale, seed, draw) == 0


def bma_accumulate(logits, proba_sum, ent_sum=None, *, one_minus_gamma, gamma_over_c, smoothed, risk_sum=None, cost=None):
    S, B, C = logits.shape
    rc = _L.oracle_bma_accumulate_f32(_p(logits), _p(proba_sum), _p(ent_sum), _p(risk_sum), _p(cost), S, B, C,
                                      one_minus_gamma, gamma_over_c, BMA_SMOOTHED if smoothed else 0)
    assert rc == 0


def leapfrog(theta, mom, grad, *, kick_coef, step_size, inv_mass, flags, want_kinetic=False):
    ke = ctypes.c_double(0.0)
    rc = _L.oracle_leapfrog_f32(_p(theta), _p(mom), _p(grad), mom.size, kick_coef, step_size, inv_mass, flags,
                                ctypes.byref(ke) if want_kinetic else None)
    assert rc == 0
    return ke.value


def sumsq(x):
    out = ctypes.c_double(0.0)
    assert _L.oracle_sumsq_f32(_p(x), x.size, ctypes.byref(out)) == 0
    return out.value


def step_scalars(lr, momentum, weight_decay, n_train):
    """Host-side scalar preparation, float64 -> fp32 exactly as the reference's Python does
    (optim_sghmc.py:48,53,64): returns kwargs for sgmcmc_step."""
    import math
    return dict(lr=lr, mu=momentum, c_wd=(weight_decay / n_train) if weight_decay != 0 else 0.0,
                c_noise=math.sqrt(2 * (1 - momentum) * lr), n_train=float(n_train))


def _bn_dims(x):
    N, C = x.shape[0], x.shape[1]
    return N, C, x.size // (N * C)


def bn_relu_fwd(x, gamma, beta, running_mean=None, running_var=None, *, eps=1e-5, momentum=0.1, relu=True):
    """K6 forward: returns (y, save_mean, save_invstd); running statistics are updated in place when given."""
    N, C, HW = _bn_dims(x)
    y, sm, si = np.empty_like(x), np.empty(C, np.float32), np.empty(C, np.float32)
    rc = _L.oracle_bn_relu_fwd_f32(_p(x), _p(y), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(sm), _p(si),
                                   N, C, HW, eps, momentum, int(relu))
    assert rc == 0
    return y, sm, si


def bn_relu_bwd(x, dy, gamma, beta, save_mean, save_invstd, *, relu=True, gates=None):
    """K6 backward: returns (dx, dgamma, dbeta). gates=(idx int32 ascending, open uint8): listed elements' ReLU gates given."""
    N, C, HW = _bn_dims(x)
    dx, dg, db = np.empty_like(x), np.empty(C, np.float32), np.empty(C, np.float32)
    if gates is not None:
        gi, go = np.ascontiguousarray(gates[0], np.int32), np.ascontiguousarray(gates[1], np.uint8)
        rc = _L.oracle_bn_relu_bwd_gated_f32(_p(x), _p(dy), _p(dx), _p(gamma), _p(beta), _p(save_mean), _p(save_invstd),
                                             _p(dg), _p(db), N, C, HW, int(relu), gi.ctypes.data, go.ctypes.data, gi.size)
        assert rc == 0
        return dx, dg, db
    rc = _L.oracle_bn_relu_bwd_f32(_p(x), _p(dy), _p(dx), _p(gamma), _p(beta), _p(save_mean), _p(save_invstd), _p(dg),
                                   _p(db), N, C, HW, int(relu))
    assert rc == 0
    return dx, dg, db


def bn_relu_eval(x, gamma, beta, running_mean, running_var, *, eps=1e-5, relu=True):
    N, C, HW = _bn_dims(x)
    y = np.empty_like(x)
    rc = _L.oracle_bn_relu_eval_f32(_p(x), _p(y), _p(gamma), _p(beta), _p(running_mean), _p(running_var), N, C, HW, eps,
                                    int(relu))
    assert rc == 0
    return y


_L.oracle_conv_wgrad_f32.argtypes = [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _i64]


def conv_wgrad(x, dy, ksize=3, stride=1):
    """dw[Cout, Cin, k, k] of a k x k / pad k // 2 convolution: x [N, Cin, H, W], dy [N, Cout, H/stride, W/stride]."""
    N, Cin, H, W = x.shape
    Cout = dy.shape[1]
    assert dy.shape == (N, Cout, H // stride, W // stride)
    dw = np.empty((Cout, Cin, ksize, ksize), np.float32)
    rc = _L.oracle_conv_wgrad_f32(_p(x), _p(dy), _p(dw), N, Cin, Cout, H, W, ksize, stride)
    assert rc == 0
    return dw


_L.oracle_conv3x3_f32.argtypes = [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32, _i64]


def conv3x3(x, w, flip=False, stride=1):
    """y = conv2d(x, w, stride, pad 1); flip: the input gradient for the output gradient `x` of the layer whose weight is `w`."""
    N, Cin, H, W = x.shape
    Cout = w.shape[1 if flip else 0]
    assert w.shape[0 if flip else 1] == Cin and w.shape[2:] == (3, 3)
    y = np.empty((N, Cout, H * stride, W * stride) if flip else (N, Cout, H // stride, W // stride), np.float32)
    rc = _L.oracle_conv3x3_f32(_p(x), _p(w), _p(y), N, Cin, Cout, H, W, int(flip), stride)
    assert rc == 0
    return y


_L.oracle_conv1x1s2_f32.argtypes = [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i32]


def conv1x1s2(x, w, flip=False):
    """y = conv2d(x, w, stride 2) for w [Cout, Cin, 1, 1]; flip: the input gradient for the output gradient `x`."""
    N, Cin, H, W = x.shape
    Cout = w.shape[1 if flip else 0]
    assert w.shape[0 if flip else 1] == Cin and w.shape[2:] == (1, 1)
    y = np.empty((N, Cout, 2 * H, 2 * W) if flip else (N, Cout, H // 2, W // 2), np.float32)
    rc = _L.oracle_conv1x1s2_f32(_p(x), _p(np.ascontiguousarray(w.reshape(w.shape[0], w.shape[1]))), _p(y), N, Cin, Cout, H, W, int(flip))
    assert rc == 0
    return y


_L.oracle_preact_fwd_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _f, _f, _i32]
_L.oracle_preact_bwd_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64]
_L.oracle_bn_bwd_dx_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64]


def preact_fwd(x, w, *, bn=None, addend=None, stride=1, eps=1e-5, momentum=0.1, running=None):
    """K10 forward: y = conv3x3(relu(bn(x)), w, stride) (+ addend). bn = (gamma, beta) or None (the stem); running = (mean, var)
    updated in place. Returns (y, sums float64 [Cout, 2] = (sum y, sum y^2), save [4, Cin] | None)."""
    N, Cin, H, W = x.shape
    Cout = w.shape[0]
    y = np.empty((N, Cout, H // stride, W // stride), np.float32)
    sums = np.empty((Cout, 2), np.float64)
    save = np.empty((4, Cin), np.float32) if bn is not None else None
    h = np.empty_like(x) if bn is not None else None
    gamma, beta = bn if bn is not None else (None, None)
    rm, rv = running if running is not None else (None, None)
    rc = _L.oracle_preact_fwd_f32(_p(x), _p(w), _p(addend), _p(y), _p(h), _p(gamma), _p(beta), _p(rm), _p(rv), _p(save), sums.ctypes.data,
                                  N, Cin, Cout, H, W, stride, eps, momentum, int(bn is not None))
    assert rc == 0
    return y, sums, save


def preact_bwd(dy, w, xin, save, *, stride=1):
    """K10 backward, first half: (g, sums float64 [Cx, 2]) - the gated input gradient and (sum g, sum g * (xin - mean))."""
    N, Cd, H, W = dy.shape
    Cx = xin.shape[1]
    assert xin.shape == (N, Cx, H * stride, W * stride) and w.shape == (Cd, Cx, 3, 3) and save.shape == (4, Cx)
    g = np.empty_like(xin)
    sums = np.empty((Cx, 2), np.float64)
    rc = _L.oracle_preact_bwd_f32(_p(dy), _p(w), _p(xin), _p(save), _p(g), sums.ctypes.data, N, Cd, Cx, H, W, stride)
    assert rc == 0
    return g, sums


def bn_bwd_dx(x, g, gamma, save, sums, dz=None):
    """K10 backward, second half: (dx, dgamma, dbeta) from the gated gradient and its sums."""
    N, C, HW = _bn_dims(x)
    dx, dg, db = np.empty_like(x), np.empty(C, np.float32), np.empty(C, np.float32)
    rc = _L.oracle_bn_bwd_dx_f32(_p(x), _p(g), _p(dz), _p(dx), _p(gamma), _p(save), np.ascontiguousarray(sums).ctypes.data, _p(dg), _p(db),
                                 N, C, HW)
    assert rc == 0
    return dx, dg, db


_L.oracle_bn_relu_pool_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _f, _f]
_L.oracle_fc_ce_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64]
_L.oracle_bn_relu_pool_bwd_f32.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64]


def bn_relu_pool(z, gamma, beta, *, eps=1e-5, momentum=0.1, running=None):
    """K11: (pooled [N, C], save [4, C]) = mean over the map of relu(batch_norm(z)), training mode."""
    N, C, HW = _bn_dims(z)
    h, save, pooled = np.empty_like(z), np.empty((4, C), np.float32), np.empty((N, C), np.float32)
    rm, rv = running if running is not None else (None, None)
    rc = _L.oracle_bn_relu_pool_f32(_p(z), _p(h), _p(gamma), _p(beta), _p(rm), _p(rv), _p(save), _p(pooled), N, C, HW, eps, momentum)
    assert rc == 0
    return pooled, save


def fc_ce(p, W, b, target, ignore_index=-100):
    """K11: (loss, logits, dW, db, dp) of mean cross entropy over fc(p)."""
    N, C = p.shape
    K = W.shape[0]
    t = np.ascontiguousarray(target, np.int64)
    loss, logits = np.empty(1, np.float32), np.empty((N, K), np.float32)
    dW, db, dp = np.empty_like(W), (None if b is None else np.empty_like(b)), np.empty_like(p)
    rc = _L.oracle_fc_ce_f32(_p(p), _p(W), _p(b), t.ctypes.data, _p(loss), _p(logits), _p(dW), _p(db), _p(dp), N, C, K, ignore_index)
    assert rc == 0
    return float(loss[0]), logits, dW, db, dp


def bn_relu_pool_bwd(z, dpooled, gamma, beta, save):
    """K11: (dz, dgamma, dbeta) of bn_relu_pool for the pooled gradient."""
    N, C, HW = _bn_dims(z)
    dy, dz, dg, db = np.empty_like(z), np.empty_like(z), np.empty(C, np.float32), np.empty(C, np.float32)
    rc = _L.oracle_bn_relu_pool_bwd_f32(_p(z), _p(dpooled), _p(dy), _p(gamma), _p(beta), _p(np.ascontiguousarray(save)), _p(dz), _p(dg), _p(db), N, C, HW)
    assert rc == 0
    return dz, dg, db


_L.oracle_conv1x1_f32.argtypes = [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i32]


def conv1x1(x, w, flip=False):
    """K12: y = conv2d(x, w) for w [Cout, Cin, 1, 1]; flip: the input gradient for the output gradient `x`."""
    N, Cin, H, W = x.shape
    Cout = w.shape[1 if flip else 0]
    assert w.shape[0 if flip else 1] == Cin and w.shape[2:] == (1, 1)
    y = np.empty((N, Cout, H, W), np.float32)
    rc = _L.oracle_conv1x1_f32(_p(x), _p(np.ascontiguousarray(w.reshape(w.shape[0], w.shape[1]))), _p(y), N, Cin, Cout, H * W, int(flip))
    assert rc == 0
    return y
