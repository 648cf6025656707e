// ursa_bn.hip — K6: BatchNorm2d (+ ReLU) of the benchmark networks' pre-activation blocks, NCHW fp32, gfx950.
//
// Every BatchNorm of PreResNet / WideResNet is followed by a ReLU (URSABench/models/preresnet.py:40-41,45-46,
// 76-85,146; wideresnet.py:47,49,117). Stock PyTorch-ROCm runs that pair as 2-4 MIOpen launches + 1 ATen clamp
// forward and 2-4 MIOpen launches + 1 ATen threshold backward per layer: 31 % of the kernel time of a PreResNet-20
// training step (profiles/r03_bench_kernel_stats.csv), at 5-19 us per launch on 2-8 MB activations. Here: two
// launches forward (per-channel partial statistics; normalise + ReLU), two backward (partial sums of dy' and
// dy'*xhat with the ReLU mask recomputed from x; dx), one in evaluation mode. All are HBM / cache streaming
// kernels: float4 per lane, 256-thread workgroups, grid = (splits, channels) with the channel's N*H*W run cut into
// `splits` contiguous chunks so that >= 512 workgroups stream even for 16-channel layers.
//
// Arithmetic = torch's CPU BatchNorm (the reference's path), bit for bit given the same x, mean and invstd:
//     alpha = invstd * gamma ; beta' = fma(-mean, alpha, beta) ; y = fma(x, alpha, beta')
// (found by probing torch 2.10's CPU kernel: 0 of 2,097,152 outputs differ with this form, 20-50 % with any other
// association). That matters beyond rounding: the ReLU that follows turns a last-bit difference of a
// pre-activation near zero into an open / closed gate, i.e. an O(1) change of that element's gradient.
// Statistics: torch's CPU kernel accumulates in double and its batch mean is the correctly rounded exact mean;
// here every thread accumulates sum x and sum x^2 in DOUBLE (gfx950's vector fp64 rate makes that free next to the
// loads), the workgroup partials {s1, s2} are doubles, and the prologue of EVERY workgroup of the second launch
// (<= 64 partials per channel: one wave) merges them in double in a fixed lane order: mean and invstd come out
// correctly rounded (mean always equal to torch's, invstd in > 90 % of channels - torch's own variance is not exact),
// all workgroups of a channel normalise with identical scalars, and no atomics, grid sync or third launch are
// needed. Deterministic. In double there is no cancellation in s2/n - mean^2 up to |mean| / std ~ 1e6.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/ursa_hip.h"

namespace {

constexpr int kBnBlock = 256;
constexpr int kBnMaxSplit = 64;     // partials per channel: merged by one wave
constexpr int kBnTargetWgs = 1024;  // 4 workgroups per CU

struct BnGeom {
    int C;            // channels
    int hw;           // H*W / V  (V = 4: float4 units, V = 1: floats)
    int hw_shift;     // log2(hw) if hw is a power of two, else -1
    int chunk;        // units per workgroup (multiple of kBnBlock)
    int64_t per_ch;   // N * hw
    // Not geometry, but every K6 kernel already receives this struct: the optional gate scalars (ursa_hip.h, `save_gate` / `gate`).
    // Forward: alpha_c = invstd_c * gamma_c and beta'_c = fma(-mean_c, alpha_c, beta_c) as THIS forward used them are stored to
    // gate_out[c], gate_out[C + c]; backward: with gate_in the ReLU gate is recomputed from those instead of from the live
    // gamma / beta - a parameter changed in place between forward and backward by a raw-pointer kernel (which autograd's version
    // counters cannot see) then cannot move a gate away from the one the forward took.
    float* gate_out;
    const float* gate_in;
    int cfirst;       // launch geometry: 0 = grid (splits, channels), 1 = grid (channels, splits) - see bn_channel_first()
};

template <int V> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<1> { using T = float; };

__device__ inline float comp(const float4& v, int k) { return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w; }
__device__ inline float comp(const float& v, int) { return v; }
__device__ inline void setc(float4& v, int k, float a) { if (k == 0) v.x = a; else if (k == 1) v.y = a; else if (k == 2) v.z = a; else v.w = a; }
__device__ inline void setc(float& v, int, float a) { v = a; }

__device__ inline float4 vadd(const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ inline float vadd(const float& a, const float& b) { return a + b; }

// unit index i of channel c -> unit offset in the NCHW tensor
__device__ inline int64_t bn_off(const BnGeom& g, int c, int64_t i)
{
    int64_t n;
    if (g.hw_shift >= 0) n = i >> g.hw_shift; else n = i / g.hw;
    const int64_t j = i - n * g.hw;
    return (n * g.C + c) * (int64_t)g.hw + j;
}

// Which workgroup handles which (channel, split): with cfirst the CHANNEL is the fast grid dimension, so workgroups that start
// together read adjacent H*W planes of the same images (consecutive channels are adjacent in NCHW) - the launch as a whole sweeps
// contiguous windows of the tensor - instead of one channel's chunks that sit N/S images (MBs) apart.
__device__ __forceinline__ int bn_channel(const BnGeom& g) { return g.cfirst ? (int)blockIdx.x : (int)blockIdx.y; }
__device__ __forceinline__ int bn_split(const BnGeom& g) { return g.cfirst ? (int)blockIdx.y : (int)blockIdx.x; }
__device__ __forceinline__ int bn_nsplit(const BnGeom& g) { return g.cfirst ? (int)gridDim.y : (int)gridDim.x; }

__device__ __forceinline__ void bn_save_gate(const BnGeom& g, int c, float alpha, float shift)
{
    if (g.gate_out) { g.gate_out[c] = alpha; g.gate_out[g.C + c] = shift; }
}
// scale / shift of the forward's y = fma(x, scale, shift): the saved ones if given, else the forward's own expressions on the
// live parameters (same bits as long as nobody changed them in between)
__device__ __forceinline__ void bn_gate_scalars(const BnGeom& g, int c, float mean, float invstd, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float& scale, float& shift)
{
    if (g.gate_in) { scale = g.gate_in[c]; shift = g.gate_in[g.C + c]; }
    else { scale = invstd * gamma[c]; shift = fmaf(-mean, scale, beta[c]); }
}

// Sum of a double over the 64 lanes of a wave on the VALU (DPP butterflies inside a 16-lane row, v_readlane across
// rows: no LDS round trip per step as with ds_bpermute-based __shfl_xor); every lane ends with the same bits, fixed tree.
template <int CTRL>
__device__ __forceinline__ float bn_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ double bn_wave_sum(double v)
{
    // the same tree on both 32-bit halves of the double
#define URSA_BN_DPP64(CTRL) do { \
        const long long b = __builtin_bit_cast(long long, v); \
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffll), CTRL, 0xF, 0xF, true); \
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true); \
        v += __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo); } while (0)
    URSA_BN_DPP64(0xB1);
    URSA_BN_DPP64(0x4E);
    URSA_BN_DPP64(0x141);
    URSA_BN_DPP64(0x140);
#undef URSA_BN_DPP64
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = (int)(b & 0xffffffffll), hi = (int)(b >> 32);
    double r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        r[k] = __builtin_bit_cast(double, ((long long)__builtin_amdgcn_readlane(hi, 16 * k) << 32) |
                                              (unsigned int)__builtin_amdgcn_readlane(lo, 16 * k));
    return (r[0] + r[1]) + (r[2] + r[3]);
}

// sums of two values over the workgroup, same bits in every thread, fixed order; sh: 2 * kBnBlock/64 floats
template <class F>
__device__ __forceinline__ void bn_block_sum2(F& a, F& b, F* sh)
{
    a = bn_wave_sum(a);
    b = bn_wave_sum(b);
    if ((threadIdx.x & 63) == 0) { sh[2 * (threadIdx.x >> 6)] = a; sh[2 * (threadIdx.x >> 6) + 1] = b; }
    __syncthreads();
    a = b = F(0);
#pragma unroll
    for (int w = 0; w < kBnBlock / 64; ++w) { a += sh[2 * w]; b += sh[2 * w + 1]; }
}

__device__ inline float bn_relu_fwd(float v) { return v < 0.f ? 0.f : v; }   // NaN stays NaN

// Streaming accesses for tensors beyond the Infinity Cache (NT = true): nontemporal loads / stores
// (per-component builtins: the compiler merges them into one global_load/store_dwordx4 ... nt, as in ursa_kernels.hip)
template <bool NT> __device__ __forceinline__ float4 bn_ld(const float4* p)
{
    if (!NT) return *p;
    float4 r;
    r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y);
    r.z = __builtin_nontemporal_load(&p->z); r.w = __builtin_nontemporal_load(&p->w);
    return r;
}
template <bool NT> __device__ __forceinline__ void bn_st(float4* p, const float4& v)
{
    if (!NT) { *p = v; return; }
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
}


// ---- forward, launch 1: partial statistics (ADD: of z = x + addend, the residual sum, which is written out) --------
template <int V, bool ADD>
__global__ __launch_bounds__(kBnBlock) void k_bn_stats(const float* __restrict__ x, const float* __restrict__ addend,
                                                       float* __restrict__ z, double2* __restrict__ partial, BnGeom g)
{
    using T = typename Vec<V>::T;
    __shared__ double sh[2 * kBnBlock / 64];
    const T* __restrict__ xv = reinterpret_cast<const T*>(x);
    const T* __restrict__ av = reinterpret_cast<const T*>(addend);
    T* __restrict__ zv = reinterpret_cast<T*>(z);
    const int c = bn_channel(g);
    const int64_t lo = (int64_t)bn_split(g) * g.chunk;
    const int64_t hi = lo + g.chunk < g.per_ch ? lo + g.chunk : g.per_ch;
    double s1 = 0.0, s2 = 0.0;
    // batches of four (ADD: eight) loads in flight, the last batch predicated: a chunk of two units per thread (the workload's
    // 32x32 layers: 64 splits of a 16-channel activation) issues both loads at once instead of one after the other (round 4's
    // tail loop). Accumulation order per thread is unchanged (ascending unit index): the same doubles.
    for (int64_t i = lo + threadIdx.x; i < hi; i += 4 * kBnBlock) {
        T v[4], w[4];
        int64_t o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t iu = i + u * kBnBlock;
            if (iu < hi) { o[u] = bn_off(g, c, iu); v[u] = xv[o[u]]; if (ADD) w[u] = av[o[u]]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u * kBnBlock < hi) {
                if (ADD) { v[u] = vadd(v[u], w[u]); zv[o[u]] = v[u]; }
#pragma unroll
                for (int k = 0; k < V; ++k) { const double d = (double)comp(v[u], k); s1 += d; s2 = fma(d, d, s2); }
            }
        }
    }
    bn_block_sum2(s1, s2, sh);
    if (threadIdx.x == 0) partial[(int64_t)c * bn_nsplit(g) + bn_split(g)] = make_double2(s1, s2);
}

// Merge of a channel's partials by wave 0 of the calling workgroup, in double; same result in lanes 0..63.
__device__ inline void bn_merge(const double2* __restrict__ partial, int c, int S, double n, double& mean, double& var)
{
    double a = 0.0, b = 0.0;
    if ((int)threadIdx.x < S) { const double2 p = partial[(int64_t)c * S + threadIdx.x]; a = p.x; b = p.y; }
    a = bn_wave_sum(a);
    b = bn_wave_sum(b);
    mean = a / n;
    var = b / n - mean * mean;
    if (var < 0.0) var = 0.0;
}

// ---- forward, launch 2: merge, normalise (+ ReLU), running statistics ------------------------------------------
template <int V, bool RELU>
__global__ __launch_bounds__(kBnBlock) void k_bn_fwd_apply(const float* __restrict__ x, float* __restrict__ y,
                                                           const double2* __restrict__ partial, int S,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                           float eps, float momentum, BnGeom g)
{
    using T = typename Vec<V>::T;
    __shared__ float sh[2];
    const int c = bn_channel(g);
    const T* __restrict__ xv = reinterpret_cast<const T*>(x);
    T* __restrict__ yv = reinterpret_cast<T*>(y);
    const int64_t lo = (int64_t)bn_split(g) * g.chunk;
    const int64_t hi = lo + g.chunk < g.per_ch ? lo + g.chunk : g.per_ch;
    // The first batch of x loads does not depend on the statistics: issue it BEFORE the merge prologue, so that the prologue's
    // own dependent chain (partials load -> wave sums -> LDS -> barrier, ~1 us) runs under the loads' latency instead of in
    // front of it (round 4 merged first). At the workload's layers a workgroup's whole chunk is this one batch.
    const int64_t i0 = lo + threadIdx.x;
    T v0[4];
    int64_t o0[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int64_t iu = i0 + u * kBnBlock;
        if (iu < hi) { o0[u] = bn_off(g, c, iu); v0[u] = xv[o0[u]]; }
    }
    if (threadIdx.x < 64) {
        double mean, var;
        const double n = (double)g.per_ch * V;
        bn_merge(partial, c, S, n, mean, var);
        if (threadIdx.x == 0) {
            const float meanf = (float)mean;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float alpha = invstd * gamma[c];
            sh[0] = alpha;
            sh[1] = fmaf(-meanf, alpha, beta[c]);
            if (bn_split(g) == 0) {
                save_mean[c] = meanf;
                save_invstd[c] = invstd;
                bn_save_gate(g, c, alpha, sh[1]);
                if (running_mean) {      // torch: running = momentum * batch + (1 - momentum) * running, unbiased variance
                    running_mean[c] = momentum * meanf + (1.0f - momentum) * running_mean[c];
                    running_var[c] = momentum * (float)(var * (n / (n - 1.0))) + (1.0f - momentum) * running_var[c];
                }
            }
        }
    }
    __syncthreads();
    const float scale = sh[0], shift = sh[1];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (i0 + u * kBnBlock < hi) {
#pragma unroll
            for (int k = 0; k < V; ++k) { const float t = fmaf(comp(v0[u], k), scale, shift); setc(v0[u], k, RELU ? bn_relu_fwd(t) : t); }
            yv[o0[u]] = v0[u];
        }
    }
    for (int64_t i = i0 + 4 * kBnBlock; i < hi; i += 4 * kBnBlock) {
        T v[4];
        int64_t o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t iu = i + u * kBnBlock;
            if (iu < hi) { o[u] = bn_off(g, c, iu); v[u] = xv[o[u]]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i + u * kBnBlock < hi) {
#pragma unroll
                for (int k = 0; k < V; ++k) { const float t = fmaf(comp(v[u], k), scale, shift); setc(v[u], k, RELU ? bn_relu_fwd(t) : t); }
                yv[o[u]] = v[u];
            }
        }
    }
}

// ---- K13: the statistics launch's partial sums -> the saved [4][C] block, nothing normalised ---------------------------------
// What k_bn_fwd_apply's prologue computes, alone: one 64-thread workgroup per channel, the same merge (bn_merge), the same scalars
// and running statistics, so that a consumer which applies y = relu(fma(x, save[2][c], save[3][c])) itself (ursa_preact_conv1x1_f32
// while it stages x) produces the bits of K6's second launch.
__global__ __launch_bounds__(64) void k_bn_finalize(const double2* __restrict__ partial, int S, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ running_mean,
                                                    float* __restrict__ running_var, float* __restrict__ save, float eps, float momentum,
                                                    double n, int C)
{
    const int c = blockIdx.x;
    double mean, var;
    bn_merge(partial, c, S, n, mean, var);
    if (threadIdx.x == 0) {
        const float meanf = (float)mean;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float alpha = invstd * gamma[c];
        save[c] = meanf;
        save[C + c] = invstd;
        save[2 * C + c] = alpha;
        save[3 * C + c] = fmaf(-meanf, alpha, beta[c]);
        if (running_mean) {
            running_mean[c] = momentum * meanf + (1.0f - momentum) * running_mean[c];
            running_var[c] = momentum * (float)(var * (n / (n - 1.0))) + (1.0f - momentum) * running_var[c];
        }
    }
}

// ---- K14: a producer's per-workgroup sums of the BatchNorm backward -> the per-channel scalars of K6's dx expression --------
// What k_bn_bwd_dx<BIGS>'s prologue computes, alone (one workgroup per channel, every thread adds its partials in ascending order,
// then the workgroup's fixed tree): coef[0][c] = gm = sum / n, coef[1][c] = kk = dotp * invstd^2 / n, coef[2][c] = gamma,
// dbeta = sum, dgamma = dotp * invstd - for a consumer that applies
// dx = (((g - gm) - (x - mean) * kk) * invstd) * gamma itself (ursa_preact_conv1x1_bwd_dx_f32).
__global__ __launch_bounds__(kBnBlock) void k_bn_bwd_coef(const double2* __restrict__ partial, int S, const float* __restrict__ save,
                                                          const float* __restrict__ gamma, float* __restrict__ coef,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, double n, int C)
{
    __shared__ double shd[2 * kBnBlock / 64];
    const int c = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int t = threadIdx.x; t < S; t += kBnBlock) { const double2 p = partial[(int64_t)c * S + t]; a += p.x; b += p.y; }
    bn_block_sum2(a, b, shd);
    if (threadIdx.x == 0) {
        const double iv = (double)save[C + c];
        coef[c] = (float)(a / n);
        coef[C + c] = (float)(b * iv * iv / n);
        coef[2 * C + c] = gamma[c];
        dbeta[c] = (float)a;
        dgamma[c] = (float)(b * iv);
    }
}

template <bool NT> __device__ __forceinline__ float ev_ld(const float* p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ float4 ev_ld(const float4* p) { return bn_ld<NT>(p); }
template <bool NT> __device__ __forceinline__ void ev_st(float* p, const float& v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __device__ __forceinline__ void ev_st(float4* p, const float4& v) { bn_st<NT>(p, v); }

// ---- evaluation mode: y = relu(gamma * (x - running_mean) / sqrt(running_var + eps) + beta), one launch;
//      ADD: of z = x + addend, which is written out too -----------------------------------------------------------
template <int V, bool RELU, bool ADD, bool NT = false>
__global__ __launch_bounds__(kBnBlock) void k_bn_eval(const float* __restrict__ x, const float* __restrict__ addend,
                                                      float* __restrict__ z, float* __restrict__ y,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ running_mean,
                                                      const float* __restrict__ running_var, float eps, BnGeom g)
{
    using T = typename Vec<V>::T;
    const int c = bn_channel(g);
    const float invstd = 1.0f / sqrtf(running_var[c] + eps);
    const float scale = invstd * gamma[c];
    const float shift = fmaf(-running_mean[c], scale, beta[c]);
    const T* __restrict__ xv = reinterpret_cast<const T*>(x);
    const T* __restrict__ av = reinterpret_cast<const T*>(addend);
    T* __restrict__ zv = reinterpret_cast<T*>(z);
    T* __restrict__ yv = reinterpret_cast<T*>(y);
    const int64_t lo = (int64_t)bn_split(g) * g.chunk;
    const int64_t hi = lo + g.chunk < g.per_ch ? lo + g.chunk : g.per_ch;
    int64_t i = lo + threadIdx.x;
    for (; i + 3 * kBnBlock < hi; i += 4 * kBnBlock) {
        T v[4], w[4];
        int64_t o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { o[u] = bn_off(g, c, i + u * kBnBlock); v[u] = ev_ld<NT>(xv + o[u]); if (ADD) w[u] = ev_ld<NT>(av + o[u]); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (ADD) { v[u] = vadd(v[u], w[u]); ev_st<NT>(zv + o[u], v[u]); }
#pragma unroll
            for (int k = 0; k < V; ++k) { const float t = fmaf(comp(v[u], k), scale, shift); setc(v[u], k, RELU ? bn_relu_fwd(t) : t); }
            ev_st<NT>(yv + o[u], v[u]);
        }
    }
    for (; i < hi; i += kBnBlock) {
        const int64_t o = bn_off(g, c, i);
        T v = xv[o];
        if (ADD) { v = vadd(v, av[o]); zv[o] = v; }
#pragma unroll
        for (int k = 0; k < V; ++k) { const float t = fmaf(comp(v, k), scale, shift); setc(v, k, RELU ? bn_relu_fwd(t) : t); }
        yv[o] = v;
    }
}

// ---- evaluation mode, LINEAR form (round 5; float4 path): the launch needs no per-channel reduction, so nothing ties a
//      workgroup to a channel. The channel-grid kernel above walks channel c as N runs of H*W floats that sit C*H*W apart
//      (4 KB pieces at a 64 KB - 2.6 MB stride: 0.57 of the HBM peak at [128, 640, 8, 8], 0.69 at [1024, 64, 32, 32]); this one
//      streams the tensor front to back like K1 - workgroup b owns float4 [b * 1024, (b + 1) * 1024), one contiguous 16 KB - and
//      looks the channel of every float4 up: plane = float4 index / (H*W/4) = n * C + c. The (scale, shift) pairs of the planes a
//      workgroup touches (at most 1024 / (H*W/4) + 1) are built once per workgroup in LDS, AFTER the workgroup's loads have been
//      issued (their latency covers the table's dependent chain: parameter loads -> rsqrt -> LDS -> barrier). Same arithmetic
//      per element as k_bn_eval (and as torch's CPU kernel): invstd = 1 / sqrtf(var + eps); alpha = invstd * gamma;
//      beta' = fma(-mean, alpha, beta); y = fma(x, alpha, beta') - the same bits.
constexpr int kEvalU = 4;
constexpr int kEvalSpan = kBnBlock * kEvalU;
template <bool RELU, bool ADD, bool NT>
__global__ __launch_bounds__(kBnBlock) void k_bn_eval_lin(const float4* __restrict__ xv, const float4* __restrict__ av,
                                                          float4* __restrict__ zv, float4* __restrict__ yv,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ running_mean,
                                                          const float* __restrict__ running_var, float eps, int C, int hw4,
                                                          int hw_shift, int64_t total4)
{
    __shared__ float2 tab[kEvalSpan + 1];
    const int64_t base = (int64_t)blockIdx.x * kEvalSpan;
    const int64_t end = base + kEvalSpan < total4 ? base + kEvalSpan : total4;
    float4 v[kEvalU], w[kEvalU];
#pragma unroll
    for (int u = 0; u < kEvalU; ++u) {
        const int64_t i = base + threadIdx.x + u * kBnBlock;
        if (i < end) { v[u] = ev_ld<NT>(xv + i); if (ADD) w[u] = ev_ld<NT>(av + i); }
    }
    const int64_t p0 = hw_shift >= 0 ? (base >> hw_shift) : (base / hw4);
    const int64_t p1 = hw_shift >= 0 ? ((end - 1) >> hw_shift) : ((end - 1) / hw4);
    const int np = (int)(p1 - p0) + 1;
    for (int k = threadIdx.x; k < np; k += kBnBlock) {
        const int c = (int)((p0 + k) % C);
        const float invstd = 1.0f / sqrtf(running_var[c] + eps);
        const float scale = invstd * gamma[c];
        tab[k] = make_float2(scale, fmaf(-running_mean[c], scale, beta[c]));
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kEvalU; ++u) {
        const int64_t i = base + threadIdx.x + u * kBnBlock;
        if (i < end) {
            const float2 ss = tab[(int)((hw_shift >= 0 ? (i >> hw_shift) : (i / hw4)) - p0)];
            if (ADD) { v[u] = vadd(v[u], w[u]); ev_st<NT>(zv + i, v[u]); }
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float t = fmaf(comp(v[u], k), ss.x, ss.y); setc(v[u], k, RELU ? bn_relu_fwd(t) : t); }
            ev_st<NT>(yv + i, v[u]);
        }
    }
}

// ---- parity instrument (ursa_bn_relu_bwd_gated_f32): a sorted list of element offsets whose ReLU gate is GIVEN instead
//      of recomputed - the reference CPU run's gates at the pre-activations within rounding of zero, where MIOpen's and
//      oneDNN's convolutions (inputs of this layer) decide the sign by their last bits. Binary search per element: this
//      form is for comparisons against the reference, never for the timed path.
struct BnGates {
    const int32_t* idx;     // ascending element offsets into the [N, C, HW] tensor; entries == INT32_MAX are padding
    const uint8_t* open;    // 1: dy passes, 0: blocked
    int n;
};

__device__ inline bool bn_gate(bool computed, int64_t e, const BnGates& gt)
{
    int lo = 0, hi = gt.n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int64_t)gt.idx[mid] < e) lo = mid + 1; else hi = mid;
    }
    return (lo < gt.n && (int64_t)gt.idx[lo] == e) ? gt.open[lo] != 0 : computed;
}

// ---- backward, launch 1: partial sums of dy' and dy' * (x - mean), in double (dy' = dy where the ReLU was open) ------
template <int V, bool RELU, bool GATED = false>
__global__ __launch_bounds__(kBnBlock) void k_bn_bwd_reduce(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ save_mean,
                                                            const float* __restrict__ save_invstd,
                                                            double2* __restrict__ partial, BnGeom g, BnGates gt = BnGates{})
{
    using T = typename Vec<V>::T;
    __shared__ double sh[2 * kBnBlock / 64];
    const int c = bn_channel(g);
    const float mean = save_mean[c], invstd = save_invstd[c];
    float scale, shift;                                      // the forward's own scalars (saved, or its expressions on the live
    bn_gate_scalars(g, c, mean, invstd, gamma, beta, scale, shift);      // parameters): same bits, same gates
    const double meand = (double)mean;
    const T* __restrict__ xv = reinterpret_cast<const T*>(x);
    const T* __restrict__ dv = reinterpret_cast<const T*>(dy);
    const int64_t lo = (int64_t)bn_split(g) * g.chunk;
    const int64_t hi = lo + g.chunk < g.per_ch ? lo + g.chunk : g.per_ch;
    double s1 = 0.0, s2 = 0.0;
    int64_t i = lo + threadIdx.x;
    for (; i + kBnBlock < hi; i += 2 * kBnBlock) {           // four loads in flight
        T a[2], b[2];
        int64_t o[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { o[u] = bn_off(g, c, i + u * kBnBlock); a[u] = xv[o[u]]; b[u] = dv[o[u]]; }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float xe = comp(a[u], k);
                float ge = comp(b[u], k);
                bool open = fmaf(xe, scale, shift) > 0.f;
                if (GATED) open = bn_gate(open, o[u] * V + k, gt);
                if (RELU && !open) ge = 0.f;
                s1 += (double)ge;
                s2 = fma((double)ge, (double)xe - meand, s2);
            }
    }
    for (; i < hi; i += kBnBlock) {
        const int64_t o = bn_off(g, c, i);
        const T a = xv[o], b = dv[o];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xe = comp(a, k);
            float ge = comp(b, k);
            bool open = fmaf(xe, scale, shift) > 0.f;
            if (GATED) open = bn_gate(open, o * V + k, gt);
            if (RELU && !open) ge = 0.f;
            s1 += (double)ge;
            s2 = fma((double)ge, (double)xe - meand, s2);
        }
    }
    bn_block_sum2(s1, s2, sh);
    if (threadIdx.x == 0) partial[(int64_t)c * bn_nsplit(g) + bn_split(g)] = make_double2(s1, s2);
}

// ---- backward, launch 2: torch's CPU association (native_batch_norm_backward, training):
//      sum = sum dy' ; dotp = sum dy' (x - mean)                      (double, merged here)
//      dbeta = sum ; dgamma = dotp * invstd ; gm = sum / n ; k = dotp * invstd^2 / n
//      dx = ((dy' - gm) - (x - mean) * k) * invstd * gamma     (RES: + dz, the gradient that reaches the residual sum
//      z = x on its other path: the accumulation autograd would run as a separate add launch)
// BIGS: S > 64 partial sums per channel (K10: one per workgroup of the convolution launch that left them): every thread adds
// its partials in ascending order (t, t + 256, ...), then the workgroup's fixed tree - every workgroup of the channel the same.
template <int V, bool RELU, bool RES, bool GATED = false, bool BIGS = false>
__global__ __launch_bounds__(kBnBlock) void k_bn_bwd_dx(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ dz, float* __restrict__ dx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                        const float* __restrict__ save_invstd,
                                                        const double2* __restrict__ partial, int S,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta, BnGeom g,
                                                        BnGates gt = BnGates{})
{
    using T = typename Vec<V>::T;
    __shared__ float sh[2];
    const int c = bn_channel(g);
    const float mean = save_mean[c], invstd = save_invstd[c], w = gamma[c];
    const T* __restrict__ xv = reinterpret_cast<const T*>(x);
    const T* __restrict__ dv = reinterpret_cast<const T*>(dy);
    const T* __restrict__ rv = reinterpret_cast<const T*>(dz);
    T* __restrict__ ov = reinterpret_cast<T*>(dx);
    const int64_t lo = (int64_t)bn_split(g) * g.chunk;
    const int64_t hi = lo + g.chunk < g.per_ch ? lo + g.chunk : g.per_ch;
    // first batch of loads (x, dy, dz of two units) BEFORE the merge prologue: see k_bn_fwd_apply
    const int64_t i0 = lo + threadIdx.x;
    T a0[2], b0[2], r0[2];
    int64_t o0[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int64_t iu = i0 + u * kBnBlock;
        if (iu < hi) { o0[u] = bn_off(g, c, iu); a0[u] = xv[o0[u]]; b0[u] = dv[o0[u]]; if (RES) r0[u] = rv[o0[u]]; }
    }
    if constexpr (BIGS) {
        __shared__ double shd[2 * kBnBlock / 64];
        double a = 0.0, b = 0.0;
        for (int t = threadIdx.x; t < S; t += kBnBlock) { const double2 p = partial[(int64_t)c * S + t]; a += p.x; b += p.y; }
        bn_block_sum2(a, b, shd);
        if (threadIdx.x == 0) {
            const double n = (double)g.per_ch * V, iv = (double)invstd;
            sh[0] = (float)(a / n);
            sh[1] = (float)(b * iv * iv / n);
            if (bn_split(g) == 0) { dbeta[c] = (float)a; dgamma[c] = (float)(b * iv); }
        }
    } else if (threadIdx.x < 64) {
        double a = 0.0, b = 0.0;
        if ((int)threadIdx.x < S) { const double2 p = partial[(int64_t)c * S + threadIdx.x]; a = p.x; b = p.y; }
        a = bn_wave_sum(a);
        b = bn_wave_sum(b);
        if (threadIdx.x == 0) {
            const double n = (double)g.per_ch * V, iv = (double)invstd;
            sh[0] = (float)(a / n);
            sh[1] = (float)(b * iv * iv / n);
            if (bn_split(g) == 0) { dbeta[c] = (float)a; dgamma[c] = (float)(b * iv); }
        }
    }
    __syncthreads();
    const float gm = sh[0], kk = sh[1];
    float scale, shift;
    bn_gate_scalars(g, c, mean, invstd, gamma, beta, scale, shift);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (i0 + u * kBnBlock < hi) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float xe = comp(a0[u], k);
                float ge = comp(b0[u], k);
                bool open = fmaf(xe, scale, shift) > 0.f;
                if (GATED) open = bn_gate(open, o0[u] * V + k, gt);
                if (RELU && !open) ge = 0.f;
                setc(b0[u], k, (((ge - gm) - (xe - mean) * kk) * invstd) * w);
            }
            ov[o0[u]] = RES ? vadd(r0[u], b0[u]) : b0[u];
        }
    }
    for (int64_t i = i0 + 2 * kBnBlock; i < hi; i += 2 * kBnBlock) {
        T a[2], b[2], r[2];
        int64_t o[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t iu = i + u * kBnBlock;
            if (iu < hi) { o[u] = bn_off(g, c, iu); a[u] = xv[o[u]]; b[u] = dv[o[u]]; if (RES) r[u] = rv[o[u]]; }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (i + u * kBnBlock < hi) {
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float xe = comp(a[u], k);
                    float ge = comp(b[u], k);
                    bool open = fmaf(xe, scale, shift) > 0.f;
                    if (GATED) open = bn_gate(open, o[u] * V + k, gt);
                    if (RELU && !open) ge = 0.f;
                    setc(b[u], k, (((ge - gm) - (xe - mean) * kk) * invstd) * w);
                }
                ov[o[u]] = RES ? vadd(r[u], b[u]) : b[u];
            }
        }
    }
}

// ---- one-pass forms: a channel that fits one workgroup's registers (N*H*W <= 4 * kOneBlock * kOneEpt elements: the
//      16x16 and 8x8 stages at batch 128) is read ONCE - one launch instead of two, no partials, no second read.
//      Same arithmetic (double sums, one rounding), so the same floats as the two-launch form. -----------------------
constexpr int kOneBlock = 1024;
constexpr int kOneEpt = 8;          // most float4 per thread held in registers (2, 4 or 8 by channel size)
constexpr int kOneMinC = 48;        // fewer channels = fewer workgroups than that: the two-launch form spreads wider

// 32-bit form of bn_off for tensors the host has checked to hold < 2^31 float4 (one-pass kernels: saves the 64-bit
// address arithmetic's registers)
__device__ __forceinline__ int bn_off32(const BnGeom& g, int c, int i)
{
    const int n = g.hw_shift >= 0 ? (i >> g.hw_shift) : (i / g.hw);
    return (n * g.C + c) * g.hw + (i - n * g.hw);
}

__device__ __forceinline__ void bn_block_sum2_one(double& a, double& b, double* sh /* 2 * kOneBlock/64 */)
{
    a = bn_wave_sum(a);
    b = bn_wave_sum(b);
    if ((threadIdx.x & 63) == 0) { sh[2 * (threadIdx.x >> 6)] = a; sh[2 * (threadIdx.x >> 6) + 1] = b; }
    __syncthreads();
    a = b = 0.0;
#pragma unroll
    for (int w = 0; w < kOneBlock / 64; ++w) { a += sh[2 * w]; b += sh[2 * w + 1]; }
}

template <bool RELU, bool ADD, int EPT>
__global__ __launch_bounds__(kOneBlock) void k_bn_fwd_one(const float* __restrict__ x, const float* __restrict__ addend,
                                                         float* __restrict__ z, float* __restrict__ y,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                                         float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                         float eps, float momentum, BnGeom g)
{
    __shared__ double sh[2 * kOneBlock / 64];
    __shared__ float shf[2];
    const float4* __restrict__ xv = reinterpret_cast<const float4*>(x);
    const float4* __restrict__ av = reinterpret_cast<const float4*>(addend);
    float4* __restrict__ zv = reinterpret_cast<float4*>(z);
    float4* __restrict__ yv = reinterpret_cast<float4*>(y);
    const int c = blockIdx.x, per_ch = (int)g.per_ch;
    float4 v[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = threadIdx.x + u * kOneBlock;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < per_ch) {
            const int o = bn_off32(g, c, i);
            v[u] = xv[o];
            if (ADD) v[u] = vadd(v[u], av[o]);
        }
    }
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {             // lanes beyond the channel hold zeros: they add nothing
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double d = (double)comp(v[u], k); s1 += d; s2 = fma(d, d, s2); }
    }
    bn_block_sum2_one(s1, s2, sh);
    if (threadIdx.x == 0) {
        const double n = (double)g.per_ch * 4.0;
        const double mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0.0) var = 0.0;
        const float meanf = (float)mean;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float alpha = invstd * gamma[c];
        shf[0] = alpha;
        shf[1] = fmaf(-meanf, alpha, beta[c]);
        save_mean[c] = meanf;
        save_invstd[c] = invstd;
        bn_save_gate(g, c, alpha, shf[1]);
        if (running_mean) {
            running_mean[c] = momentum * meanf + (1.0f - momentum) * running_mean[c];
            running_var[c] = momentum * (float)(var * (n / (n - 1.0))) + (1.0f - momentum) * running_var[c];
        }
    }
    __syncthreads();
    const float scale = shf[0], shift = shf[1];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = threadIdx.x + u * kOneBlock;
        if (i < per_ch) {
            const int o = bn_off32(g, c, i);
            if (ADD) zv[o] = v[u];
            float4 r = v[u];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float t = fmaf(comp(v[u], k), scale, shift); setc(r, k, RELU ? bn_relu_fwd(t) : t); }
            yv[o] = r;
        }
    }
}

template <bool RELU, bool RES, int EPT>
__global__ __launch_bounds__(kOneBlock) void k_bn_bwd_one(const float* __restrict__ x, const float* __restrict__ dy,
                                                         const float* __restrict__ dz, float* __restrict__ dx,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ save_mean,
                                                         const float* __restrict__ save_invstd, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, BnGeom g)
{
    __shared__ double sh[2 * kOneBlock / 64];
    __shared__ float shf[2];
    const float4* __restrict__ xv = reinterpret_cast<const float4*>(x);
    const float4* __restrict__ dv = reinterpret_cast<const float4*>(dy);
    const float4* __restrict__ rv = reinterpret_cast<const float4*>(dz);
    float4* __restrict__ ov = reinterpret_cast<float4*>(dx);
    const int c = blockIdx.x, per_ch = (int)g.per_ch;
    const float mean = save_mean[c], invstd = save_invstd[c], w = gamma[c];
    float scale, shift;
    bn_gate_scalars(g, c, mean, invstd, gamma, beta, scale, shift);
    const double meand = (double)mean;
    float4 a[EPT], b[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = threadIdx.x + u * kOneBlock;
        a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < per_ch) { const int o = bn_off32(g, c, i); a[u] = xv[o]; b[u] = dv[o]; }
    }
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xe = comp(a[u], k);
            float ge = comp(b[u], k);                // zero beyond the channel: adds nothing
            if (RELU && !(fmaf(xe, scale, shift) > 0.f)) ge = 0.f;
            setc(b[u], k, ge);
            s1 += (double)ge;
            s2 = fma((double)ge, (double)xe - meand, s2);
        }
    }
    bn_block_sum2_one(s1, s2, sh);
    if (threadIdx.x == 0) {
        const double n = (double)g.per_ch * 4.0, iv = (double)invstd;
        shf[0] = (float)(s1 / n);
        shf[1] = (float)(s2 * iv * iv / n);
        dbeta[c] = (float)s1;
        dgamma[c] = (float)(s2 * iv);
    }
    __syncthreads();
    const float gm = shf[0], kk = shf[1];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = threadIdx.x + u * kOneBlock;
        if (i < per_ch) {
            const int o = bn_off32(g, c, i);
            float4 r = b[u];
#pragma unroll
            for (int k = 0; k < 4; ++k) setc(r, k, (((comp(b[u], k) - gm) - (comp(a[u], k) - mean) * kk) * invstd) * w);
            ov[o] = RES ? vadd(rv[o], r) : r;
        }
    }
}


// ---- held forms: a channel too large for one workgroup, read ONCE -------------------------------------------------------
// The two-launch form reads its inputs twice; beyond the 256 MiB Infinity Cache that is 1.5x (forward) / 1.67x (backward)
// the algorithmic bytes from HBM. Here the channel's S workgroups each load their chunk into REGISTERS, publish their
// double partial sums, gather all S partials of the channel (every workgroup by itself, fixed lane order: identical
// scalars, no broadcast) and finish from registers: one launch, every byte read once.
//   * Work is handed out by ticket (one atomic per workgroup), channel-major: ticket t -> channel t / S, chunk t % S. A
//     workgroup only ever waits for partials of its own channel, and tickets are drawn in order by workgroups that are
//     RUNNING - so ONE launch cannot deadlock as long as S workgroups of it can be resident at once WHEREVER the hardware
//     puts them: workgroups go to the 8 XCDs round-robin by index while tickets are drawn in start order, so all S pieces
//     of a channel can land on the slowest XCD, which holds 32 forward / 64 backward workgroups - hence S <= 32 / 64
//     (kHeld*MaxSplit below). SEVERAL held launches in flight at once can starve one another (each keeps some pieces
//     resident and waits for the rest): safe only while sum (S_k - 1) stays below those 32 / 64, and measured to fail
//     otherwise (tools/exp/bn_held_concurrency.py: two to eight launches at the largest splits ran into the bounded wait in
//     8-23 of 40 trials). The contract is therefore ONE held launch in flight per device; callers that overlap BatchNorm
//     launches (ChainGroup's branches, bn_update_many's member streams) do not pass URSA_BN_HELD. A ticket queue per XCD (a
//     channel's workgroups on one L2) was tried: no faster, and it needs 8 x 63 + 64 resident workgroups. The wait is bounded all
//     the same: after ~3 s a workgroup raises the err word and POISONS its sums (NaN: bn_gather), so a broken contract ends in
//     NaN outputs and statistics - loud on the device itself - never in plausible wrong numbers and never in a hung GPU.
//   * Hand-off of the partials: each is two 8-byte words stored with agent-scope atomic stores as bits(value) XOR a NaN
//     payload no sum can produce, so that ZERO means "not there yet"; wave 0 of every workgroup polls the channel's S
//     slots (lane i polls slot i, agent-scope 8-byte loads, s_sleep between polls) until none is zero - the poll IS the
//     read, there is no flag, no ordering between the two words is needed and nothing else is fetched.
//   * ws (slots and counters) must be ZERO at launch and is zero again when the launch has drained: the last workgroup to
//     have gathered a channel clears its slots, the one that clears the last channel re-arms the ticket queues.
//   * A workgroup of the 256-thread form spends ~28 us per chunk (tools/exp/bn_held_timeline.hip: ticket 3.4, load +
//     reduce 10.1, gather 8.5, scalars 1.6, apply 4.2), half of it waiting. The chunk ALSO lives in LDS: LX further float4 per thread
//     arrive by LDS-DMA (global_load_lds_dwordx4: no register on the way), 9 per thread forward, 4 + 4 backward (64 KB, 2 per
//     CU): with 256-thread workgroups 268 MB forward 136 -> 123 us, backward 176 -> 159 us. Not in the residual forward.
//     (The forward's workgroup shape was re-measured after that and is now 512 threads, one workgroup per CU: below.)
// Same arithmetic as the two-launch form (double sums rounded once), hence the same floats.
constexpr int kSyncStride = 32;          // uint32 per 128-byte line: every counter on a line of its own (atomics and polls to
                                         // ONE line serialise at ~11.5 ns each whichever word they hit)
struct BnSync {
    uint32_t ticket;
    uint32_t pad0[kSyncStride - 1];
    uint32_t done;                              // channels fully gathered
    uint32_t err;                               // sticky: a bounded wait ran out / a workgroup found no work (never in a correct run)
    uint32_t pad[kSyncStride - 2];
    uint32_t left[1][kSyncStride];              // [C]: workgroups that have gathered channel c
};
constexpr unsigned long long kSlotXor = 0xFFF8DEADBEEF0001ull;   // a NaN payload with low mantissa bits set: no sum of floats has it

// Workgroup shapes (compile-time A/B: tools/exp/bn_held_fwd_ab.sh, profiles/r04_bn_held_shapes_ab.json). Forward: 512 threads,
// up to 32 float4 of x per thread in registers + 9 in LDS (72 KB), no register cap (<= 256 VGPRs, no spills): ONE workgroup
// per CU. With 256 threads, 16 + 9 float4 and four workgroups per CU (<= 128 VGPRs: 8 spilled) the same CU held more
// bytes and the 268 MB layer took 126 us instead of 108-113: fewer, larger pieces per channel to wait for and no scratch
// traffic count for more than bytes resident. Backward: 512 threads, 8 + 8 float4 of x and dy in registers + 4 + 4 in LDS
// (64 KB), two workgroups per CU - every larger or smaller shape tried was slower (160 us at 268 MB; 165-176 with 12-16
// float4 in registers or 2-9 in LDS, 225 with 1024 threads, 263 with 256).
#ifndef URSA_HELD_FWD_BLOCK
#define URSA_HELD_FWD_BLOCK 512
#endif
#ifndef URSA_HELD_FWD_EPT
#define URSA_HELD_FWD_EPT 32
#endif
#ifndef URSA_HELD_FWD_LX
#define URSA_HELD_FWD_LX 9
#endif
#ifndef URSA_HELD_BWD_BLOCK
#define URSA_HELD_BWD_BLOCK 512
#endif
#ifndef URSA_HELD_BWD_EPT
#define URSA_HELD_BWD_EPT 8
#endif
#ifndef URSA_HELD_BWD_LX
#define URSA_HELD_BWD_LX 4
#endif
constexpr int kHeldFwdBlock = URSA_HELD_FWD_BLOCK, kHeldFwdEpt = URSA_HELD_FWD_EPT, kHeldFwdLx = URSA_HELD_FWD_LX;
constexpr int kHeldBwdBlock = URSA_HELD_BWD_BLOCK, kHeldBwdEpt = URSA_HELD_BWD_EPT, kHeldBwdLx = URSA_HELD_BWD_LX;
constexpr uint32_t kHeldSpinLimit = 1u << 21;            // x ~1.7 us of s_sleep

typedef unsigned long long bn_u64;

// thread 0: draw a work item. Returns false (and raises err) if every queue is empty: only when grid != C * S.
__device__ __forceinline__ bool bn_take_item(BnSync* sy, int C, int S, int& c, int& sp)
{
    const uint32_t t = __hip_atomic_fetch_add(&sy->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < (uint32_t)C * (uint32_t)S) { c = (int)(t / (uint32_t)S); sp = (int)(t % (uint32_t)S); return true; }
    __hip_atomic_store(&sy->err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return false;
}

__device__ __forceinline__ void bn_publish(bn_u64* slot, double a, double b)
{
    __hip_atomic_store(slot, __builtin_bit_cast(bn_u64, a) ^ kSlotXor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(slot + 1, __builtin_bit_cast(bn_u64, b) ^ kSlotXor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave 0: lane i < S polls slot i of the channel until both words are there; returns the channel's sums in every lane
// A wait that runs out (never in a correct run: only when the pieces of a channel cannot all become resident, i.e. the
// "one held launch in flight" contract was broken) raises the err word AND POISONS THE SUMS: a = b = NaN, so everything this
// workgroup derives from them - y / dx of its whole chunk, save_mean / save_invstd / running statistics / dgamma / dbeta of
// its channel - is NaN. The loss and every later step are then NaN on the device itself: the failure cannot pass for a
// result, whether or not anybody reads the err word (the round-4 form went on with incomplete sums).
__device__ __forceinline__ void bn_gather(const bn_u64* slots, int S, uint32_t* err, double& a, double& b)
{
    bn_u64 wa = 1, wb = 1;                       // lanes >= S: "there", contribute zeros below
    const bool mine = (int)threadIdx.x < S;
    if (mine) wa = wb = 0;
    uint32_t spins = 0;
    bool starved = false;                        // wave-uniform (every lane counts the same polls)
    for (;;) {
        if (mine && (wa == 0 || wb == 0)) {
            wa = __hip_atomic_load(slots + 2 * threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wb = __hip_atomic_load(slots + 2 * threadIdx.x + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (__builtin_amdgcn_ballot_w64(wa == 0 || wb == 0) == 0) break;
        if (spins < 24) __builtin_amdgcn_s_sleep(6); else __builtin_amdgcn_s_sleep(64);   // the channel's workgroups start together: short waits first
        if (++spins > kHeldSpinLimit) { if (threadIdx.x == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); starved = true; break; }
    }
    a = mine ? __builtin_bit_cast(double, wa ^ kSlotXor) : 0.0;
    b = mine ? __builtin_bit_cast(double, wb ^ kSlotXor) : 0.0;
    a = bn_wave_sum(a);
    b = bn_wave_sum(b);
    if (starved) a = b = __builtin_nan("");
}

// wave 0, after its gather: count this workgroup out of channel c; the last one out clears the channel's slots and counter,
// and the one that clears the last channel re-arms the ticket queues. `ret` = the returned count (issued early, used late).
__device__ __forceinline__ uint32_t bn_leave_issue(BnSync* sy, int c)
{
    uint32_t r = 0;
    if (threadIdx.x == 0) r = __hip_atomic_fetch_add(&sy->left[c][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return r;
}
__device__ __forceinline__ void bn_leave_finish(BnSync* sy, bn_u64* slots, int c, int C, int S, uint32_t ret)
{
    const bool last = __builtin_amdgcn_readfirstlane((int)ret) == S - 1;        // wave-uniform (lane 0's value)
    if (!last) return;
    if ((int)threadIdx.x < S) {
        __hip_atomic_store(slots + 2 * threadIdx.x, (bn_u64)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slots + 2 * threadIdx.x + 1, (bn_u64)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
        __hip_atomic_store(&sy->left[c][0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_fetch_add(&sy->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (uint32_t)C - 1u) {
            __hip_atomic_store(&sy->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&sy->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int BLOCK>
__device__ __forceinline__ void bn_block_sum2_n(double& a, double& b, double* sh /* 2 * BLOCK/64 */)
{
    a = bn_wave_sum(a);
    b = bn_wave_sum(b);
    if ((threadIdx.x & 63) == 0) { sh[2 * (threadIdx.x >> 6)] = a; sh[2 * (threadIdx.x >> 6) + 1] = b; }
    __syncthreads();
    a = b = 0.0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) { a += sh[2 * w]; b += sh[2 * w + 1]; }
}

// LDS-DMA (global_load_lds_dwordx4): a load that lands in LDS without passing through registers - lane l of the wave writes
// 16 bytes at (wave-uniform base) + 16 l. The held forms use it to hold MORE of the channel per workgroup than the register
// file alone allows (what bounds them is bytes resident x 1 / residency time, §held forms): LX further float4 per thread.
typedef __attribute__((address_space(1))) const void* bn_gptr;
typedef __attribute__((address_space(3))) void* bn_lptr;
template <bool NT> __device__ __forceinline__ void bn_ld_lds(const float4* src, float4* wave_base_in_lds)
{
    __builtin_amdgcn_global_load_lds((bn_gptr)src, (bn_lptr)wave_base_in_lds, 16, 0, NT ? 2 : 0);
}

template <bool RELU, bool ADD, int EPT, int LX, bool NT>
#ifdef URSA_HELD_FWD_MIN_WAVES            // (A/B only: round 4's first shape was 256 threads with 4 waves per SIMD = <= 128 VGPRs)
__global__ __launch_bounds__(kHeldFwdBlock, URSA_HELD_FWD_MIN_WAVES) void k_bn_fwd_held(
#else
__global__ __launch_bounds__(kHeldFwdBlock) void k_bn_fwd_held(         // no register cap: one workgroup per CU (see the shapes above)
#endif
                                                                     const float* __restrict__ x, const float* __restrict__ addend,
                                                              float* __restrict__ z, float* __restrict__ y,
                                                              bn_u64* slots_base, BnSync* sync,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ running_mean, float* __restrict__ running_var,
                                                              float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                              float eps, float momentum, BnGeom g, int S)
{
    static_assert(!(ADD && LX > 0), "the residual form holds nothing in LDS (its addend would need the registers the chunk lives in)");
    __shared__ double sh[2 * kHeldFwdBlock / 64];
    __shared__ float shf[2];
    __shared__ int sh_item[2];
    __shared__ float4 hold[LX > 0 ? LX * kHeldFwdBlock : 1];
    if (threadIdx.x == 0) {
        int c0 = -1, sp0 = 0;
        bn_take_item(sync, g.C, S, c0, sp0);
        sh_item[0] = c0; sh_item[1] = sp0;
    }
    __syncthreads();
    const int c = sh_item[0], sp = sh_item[1];
    if (c < 0) return;                                   // (grid != C * S: flagged in err)
    const float4* __restrict__ xv = reinterpret_cast<const float4*>(x);
    const float4* __restrict__ av = reinterpret_cast<const float4*>(addend);
    float4* __restrict__ zv = reinterpret_cast<float4*>(z);
    float4* __restrict__ yv = reinterpret_cast<float4*>(y);
    const int lo = sp * g.chunk;
    const int hi = lo + g.chunk < (int)g.per_ch ? lo + g.chunk : (int)g.per_ch;
    // thread t owns float4 lo + t + u * 256 of the chunk: u < EPT in registers, EPT <= u < EPT + LX in LDS (DMA first: no registers)
#pragma unroll
    for (int u = 0; u < LX; ++u) {
        const int i = lo + threadIdx.x + (EPT + u) * kHeldFwdBlock;
        if (i < hi) bn_ld_lds<NT>(xv + bn_off32(g, c, i), &hold[u * kHeldFwdBlock + (threadIdx.x & ~63)]);
    }
    float4 v[EPT];                       // (offsets are recomputed for the stores rather than kept live)
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = lo + threadIdx.x + u * kHeldFwdBlock;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < hi) {
            const int o = bn_off32(g, c, i);
            v[u] = bn_ld<NT>(xv + o);
            if (ADD) v[u] = vadd(v[u], bn_ld<NT>(av + o));
        }
    }
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        if (ADD) { const int i = lo + threadIdx.x + u * kHeldFwdBlock; if (i < hi) bn_st<NT>(zv + bn_off32(g, c, i), v[u]); }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double d = (double)comp(v[u], k); s1 += d; s2 = fma(d, d, s2); }
    }
    if (LX > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's own DMA has landed (it only reads its own lanes' slots)
#pragma unroll
        for (int u = 0; u < LX; ++u) {
            const int i = lo + threadIdx.x + (EPT + u) * kHeldFwdBlock;
            if (i < hi) {
                const float4 w = hold[u * kHeldFwdBlock + threadIdx.x];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const double d = (double)comp(w, k); s1 += d; s2 = fma(d, d, s2); }
            }
        }
    }
    bn_block_sum2_n<kHeldFwdBlock>(s1, s2, sh);
    bn_u64* slots = slots_base + (int64_t)c * kBnMaxSplit * 2;
    if (threadIdx.x == 0) bn_publish(slots + 2 * sp, s1, s2);
    uint32_t left = 0;
    if (threadIdx.x < 64) {
        double a, b;
        bn_gather(slots, S, &sync->err, a, b);
        left = bn_leave_issue(sync, c);
        if (threadIdx.x == 0) {
            const double n = (double)g.per_ch * 4.0;
            const double mean = a / n;
            double var = b / n - mean * mean;
            if (var < 0.0) var = 0.0;
            const float meanf = (float)mean;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float alpha = invstd * gamma[c];
            shf[0] = alpha;
            shf[1] = fmaf(-meanf, alpha, beta[c]);
            if (sp == 0 || mean != mean) {                       // (a starved piece - NaN sums - poisons the channel's statistics too)
                save_mean[c] = meanf;
                save_invstd[c] = invstd;
                bn_save_gate(g, c, alpha, shf[1]);
                if (running_mean) {
                    running_mean[c] = momentum * meanf + (1.0f - momentum) * running_mean[c];
                    running_var[c] = momentum * (float)(var * (n / (n - 1.0))) + (1.0f - momentum) * running_var[c];
                }
            }
        }
    }
    __syncthreads();
    const float scale = shf[0], shift = shf[1];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = lo + threadIdx.x + u * kHeldFwdBlock;
        if (i < hi) {
            float4 r;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float t = fmaf(comp(v[u], k), scale, shift); setc(r, k, RELU ? bn_relu_fwd(t) : t); }
            bn_st<NT>(yv + bn_off32(g, c, i), r);
        }
    }
#pragma unroll
    for (int u = 0; u < LX; ++u) {
        const int i = lo + threadIdx.x + (EPT + u) * kHeldFwdBlock;
        if (i < hi) {
            const float4 w = hold[u * kHeldFwdBlock + threadIdx.x];
            float4 r;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float t = fmaf(comp(w, k), scale, shift); setc(r, k, RELU ? bn_relu_fwd(t) : t); }
            bn_st<NT>(yv + bn_off32(g, c, i), r);
        }
    }
    if (threadIdx.x < 64) bn_leave_finish(sync, slots, c, g.C, S, left);
}

template <bool RELU, bool RES, int EPT, int LX, bool NT>
__global__ __launch_bounds__(kHeldBwdBlock) void k_bn_bwd_held(const float* __restrict__ x, const float* __restrict__ dy,
                                                              const float* __restrict__ dz, float* __restrict__ dx,
                                                              bn_u64* slots_base, BnSync* sync,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta, BnGeom g, int S)
{
    __shared__ double sh[2 * kHeldBwdBlock / 64];
    __shared__ float shf[2];
    __shared__ int sh_item[2];
    __shared__ float4 holdx[LX > 0 ? LX * kHeldBwdBlock : 1], holdd[LX > 0 ? LX * kHeldBwdBlock : 1];
    if (threadIdx.x == 0) {
        int c0 = -1, sp0 = 0;
        bn_take_item(sync, g.C, S, c0, sp0);
        sh_item[0] = c0; sh_item[1] = sp0;
    }
    __syncthreads();
    const int c = sh_item[0], sp = sh_item[1];
    if (c < 0) return;
    const float4* __restrict__ xv = reinterpret_cast<const float4*>(x);
    const float4* __restrict__ dv = reinterpret_cast<const float4*>(dy);
    const float4* __restrict__ rv = reinterpret_cast<const float4*>(dz);
    float4* __restrict__ ov = reinterpret_cast<float4*>(dx);
    const float mean = save_mean[c], invstd = save_invstd[c], w = gamma[c];
    float scale, shift;
    bn_gate_scalars(g, c, mean, invstd, gamma, beta, scale, shift);
    const double meand = (double)mean;
    const int lo = sp * g.chunk;
    const int hi = lo + g.chunk < (int)g.per_ch ? lo + g.chunk : (int)g.per_ch;
#pragma unroll
    for (int u = 0; u < LX; ++u) {                   // the LDS-held part of the chunk: x and dy by DMA, no registers
        const int i = lo + threadIdx.x + (EPT + u) * kHeldBwdBlock;
        if (i < hi) {
            const int o = bn_off32(g, c, i);
            bn_ld_lds<NT>(xv + o, &holdx[u * kHeldBwdBlock + (threadIdx.x & ~63)]);
            bn_ld_lds<NT>(dv + o, &holdd[u * kHeldBwdBlock + (threadIdx.x & ~63)]);
        }
    }
    float4 a[EPT], b[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = lo + threadIdx.x + u * kHeldBwdBlock;
        a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < hi) { const int o = bn_off32(g, c, i); a[u] = bn_ld<NT>(xv + o); b[u] = bn_ld<NT>(dv + o); }
    }
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xe = comp(a[u], k);
            float ge = comp(b[u], k);                // zero beyond the chunk: adds nothing
            if (RELU && !(fmaf(xe, scale, shift) > 0.f)) ge = 0.f;
            setc(b[u], k, ge);
            s1 += (double)ge;
            s2 = fma((double)ge, (double)xe - meand, s2);
        }
    }
    if (LX > 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < LX; ++u) {
            const int i = lo + threadIdx.x + (EPT + u) * kHeldBwdBlock;
            if (i < hi) {
                const float4 xa = holdx[u * kHeldBwdBlock + threadIdx.x];
                float4 gb = holdd[u * kHeldBwdBlock + threadIdx.x];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float xe = comp(xa, k);
                    float ge = comp(gb, k);
                    if (RELU && !(fmaf(xe, scale, shift) > 0.f)) ge = 0.f;
                    setc(gb, k, ge);
                    s1 += (double)ge;
                    s2 = fma((double)ge, (double)xe - meand, s2);
                }
                holdd[u * kHeldBwdBlock + threadIdx.x] = gb;          // the gated gradient, as b[] holds it for the register part
            }
        }
    }
    bn_block_sum2_n<kHeldBwdBlock>(s1, s2, sh);
    bn_u64* slots = slots_base + (int64_t)c * kBnMaxSplit * 2;
    if (threadIdx.x == 0) bn_publish(slots + 2 * sp, s1, s2);
    uint32_t left = 0;
    if (threadIdx.x < 64) {
        double sa, sb;
        bn_gather(slots, S, &sync->err, sa, sb);
        left = bn_leave_issue(sync, c);
        if (threadIdx.x == 0) {
            const double n = (double)g.per_ch * 4.0, iv = (double)invstd;
            shf[0] = (float)(sa / n);
            shf[1] = (float)(sb * iv * iv / n);
            if (sp == 0 || sa != sa) { dbeta[c] = (float)sa; dgamma[c] = (float)(sb * iv); }      // (starved: NaN, from any piece)
        }
    }
    __syncthreads();
    const float gm = shf[0], kk = shf[1];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = lo + threadIdx.x + u * kHeldBwdBlock;
        if (i < hi) {
            const int o = bn_off32(g, c, i);
            float4 r;
#pragma unroll
            for (int k = 0; k < 4; ++k) setc(r, k, (((comp(b[u], k) - gm) - (comp(a[u], k) - mean) * kk) * invstd) * w);
            if (RES) r = vadd(bn_ld<NT>(rv + o), r);
            bn_st<NT>(ov + o, r);
        }
    }
#pragma unroll
    for (int u = 0; u < LX; ++u) {
        const int i = lo + threadIdx.x + (EPT + u) * kHeldBwdBlock;
        if (i < hi) {
            const int o = bn_off32(g, c, i);
            const float4 xa = holdx[u * kHeldBwdBlock + threadIdx.x], gb = holdd[u * kHeldBwdBlock + threadIdx.x];
            float4 r;
#pragma unroll
            for (int k = 0; k < 4; ++k) setc(r, k, (((comp(gb, k) - gm) - (comp(xa, k) - mean) * kk) * invstd) * w);
            if (RES) r = vadd(bn_ld<NT>(rv + o), r);
            bn_st<NT>(ov + o, r);
        }
    }
    if (threadIdx.x < 64) bn_leave_finish(sync, slots, c, g.C, S, left);
}

#ifdef URSA_DEBUG_KNOBS     // parked experiment (-2 % on the workload as built): libursa_hip_knobs.so only, not in the product ABI
// ---- NHWC twins: the second launch of the two-launch form also stores its output channels-last ----------------------------
// MIOpen's fastest weight-gradient kernel for these networks (igemm_wrw_gtcx35_nhwc) wants NHWC operands and, handed NCHW
// tensors, transposes both of them itself: 15 % of a PreResNet-20 training step's kernel time (20 % with its zero-fills),
// 16 % of a PreResNet-164 HMC step. Both operands of every 3x3 weight gradient are K6 outputs - the convolution's input
// is a forward y, its output gradient a backward dx - so K6 can hand them over in NHWC as a second store
// (ursabench_amd/fused_conv.py takes the weight gradient on those twins; forward and backward-data stay on the NCHW
// Winograd kernels).
// The second launches in a FOUR-CHANNEL form: a workgroup owns channels 4cg..4cg+3 and 256 float4 columns of them; its four
// waves merge the four channels' partials side by side (wave w: channel 4cg + w - the same sums in the same lane order as
// bn_merge, so the same floats, and no longer than the one-channel merge of the plain kernels); a thread loads one float4
// from each of the four rows, transposes the 4x4 block in registers and stores four NCHW float4 as before plus four
// NHWC float4 (channels 4cg..4cg+3 of positions 4j..4j+3: 16 contiguous bytes each).

template <bool RELU>
__global__ __launch_bounds__(kBnBlock) void k_bn_fwd_apply_4(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ yt,
                                                            const double2* __restrict__ partial, int S,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ running_mean, float* __restrict__ running_var,
                                                            float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                            float eps, float momentum, BnGeom g)
{
    __shared__ float sh[8];
    const float4* __restrict__ xv = reinterpret_cast<const float4*>(x);
    float4* __restrict__ yv = reinterpret_cast<float4*>(y);
    float4* __restrict__ tv = reinterpret_cast<float4*>(yt);
    const int c0 = 4 * blockIdx.y, C = g.C, hw = g.hw;
    const int64_t i = (int64_t)blockIdx.x * kBnBlock + threadIdx.x;            // float4 column of the channel rows: (n, j)
    const bool active = i < g.per_ch;
    int64_t n = 0, j = 0;
    float4 v[4];
    if (active) {
        if (g.hw_shift >= 0) n = i >> g.hw_shift; else n = i / hw;
        j = i - n * hw;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = xv[(n * C + c0 + r) * hw + j];                 // in flight under the merge
    }
    {
        const int c = c0 + (threadIdx.x >> 6), lane = threadIdx.x & 63;                   // wave w merges channel c0 + w
        double a = 0.0, b = 0.0;
        if (lane < S) { const double2 p = partial[(int64_t)c * S + lane]; a = p.x; b = p.y; }
        a = bn_wave_sum(a);
        b = bn_wave_sum(b);
        if (lane == 0) {
            const double cnt = (double)g.per_ch * 4.0;
            const double mean = a / cnt;
            double var = b / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            const float meanf = (float)mean;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float alpha = invstd * gamma[c];
            sh[2 * (threadIdx.x >> 6)] = alpha;
            sh[2 * (threadIdx.x >> 6) + 1] = fmaf(-meanf, alpha, beta[c]);
            if (blockIdx.x == 0) {
                save_mean[c] = meanf;
                save_invstd[c] = invstd;
                bn_save_gate(g, c, alpha, sh[2 * (threadIdx.x >> 6) + 1]);
                if (running_mean) {
                    running_mean[c] = momentum * meanf + (1.0f - momentum) * running_mean[c];
                    running_var[c] = momentum * (float)(var * (cnt / (cnt - 1.0))) + (1.0f - momentum) * running_var[c];
                }
            }
        }
    }
    __syncthreads();
    if (!active) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float scale = sh[2 * r], shift = sh[2 * r + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float t = fmaf(comp(v[r], k), scale, shift); setc(v[r], k, RELU ? bn_relu_fwd(t) : t); }
        yv[(n * C + c0 + r) * hw + j] = v[r];
    }
    const int64_t tb = (n * hw + j) * 4 * (C / 4) + blockIdx.y;        // float4 index of (n, position 4j, channels c0..c0+3)
#pragma unroll
    for (int q = 0; q < 4; ++q)
        tv[tb + (int64_t)q * (C / 4)] = make_float4(comp(v[0], q), comp(v[1], q), comp(v[2], q), comp(v[3], q));
}

template <bool RELU, bool RES>
__global__ __launch_bounds__(kBnBlock) void k_bn_bwd_dx_4(const float* __restrict__ x, const float* __restrict__ dy, const float* __restrict__ dz,
                                                         float* __restrict__ dx, float* __restrict__ dxt, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                         const float* __restrict__ save_invstd, const double2* __restrict__ partial, int S,
                                                         float* __restrict__ dgamma, float* __restrict__ dbeta, BnGeom g)
{
    __shared__ float sh[8];
    const float4* __restrict__ xv = reinterpret_cast<const float4*>(x);
    const float4* __restrict__ dv = reinterpret_cast<const float4*>(dy);
    const float4* __restrict__ rv = reinterpret_cast<const float4*>(dz);
    float4* __restrict__ ov = reinterpret_cast<float4*>(dx);
    float4* __restrict__ tv = reinterpret_cast<float4*>(dxt);
    const int c0 = 4 * blockIdx.y, C = g.C, hw = g.hw;
    const int64_t i = (int64_t)blockIdx.x * kBnBlock + threadIdx.x;
    const bool active = i < g.per_ch;
    int64_t n = 0, j = 0;
    float4 a[4], b[4], rz[4];
    if (active) {
        if (g.hw_shift >= 0) n = i >> g.hw_shift; else n = i / hw;
        j = i - n * hw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t o = (n * C + c0 + r) * hw + j;
            a[r] = xv[o]; b[r] = dv[o];
            if (RES) rz[r] = rv[o];
        }
    }
    {
        const int c = c0 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        double sa = 0.0, sb = 0.0;
        if (lane < S) { const double2 p = partial[(int64_t)c * S + lane]; sa = p.x; sb = p.y; }
        sa = bn_wave_sum(sa);
        sb = bn_wave_sum(sb);
        if (lane == 0) {
            const double cnt = (double)g.per_ch * 4.0, iv = (double)save_invstd[c];
            sh[2 * (threadIdx.x >> 6)] = (float)(sa / cnt);
            sh[2 * (threadIdx.x >> 6) + 1] = (float)(sb * iv * iv / cnt);
            if (blockIdx.x == 0) { dbeta[c] = (float)sa; dgamma[c] = (float)(sb * iv); }
        }
    }
    __syncthreads();
    if (!active) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = c0 + r;
        const float mean = save_mean[c], invstd = save_invstd[c], w = gamma[c];
        float scale, shift;
        bn_gate_scalars(g, c, mean, invstd, gamma, beta, scale, shift);
        const float gm = sh[2 * r], kk = sh[2 * r + 1];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xe = comp(a[r], k);
            float ge = comp(b[r], k);
            if (RELU && !(fmaf(xe, scale, shift) > 0.f)) ge = 0.f;
            setc(b[r], k, (((ge - gm) - (xe - mean) * kk) * invstd) * w);
        }
        if (RES) b[r] = vadd(rz[r], b[r]);
        ov[(n * C + c0 + r) * hw + j] = b[r];
    }
    const int64_t tb = (n * hw + j) * 4 * (C / 4) + blockIdx.y;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        tv[tb + (int64_t)q * (C / 4)] = make_float4(comp(b[0], q), comp(b[1], q), comp(b[2], q), comp(b[3], q));
}

#endif  // URSA_DEBUG_KNOBS (NHWC twins)

// ---- host side ----------------------------------------------------------------------------------------------------
struct BnPlan {
    BnGeom g;
    int S;       // workgroups (= partials) per channel
    int V;
};

inline bool bn_eval_no_nt()           // knob (A/B): URSA_BN_EVAL_NO_NT=1 keeps the evaluation launch's accesses temporal
{
#ifdef URSA_DEBUG_KNOBS
    static const bool v = [] { const char* e = getenv("URSA_BN_EVAL_NO_NT"); return e && e[0] && e[0] != '0'; }();
    return v;
#else
    return false;
#endif
}
inline bool bn_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool bn_aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

// One geometry per (N, C, HW): the backward's partials are laid out by the same S as its own first launch, and the
// forward's by its own; nothing is shared between calls except through `ws` within one call.
// Channel-first grids up to 16 MiB of activation - the workload's layers. Measured (tools/exp/bn_cfirst_ab.py,
// profiles/r05_bn_cfirst_ab.json; us per call inside a hipGraph, grid (channels, splits) vs (splits, channels)): [128,16,32,32]
// forward 6.27 vs 6.84, residual forward 7.92 vs 10.1, backward 6.93 vs 7.61; [128,32,16,16] 5.22 vs 5.66 / 5.56 vs 5.98 (residual
// backward 5.64 vs 6.50); neutral at 34-134 MB; +14...29 % SLOWER at 268 MB (there the runs of one channel are what streams well):
// hence the bound. Same partial sums in the same slots either way: the same floats. (Knobs build: URSA_BN_CFIRST_MAX_MIB, -1 = never.)
inline bool bn_channel_first(int64_t bytes)
{
    int64_t max_mib = 16;
#ifdef URSA_DEBUG_KNOBS
    static const int64_t knob = [] { const char* e = getenv("URSA_BN_CFIRST_MAX_MIB"); return e && e[0] ? (int64_t)atoll(e) : (int64_t)-2; }();
    if (knob != -2) max_mib = knob;
#endif
    return max_mib >= 0 && bytes <= (max_mib << 20);
}

inline int bn_plan(int64_t N, int64_t C, int64_t HW, bool vec_ok, BnPlan* p)
{
    if (N <= 0 || C <= 0 || HW <= 0 || C > 65535 || HW > (1 << 30) || N > (1ll << 31)) return URSA_ESIZE;
    if (N * C > (1ll << 40) / HW) return URSA_ESIZE;
    const int V = (vec_ok && (HW & 3) == 0) ? 4 : 1;
    const int64_t hw = HW / V;
    const int64_t per_ch = N * hw;
#ifdef URSA_DEBUG_KNOBS                    // geometry experiments (tools/exp/bn_fused_bench.py): libursa_hip_knobs.so only
    static const int target = [] { const char* e = getenv("URSA_BN_TARGET_WGS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : kBnTargetWgs; }();
#else
    constexpr int target = kBnTargetWgs;
#endif
    int64_t S = (target + C - 1) / C;
    if (S > kBnMaxSplit) S = kBnMaxSplit;
    if (S < 1) S = 1;
    int64_t chunk = (per_ch + S - 1) / S;
    chunk = (chunk + kBnBlock - 1) / kBnBlock * kBnBlock;
    if (chunk > (1ll << 30)) return URSA_ESIZE;
    S = (per_ch + chunk - 1) / chunk;
    p->V = V;
    p->S = (int)S;
    p->g.C = (int)C;
    p->g.hw = (int)hw;
    p->g.hw_shift = -1;
    for (int s = 0; s < 31; ++s) if ((1ll << s) == hw) p->g.hw_shift = s;
    p->g.chunk = (int)chunk;
    p->g.per_ch = per_ch;
    p->g.gate_out = nullptr;
    p->g.gate_in = nullptr;
    p->g.cfirst = bn_channel_first(per_ch * C * (V == 4 ? 16 : 4)) ? 1 : 0;
    return URSA_OK;
}

inline int bn_launch_status() { return (int)hipGetLastError(); }

#ifdef URSA_DEBUG_KNOBS
// NHWC twin (see k_bn_fwd_apply_4): float4 accesses, channels a multiple of 4, 16-byte aligned twin, a grid that fits.
inline bool bn_twin_ok(const BnPlan& p, const void* twin)
{
    return p.V == 4 && (p.g.C & 3) == 0 && bn_aligned16(twin) && (p.g.per_ch + kBnBlock - 1) / kBnBlock < (1ll << 31);
}
#endif

// Held form (one launch, inputs read once): float4 accesses, 32-bit float4 offsets, the channel cut into 2 <= S <= 64
// register-sized chunks, enough workgroups to fill the chip, and an activation large enough that the second read of the
// two-launch form costs more than the wait (measured: tools/exp/bn_fused_bench.py). The caller vouches for zeroed sync
// words with URSA_BN_HELD.
// Measured, us per call inside a hipGraph, held / two-launch (tools/exp/bn_held_fwd_ab.py, profiles/r04_bn_held_shapes_ab.json;
// plain and residual forms):
//   [1024,64,32,32] 268 MB  forward 108 / 152 (205 / 251)   backward 156 / 253        [1024,128,16,16] 134 MB  51 / 60   84 / 114
//   [128,160,32,32]  84 MB           34 /  41 ( 74 /  73)             48 /  65        [256,64,32,32]    67 MB  25 / 32   43 /  51
//   [128,96,32,32]   50 MB           25 /  27 ( 37 /  41)             33 / (41)       [512,16,32,32]    34 MB  17 / 16   24 / (27)
// The backward saves 8 of 20 B/element and wins from 24 MiB on; the forward saves 4 of 12: from 48 MiB on (below that it is
// a tie or, with few channels, a loss); the residual forward 4 of 20 (the two-launch form's second pass over z): from
// 32 MiB on (34 MB: 23-26 vs 27-31 us).
constexpr int64_t kHeldMinFloat4Bwd = (24ll << 20) / 16;      // 24 MiB of activation
constexpr int64_t kHeldMinFloat4Fwd = (48ll << 20) / 16;      // 48 MiB
constexpr int64_t kHeldMinFloat4FwdAdd = (32ll << 20) / 16;   // the residual forward (20 B/element two-launch, 16 held): 32 MiB
struct BnHeld { int S, chunk, ept, lx; };
struct BnHeldShape { int ept, lx; };       // a compiled kernel shape: float4 per thread in registers / in LDS
// `shapes`: the compiled shapes, largest capacity first. The LARGEST chunk that still splits the channel in two: fewer,
// larger pieces per channel = fewer workgroups to wait for. (Measured the other way round - the smallest chunk that still
// gives <= 64 pieces, for more workgroups - [1024,128,16,16] 59 -> 123 us forward, 96 -> 129 backward.) The kernel that runs
// is the smallest compiled shape that holds the chunk.
template <int NSHAPES>
inline bool bn_held_plan(const BnPlan& p, uint32_t flags, int block, const BnHeldShape (&shapes)[NSHAPES], bool lds_ok,
                         int max_split, int resident, int64_t min_float4, BnHeld* h)
{
    if (!(flags & URSA_BN_HELD) || (flags & URSA_BN_TWO_LAUNCH) || p.V != 4) return false;
    const int64_t per_ch = p.g.per_ch;
#ifdef URSA_DEBUG_KNOBS                    // experiments (tools/exp/bn_held_ab.py): the size from which the held form is taken, in MiB;
    static const int64_t forced = [] { const char* e = getenv("URSA_BN_HELD_MIN_MIB"); return e && e[0] ? ((int64_t)atoll(e) << 20) / 16 : (int64_t)-1; }();
    if (forced >= 0) min_float4 = forced;
    static const bool no_lds = [] { const char* e = getenv("URSA_BN_HELD_NO_LDS"); return e && e[0] && e[0] != '0'; }();   // registers only
    if (no_lds) lds_ok = false;
#endif
    if (per_ch * p.g.C >= (1ll << 31) || per_ch * p.g.C < min_float4) return false;
    // Per shape: pieces S, chunk, and how full the launch's last round of `resident` workgroups is. The first (largest)
    // shape whose rounds are >= 80 % full is taken - [128,160,32,32] in two pieces per channel is 320 workgroups on 256 CUs,
    // 1.25 rounds: three pieces (480) run the residual form in 72 us instead of 79 - else the fullest.
    int pick = -1;
    double pick_fill = 0.0;
    int64_t pick_S = 0, pick_chunk = 0;
    for (int k = 0; k < NSHAPES; ++k) {
        if (shapes[k].lx > 0 && !lds_ok) continue;
        const int64_t cap = (int64_t)block * (shapes[k].ept + shapes[k].lx);
        int64_t S = (per_ch + cap - 1) / cap;
        if (S < 2) continue;                                    // the channel fits one such piece: a smaller shape
        if (S > max_split) break;                               // (smaller shapes only give more pieces)
        int64_t chunk = (per_ch + S - 1) / S;
        chunk = (chunk + block - 1) / block * block;
        S = (per_ch + chunk - 1) / chunk;
        const int64_t wgs = S * p.g.C;
        if (S < 2 || S > max_split || wgs < 256 || wgs >= (1ll << 31)) continue;
        const double fill = resident > 0 ? (double)wgs / (double)((wgs + resident - 1) / resident * resident) : 1.0;
        if (pick < 0 || fill > pick_fill) { pick = k; pick_fill = fill; pick_S = S; pick_chunk = chunk; }
        if (fill >= 0.8) break;
    }
    if (pick < 0) return false;
    const int need = (int)(pick_chunk / block);
    int best = pick;                                            // the smallest compiled shape that holds `need` float4 per thread
    for (int m = 0; m < NSHAPES; ++m) {
        if (shapes[m].lx > 0 && !lds_ok) continue;
        const int cm = shapes[m].ept + shapes[m].lx, cb = shapes[best].ept + shapes[best].lx;
        if (cm >= need && (cm < cb || (cm == cb && shapes[m].lx < shapes[best].lx))) best = m;
    }
    h->S = (int)pick_S;
    h->chunk = (int)pick_chunk;
    h->ept = shapes[best].ept;
    h->lx = shapes[best].lx;
    return true;
}
constexpr BnHeldShape kHeldFwdShapes[] = {{kHeldFwdEpt, kHeldFwdLx}, {kHeldFwdEpt, 0}, {kHeldFwdEpt / 2, kHeldFwdLx}, {kHeldFwdEpt / 2, 0}};
constexpr BnHeldShape kHeldBwdShapes[] = {{kHeldBwdEpt, kHeldBwdLx}, {kHeldBwdEpt, 0}};
// Most pieces per channel = the workgroups ONE XCD holds (32 CUs: one forward workgroup per CU, two backward ones): all
// pieces of a channel may land on one XCD (see "held forms" above).
constexpr int kHeldFwdMaxSplit = 32, kHeldBwdMaxSplit = kBnMaxSplit;

// One-pass form: float4 accesses, the channel fits one workgroup's registers, 32-bit float4 offsets suffice, and there
// are enough channels (= workgroups) to spread over the CUs. Measured (tools/exp/bn_fused_bench.py): [128,64,8,8] forward
// 6.7 -> 5.8 us, backward 5.4 -> 3.8; [128,320,16,16] 23 -> 19 / 34 -> 26; with only 32 channels ([128,32,16,16]) the
// two-launch form is as fast forward and faster backward.
inline bool bn_one_pass(const BnPlan& p, uint32_t flags)
{
    return p.V == 4 && p.g.per_ch <= (int64_t)kOneBlock * kOneEpt && p.g.C >= kOneMinC && p.g.per_ch * p.g.C < (1ll << 31) &&
           !(flags & URSA_BN_TWO_LAUNCH);
}

}  // namespace

extern "C" {

static int bn_fwd_impl(const float* x, const float* addend, float* z_out, float* y, float* y_nhwc,
                       const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* save_gate, float* ws,
                       int64_t N, int64_t C, int64_t HW, float eps, float momentum, uint32_t flags, ursa_stream_t stream)
{
    if (flags & ~URSA_BN_ALLFLAGS) return URSA_EFLAGS;
    if (N == 0 || C == 0 || HW == 0) return (N < 0 || C < 0 || HW < 0) ? URSA_ESIZE : URSA_OK;
    if (!x || !y || !gamma || !beta || !save_mean || !save_invstd || !ws) return URSA_ENULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return URSA_ENULL;
    if ((addend == nullptr) != (z_out == nullptr)) return URSA_ENULL;
    if (!bn_aligned4(x) || !bn_aligned4(y) || !bn_aligned4(addend) || !bn_aligned4(z_out) || !bn_aligned16(ws)) return URSA_EALIGN;
    if (N > 0 && HW > 0 && N * HW < 2) return URSA_EVALUE;        // torch: "Expected more than 1 value per channel"
    BnPlan p;
    const int rc = bn_plan(N, C, HW, bn_aligned16(x) && bn_aligned16(y) && bn_aligned16(addend) && bn_aligned16(z_out), &p);
    if (rc) return rc;
    if (!bn_aligned4(save_gate)) return URSA_EALIGN;
    p.g.gate_out = save_gate;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = p.g.cfirst ? dim3(p.g.C, p.S) : dim3(p.S, p.g.C), block(kBnBlock);
    double2* part = reinterpret_cast<double2*>(ws);
    const bool relu = flags & URSA_BN_RELU;
#ifdef URSA_DEBUG_KNOBS
    if (y_nhwc) {
        // NHWC-twin form: the plain statistics launch, then the four-channel launch that stores y twice
        if (!bn_twin_ok(p, y_nhwc)) return URSA_EVALUE;
        if (addend) hipLaunchKernelGGL((k_bn_stats<4, true>), grid, block, 0, st, x, addend, z_out, part, p.g);
        else hipLaunchKernelGGL((k_bn_stats<4, false>), grid, block, 0, st, x, addend, z_out, part, p.g);
        const dim3 g4((unsigned)((p.g.per_ch + kBnBlock - 1) / kBnBlock), (unsigned)(p.g.C / 4));
        const float* in2t = addend ? z_out : x;
        if (relu) hipLaunchKernelGGL((k_bn_fwd_apply_4<true>), g4, block, 0, st, in2t, y, y_nhwc, part, p.S, gamma, beta, running_mean, running_var,
                                     save_mean, save_invstd, eps, momentum, p.g);
        else hipLaunchKernelGGL((k_bn_fwd_apply_4<false>), g4, block, 0, st, in2t, y, y_nhwc, part, p.S, gamma, beta, running_mean, running_var,
                                save_mean, save_invstd, eps, momentum, p.g);
        return bn_launch_status();
    }
#else
    if (y_nhwc) return URSA_EVALUE;
#endif
    BnHeld hd;
    if (!bn_one_pass(p, flags) && bn_held_plan(p, flags, kHeldFwdBlock, kHeldFwdShapes, /*lds_ok=*/addend == nullptr, kHeldFwdMaxSplit, /*resident=*/256, addend ? kHeldMinFloat4FwdAdd : kHeldMinFloat4Fwd, &hd)) {
        BnGeom gh = p.g;
        gh.chunk = hd.chunk;
        // the held form's own part of ws (slots, then counters): the two-launch form's partials never touch it
        bn_u64* slots = reinterpret_cast<bn_u64*>(ws + (int64_t)C * kBnMaxSplit * 4);
        BnSync* sync = reinterpret_cast<BnSync*>(ws + (int64_t)C * kBnMaxSplit * 8);
        const dim3 gg((unsigned)(hd.S * p.g.C)), bb(kHeldFwdBlock);
        // beyond the 256 MiB Infinity Cache the streams go past it (as K1-K4 do)
        const bool nt = p.g.per_ch * p.g.C * 16 * (addend ? 4 : 2) > (256ll << 20);
#define URSA_BN_HELD_E(R, A, E, L) do { \
    if (nt) hipLaunchKernelGGL((k_bn_fwd_held<R, A, E, L, true>), gg, bb, 0, st, x, addend, z_out, y, slots, sync, gamma, beta, \
                               running_mean, running_var, save_mean, save_invstd, eps, momentum, gh, hd.S); \
    else hipLaunchKernelGGL((k_bn_fwd_held<R, A, E, L, false>), gg, bb, 0, st, x, addend, z_out, y, slots, sync, gamma, beta, \
                            running_mean, running_var, save_mean, save_invstd, eps, momentum, gh, hd.S); } while (0)
#define URSA_BN_HELD_F(R, A) do { if (hd.ept == kHeldFwdEpt / 2) URSA_BN_HELD_E(R, A, kHeldFwdEpt / 2, 0); else URSA_BN_HELD_E(R, A, kHeldFwdEpt, 0); } while (0)
#define URSA_BN_HELD_L(R) do { if (hd.ept == kHeldFwdEpt / 2) URSA_BN_HELD_E(R, false, kHeldFwdEpt / 2, kHeldFwdLx); else URSA_BN_HELD_E(R, false, kHeldFwdEpt, kHeldFwdLx); } while (0)
        if (hd.lx > 0) { if (relu) URSA_BN_HELD_L(true); else URSA_BN_HELD_L(false); }
        else if (relu) { if (addend) URSA_BN_HELD_F(true, true); else URSA_BN_HELD_F(true, false); }
        else           { if (addend) URSA_BN_HELD_F(false, true); else URSA_BN_HELD_F(false, false); }
#undef URSA_BN_HELD_L
#undef URSA_BN_HELD_F
#undef URSA_BN_HELD_E
        return bn_launch_status();
    }
    if (bn_one_pass(p, flags)) {
        const dim3 g1(p.g.C), b1(kOneBlock);
#define URSA_BN_ONE_E(R, A, E) hipLaunchKernelGGL((k_bn_fwd_one<R, A, E>), g1, b1, 0, st, x, addend, z_out, y, gamma, beta, running_mean, \
                                                  running_var, save_mean, save_invstd, eps, momentum, p.g)
#define URSA_BN_ONE(R, A) do { if (p.g.per_ch <= 2 * kOneBlock) URSA_BN_ONE_E(R, A, 2); else if (p.g.per_ch <= 4 * kOneBlock) URSA_BN_ONE_E(R, A, 4); \
                               else URSA_BN_ONE_E(R, A, 8); } while (0)
        if (relu) { if (addend) URSA_BN_ONE(true, true); else URSA_BN_ONE(true, false); }
        else      { if (addend) URSA_BN_ONE(false, true); else URSA_BN_ONE(false, false); }
#undef URSA_BN_ONE
#undef URSA_BN_ONE_E
        return bn_launch_status();
    }
    const float* in2 = addend ? z_out : x;                        // what the second launch normalises
#define URSA_BN_FWD(V, R) \
    hipLaunchKernelGGL((k_bn_fwd_apply<V, R>), grid, block, 0, st, in2, y, part, p.S, gamma, beta, running_mean, running_var, \
                       save_mean, save_invstd, eps, momentum, p.g)
#define URSA_BN_STATS(V) do { \
    if (addend) hipLaunchKernelGGL((k_bn_stats<V, true>), grid, block, 0, st, x, addend, z_out, part, p.g); \
    else hipLaunchKernelGGL((k_bn_stats<V, false>), grid, block, 0, st, x, addend, z_out, part, p.g); } while (0)
    if (p.V == 4) {
        URSA_BN_STATS(4);
        if (relu) URSA_BN_FWD(4, true); else URSA_BN_FWD(4, false);
    } else {
        URSA_BN_STATS(1);
        if (relu) URSA_BN_FWD(1, true); else URSA_BN_FWD(1, false);
    }
#undef URSA_BN_STATS
#undef URSA_BN_FWD
    return bn_launch_status();
}

int ursa_bn_relu_fwd_f32(const float* x, const float* addend, float* z_out, float* y, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* save_gate, float* ws,
                         int64_t N, int64_t C, int64_t HW, float eps, float momentum, uint32_t flags, ursa_stream_t stream)
{
    return bn_fwd_impl(x, addend, z_out, y, nullptr, gamma, beta, running_mean, running_var, save_mean, save_invstd, save_gate, ws, N, C, HW,
                       eps, momentum, flags, stream);
}

// K13: K6's FIRST forward launch (+ the merge of its partial sums) alone: batch statistics of x (or of z = x + addend, stored to
// z_out), the [4][C] block (mean, invstd, scale, shift) and the running statistics - exactly the scalars ursa_bn_relu_fwd_f32's
// two-launch form produces - and no normalised activation: the consumer applies relu(fma(x, scale, shift)) itself.
int ursa_bn_stats_f32(const float* x, const float* addend, float* z_out, const float* gamma, const float* beta, float* running_mean,
                      float* running_var, float* save, float* ws, int64_t N, int64_t C, int64_t HW, float eps, float momentum,
                      ursa_stream_t stream)
{
    if (N == 0 || C == 0 || HW == 0) return (N < 0 || C < 0 || HW < 0) ? URSA_ESIZE : URSA_OK;
    if (N < 0 || C < 0 || HW < 0) return URSA_ESIZE;
    if (!x || !gamma || !beta || !save || !ws) return URSA_ENULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return URSA_ENULL;
    if ((addend == nullptr) != (z_out == nullptr)) return URSA_ENULL;
    if (!bn_aligned4(x) || !bn_aligned4(addend) || !bn_aligned4(z_out) || !bn_aligned4(save) || !bn_aligned16(ws)) return URSA_EALIGN;
    if (N * HW < 2) return URSA_EVALUE;
    BnPlan p;
    const int rc = bn_plan(N, C, HW, bn_aligned16(x) && bn_aligned16(addend) && bn_aligned16(z_out), &p);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = p.g.cfirst ? dim3(p.g.C, p.S) : dim3(p.S, p.g.C), block(kBnBlock);
    double2* part = reinterpret_cast<double2*>(ws);
#define URSA_BN_STATS(V) do { \
    if (addend) hipLaunchKernelGGL((k_bn_stats<V, true>), grid, block, 0, st, x, addend, z_out, part, p.g); \
    else hipLaunchKernelGGL((k_bn_stats<V, false>), grid, block, 0, st, x, addend, z_out, part, p.g); } while (0)
    if (p.V == 4) URSA_BN_STATS(4); else URSA_BN_STATS(1);
#undef URSA_BN_STATS
    hipLaunchKernelGGL(k_bn_finalize, dim3((unsigned)C), dim3(64), 0, st, part, p.S, gamma, beta, running_mean, running_var, save, eps, momentum,
                       (double)p.g.per_ch * p.V, (int)C);
    return bn_launch_status();
}

// K14: the merge of a producer's partial sums (partial: [C][nl] pairs of doubles (sum g, sum g * (x - mean)), any nl >= 1) into
// coef [3][C] = (gm, kk, gamma), dgamma and dbeta; n_per_channel = N * HW of the normalised tensor.
int ursa_bn_bwd_coef_f32(const double* partial, int64_t nl, const float* bn_save, const float* gamma, float* coef, float* dgamma,
                         float* dbeta, int64_t n_per_channel, int64_t C, ursa_stream_t stream)
{
    if (!partial || !bn_save || !gamma || !coef || !dgamma || !dbeta) return URSA_ENULL;
    if (nl < 1 || n_per_channel < 1 || C < 1) return URSA_ESIZE;
    if (!bn_aligned16(partial) || !bn_aligned4(bn_save) || !bn_aligned4(gamma) || !bn_aligned4(coef) || !bn_aligned4(dgamma) || !bn_aligned4(dbeta))
        return URSA_EALIGN;
    if (nl > (1 << 24) || C > 65535) return URSA_EVALUE;
    hipLaunchKernelGGL(k_bn_bwd_coef, dim3((unsigned)C), dim3(kBnBlock), 0, (hipStream_t)stream, reinterpret_cast<const double2*>(partial), (int)nl,
                       bn_save, gamma, coef, dgamma, dbeta, (double)n_per_channel, (int)C);
    return bn_launch_status();
}

#ifdef URSA_DEBUG_KNOBS
int ursa_bn_relu_fwd_nhwc_f32(const float* x, const float* addend, float* z_out, float* y, float* y_nhwc,
                              const float* gamma, const float* beta, float* running_mean, float* running_var, float* save_mean,
                              float* save_invstd, float* ws, int64_t N, int64_t C, int64_t HW, float eps, float momentum,
                              uint32_t flags, ursa_stream_t stream)
{
    if (!y_nhwc) return URSA_ENULL;
    return bn_fwd_impl(x, addend, z_out, y, y_nhwc, gamma, beta, running_mean, running_var, save_mean, save_invstd, nullptr, ws, N, C, HW,
                       eps, momentum, flags, stream);
}
#endif

int ursa_bn_relu_eval_f32(const float* x, const float* addend, float* z_out, float* y, const float* gamma, const float* beta,
                          const float* running_mean, const float* running_var, int64_t N, int64_t C, int64_t HW, float eps,
                          uint32_t flags, ursa_stream_t stream)
{
    if (flags & ~URSA_BN_ALLFLAGS) return URSA_EFLAGS;
    if (N == 0 || C == 0 || HW == 0) return (N < 0 || C < 0 || HW < 0) ? URSA_ESIZE : URSA_OK;
    if (!x || !y || !gamma || !beta || !running_mean || !running_var) return URSA_ENULL;
    if ((addend == nullptr) != (z_out == nullptr)) return URSA_ENULL;
    if (!bn_aligned4(x) || !bn_aligned4(y) || !bn_aligned4(addend) || !bn_aligned4(z_out)) return URSA_EALIGN;
    BnPlan p;
    const int rc = bn_plan(N, C, HW, bn_aligned16(x) && bn_aligned16(y) && bn_aligned16(addend) && bn_aligned16(z_out), &p);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = p.g.cfirst ? dim3(p.g.C, p.S) : dim3(p.S, p.g.C), block(kBnBlock);
    const bool relu = flags & URSA_BN_RELU;
    // beyond the 256 MiB Infinity Cache the streams go past it (as K1-K4 and the held forms do)
    const bool nt = p.V == 4 && N * C * HW * 4 * (addend ? 4 : 2) > (256ll << 20) && !bn_eval_no_nt();
#ifdef URSA_DEBUG_KNOBS                    // A/B (tools/exp/bn_eval_lin_ab.py): URSA_BN_EVAL_GRID=1 keeps the channel-grid kernel
    static const bool eval_grid = [] { const char* e = getenv("URSA_BN_EVAL_GRID"); return e && e[0] && e[0] != '0'; }();
#else
    constexpr bool eval_grid = false;
#endif
    // Linear form from 16 MiB of activation on (8 MiB with an addend): measured over 13 layer shapes with the linear form forced
    // everywhere (tools/exp/bn_eval_lin_ab.py, profiles/r05_bn_eval_lin_ab.json): 67-268 MB activations 20.3 vs 24.8 us
    // ([4096,64,8,8]: 0.83 vs 0.68 of the HBM peak), 38.5 vs 41.4, 88 vs 95 us (0.76 vs 0.70); residual form 80.7 vs 123.7 us at
    // [4096,32,16,16]; the two 1 GB residual cases lose 3 %. Below ~10 MB the launch is latency-bound and the table's barrier costs
    // more than the channel grid's per-thread scalars (2 MB: 2.7 vs 2.2 us): those keep the channel grid.
    const bool eval_lin = p.V == 4 && !eval_grid && p.g.per_ch * p.g.C * 16 >= ((addend ? 8ll : 16ll) << 20);
    if (eval_lin) {
        // one contiguous 16 KB span per workgroup (see k_bn_eval_lin)
        const int64_t total4 = p.g.per_ch * p.g.C;
        const int64_t wgs = (total4 + kEvalSpan - 1) / kEvalSpan;
        if (wgs < (1ll << 31)) {
            const dim3 gl((unsigned)wgs);
            const float4* x4 = reinterpret_cast<const float4*>(x);
            const float4* a4 = reinterpret_cast<const float4*>(addend);
            float4* z4 = reinterpret_cast<float4*>(z_out);
            float4* y4 = reinterpret_cast<float4*>(y);
#define URSA_BN_EVAL_L(R, A, T) hipLaunchKernelGGL((k_bn_eval_lin<R, A, T>), gl, block, 0, st, x4, a4, z4, y4, gamma, beta, running_mean, \
                                                   running_var, eps, p.g.C, p.g.hw, p.g.hw_shift, total4)
#define URSA_BN_EVAL_L2(R, A) do { if (nt) URSA_BN_EVAL_L(R, A, true); else URSA_BN_EVAL_L(R, A, false); } while (0)
            if (relu) { if (addend) URSA_BN_EVAL_L2(true, true); else URSA_BN_EVAL_L2(true, false); }
            else      { if (addend) URSA_BN_EVAL_L2(false, true); else URSA_BN_EVAL_L2(false, false); }
#undef URSA_BN_EVAL_L2
#undef URSA_BN_EVAL_L
            return bn_launch_status();
        }
    }
#define URSA_BN_EVAL(V, R, A) do { \
    if (nt) hipLaunchKernelGGL((k_bn_eval<V, R, A, true>), grid, block, 0, st, x, addend, z_out, y, gamma, beta, running_mean, running_var, eps, p.g); \
    else hipLaunchKernelGGL((k_bn_eval<V, R, A, false>), grid, block, 0, st, x, addend, z_out, y, gamma, beta, running_mean, running_var, eps, p.g); } while (0)
#define URSA_BN_EVAL2(V, R) do { if (addend) URSA_BN_EVAL(V, R, true); else URSA_BN_EVAL(V, R, false); } while (0)
    if (p.V == 4) { if (relu) URSA_BN_EVAL2(4, true); else URSA_BN_EVAL2(4, false); }
    else          { if (relu) URSA_BN_EVAL2(1, true); else URSA_BN_EVAL2(1, false); }
#undef URSA_BN_EVAL2
#undef URSA_BN_EVAL
    return bn_launch_status();
}

static int bn_bwd_impl(const float* x, const float* dy, const float* dz, float* dx, float* dx_nhwc,
                       const float* gamma, const float* beta,
                       const float* save_mean, const float* save_invstd, const float* gate, float* dgamma, float* dbeta, float* ws,
                       int64_t N, int64_t C, int64_t HW, uint32_t flags, const BnGates* gates, ursa_stream_t stream)
{
    if (flags & ~URSA_BN_ALLFLAGS) return URSA_EFLAGS;
    if (N == 0 || C == 0 || HW == 0) return (N < 0 || C < 0 || HW < 0) ? URSA_ESIZE : URSA_OK;
    if (!x || !dy || !dx || !gamma || !beta || !save_mean || !save_invstd || !dgamma || !dbeta || !ws) return URSA_ENULL;
    if (!bn_aligned4(x) || !bn_aligned4(dy) || !bn_aligned4(dz) || !bn_aligned4(dx) || !bn_aligned16(ws)) return URSA_EALIGN;
    BnPlan p;
    const int rc = bn_plan(N, C, HW, bn_aligned16(x) && bn_aligned16(dy) && bn_aligned16(dz) && bn_aligned16(dx), &p);
    if (rc) return rc;
    if (!bn_aligned4(gate)) return URSA_EALIGN;
    p.g.gate_in = gate;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = p.g.cfirst ? dim3(p.g.C, p.S) : dim3(p.S, p.g.C), block(kBnBlock);
    double2* part = reinterpret_cast<double2*>(ws);
    const bool relu = flags & URSA_BN_RELU;
#ifdef URSA_DEBUG_KNOBS
    if (dx_nhwc) {
        // NHWC-twin form: the plain reduction launch, then the four-channel launch that stores dx twice
        if (gates || !bn_twin_ok(p, dx_nhwc)) return URSA_EVALUE;
        const dim3 g4((unsigned)((p.g.per_ch + kBnBlock - 1) / kBnBlock), (unsigned)(p.g.C / 4));
#define URSA_BN_BWD_T(R) do { \
    hipLaunchKernelGGL((k_bn_bwd_reduce<4, R>), grid, block, 0, st, x, dy, gamma, beta, save_mean, save_invstd, part, p.g, BnGates{}); \
    if (dz) hipLaunchKernelGGL((k_bn_bwd_dx_4<R, true>), g4, block, 0, st, x, dy, dz, dx, dx_nhwc, gamma, beta, save_mean, save_invstd, part, p.S, \
                               dgamma, dbeta, p.g); \
    else hipLaunchKernelGGL((k_bn_bwd_dx_4<R, false>), g4, block, 0, st, x, dy, dz, dx, dx_nhwc, gamma, beta, save_mean, save_invstd, part, p.S, \
                            dgamma, dbeta, p.g); } while (0)
        if (relu) URSA_BN_BWD_T(true); else URSA_BN_BWD_T(false);
#undef URSA_BN_BWD_T
        return bn_launch_status();
    }
#else
    if (dx_nhwc) return URSA_EVALUE;
#endif
    if (gates) {                                                  // parity instrument: always the two-launch kernels
        if (!relu) return URSA_EFLAGS;
        const BnGates gt = *gates;
#define URSA_BN_BWD_G(V) do { \
    hipLaunchKernelGGL((k_bn_bwd_reduce<V, true, true>), grid, block, 0, st, x, dy, gamma, beta, save_mean, save_invstd, part, p.g, gt); \
    if (dz) hipLaunchKernelGGL((k_bn_bwd_dx<V, true, true, true>), grid, block, 0, st, x, dy, dz, dx, gamma, beta, save_mean, \
                               save_invstd, part, p.S, dgamma, dbeta, p.g, gt); \
    else hipLaunchKernelGGL((k_bn_bwd_dx<V, true, false, true>), grid, block, 0, st, x, dy, dz, dx, gamma, beta, save_mean, \
                            save_invstd, part, p.S, dgamma, dbeta, p.g, gt); } while (0)
        if (p.V == 4) URSA_BN_BWD_G(4); else URSA_BN_BWD_G(1);
#undef URSA_BN_BWD_G
        return bn_launch_status();
    }
    BnHeld hd;
    if (!bn_one_pass(p, flags) && bn_held_plan(p, flags, kHeldBwdBlock, kHeldBwdShapes, /*lds_ok=*/true, kHeldBwdMaxSplit, /*resident=*/0 /* the largest shape, as measured */, kHeldMinFloat4Bwd, &hd)) {
        BnGeom gh = p.g;
        gh.chunk = hd.chunk;
        // the held form's own part of ws (slots, then counters): the two-launch form's partials never touch it
        bn_u64* slots = reinterpret_cast<bn_u64*>(ws + (int64_t)C * kBnMaxSplit * 4);
        BnSync* sync = reinterpret_cast<BnSync*>(ws + (int64_t)C * kBnMaxSplit * 8);
        const dim3 gg((unsigned)(hd.S * p.g.C)), bb(kHeldBwdBlock);
        const bool nt = p.g.per_ch * p.g.C * 16 * (dz ? 4 : 3) > (256ll << 20);
#define URSA_BN_HELD_E(R, A, E, L) do { \
    if (nt) hipLaunchKernelGGL((k_bn_bwd_held<R, A, E, L, true>), gg, bb, 0, st, x, dy, dz, dx, slots, sync, gamma, beta, save_mean, \
                               save_invstd, dgamma, dbeta, gh, hd.S); \
    else hipLaunchKernelGGL((k_bn_bwd_held<R, A, E, L, false>), gg, bb, 0, st, x, dy, dz, dx, slots, sync, gamma, beta, save_mean, \
                            save_invstd, dgamma, dbeta, gh, hd.S); } while (0)
#define URSA_BN_HELD_B(R, A) do { if (hd.lx > 0) URSA_BN_HELD_E(R, A, kHeldBwdEpt, kHeldBwdLx); else URSA_BN_HELD_E(R, A, kHeldBwdEpt, 0); } while (0)
        if (relu) { if (dz) URSA_BN_HELD_B(true, true); else URSA_BN_HELD_B(true, false); }
        else      { if (dz) URSA_BN_HELD_B(false, true); else URSA_BN_HELD_B(false, false); }
#undef URSA_BN_HELD_B
#undef URSA_BN_HELD_E
        return bn_launch_status();
    }
    if (bn_one_pass(p, flags)) {
        const dim3 g1(p.g.C), b1(kOneBlock);
#define URSA_BN_ONE_E(R, A, E) hipLaunchKernelGGL((k_bn_bwd_one<R, A, E>), g1, b1, 0, st, x, dy, dz, dx, gamma, beta, save_mean, save_invstd, \
                                                  dgamma, dbeta, p.g)
#define URSA_BN_ONE(R, A) do { if (p.g.per_ch <= 2 * kOneBlock) URSA_BN_ONE_E(R, A, 2); else if (p.g.per_ch <= 4 * kOneBlock) URSA_BN_ONE_E(R, A, 4); \
                               else URSA_BN_ONE_E(R, A, 8); } while (0)
        if (relu) { if (dz) URSA_BN_ONE(true, true); else URSA_BN_ONE(true, false); }
        else      { if (dz) URSA_BN_ONE(false, true); else URSA_BN_ONE(false, false); }
#undef URSA_BN_ONE
#undef URSA_BN_ONE_E
        return bn_launch_status();
    }
#define URSA_BN_BWD(V, R) do { \
    hipLaunchKernelGGL((k_bn_bwd_reduce<V, R>), grid, block, 0, st, x, dy, gamma, beta, save_mean, save_invstd, part, p.g, BnGates{}); \
    if (dz) hipLaunchKernelGGL((k_bn_bwd_dx<V, R, true>), grid, block, 0, st, x, dy, dz, dx, gamma, beta, save_mean, save_invstd, \
                               part, p.S, dgamma, dbeta, p.g, BnGates{}); \
    else hipLaunchKernelGGL((k_bn_bwd_dx<V, R, false>), grid, block, 0, st, x, dy, dz, dx, gamma, beta, save_mean, save_invstd, \
                            part, p.S, dgamma, dbeta, p.g, BnGates{}); } while (0)
    if (p.V == 4) { if (relu) URSA_BN_BWD(4, true); else URSA_BN_BWD(4, false); }
    else          { if (relu) URSA_BN_BWD(1, true); else URSA_BN_BWD(1, false); }
#undef URSA_BN_BWD
    return bn_launch_status();
}

int ursa_bn_relu_bwd_f32(const float* x, const float* dy, const float* dz, float* dx, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd, const float* gate, float* dgamma, float* dbeta, float* ws,
                         int64_t N, int64_t C, int64_t HW, uint32_t flags, ursa_stream_t stream)
{
    return bn_bwd_impl(x, dy, dz, dx, nullptr, gamma, beta, save_mean, save_invstd, gate, dgamma, dbeta, ws, N, C, HW, flags, nullptr, stream);
}

#ifdef URSA_DEBUG_KNOBS
int ursa_bn_relu_bwd_nhwc_f32(const float* x, const float* dy, const float* dz, float* dx, float* dx_nhwc,
                              const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                              float* dgamma, float* dbeta, float* ws, int64_t N, int64_t C, int64_t HW, uint32_t flags,
                              ursa_stream_t stream)
{
    if (!dx_nhwc) return URSA_ENULL;
    return bn_bwd_impl(x, dy, dz, dx, dx_nhwc, gamma, beta, save_mean, save_invstd, nullptr, dgamma, dbeta, ws, N, C, HW, flags, nullptr,
                       stream);
}
#endif

int ursa_bn_relu_bwd_gated_f32(const float* x, const float* dy, const float* dz, float* dx, const float* gamma,
                               const float* beta, const float* save_mean, const float* save_invstd, const float* gate, float* dgamma,
                               float* dbeta, float* ws, int64_t N, int64_t C, int64_t HW, uint32_t flags,
                               const int32_t* gate_idx, const uint8_t* gate_open, int64_t n_gates, ursa_stream_t stream)
{
    if (n_gates < 0 || n_gates > (1 << 30)) return URSA_ESIZE;
    if (n_gates > 0 && (!gate_idx || !gate_open)) return URSA_ENULL;
    if (!bn_aligned4(gate_idx)) return URSA_EALIGN;
    if (N > 0 && C > 0 && HW > 0 && N * C > (int64_t)0x7ffffffe / HW) return URSA_ESIZE;      // 32-bit element offsets
    const BnGates gt{gate_idx, gate_open, (int)n_gates};
    return bn_bwd_impl(x, dy, dz, dx, nullptr, gamma, beta, save_mean, save_invstd, gate, dgamma, dbeta, ws, N, C, HW, flags, &gt, stream);
}

// ---- K10's ends of a chain of fused units: K6's SECOND launches alone, fed by partial sums that a convolution launch left
//      (ursa_preact_conv3x3_f32: `partial` = its out_partial, [C][nl] pairs of doubles, nl <= 16) ------------------------------
// forward: merge + normalise (+ ReLU) + running statistics; save: [4][C] = mean, invstd, alpha, beta' (K10's bn_save layout)
int ursa_bn_apply_f32(const float* x, float* y, const double* partial, int32_t nl, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float* save, int64_t N, int64_t C, int64_t HW, float eps,
                      float momentum, uint32_t flags, ursa_stream_t stream)
{
    if (flags & ~URSA_BN_RELU) return URSA_EFLAGS;
    if (N <= 0 || C <= 0 || HW <= 0) return URSA_ESIZE;
    if (!x || !y || !partial || !gamma || !beta || !save) return URSA_ENULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return URSA_ENULL;
    if (nl < 1 || nl > kBnMaxSplit) return URSA_ESIZE;
    if (!bn_aligned4(x) || !bn_aligned4(y) || !bn_aligned4(save) || !bn_aligned16(partial)) return URSA_EALIGN;
    if (N * HW < 2) return URSA_EVALUE;
    BnPlan p;
    const int rc = bn_plan(N, C, HW, bn_aligned16(x) && bn_aligned16(y), &p);
    if (rc) return rc;
    p.g.gate_out = save + 2 * C;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = p.g.cfirst ? dim3(p.g.C, p.S) : dim3(p.S, p.g.C), block(kBnBlock);
    const double2* part = reinterpret_cast<const double2*>(partial);
#define URSA_BN_APPLY(V, R) hipLaunchKernelGGL((k_bn_fwd_apply<V, R>), grid, block, 0, st, x, y, part, (int)nl, gamma, beta, running_mean, \
                                               running_var, save, save + C, eps, momentum, p.g)
    if (p.V == 4) { if (flags & URSA_BN_RELU) URSA_BN_APPLY(4, true); else URSA_BN_APPLY(4, false); }
    else          { if (flags & URSA_BN_RELU) URSA_BN_APPLY(1, true); else URSA_BN_APPLY(1, false); }
#undef URSA_BN_APPLY
    return bn_launch_status();
}

// backward: dx = (((g - gm) - (x - mean) * k) * invstd) * gamma (+ dz), dgamma, dbeta from the merged sums; g = the ALREADY
// gated output gradient (ursa_preact_conv3x3_f32 with URSA_PREACT_BNBWD stored it), so no gate is recomputed here
int ursa_bn_bwd_dx_f32(const float* x, const float* g, const float* dz, float* dx, const float* gamma, const float* save,
                       const double* partial, int32_t nl, float* dgamma, float* dbeta, int64_t N, int64_t C, int64_t HW,
                       ursa_stream_t stream)
{
    if (N <= 0 || C <= 0 || HW <= 0) return URSA_ESIZE;
    if (!x || !g || !dx || !gamma || !save || !partial || !dgamma || !dbeta) return URSA_ENULL;
    if (nl < 1 || nl > 65536) return URSA_ESIZE;
    if (!bn_aligned4(x) || !bn_aligned4(g) || !bn_aligned4(dz) || !bn_aligned4(dx) || !bn_aligned4(save) || !bn_aligned16(partial)) return URSA_EALIGN;
    BnPlan p;
    const int rc = bn_plan(N, C, HW, bn_aligned16(x) && bn_aligned16(g) && bn_aligned16(dz) && bn_aligned16(dx), &p);
    if (rc) return rc;
    p.g.gate_in = save + 2 * C;                                  // (not read without RELU; keeps beta out of the launch)
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = p.g.cfirst ? dim3(p.g.C, p.S) : dim3(p.S, p.g.C), block(kBnBlock);
    const double2* part = reinterpret_cast<const double2*>(partial);
#define URSA_BN_DX(V, RES) do { \
    if (nl > kBnMaxSplit) hipLaunchKernelGGL((k_bn_bwd_dx<V, false, RES, false, true>), grid, block, 0, st, x, g, dz, dx, gamma, (const float*)nullptr, \
                                             save, save + C, part, (int)nl, dgamma, dbeta, p.g, BnGates{}); \
    else hipLaunchKernelGGL((k_bn_bwd_dx<V, false, RES>), grid, block, 0, st, x, g, dz, dx, gamma, (const float*)nullptr, save, \
                            save + C, part, (int)nl, dgamma, dbeta, p.g, BnGates{}); } while (0)
    if (p.V == 4) { if (dz) URSA_BN_DX(4, true); else URSA_BN_DX(4, false); }
    else          { if (dz) URSA_BN_DX(1, true); else URSA_BN_DX(1, false); }
#undef URSA_BN_DX
    return bn_launch_status();
}

}  // extern "C"

// =====================================================================================================================
// K11, the BatchNorm ends of the network's head (URSABench/models/preresnet.py:146-148: `x = self.bn(x); x = self.relu(x);
// x = self.avgpool(x)` on the final 8 x 8 map, and their backward): the rectified activation is only ever averaged, so it is
// neither stored (forward) nor is its gradient (backward: d relu(bn(z)) = dpooled / 64, the same number for a whole map).
namespace {

__device__ __forceinline__ float row_sum16f(float v)
{
    v += bn_dpp<0xB1>(v);
    v += bn_dpp<0x4E>(v);
    v += bn_dpp<0x141>(v);
    v += bn_dpp<0x140>(v);
    return v;
}

// forward: pooled[n][c] = (sum over the 64 positions of max(fmaf(z, alpha_c, beta'_c), 0)) / 64. grid (C, ceil(N / 16)); a 16-lane row
// owns one (n, c) map: lane i its float4 i; fp32 sums: ((x + y) + (z + w)) per lane, then the row's DPP tree. Statistics merged
// from the producer's partial sums exactly as k_bn_fwd_apply does (same scalars, same running statistics).
__global__ __launch_bounds__(kBnBlock) void k_bn_relu_pool64(const float4* __restrict__ z, const double2* __restrict__ partial, int S,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ running_mean, float* __restrict__ running_var,
                                                             float* __restrict__ save, float* __restrict__ pooled, int N, int C, float eps,
                                                             float momentum)
{
    __shared__ float sh[2];
    const int c = blockIdx.x, r = threadIdx.x >> 4, i = threadIdx.x & 15, n = blockIdx.y * 16 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) v = z[((int64_t)n * C + c) * 16 + i];           // before the merge: its dependent chain runs under the load
    if (threadIdx.x < 64) {
        double mean, var;
        const double cnt = (double)N * 64.0;
        bn_merge(partial, c, S, cnt, mean, var);
        if (threadIdx.x == 0) {
            const float meanf = (float)mean;
            const float invstd = (float)(1.0 / sqrt(var + (double)eps));
            const float alpha = invstd * gamma[c];
            sh[0] = alpha;
            sh[1] = fmaf(-meanf, alpha, beta[c]);
            if (blockIdx.y == 0) {
                save[c] = meanf;
                save[C + c] = invstd;
                save[2 * C + c] = alpha;
                save[3 * C + c] = sh[1];
                if (running_mean) {
                    running_mean[c] = momentum * meanf + (1.0f - momentum) * running_mean[c];
                    running_var[c] = momentum * (float)(var * (cnt / (cnt - 1.0))) + (1.0f - momentum) * running_var[c];
                }
            }
        }
    }
    __syncthreads();
    const float scale = sh[0], shift = sh[1];
    const float a = bn_relu_fwd(fmaf(v.x, scale, shift)), b = bn_relu_fwd(fmaf(v.y, scale, shift));
    const float d = bn_relu_fwd(fmaf(v.z, scale, shift)), e = bn_relu_fwd(fmaf(v.w, scale, shift));
    const float s = row_sum16f((a + b) + (d + e));
    if (i == 0 && n < N) pooled[(int64_t)n * C + c] = s * (1.0f / 64.0f);
}

// backward: the gradient of every position of map (n, c) is dpooled[n][c] / 64 where the ReLU was open (gate from the forward's
// saved scalars, as K6's backward). One workgroup per channel holds the channel (N <= 128 maps of 16 float4: 8 per thread) in
// registers: sums in double, the workgroup's fixed tree, then dz from registers - K6's one-pass backward with dy never stored.
constexpr int kPoolEpt = 8;
__global__ __launch_bounds__(kBnBlock) void k_bn_relu_pool64_bwd(const float4* __restrict__ z, const float* __restrict__ dpooled,
                                                                 const float* __restrict__ gamma, const float* __restrict__ save,
                                                                 float4* __restrict__ dz, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                 int N, int C)
{
    __shared__ double shd[2 * kBnBlock / 64];
    const int c = blockIdx.x;
    const float mean = save[c], invstd = save[C + c], scale = save[2 * C + c], shift = save[3 * C + c], w = gamma[c];
    const double meand = (double)mean;
    float4 x[kPoolEpt], g[kPoolEpt];
    const int total = N * 16;
#pragma unroll
    for (int k = 0; k < kPoolEpt; ++k) {
        const int e = threadIdx.x + k * kBnBlock;
        if (e < total) {
            const int n = e >> 4;
            x[k] = z[((int64_t)n * C + c) * 16 + (e & 15)];
            const float d = dpooled[(int64_t)n * C + c] * (1.0f / 64.0f);
            g[k] = make_float4(d, d, d, d);
        }
    }
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < kPoolEpt; ++k) {
        if (threadIdx.x + k * kBnBlock < total) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float xe = comp(x[k], q);
                const float ge = fmaf(xe, scale, shift) > 0.f ? comp(g[k], q) : 0.f;
                setc(g[k], q, ge);
                s1 += (double)ge;
                s2 = fma((double)ge, (double)xe - meand, s2);
            }
        }
    }
    bn_block_sum2(s1, s2, shd);
    const double cnt = (double)N * 64.0, iv = (double)invstd;
    const float gm = (float)(s1 / cnt), kk = (float)(s2 * iv * iv / cnt);
    if (threadIdx.x == 0) { dbeta[c] = (float)s1; dgamma[c] = (float)(s2 * iv); }
#pragma unroll
    for (int k = 0; k < kPoolEpt; ++k) {
        const int e = threadIdx.x + k * kBnBlock;
        if (e < total) {
            float4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) setc(o, q, (((comp(g[k], q) - gm) - (comp(x[k], q) - mean) * kk) * invstd) * w);
            dz[((int64_t)(e >> 4) * C + c) * 16 + (e & 15)] = o;
        }
    }
}

}  // namespace

extern "C" {

int ursa_bn_relu_pool_f32(const float* z, const double* partial, int32_t nl, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float* save, float* pooled, int64_t N, int64_t C, int64_t HW, float eps, float momentum,
                          ursa_stream_t stream)
{
    if (N <= 0 || C <= 0 || HW <= 0) return URSA_ESIZE;
    if (!z || !partial || !gamma || !beta || !save || !pooled) return URSA_ENULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return URSA_ENULL;
    if (HW != 64 || C > 65535 || N > (1 << 20)) return URSA_EVALUE;           // the 8 x 8 map the CIFAR networks end with
    if (nl < 1 || nl > kBnMaxSplit) return URSA_ESIZE;
    if (!bn_aligned16(z) || !bn_aligned16(partial) || !bn_aligned4(save) || !bn_aligned4(pooled)) return URSA_EALIGN;
    hipLaunchKernelGGL(k_bn_relu_pool64, dim3((unsigned)C, (unsigned)((N + 15) / 16)), dim3(kBnBlock), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(z), reinterpret_cast<const double2*>(partial), (int)nl, gamma, beta, running_mean,
                       running_var, save, pooled, (int)N, (int)C, eps, momentum);
    return bn_launch_status();
}

int ursa_bn_relu_pool_bwd_f32(const float* z, const float* dpooled, const float* gamma, const float* save, float* dz, float* dgamma,
                              float* dbeta, int64_t N, int64_t C, int64_t HW, ursa_stream_t stream)
{
    if (N <= 0 || C <= 0 || HW <= 0) return URSA_ESIZE;
    if (!z || !dpooled || !gamma || !save || !dz || !dgamma || !dbeta) return URSA_ENULL;
    if (HW != 64 || C > 65535 || N * 16 > (int64_t)kBnBlock * kPoolEpt) return URSA_EVALUE;   // a channel in one workgroup's registers: N <= 128
    if (!bn_aligned16(z) || !bn_aligned16(dz) || !bn_aligned4(dpooled) || !bn_aligned4(save)) return URSA_EALIGN;
    hipLaunchKernelGGL(k_bn_relu_pool64_bwd, dim3((unsigned)C), dim3(kBnBlock), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(z), dpooled,
                       gamma, save, reinterpret_cast<float4*>(dz), dgamma, dbeta, (int)N, (int)C);
    return bn_launch_status();
}

}  // extern "C"
