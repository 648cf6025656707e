// ursa_conv1x1.hip — K12: the 1x1 / stride 1 convolutions of the Bottleneck pre-activation ResNets (PreResNet-164, BASELINE
// configs[4]), NCHW fp32, gfx950: forward, input gradient and weight gradient.
//
// What it replaces: `self.conv1(out)` / `self.conv3(out)` and the stride-1 `downsample` (URSABench/models/preresnet.py:56,62,
// 76-87,130-136) and their halves of ATen's convolution_backward inside `loss.backward()` / hamiltorch's potential gradient
// (URSABench/inference/hmc.py:71-75). On this stack MIOpen runs them as NCHW->NHWC transposes + an implicit GEMM or a rocBLAS GEMM
// + a transpose back: `batched_transpose_32x32_dword` alone was 12.4 % of the C5 configuration's kernel time, the GEMMs 19 %, the
// weight-gradient implicit GEMMs 7.5 % (profiles/r05_c5_kernel_stats.csv). A 1x1 convolution IS a GEMM on the NCHW planes as they
// lie - per image Y[Cout x HW] = W[Cout x Cin] X[Cin x HW] - so nothing needs transposing.
//
//     y[n][o][p]  = sum_i w[o][i] x[n][i][p]                                  (forward; fma chain over i in groups of 4, ascending)
//     dx[n][i][p] = sum_o w[o][i] dy[n][o][p]                                 (FLIP: the same kernel reading w transposed)
//     dw[o][i]    = sum_{n, p} dy[n][o][p] x[n][i][p]                         (weight gradient: K split over workgroups, fixed order)
//
// Exact fp32 on v_mfma_f32_16x16x4_f32 (every product rounded once, fma chains). These layers are bound by memory, not by the
// matrix pipe (16 -> 64 channels: 2,048 flops per 320 bytes): a workgroup walks whole images - weights gathered into registers
// once, 64-position chunks of x staged through LDS with the next chunk's loads issued before this chunk's matrix work.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ursa_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---- forward / input gradient ---------------------------------------------------------------------------------------------
// A = x: lane (j = lane & 15: position j of a run of 16, k = lane >> 4: channel 4 g + k) reads ONE float of the staged chunk per
// MFMA; B = w: every lane keeps its (output channel j of a 16-channel tile, channel k of every group) weights in registers for the
// whole launch; D: a lane holds four adjacent positions of one output channel: one float4 store.
// Workgroup = 4 waves; a chunk = 64 positions (4 runs) x CK = min(CI, 64) channels in LDS at a row pitch of 80 floats (16 mod 64:
// the 16 positions x 4 channel planes of an A read fall into 64 distinct banks). MT = CO / 16 output tiles: MT >= 4: wave w owns
// tiles w, w + 4, ... and all 4 runs; MT = 2: tile w % 2, runs {w / 2, w / 2 + 2}; MT = 1: run w.
//
// XBN (K13, the Bottleneck networks' `conv(relu(bn(x)))`): x is the BatchNorm's INPUT and `bn` the [4][CI] block K6's statistics
// launch saved (rows 2, 3: the scale / shift of y = fma(x, scale, shift)); a staged row becomes relu(fma(x, scale, shift)) - K6's
// own expression, so the result has the bits of K6's second launch followed by this kernel - and the normalised activation is
// never stored.
__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }   // NaN stays NaN (K6's bn_relu_fwd)
__device__ __forceinline__ f32x4 bn_relu4(const f32x4& v, float scale, float shift)
{
    return f32x4{relu_nan(fmaf(v.x, scale, shift)), relu_nan(fmaf(v.y, scale, shift)), relu_nan(fmaf(v.z, scale, shift)),
                 relu_nan(fmaf(v.w, scale, shift))};
}

//
// EPI (K14, FLIP only: the input gradient of a layer with a BatchNorm + ReLU in front, `bn` = that BatchNorm's saved block, CO its
// channels): the result dh = conv^T(dy, w) is NOT stored. The layers this is for widen on the way back (16 -> 64 channels: dy is a
// quarter of dh), and K6's backward needs the whole batch's two sums before it can produce any dx - so instead of writing dh for K6
// to read twice, the (memory-bound, cheap) GEMM runs twice:
//   EPI 1: the ReLU gate from the BatchNorm's input `aux` at the result's positions, and this workgroup's
//          (sum g, sum g * (aux - mean)) in double per channel -> partial[c][workgroup]: the two sums of K6's first backward launch;
//   EPI 2: the same dh again (same bits), gated, then K6's dx expression
//          dx = (((g - gm) - (aux - mean) * kk) * invstd) * gamma (+ dz) with the per-channel (gm, kk, gamma) from `coef`
//          (ursa_bn_bwd_coef_f32 between the two launches), stored to y.
struct Epi1x1 {
    const float* aux;      // the BatchNorm's input, [N, CO, HW]
    const float* dz;       // EPI 2: the gradient reaching the residual sum on its other path, or nullptr
    const float* coef;     // EPI 2: [3][CO] gm, kk, gamma
    double2* partial;      // EPI 1: [CO][gridDim.x * gridDim.y]
};

template <int CI, int CO, int HW, bool FLIP, bool XBN = false, int EPI = 0>
__global__ __launch_bounds__(kThreads, (EPI && CO < 256) ? 2 : 1) void k_conv1x1(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int N,
                                                      int ipw, int npc,        // npc: position chunks of an image this workgroup walks (grid.y covers the rest)
                                                      const float* __restrict__ bn, const Epi1x1 ep)
{
    constexpr int CK = cmin(CI, 64), NCK = CI / CK, P = 64, PITCH = 80;
    constexpr int MT = CO / 16, NT = cmax(1, MT / 4), WPT = cmax(1, 4 / MT), NR = 4 / WPT;   // tiles per wave, waves per tile, runs per wave
    constexpr int KG = CI / 4, KGC = CK / 4;
    constexpr int NV = CK * (P / 4) / kThreads;                // float4 a thread stages per chunk
    static_assert(CI % 16 == 0 && CO % 16 == 0 && HW % P == 0 && CI % CK == 0 && (CK * (P / 4)) % kThreads == 0, "geometry");
    static_assert(MT == 1 || MT == 2 || MT % 4 == 0, "output tiles per workgroup");
    static_assert(!(XBN && FLIP), "the BatchNorm sits in front of the forward layer");
    static_assert(EPI == 0 || (FLIP && !XBN), "the backward epilogues belong to the flipped launch");
    __shared__ __attribute__((aligned(16))) float xs[CK * PITCH];
    __shared__ float tab[XBN ? 2 * CI : 1];                    // scale[CI], shift[CI]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 15, k = lane >> 4;
    const int tile0 = MT >= 4 ? wave : wave % MT, run0 = MT >= 4 ? 0 : wave / MT;
    if constexpr (XBN) {
        for (int i = tid; i < 2 * CI; i += kThreads) tab[i] = bn[2 * CI + i];
        __syncthreads();
    }

    float wr[NT][KG];                                          // w[o = tile * 16 + j][i = 4 g + k]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const int o = (tile0 + 4 * t) * 16 + j, i = 4 * g + k;
            wr[t][g] = FLIP ? w[(size_t)i * CO + o] : w[(size_t)o * CI + i];
        }

    const int n0 = blockIdx.x * ipw;
    const int n1 = n0 + ipw < N ? n0 + ipw : N;
    const int pc0 = blockIdx.y * npc;                          // this workgroup's position chunks of every image: [pc0, pc0 + npc)
    const int steps = (n1 - n0) * npc * NCK;                   // (image, position chunk, channel chunk) in this order, channel fastest
    f32x4 vx[NV];
    auto load = [&](int s) {                                   // every thread's loads of step s, issued together
        const int n = n0 + s / (npc * NCK), pc = pc0 + (s / NCK) % npc, cc = s % NCK;
        const float* src = x + ((size_t)n * CI + cc * CK) * HW + pc * P;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * kThreads, c = idx / (P / 4), q = idx % (P / 4);
            vx[v] = *reinterpret_cast<const f32x4*>(src + (size_t)c * HW + 4 * q);
        }
    };
    auto stage = [&](int cc) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int idx = tid + v * kThreads, c = idx / (P / 4), q = idx % (P / 4);
            if constexpr (XBN) vx[v] = bn_relu4(vx[v], tab[cc * CK + c], tab[CI + cc * CK + c]);
            *reinterpret_cast<f32x4*>(xs + c * PITCH + 4 * q) = vx[v];
        }
    };
    f32x4 acc[NT][NR];
    // EPI: this lane's output channels (one per tile it owns) are the same for the whole launch
    float e_mean[NT], e_scale[NT], e_shift[NT], e_invstd[NT], e_gm[NT], e_kk[NT], e_w[NT];
    double s1[NT], s2[NT];
    if constexpr (EPI != 0) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int o = (tile0 + 4 * t) * 16 + j;
            e_mean[t] = bn[o], e_invstd[t] = bn[CO + o], e_scale[t] = bn[2 * CO + o], e_shift[t] = bn[3 * CO + o];
            s1[t] = 0.0, s2[t] = 0.0;
            if constexpr (EPI == 2) e_gm[t] = ep.coef[o], e_kk[t] = ep.coef[CO + o], e_w[t] = ep.coef[2 * CO + o];
        }
    }
    if (steps > 0) load(0);
    // EPI: the epilogue's own operands (x, dz at the result's positions) are asked for BEFORE the matrix work where they fit the
    // register file (<= 16 float4), so that the epilogue does not start with a trip to memory
    constexpr bool PREF = EPI != 0 && NT * NR * (EPI == 2 ? 2 : 1) <= 16;
    f32x4 pxa[PREF ? NT : 1][PREF ? NR : 1], pdz[PREF && EPI == 2 ? NT : 1][PREF && EPI == 2 ? NR : 1];
    for (int ip = 0; ip < (n1 - n0) * npc; ++ip) {             // (image, position chunk); the channel chunks unrolled: wr's index static
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[t][r] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PREF) {
            const int n = n0 + ip / npc, pc = pc0 + ip % npc;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const size_t off = ((size_t)n * CO + (tile0 + 4 * t) * 16 + j) * HW + pc * P + (run0 + WPT * r) * 16 + 4 * k;
                    pxa[t][r] = *reinterpret_cast<const f32x4*>(ep.aux + off);
                    if constexpr (EPI == 2) { if (ep.dz) pdz[t][r] = *reinterpret_cast<const f32x4*>(ep.dz + off); }
                }
        }
#pragma unroll
        for (int cc = 0; cc < NCK; ++cc) {
            const int s = ip * NCK + cc;
            if (s > 0) __syncthreads();                        // the previous chunk's reads are done
            stage(cc);
            __syncthreads();
            if (s + 1 < steps) load(s + 1);                    // the next chunk's rows are in flight under this chunk's matrix work
#pragma unroll
            for (int g = 0; g < KGC; ++g) {
                float a[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) a[r] = xs[(4 * g + k) * PITCH + (run0 + WPT * r) * 16 + j];
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        acc[t][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], wr[t][cc * KGC + g], acc[t][r], 0, 0, 0);
            }
        }
        const int n = n0 + ip / npc, pc = pc0 + ip % npc;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const size_t off = ((size_t)n * CO + (tile0 + 4 * t) * 16 + j) * HW + pc * P + (run0 + WPT * r) * 16 + 4 * k;
                if constexpr (EPI == 0) {
                    *reinterpret_cast<f32x4*>(y + off) = acc[t][r];
                } else {
                    f32x4 xa;
                    if constexpr (PREF) xa = pxa[t][r]; else xa = *reinterpret_cast<const f32x4*>(ep.aux + off);
                    f32x4 v = acc[t][r];
                    if constexpr (EPI == 1) {
                        const double md = (double)e_mean[t];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float ge = fmaf(xa[e], e_scale[t], e_shift[t]) > 0.f ? v[e] : 0.f;
                            s1[t] += (double)ge;
                            s2[t] = fma((double)ge, (double)xa[e] - md, s2[t]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float ge = fmaf(xa[e], e_scale[t], e_shift[t]) > 0.f ? v[e] : 0.f;
                            v[e] = (((ge - e_gm[t]) - (xa[e] - e_mean[t]) * e_kk[t]) * e_invstd[t]) * e_w[t];
                        }
                        if (ep.dz) {
                            if constexpr (PREF) v = pdz[t][r] + v; else v = *reinterpret_cast<const f32x4*>(ep.dz + off) + v;
                        }
                        *reinterpret_cast<f32x4*>(y + off) = v;
                    }
                }
            }
    }
    if constexpr (EPI == 1) {
        // a channel's four position groups (k) sit in lanes j, j + 16, j + 32, j + 48 of a wave; for MT < 4 the WPT waves
        // wave % MT, wave % MT + MT, ... hold the same tile: fixed order everywhere
        __shared__ double red[4][NT][16][2];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            double a = s1[t], b = s2[t];
            a += __shfl_xor(a, 16); b += __shfl_xor(b, 16);
            a += __shfl_xor(a, 32); b += __shfl_xor(b, 32);
            if (k == 0) red[wave][t][j][0] = a, red[wave][t][j][1] = b;
        }
        __syncthreads();
        const int nl = gridDim.x * gridDim.y, wg = blockIdx.y * gridDim.x + blockIdx.x;
        if (k == 0 && (MT >= 4 || wave < MT)) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                double a = red[wave][t][j][0], b = red[wave][t][j][1];
                if constexpr (MT < 4) {
#pragma unroll
                    for (int u = 1; u < WPT; ++u) a += red[wave + u * MT][t][j][0], b += red[wave + u * MT][t][j][1];
                }
                ep.partial[(size_t)((tile0 + 4 * t) * 16 + j) * nl + wg] = make_double2(a, b);
            }
        }
    }
}

// ---- weight gradient, first launch: partial dW of a K slice (ipw images), in K7's tile order (ursa_conv.hip) so that K7's
// second launch (k_conv_wgrad_reduce) sums the slices in ascending order and writes dW[o][i] --------------------------------
// GEMM with M = Cout, N = Cin, K = positions: lane (c = lane & 15, g = lane >> 4) reads ONE float4 of dy (its output channel c of a
// tile, positions 16 t + 4 g .. + 3) and ONE float4 of x (its input channel c of a tile): four MFMAs. Chunks of 32 positions of
// all CO + CI channel rows in LDS at a row pitch of 36 floats (36 c mod 64 runs through the 16 multiples of 4: the 16 rows' float4
// reads of one g fall into 64 distinct banks). Waves: WCO = min(4, CO / 16) across output tiles, the rest across input tiles.
// XBN: x is the BatchNorm's input, `bn` K6's saved [4][CI] block: the staged x rows become relu(fma(x, scale, shift)) (see k_conv1x1).
template <int CI, int CO, int HW, bool XBN = false>
__global__ __launch_bounds__(kThreads) void k_conv1x1_wgrad(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial,
                                                            int N, int Cout_unused, int ipw, const float* __restrict__ bn)
{
    constexpr int MTO = CO / 16, CTI = CI / 16, WCO = cmin(4, MTO), WCI = 4 / WCO;
    constexpr int TO = MTO / WCO, TI = CTI / WCI;              // output / input tiles per wave
    constexpr int PK = 32, PITCH = 36, NPC = HW / PK;
    constexpr int TOT = (CO + CI) * (PK / 4), NV = (TOT + kThreads - 1) / kThreads;
    static_assert(MTO % WCO == 0 && CTI % WCI == 0 && CTI >= WCI && HW % PK == 0, "geometry");
    static_assert(((CO + CI) * PITCH + (XBN ? 2 * CI : 0)) * 4 <= 64 * 1024, "static LDS");
    __shared__ __attribute__((aligned(16))) float sm[(CO + CI) * PITCH];     // rows [0, CO): dy, rows [CO, CO + CI): x
    __shared__ float tab[XBN ? 2 * CI : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (XBN) {
        for (int i = tid; i < 2 * CI; i += kThreads) tab[i] = bn[2 * CI + i];
        __syncthreads();
    }
    const int c = lane & 15, g = lane >> 4;
    const int wo = wave % WCO, wi = wave / WCO;
    const int n0 = blockIdx.x * ipw;
    const int n1 = n0 + ipw < N ? n0 + ipw : N;
    const int steps = (n1 - n0) * NPC;
    f32x4 v[NV];
    auto load = [&](int s) {
        const int n = n0 + s / NPC, pc = s % NPC;
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            int idx = tid + u * kThreads;
            if (TOT % kThreads != 0) idx = idx < TOT ? idx : TOT - 1;      // a tail index re-loads the last element (not staged)
            const int row = idx / (PK / 4), q = idx % (PK / 4);
            const float* src = row < CO ? dy + ((size_t)n * CO + row) * HW : x + ((size_t)n * CI + (row - CO)) * HW;
            v[u] = *reinterpret_cast<const f32x4*>(src + pc * PK + 4 * q);
        }
    };
    f32x4 acc[TO][TI];
#pragma unroll
    for (int a = 0; a < TO; ++a)
#pragma unroll
        for (int b = 0; b < TI; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (steps > 0) load(0);
    for (int s = 0; s < steps; ++s) {
        if (s > 0) __syncthreads();
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int idx = tid + u * kThreads, row = idx / (PK / 4), q = idx % (PK / 4);
            if constexpr (XBN) { if (row >= CO && row < CO + CI) v[u] = bn_relu4(v[u], tab[row - CO], tab[CI + row - CO]); }
            if (TOT % kThreads == 0 || idx < TOT) *reinterpret_cast<f32x4*>(sm + row * PITCH + 4 * q) = v[u];
        }
        __syncthreads();
        if (s + 1 < steps) load(s + 1);
#pragma unroll
        for (int t = 0; t < PK / 16; ++t) {
            f32x4 a[TO], b[TI];
#pragma unroll
            for (int p = 0; p < TO; ++p) a[p] = *reinterpret_cast<const f32x4*>(sm + ((wo * TO + p) * 16 + c) * PITCH + 16 * t + 4 * g);
#pragma unroll
            for (int p = 0; p < TI; ++p) b[p] = *reinterpret_cast<const f32x4*>(sm + (CO + (wi * TI + p) * 16 + c) * PITCH + 16 * t + 4 * g);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int p = 0; p < TO; ++p)
#pragma unroll
                    for (int q = 0; q < TI; ++q) acc[p][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[p][tt], b[q][tt], acc[p][q], 0, 0, 0);
        }
    }
    // element (pair * 4 + reg) * 64 + lane holds dW[co = cot * 16 + (lane >> 4) * 4 + reg][ci = cit * 16 + (lane & 15)], pair = cot * CTI + cit
    float* out = partial + (size_t)blockIdx.x * ((size_t)MTO * CTI * 256);
#pragma unroll
    for (int p = 0; p < TO; ++p)
#pragma unroll
        for (int q = 0; q < TI; ++q) {
            const int pair = (wo * TO + p) * CTI + wi * TI + q;
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)pair * 256 + r * 64 + lane] = acc[p][q][r];
        }
}

typedef void (*FwFn)(const float*, const float*, float*, int, int, int, const float*, Epi1x1);

// (CI, CO, H) of the launch: forward = the layer's (Cin, Cout); FLIP = (the layer's Cout, Cin). The bottleneck stages of
// PreResNet-164: 16 <-> 64 at 32 x 32, 32 <-> 128 at 16 x 16, 64 <-> 256 at 8 x 8, and each stage's first block (16 -> 16, 64 -> 32,
// 128 -> 64; the 16 -> 64 stride-1 shortcut is the first shape again).
FwFn fw_for(int64_t CI, int64_t CO, int64_t H, bool flip, bool xbn = false) {
#define URSA_1X1(ci, co, h) if (CI == ci && CO == co && H == h) return flip ? (xbn ? nullptr : (FwFn)k_conv1x1<ci, co, h * h, true>) \
                                : xbn ? (FwFn)k_conv1x1<ci, co, h * h, false, true> : (FwFn)k_conv1x1<ci, co, h * h, false>;
    URSA_1X1(64, 16, 32) URSA_1X1(16, 64, 32) URSA_1X1(128, 32, 16) URSA_1X1(32, 128, 16) URSA_1X1(256, 64, 8) URSA_1X1(64, 256, 8)
    URSA_1X1(16, 16, 32) URSA_1X1(64, 32, 32) URSA_1X1(32, 64, 32) URSA_1X1(128, 64, 16) URSA_1X1(64, 128, 16)
#undef URSA_1X1
    return nullptr;
}

// K14: the widening input gradients behind a BatchNorm (conv1 of a Bottleneck block): (CI = the layer's outputs, CO = its inputs, H)
FwFn bwd_for(int64_t CI, int64_t CO, int64_t H, int epi) {
#define URSA_1X1B(ci, co, h) if (CI == ci && CO == co && H == h) return epi == 1 ? (FwFn)k_conv1x1<ci, co, h * h, true, false, 1> : (FwFn)k_conv1x1<ci, co, h * h, true, false, 2>;
    URSA_1X1B(16, 64, 32) URSA_1X1B(32, 128, 16) URSA_1X1B(64, 256, 8) URSA_1X1B(32, 64, 32) URSA_1X1B(64, 128, 16)
#undef URSA_1X1B
    return nullptr;
}

}  // namespace

// Launch geometry: a workgroup walks `ipw` images x `npc` of an image's HW / 64 position chunks. One whole image per workgroup
// is the better walk when the batch alone gives >= 1,024 workgroups (at 1,024 rows a 32 x 32 image cut in four was SLOWER: 86.6 ->
// 100.4 us, more workgroups re-gathering the weights); small batches cut images until there are 1,024 workgroups; an 8 x 8 image
// (one chunk) is walked two at a time so that the weights are gathered once per several chunks (38.6 -> 32.5 us). tools/k12_bench.py.
static void geometry_for(int64_t N, int64_t H, int* ipw, int* npc, int* gy) {
    const int NPC = (int)(H * H / 64);
    int split = 1;
    while (split < NPC && NPC / (2 * split) >= 2 && N * split < 1024) split *= 2;
    int w = 1;
    if (NPC < 4) while (w < 8 && N / (2 * w) >= 512) w *= 2;
    *ipw = w, *npc = NPC / split, *gy = split;
}

extern "C" int ursa_conv1x1_supported(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, uint32_t flags) {
    if (flags & ~URSA_CONV_FLIP) return 0;
    return N >= 1 && N <= (1 << 20) && H == W && fw_for(Cin, Cout, H, flags & URSA_CONV_FLIP) != nullptr;
}

extern "C" int ursa_conv1x1_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                                uint32_t flags, ursa_stream_t stream) {
    if (flags & ~URSA_CONV_FLIP) return URSA_EFLAGS;
    if (!x || !w || !y) return URSA_ENULL;
    if (N < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return URSA_ESIZE;
    if (((uintptr_t)x | (uintptr_t)y) & 15 || (uintptr_t)w & 3) return URSA_EALIGN;
    if (N > (1 << 20) || H != W) return URSA_EVALUE;
    const FwFn fn = fw_for(Cin, Cout, H, flags & URSA_CONV_FLIP);
    if (!fn) return URSA_EVALUE;
    int ipw, npc, gy;
    geometry_for(N, H, &ipw, &npc, &gy);
    hipLaunchKernelGGL(fn, dim3((unsigned)((N + ipw - 1) / ipw), (unsigned)gy), dim3(kThreads), 0, (hipStream_t)stream, x, w, y, (int)N, ipw, npc,
                       (const float*)nullptr, Epi1x1{});
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? URSA_OK : (int)e;
}

// K13: y = conv1x1(relu(bn(x))) with the BatchNorm's scale / shift taken from `bn_save` (the [4][Cin] block ursa_bn_stats_f32 saved)
extern "C" int ursa_preact_conv1x1_supported(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W) {
    return N >= 1 && N <= (1 << 20) && H == W && fw_for(Cin, Cout, H, false, true) != nullptr;
}

extern "C" int ursa_preact_conv1x1_f32(const float* x, const float* bn_save, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout,
                                       int64_t H, int64_t W, ursa_stream_t stream) {
    if (!x || !bn_save || !w || !y) return URSA_ENULL;
    if (N < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return URSA_ESIZE;
    if (((uintptr_t)x | (uintptr_t)y) & 15 || ((uintptr_t)w | (uintptr_t)bn_save) & 3) return URSA_EALIGN;
    if (N > (1 << 20) || H != W) return URSA_EVALUE;
    const FwFn fn = fw_for(Cin, Cout, H, false, true);
    if (!fn) return URSA_EVALUE;
    int ipw, npc, gy;
    geometry_for(N, H, &ipw, &npc, &gy);
    hipLaunchKernelGGL(fn, dim3((unsigned)((N + ipw - 1) / ipw), (unsigned)gy), dim3(kThreads), 0, (hipStream_t)stream, x, w, y, (int)N, ipw, npc,
                       bn_save, Epi1x1{});
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? URSA_OK : (int)e;
}

// K14: the input gradient of `conv1x1(relu(bn(x)))` and the BatchNorm's backward without storing the convolution's input gradient
extern "C" int64_t ursa_preact_conv1x1_bwd_nl(int64_t N, int64_t Cd, int64_t Cx, int64_t H, int64_t W) {
    if (N < 1 || N > (1 << 20) || H != W || !bwd_for(Cd, Cx, H, 1)) return 0;
    int ipw, npc, gy;
    geometry_for(N, H, &ipw, &npc, &gy);
    return (int64_t)((N + ipw - 1) / ipw) * gy;
}

static int bwd_launch(int epi, const float* dy, const float* w, const float* x, const float* bn_save, double* out_partial, const float* coef,
                      const float* dz, float* dx, int64_t N, int64_t Cd, int64_t Cx, int64_t H, int64_t W, ursa_stream_t stream) {
    if (!dy || !w || !x || !bn_save || (epi == 1 ? !out_partial : (!coef || !dx))) return URSA_ENULL;
    if (N < 1 || Cd < 1 || Cx < 1 || H < 1 || W < 1) return URSA_ESIZE;
    if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dz | (uintptr_t)dx | (uintptr_t)out_partial) & 15 || ((uintptr_t)w | (uintptr_t)bn_save | (uintptr_t)coef) & 3)
        return URSA_EALIGN;
    if (N > (1 << 20) || H != W) return URSA_EVALUE;
    const FwFn fn = bwd_for(Cd, Cx, H, epi);
    if (!fn) return URSA_EVALUE;
    int ipw, npc, gy;
    geometry_for(N, H, &ipw, &npc, &gy);
    Epi1x1 ep;
    ep.aux = x, ep.dz = dz, ep.coef = coef, ep.partial = reinterpret_cast<double2*>(out_partial);
    hipLaunchKernelGGL(fn, dim3((unsigned)((N + ipw - 1) / ipw), (unsigned)gy), dim3(kThreads), 0, (hipStream_t)stream, dy, w, dx, (int)N, ipw, npc,
                       bn_save, ep);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? URSA_OK : (int)e;
}

extern "C" int ursa_preact_conv1x1_bwd_sums_f32(const float* dy, const float* w, const float* x, const float* bn_save, double* out_partial,
                                                int64_t N, int64_t Cd, int64_t Cx, int64_t H, int64_t W, ursa_stream_t stream) {
    return bwd_launch(1, dy, w, x, bn_save, out_partial, nullptr, nullptr, nullptr, N, Cd, Cx, H, W, stream);
}

extern "C" int ursa_preact_conv1x1_bwd_dx_f32(const float* dy, const float* w, const float* x, const float* bn_save, const float* coef,
                                              const float* dz, float* dx, int64_t N, int64_t Cd, int64_t Cx, int64_t H, int64_t W,
                                              ursa_stream_t stream) {
    return bwd_launch(2, dy, w, x, bn_save, nullptr, coef, dz, dx, N, Cd, Cx, H, W, stream);
}

// what ursa_conv.hip's plan_for() asks for a 1x1 / stride 1 weight gradient: the launch, its K slices and the floats of one slice
extern "C" __attribute__((visibility("hidden"))) int ursa_conv1x1_wgrad_plan(int64_t N, int64_t Cin, int64_t Cout, int64_t H, void (**fn)(const float*, const float*, float*, int, int, int, const float*),
                                       int* slices, int* ipw, int64_t* E, int xbn) {
    typedef void (*WgFn)(const float*, const float*, float*, int, int, int, const float*);
    WgFn f = nullptr;
#define URSA_1X1W(ci, co, h) if (Cin == ci && Cout == co && H == h) f = xbn ? (WgFn)k_conv1x1_wgrad<ci, co, h * h, true> : (WgFn)k_conv1x1_wgrad<ci, co, h * h>;
    URSA_1X1W(64, 16, 32) URSA_1X1W(16, 64, 32) URSA_1X1W(128, 32, 16) URSA_1X1W(32, 128, 16) URSA_1X1W(256, 64, 8) URSA_1X1W(64, 256, 8)
    URSA_1X1W(64, 32, 32) URSA_1X1W(128, 64, 16)
#undef URSA_1X1W
    if (!f || N < 1) return 0;
    const int64_t e = (Cout / 16) * (Cin / 16) * 256;
    int w = 1;
    while (N / (2 * w) >= 256 && ((N + w - 1) / w) * e * 4 > (16ll << 20)) w *= 2;   // as many K slices as keep the partial sums <= 16 MB (>= 256)
    *fn = f, *ipw = w, *slices = (int)((N + w - 1) / w), *E = e;
    return 1;
}
