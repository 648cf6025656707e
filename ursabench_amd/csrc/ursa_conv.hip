// K7: weight gradient of the small-channel 3x3 convolutions of the CIFAR pre-activation ResNets, NCHW fp32, gfx950.
//
// What it replaces (URSABench/inference/sghmc.py:80 `loss.backward()` -> ATen convolution_backward -> MIOpen): for
// `dW[co][ci][kh][kw] = sum_{n,oh,ow} dy[n][co][oh][ow] * x[n][ci][oh*s+kh-1][ow*s+kw-1]` at 16..64 channels MIOpen's best
// solver on this stack is an NHWC implicit GEMM with atomics: per layer two NCHW->NHWC transposes, a zero fill, the GEMM and a
// transpose back - 38 us of kernels for 0.6 GFLOP (profiles/r05_step_timeline.json: 92 of a step's 237 launches).
//
// Form. The sum is a GEMM with M = Cout, N = Cin*9, K = batch*OH*OW, taken on `v_mfma_f32_16x16x4_f32` (exact fp32: every
// product rounded once, a k-ordered fma chain - no reduced precision anywhere). One wave owns one (16 co x 16 ci) pair and
// keeps its nine taps in nine accumulators. K runs over positions in groups of 16 (four quarter-rows of four adjacent `ow`):
// lane (c = lane & 15, g = lane >> 4) reads ONE float4 of dy (its co = c, four positions) and per `kh` a six-float window of x
// (its ci = c): the nine taps of those four positions are 36 MFMAs from 7 LDS reads. x and dy tiles are staged in LDS straight
// from NCHW with coalesced float4 loads (halo columns zero), so no transposed copy of anything exists.
//
// The K split over workgroups is reduced in a FIXED order: every workgroup stores its partial dW in tile order
// (coalesced), a second launch sums the S partials of each element in ascending slice order and writes dW[co][ci][3][3].
// No atomics: the same inputs give the same bits on every run (MIOpen's solver does not).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/ursa_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;

// smallest m >= n with m % 64 == 4: plane pitch that spreads 16 planes' float4 reads over all 64 LDS banks
constexpr int pitch64p4(int n) { return ((n - 4 + 63) / 64) * 64 + 4; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// CIN: input channels (all of them are staged); COUT_WG: output channels one workgroup takes (grid.y covers the rest);
// WO: output width = height; R: output rows per band (grid.x = bands x image groups); STRIDE 1 or 2 (pad 1).
template <int CIN, int COUT_WG, int WO, int R, int STRIDE>
struct Wg {
    static constexpr int WI = WO * STRIDE;                    // input width = height
    static constexpr int RI = (R - 1) * STRIDE + 3;           // input rows a band touches
    static constexpr int WP = WI + 8;                         // row pitch: iw = -1 at column 3, iw = 0 at column 4 (16-byte aligned)
    static constexpr int XPLANE = pitch64p4(RI * WP);
    static constexpr int DPLANE = pitch64p4(R * WO);
    static constexpr int XS = CIN * XPLANE, DS = COUT_WG * DPLANE;
    static constexpr int CT = CIN / 16, MT = COUT_WG / 16;
    static constexpr int PW = CT * MT;                        // (co16, ci16) pairs per workgroup: one or more waves each
    static constexpr int KW = 4 / PW;                         // waves that share a pair (they split the K groups)
    static constexpr int G = R * WO / 16;                     // K groups (16 positions) per tile
    static constexpr int BANDS = WO / R;
    static constexpr int RED = KW > 1 ? 4 * 36 * 64 : 0;      // cross-wave sum of the shared pairs
    static constexpr int SMEM = cmax(XS + DS, RED);
    static_assert(CIN % 16 == 0 && COUT_WG % 16 == 0 && (PW == 1 || PW == 2 || PW == 4), "pairs per workgroup");
    static_assert(WO % 4 == 0 && WO % R == 0 && (R * WO) % 16 == 0 && G % KW == 0, "tile geometry");
    static_assert(SMEM * 4 <= 64 * 1024, "static LDS");
};

template <int CIN, int COUT_WG, int WO, int R, int STRIDE>
__global__ __launch_bounds__(kThreads) void k_conv3x3_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ partial, int N, int Cout, int ipw) {
    using C = Wg<CIN, COUT_WG, WO, R, STRIDE>;
    __shared__ __attribute__((aligned(16))) float smem[C::SMEM];
    float* xs = smem;
    float* ds = smem + C::XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int band = blockIdx.x % C::BANDS, ig = blockIdx.x / C::BANDS;
    const int co_base = blockIdx.y * COUT_WG;

    for (int i = tid; i < CIN * C::RI; i += kThreads) {       // halo columns: written once, no load ever touches them
        float* row = xs + (i / C::RI) * C::XPLANE + (i % C::RI) * C::WP;
        row[3] = 0.f;
        row[4 + C::WI] = 0.f;
    }

    f32x4 acc[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pair = wave % C::PW, kpart = wave / C::PW;
    const int cot = pair / C::CT, cit = pair % C::CT;
    const int c = lane & 15, g = lane >> 4;

    // One image's tiles in PH phases of RP output rows. Every thread's global loads are issued together, in the order the
    // phases need them (row-major chunks of 256 float4: x rows first needed by phase p, then dy rows of phase p); phase p stages
    // its chunks into LDS as they arrive and runs its MFMAs while the later rows are still in flight - the launch is bound by
    // the stream of its inputs or by the matrix pipe, not by their sum. Chunks of different phases touch disjoint LDS rows, so
    // one barrier per phase suffices. The next image's loads are issued before the last phase's MFMAs.
    constexpr int PH = 4, RP = R / PH, GP = C::G / PH;
    static_assert(R % PH == 0 && C::G % PH == 0 && GP % C::KW == 0 && (GP * 16) == RP * WO, "phases");
    constexpr int XROW = CIN * (C::WI / 4), DROW = COUT_WG * (WO / 4);   // float4 per staged row
    constexpr int XV = XROW * C::RI, DV = DROW * R;
    constexpr int NX = (XV + kThreads - 1) / kThreads, ND = (DV + kThreads - 1) / kThreads;
    static_assert(STRIDE == 1, "phase bookkeeping below is written for stride 1");
    f32x4 vx[NX], vd[ND];
    // first phase that reads any row of chunk k (phase p reads x rows <= RP*(p+1)+1 and dy rows < RP*(p+1))
    auto xphase = [](int k) { const int fr = k * kThreads / XROW; return fr < 2 ? 0 : (fr - 2) / RP < PH ? (fr - 2) / RP : PH - 1; };
    auto dphase = [](int k) { return (k * kThreads / DROW) / RP; };
    auto load_x = [&](int n, int k) {
        const int idx = tid + k * kThreads;
        const int c4 = idx % (C::WI / 4), ci = (idx / (C::WI / 4)) % CIN, r = idx / XROW;
        int ih = band * R - 1 + r;                             // rows outside the image: a valid row is loaded (every load
        ih = ih < 0 ? 0 : ih >= C::WI ? C::WI - 1 : ih;        // unconditional, so the waits below count exactly) and zeroed when staged
        static_assert(XV % kThreads == 0, "whole chunks");
        vx[k] = *reinterpret_cast<const f32x4*>(x + (((size_t)n * CIN + ci) * C::WI + ih) * C::WI + 4 * c4);
    };
    auto load_d = [&](int n, int k) {
        const int idx = tid + k * kThreads;
        const int c4 = idx % (WO / 4), co = (idx / (WO / 4)) % COUT_WG, r = idx / DROW;
        static_assert(DV % kThreads == 0, "whole chunks");
        vd[k] = *reinterpret_cast<const f32x4*>(dy + (((size_t)n * Cout + co_base + co) * WO + band * R + r) * WO + 4 * c4);
    };
    auto issue = [&](int n) {
#pragma unroll
        for (int p = 0; p < PH; ++p) {
#pragma unroll
            for (int k = 0; k < NX; ++k)
                if (xphase(k) == p) load_x(n, k);
#pragma unroll
            for (int k = 0; k < ND; ++k)
                if (dphase(k) == p) load_d(n, k);
            __builtin_amdgcn_sched_barrier(0);                 // keep the issue order: in-order returns are what the phases wait on
        }
    };
    auto stage = [&](int p) {
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            if (xphase(k) != p) continue;
            const int idx = tid + k * kThreads;
            const int c4 = idx % (C::WI / 4), ci = (idx / (C::WI / 4)) % CIN, r = idx / XROW;
            const int ih = band * R - 1 + r;
            const bool in = ih >= 0 && ih < C::WI;
            *reinterpret_cast<f32x4*>(xs + ci * C::XPLANE + r * C::WP + 4 + 4 * c4) = in ? vx[k] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            if (dphase(k) != p) continue;
            const int idx = tid + k * kThreads;
            const int c4 = idx % (WO / 4), co = (idx / (WO / 4)) % COUT_WG, r = idx / DROW;
            *reinterpret_cast<f32x4*>(ds + co * C::DPLANE + r * WO + 4 * c4) = vd[k];
        }
    };

    const int n0 = ig * ipw;
    const int n1 = n0 + ipw < N ? n0 + ipw : N;                // this workgroup's images: [n0, n1), uniform
    if (n0 < n1) issue(n0);
    for (int n = n0; n < n1; ++n) {
        if (n > n0) __syncthreads();                           // the previous image's LDS reads are done
#pragma unroll
        for (int p = 0; p < PH; ++p) {
            stage(p);
            __syncthreads();
            if (p == PH - 1 && n + 1 < n1) issue(n + 1);
            for (int t = p * GP + kpart; t < (p + 1) * GP; t += C::KW) {
                const int q = t * 4 + g;                       // this lane group's quarter-row: four adjacent ow of one row
                const int row = q / (WO / 4), ow0 = 4 * (q % (WO / 4));
                const f32x4 a = *reinterpret_cast<const f32x4*>(ds + (cot * 16 + c) * C::DPLANE + row * WO + ow0);
                const float* xb = xs + (cit * 16 + c) * C::XPLANE + row * C::WP + ow0 + 3;   // iw = ow0 - 1
#pragma unroll
                for (int kh = 0; kh < 3; ++kh) {
                    const float* pw = xb + kh * C::WP;
                    const float w0 = pw[0], w5 = pw[5];
                    const f32x4 m = *reinterpret_cast<const f32x4*>(pw + 1);
                    const float win[6] = {w0, m.x, m.y, m.z, m.w, w5};
#pragma unroll
                    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                        for (int kw = 0; kw < 3; ++kw)
                            acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt], win[tt + kw], acc[kh * 3 + kw], 0, 0, 0);
                }
            }
        }
    }

    // partial dW of this K slice, tile order: element ((pair * 9 + tap) * 4 + reg) * 64 + lane holds
    // dW[co = cot*16 + (lane >> 4)*4 + reg][ci = cit*16 + (lane & 15)][tap]
    const size_t E = (size_t)Cout * CIN * 9;
    float* out = partial + (size_t)blockIdx.x * E;
    if constexpr (C::KW == 1) {
        const int pair_g = (blockIdx.y * C::MT + cot) * C::CT + cit;
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)pair_g * 2304 + (j * 4 + r) * 64 + lane] = acc[j][r];
    } else {
        __syncthreads();                                       // every wave is done with the tiles: reuse them
        float* red = smem;
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * 36 + j * 4 + r) * 64 + lane] = acc[j][r];
        __syncthreads();
        for (int e = tid; e < C::PW * 2304; e += kThreads) {
            const int p = e / 2304, rem = e % 2304;
            float s = red[p * 2304 + rem];                     // waves p, p + PW, p + 2 PW, ...: ascending K part
#pragma unroll
            for (int kp = 1; kp < C::KW; ++kp) s += red[(p + C::PW * kp) * 2304 + rem];
            const int pair_g = (blockIdx.y * C::MT + p / C::CT) * C::CT + p % C::CT;
            out[(size_t)pair_g * 2304 + rem] = s;
        }
    }
}

// Second launch: dW[co][ci][tap] = sum over the S slices, ascending. 16 float4 columns x 16 slice classes per workgroup.
__global__ __launch_bounds__(kThreads) void k_conv_wgrad_reduce(const float* __restrict__ partial, float* __restrict__ dw, int S,
                                                                 int E4, int Cin) {
    __shared__ f32x4 red[16][16];
    const int tid = threadIdx.x, col = tid & 15, sp = tid >> 4;
    const int e4 = blockIdx.x * 16 + col;
    const f32x4* p4 = reinterpret_cast<const f32x4*>(partial);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int k = sp;
    for (; k + 16 * 7 < S; k += 16 * 8) {                     // eight loads in flight, added in ascending slice order
        f32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p4[(size_t)(k + 16 * u) * E4 + e4];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < S; k += 16) s += p4[(size_t)k * E4 + e4];
    red[sp][col] = s;
    __syncthreads();
    if (sp == 0) {
        f32x4 t = red[0][col];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][col];
        const int e = e4 * 4, lane = e & 63, reg = (e >> 6) & 3, tap = (e >> 8) % 9, pr = (e >> 8) / 9;
        const int CT = Cin / 16, cot = pr / CT, cit = pr % CT;
        const int co = cot * 16 + (lane >> 4) * 4 + reg, ci = cit * 16 + (lane & 15);
        float* o = dw + ((size_t)co * Cin + ci) * 9 + tap;
        o[0] = t.x, o[9] = t.y, o[18] = t.z, o[27] = t.w;
    }
}

struct Plan {
    int slices;       // grid.x = K slices = partial copies
    int gy;           // grid.y
    int ipw;          // images per workgroup
    void (*fn)(const float*, const float*, float*, int, int, int);
};

// Which shapes K7 takes: the three stages of the CIFAR pre-activation ResNets (URSABench/models/preresnet.py:62-64,149-151),
// stride 1. Anything else -> slices = 0, and the caller keeps MIOpen's weight gradient.
Plan plan_for(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int stride) {
    Plan p = {0, 0, 0, nullptr};
    if (N < 1 || N > (1 << 20) || H != W) return p;
    int bands = 0;
    if (stride == 1 && Cin == 16 && Cout == 16 && W == 32) {
        p.ipw = 1, bands = 4, p.gy = 1, p.fn = k_conv3x3_wgrad<16, 16, 32, 8, 1>;
    } else if (stride == 1 && Cin == 32 && Cout == 32 && W == 16) {
        p.ipw = 1, bands = 2, p.gy = 1, p.fn = k_conv3x3_wgrad<32, 32, 16, 8, 1>;
    } else if (stride == 1 && Cin == 64 && Cout == 64 && W == 8) {
        p.ipw = 2, bands = 1, p.gy = 4, p.fn = k_conv3x3_wgrad<64, 16, 8, 8, 1>;
    } else {
        return p;
    }
#ifdef URSA_DEBUG_KNOBS
    if (const char* e = getenv("URSA_CONV_IPW")) {            // images per workgroup: fewer, longer K slices
        const int v = atoi(e);
        if (v >= 1 && v <= 64) p.ipw = v;
    }
#endif
    p.slices = (int)((N + p.ipw - 1) / p.ipw) * bands;
    return p;
}

}  // namespace

extern "C" int64_t ursa_conv3x3_wgrad_ws_floats(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int32_t stride) {
    const Plan p = plan_for(N, Cin, Cout, H, W, stride);
    return p.slices ? (int64_t)p.slices * Cout * Cin * 9 : 0;
}

extern "C" int ursa_conv3x3_wgrad_f32(const float* x, const float* dy, float* dw, float* ws, int64_t ws_floats, int64_t N,
                                      int64_t Cin, int64_t Cout, int64_t H, int64_t W, int32_t stride, ursa_stream_t stream) {
    if (!x || !dy || !dw || !ws) return URSA_ENULL;
    if (N < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return URSA_ESIZE;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)ws) & 15) return URSA_EALIGN;
    if ((uintptr_t)dw & 3) return URSA_EALIGN;
    const Plan p = plan_for(N, Cin, Cout, H, W, stride);
    if (!p.slices) return URSA_EVALUE;                         // shape not covered: ursa_conv3x3_wgrad_ws_floats() said 0
    const int64_t E = Cout * Cin * 9;
    if (ws_floats < (int64_t)p.slices * E) return URSA_ESIZE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(p.fn, dim3(p.slices * 1, p.gy), dim3(kThreads), 0, s, x, dy, ws, (int)N, (int)Cout, p.ipw);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_conv_wgrad_reduce, dim3((unsigned)(E / 64)), dim3(kThreads), 0, s, ws, dw, p.slices, (int)(E / 4), (int)Cin);
    e = hipGetLastError();
    return e == hipSuccess ? URSA_OK : (int)e;
}
