// K7: weight gradient of the small-channel convolutions of the CIFAR pre-activation ResNets, NCHW fp32, gfx950.
//
// What it replaces (URSABench/inference/sghmc.py:80 `loss.backward()` -> ATen convolution_backward -> MIOpen): for
// `dW[co][ci][kh][kw] = sum_{n,oh,ow} dy[n][co][oh][ow] * x[n][ci][oh*s+kh-p][ow*s+kw-p]` at 16..64 channels MIOpen's best
// solver on this stack is an NHWC implicit GEMM with atomics: per layer two NCHW->NHWC transposes, a zero fill, the GEMM and a
// transpose back - 38 us of kernels for 0.6 GFLOP (profiles/r05_step_timeline.json: 92 of a step's 237 launches).
//
// Form. The sum is a GEMM with M = Cout, N = Cin*taps, K = batch*OH*OW, taken on `v_mfma_f32_16x16x4_f32` (exact fp32: every
// product rounded once, a k-ordered fma chain - no reduced precision anywhere). One wave owns one (16 co x 16 ci) pair and
// keeps its taps in one accumulator each. K runs over positions in groups of 16 (four quarter-rows of four adjacent `ow`):
// lane (c = lane & 15, g = lane >> 4) reads ONE float4 of dy (its co = c, four positions) and per `kh` a window of x
// (its ci = c; six floats at stride 1, nine at stride 2): the nine taps of those four positions are 36 MFMAs from 7 (10) LDS
// reads. x and dy tiles are staged in LDS straight from NCHW with coalesced float4 loads (halo columns zero), so no
// transposed copy of anything exists. A 1x1 / stride 2 shortcut convolution is the centre tap of the 3x3 / stride 2 form.
//
// The K split over workgroups is reduced in a FIXED order: every workgroup stores its partial dW in tile order
// (coalesced), a second launch sums the S partials of each element in ascending slice order and writes dW[co][ci][kh][kw].
// No atomics: the same inputs give the same bits on every run (MIOpen's solver does not). The second launch can be
// deferred and taken for many layers at once (ursa_conv_wgrad_reduce_f32): a backward pass then costs one launch per layer
// plus one.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/ursa_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
// Registers: left to itself the compiler gives these kernels up to 256 VGPRs PLUS the accumulators' 16 - 38 AGPRs - more than half of
// a SIMD's 512, i.e. ONE wave per SIMD, one workgroup per CU at a time - although nothing needs that many live values. Asking for two
// waves per SIMD (`__launch_bounds__(256, 2)`) makes it allocate <= 256 in all (no AGPRs, no scratch except where noted) and two to
// three workgroups share a CU: the paired backward launch 19.9 / 20.7 / 20.9 -> 17.7 / 17.8 / 17.6 us (its two roles really overlap),
// the 64-channel evaluation unit 190.6 -> 167.0 us at 4,096 rows; a launch with one workgroup per CU does not change
// (profiles/r06_minw_ab.json). URSA_MINW=1 (knobs build: make KNOBS_EXTRA=-DURSA_MINW=1) restores the compiler's choice.
#ifndef URSA_MINW
#define URSA_MINW 2
#endif
// (the strided 64 -> 32 input gradient with BatchNorm-backward sums spills 100 bytes per lane under that cap and is slower alone,
// 9.1 -> 10.9 us: it keeps the compiler's allocation; inside the paired launch the cap still pays, 16.9 -> 15.9 us)
constexpr int minw_for(int CIN, int MODE, int EPI) { return (CIN == 64 && MODE == 2 && EPI == 3) ? 1 : URSA_MINW; }

// smallest m >= n with m % 64 == 4: plane pitch that spreads 16 planes' float4 reads over all 64 LDS banks
constexpr int pitch64p4(int n) { return ((n - 4 + 63) / 64) * 64 + 4; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---------------------------------------------------------------------------------------------------------------------
// K10 (fused pre-activation unit, below): what K8 / K7 are told when the BatchNorm + ReLU that precedes a convolution
// (URSABench/models/preresnet.py:40-42,45-47) is applied while the input tile is staged, and the statistics the NEXT
// BatchNorm needs - or the two sums its backward needs - are taken from the accumulators before they are stored.
typedef unsigned long long u64;
typedef u64 __attribute__((address_space(1))) gu64;
typedef unsigned int __attribute__((address_space(1))) gu32;
constexpr int kLines = 16;                  // most partial sums per channel a consumer merges (one 16-lane DPP row)
constexpr u64 kSlotXor = 0xFFF8DEADBEEF0001ull;   // a NaN payload no sum of floats has: a published word is never zero
constexpr unsigned kPollLimit = 1u << 20;

struct Fuse {
    // PRO: y = conv(relu(bn(x))): the batch statistics of x arrive as per-line partial sums (sum x, sum x^2 in double)
    const double2* in_partial;              // [CIN][in_nl]
    const float* gamma;                     // [CIN]
    const float* beta;
    float* running_mean;                    // or null
    float* running_var;
    float* save;                            // [4][CIN]: mean, invstd, alpha = invstd * gamma, beta' = fma(-mean, alpha, beta)
    double in_count;                        // N * H * W of x
    float eps, momentum;
    int in_nl;
    // EPI
    int nl;                                 // lines = partial sums per output channel this launch leaves in out_partial
    int line_sz;                            // workgroups (grid.x indices) per line
    const float* aux;                       // EPI 2: the addend (shape of y); EPI 3: the BatchNorm input at y's positions
    const float* bsave;                     // EPI 3: [4][Cout] of that BatchNorm (mean, invstd, alpha, beta')
    u64* slots;                             // [Cout][grid.x][2]: zero at launch, zero again when the launch has drained
    double2* out_partial;                   // [Cout][nl]
    unsigned int* tickets;                  // [grid.y][kLines][32]: one counter per 128-byte line, zero at launch and afterwards
    unsigned int* err;                      // sticky: a bounded poll ran out (never in a correct run); the sums are NaN then
};

// sum of a double over the 16 lanes of a DPP row, the same bits in all 16, fixed tree (the first four steps of K6's
// bn_wave_sum: merging <= 16 partial sums here gives the doubles K6's own merge gives for the same 16 numbers)
__device__ __forceinline__ double row_sum16(double v)
{
#define URSA_ROW_DPP64(CTRL) do { \
        const long long b_ = __builtin_bit_cast(long long, v); \
        const int lo_ = __builtin_amdgcn_update_dpp(0, (int)(b_ & 0xffffffffll), CTRL, 0xF, 0xF, true); \
        const int hi_ = __builtin_amdgcn_update_dpp(0, (int)(b_ >> 32), CTRL, 0xF, 0xF, true); \
        v += __builtin_bit_cast(double, ((long long)hi_ << 32) | (unsigned int)lo_); } while (0)
    URSA_ROW_DPP64(0xB1);
    URSA_ROW_DPP64(0x4E);
    URSA_ROW_DPP64(0x141);
    URSA_ROW_DPP64(0x140);
#undef URSA_ROW_DPP64
    return v;
}

__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }   // NaN stays NaN (K6's bn_relu_fwd)
__device__ __forceinline__ f32x4 bn_relu4(const f32x4& v, float scale, float shift)
{
    return f32x4{relu_nan(fmaf(v.x, scale, shift)), relu_nan(fmaf(v.y, scale, shift)), relu_nan(fmaf(v.z, scale, shift)),
                 relu_nan(fmaf(v.w, scale, shift))};
}

// End of a launch that leaves per-channel sums (EPI): every thread holds (s1, s2) of ITS output channel (channel
// co_base + 16 * (wave % MT) + (lane & 15)). The workgroups of one grid.y row are cut into <= 16 "lines" of line_sz consecutive
// grid.x indices; a line's sums are added in ascending grid.x order by ONE of its workgroups and stored as the line's partial
// sum: a fixed summation order whatever the arrival order, no atomics on the data.
//   * Who adds: the workgroup of the line that STARTED last. Every workgroup takes a ticket on its line's counter when it
//     starts (take_ticket: one relaxed agent-scope fetch-add whose result is first needed here, a whole launch later - its
//     round trip hides behind the convolution); the one that drew the last ticket knows every other workgroup of the line is
//     running, and a running workgroup of this launch never waits for anything - so waiting for THEIR sums cannot starve,
//     whatever else shares the device (other streams, graph branches), and needs no assumption about dispatch order.
//   * How the sums travel: the other workgroups store theirs to their slot with agent-scope atomic stores (they bypass the
//     non-coherent per-XCD L2 lines) as bits XOR a NaN payload, so a zero word means "not there yet", and exit. The adder
//     polls the line's slots (agent-scope loads; bounded: a poll that runs out raises *err and makes the sums NaN - never in a
//     correct run), adds them by position with its own sums (which never leave the workgroup), stores the partial sum, zeroes
//     the slots and re-arms the counter: the scratch is zero again when the launch has drained.
// One memory round trip (the poll) is what the launch pays at its end.
__device__ __forceinline__ unsigned take_ticket(const Fuse& f, int bx, int by)
{
    unsigned t = 0;
    if (threadIdx.x == 0) {
        const int line = bx / f.line_sz;
        gu32* tp = (gu32*)(f.tickets + ((size_t)by * kLines + line) * 32);
        t = __hip_atomic_fetch_add(tp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return t;
}

// EPI 3 (the input-gradient forms): no hand-over inside the launch at all. The launch that follows is per channel (K6's dx
// launch), so every workgroup simply stores its sums at out_partial[channel][grid.x index] and that launch adds a channel's
// S partial sums itself, in ascending order (k_bn_bwd_dx<.., BIGS>): nothing waits at the end of the convolution.
template <int COUT_WG, int MT, int WPC>
__device__ __forceinline__ void store_sums(const Fuse& f, double s1, double s2, float* smem, int co_base, int bx, int gx)
{
    const int tid = threadIdx.x;
    __syncthreads();                                           // every wave is done with the tile: reuse it
    double2* red = reinterpret_cast<double2*>(smem);
    red[tid] = make_double2(s1, s2);
    __syncthreads();
    if (tid < COUT_WG) {
        const int cc = tid >> 4, jj = tid & 15;
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int ws = 0; ws < WPC; ++ws)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const double2 v = red[(cc + MT * ws) * 64 + kq * 16 + jj];
                a += v.x;
                b += v.y;
            }
        f.out_partial[(size_t)(co_base + tid) * gx + bx] = make_double2(a, b);
    }
}

template <int COUT_WG, int MT, int WPC>
__device__ __forceinline__ void publish_sums(const Fuse& f, unsigned ticket, double s1, double s2, float* smem, int co_base, int bx, int by,
                                             int gx)
{
    const int tid = threadIdx.x;
    __shared__ int adder;
    __shared__ double2 own[COUT_WG];
    const int S = gx, tile = bx, line = tile / f.line_sz;
    const int first = line * f.line_sz;
    const int cnt = S - first < f.line_sz ? S - first : f.line_sz;
    __syncthreads();                                           // every wave is done with the tile: reuse it
    double2* red = reinterpret_cast<double2*>(smem);
    red[tid] = make_double2(s1, s2);
    if (tid == 0) adder = ticket == (unsigned)cnt - 1u;
    __syncthreads();
    const bool is_adder = adder != 0;
    if (tid < COUT_WG) {
        const int cc = tid >> 4, jj = tid & 15;
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int ws = 0; ws < WPC; ++ws)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const double2 v = red[(cc + MT * ws) * 64 + kq * 16 + jj];
                a += v.x;
                b += v.y;
            }
        if (is_adder) {
            own[tid] = make_double2(a, b);
        } else {
            gu64* sl = (gu64*)(f.slots + ((size_t)(co_base + tid) * S + tile) * 2);
            __hip_atomic_store(sl, __builtin_bit_cast(u64, a) ^ kSlotXor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sl + 1, __builtin_bit_cast(u64, b) ^ kSlotXor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (!is_adder) return;
    __syncthreads();
    const int mine = tile - first;                             // this workgroup's position in the line
    for (int e = tid; e < COUT_WG * 16; e += kThreads) {       // (channel, lane of its row): COUT_WG * 16 is a multiple of 256
        const int c = e >> 4, i = e & 15;
        double a = 0.0, b = 0.0;
        bool bad = false;
        for (int t = i; t < cnt; t += 16) {                   // this lane's positions, ascending
            if (t == mine) {
                a += own[c].x;
                b += own[c].y;
                continue;
            }
            gu64* sl = (gu64*)(f.slots + ((size_t)(co_base + c) * S + first + t) * 2);
            u64 wa, wb;
            unsigned spins = 0;
            for (;;) {
                wa = __hip_atomic_load(sl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                wb = __hip_atomic_load(sl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (wa != 0 && wb != 0) break;
                if (++spins > kPollLimit) { bad = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            a += __builtin_bit_cast(double, wa ^ kSlotXor);
            b += __builtin_bit_cast(double, wb ^ kSlotXor);
            __hip_atomic_store(sl, (u64)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sl + 1, (u64)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (bad) {
            a = b = __builtin_nan("");
            __hip_atomic_store((gu32*)f.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        a = row_sum16(a);
        b = row_sum16(b);
        if (i == 0) f.out_partial[(size_t)(co_base + c) * f.nl + line] = make_double2(a, b);
    }
    if (tid == 0)                                              // every workgroup of the line has counted itself: re-armed for the next launch
        __hip_atomic_store((gu32*)(f.tickets + ((size_t)by * kLines + line) * 32), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// CIN: input channels (all staged; padded to a multiple of 16 by one zero plane); COUT_WG: output channels one workgroup takes
// (grid.y covers the rest); WO: output width = height; R: output rows per band (grid.x = bands x image groups); STRIDE 1 / 2;
// TAPS 9 (3x3, pad 1) or 1 (the centre tap alone: 1x1, pad 0, at STRIDE 2); PH: phases per tile (see the kernel).
template <int CIN, int COUT_WG, int WO, int R, int STRIDE, int TAPS, int PH>
struct Wg {
    static constexpr int CINP = (CIN + 15) / 16 * 16;
    static constexpr int WI = WO * STRIDE;                    // input width = height
    static constexpr int RI = (R - 1) * STRIDE + 3;           // input rows a band touches
    static constexpr int WP = WI + 8;                         // row pitch: iw = -1 at column 3, iw = 0 at column 4 (16-byte aligned)
    static constexpr int XPLANE = pitch64p4(RI * WP);
    static constexpr int DPLANE = pitch64p4(R * WO);
    static constexpr int NPLANES = CIN + (CINP > CIN ? 1 : 0);   // + one zero plane every padded channel reads
    static constexpr int XS = NPLANES * XPLANE, DS = COUT_WG * DPLANE;
    static constexpr int CT = CINP / 16, MT = COUT_WG / 16;
    static constexpr int PW = CT * MT;                        // (co16, ci16) pairs per workgroup: one or more waves each
    static constexpr int KW = 4 / PW;                         // waves that share a pair (they split the K groups)
    static constexpr int G = R * WO / 16;                     // K groups (16 positions) per tile
    static constexpr int BANDS = WO / R;
    static constexpr int TE = TAPS * 256;                     // floats of one pair's partial tile
    static constexpr int RED = KW > 1 ? 4 * TE : 0;           // cross-wave sum of the shared pairs
    static constexpr int SMEM = cmax(XS + DS, RED);
    static constexpr int RP = R / PH, GP = G / PH;            // output rows / K groups per phase
    static_assert(COUT_WG % 16 == 0 && (PW == 1 || PW == 2 || PW == 4), "pairs per workgroup");
    static_assert(WO % 4 == 0 && WO % R == 0 && (R * WO) % 16 == 0, "tile geometry");
    static_assert(R % PH == 0 && G % PH == 0 && GP % KW == 0 && GP * 16 == RP * WO, "phases");
    static_assert((TAPS == 9) || (TAPS == 1 && STRIDE == 2), "taps");
    static_assert(SMEM * 4 <= 64 * 1024, "static LDS");
};

// XBN: the x operand is relu(fma(x, alpha[ci], beta'[ci])) - the BatchNorm + ReLU that stands in front of the layer (K10),
// applied as the tile is staged from the scale / shift the forward saved (xbn: [4][CIN], rows 2 and 3), so the normalised
// activation is never stored; the padding stays zero.
// (bx, by): the workgroup's place in the launch's logical grid - blockIdx of k_conv_wgrad; a workgroup of the paired backward
// launch (k_bwd_pair below) plays this role with indices of its own. smem: >= Wg::SMEM floats, 16-byte aligned.
template <int CIN, int COUT_WG, int WO, int R, int STRIDE, int TAPS, int PH, bool XBN>
__device__ __forceinline__ void wgrad_body(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ partial, int N,
                                           int Cout, int ipw, const float* __restrict__ xbn, float* smem, const int bx, const int by) {
    using C = Wg<CIN, COUT_WG, WO, R, STRIDE, TAPS, PH>;
    static_assert(!XBN || kThreads % (CIN * (C::WI / 4)) == 0, "XBN: a thread stages one channel");
    float* xs = smem;
    float* ds = smem + C::XS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int band = bx % C::BANDS, ig = bx / C::BANDS;
    const int co_base = by * COUT_WG;

    for (int i = tid; i < CIN * C::RI; i += kThreads) {       // halo columns: written once, no load ever touches them
        float* row = xs + (i / C::RI) * C::XPLANE + (i % C::RI) * C::WP;
        row[3] = 0.f;
        row[4 + C::WI] = 0.f;
    }
    if constexpr (C::CINP > CIN)
        for (int i = tid; i < C::XPLANE; i += kThreads) xs[CIN * C::XPLANE + i] = 0.f;

    f32x4 acc[TAPS];
#pragma unroll
    for (int j = 0; j < TAPS; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int pair = wave % C::PW, kpart = wave / C::PW;
    const int cot = pair / C::CT, cit = pair % C::CT;
    const int c = lane & 15, g = lane >> 4;
    const int xplane = (cit * 16 + c < CIN) ? cit * 16 + c : CIN;      // padded channels read the zero plane
    float xb_scale = 0.f, xb_shift = 0.f;                      // XBN: this thread's staged channel is the same for every chunk
    if constexpr (XBN) {
        const int ci_t = (tid / (C::WI / 4)) % CIN;
        xb_scale = xbn[2 * CIN + ci_t];
        xb_shift = xbn[3 * CIN + ci_t];
    }

    // One image's tiles in PH phases of RP output rows. Every thread's global loads are issued together, in the order the
    // phases need them (row-major chunks of 256 float4: x rows first needed by phase p, then dy rows of phase p); phase p stages
    // its chunks into LDS as they arrive and runs its MFMAs while the later rows are still in flight. Chunks of different
    // phases touch disjoint LDS rows, so one barrier per phase suffices. Every load is unconditional (rows outside the image
    // load a valid row and are zeroed when staged; a chunk's tail past the tile re-loads its last element), so the compiler's
    // wait counts are exact. The next image's loads are issued before the last phase's MFMAs.
    constexpr int XROW = CIN * (C::WI / 4), DROW = COUT_WG * (WO / 4);   // float4 per staged row
    constexpr int XV = XROW * C::RI, DV = DROW * R;
    constexpr int NX = (XV + kThreads - 1) / kThreads, ND = (DV + kThreads - 1) / kThreads;
    f32x4 vx[NX], vd[ND];
    // first phase that reads any row of chunk k: phase p reads x rows <= STRIDE*(RP*(p+1) - 1) + 2 and dy rows < RP*(p+1)
    auto xphase = [](int k) {
        const int fr = k * kThreads / XROW;
        int p = 0;
        while (p < PH - 1 && STRIDE * (C::RP * (p + 1) - 1) + 2 < fr) ++p;
        return p;
    };
    auto dphase = [](int k) {
        const int p = (k * kThreads / DROW) / C::RP;
        return p < PH ? p : PH - 1;
    };
    auto load_x = [&](int n, int k) {
        int idx = tid + k * kThreads;
        if (XV % kThreads != 0) idx = idx < XV ? idx : XV - 1;
        const int c4 = idx % (C::WI / 4), ci = (idx / (C::WI / 4)) % CIN, r = idx / XROW;
        int ih = band * R * STRIDE - 1 + r;
        ih = ih < 0 ? 0 : ih >= C::WI ? C::WI - 1 : ih;
        vx[k] = *reinterpret_cast<const f32x4*>(x + (((size_t)n * CIN + ci) * C::WI + ih) * C::WI + 4 * c4);
    };
    auto load_d = [&](int n, int k) {
        int idx = tid + k * kThreads;
        if (DV % kThreads != 0) idx = idx < DV ? idx : DV - 1;
        const int c4 = idx % (WO / 4), co = (idx / (WO / 4)) % COUT_WG, r = idx / DROW;
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
            const int ih = band * R * STRIDE - 1 + r;
            const bool in = ih >= 0 && ih < C::WI;
            if (XV % kThreads == 0 || idx < XV)
                *reinterpret_cast<f32x4*>(xs + ci * C::XPLANE + r * C::WP + 4 + 4 * c4) =
                    in ? (XBN ? bn_relu4(vx[k], xb_scale, xb_shift) : vx[k]) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            if (dphase(k) != p) continue;
            const int idx = tid + k * kThreads;
            const int c4 = idx % (WO / 4), co = (idx / (WO / 4)) % COUT_WG, r = idx / DROW;
            if (DV % kThreads == 0 || idx < DV) *reinterpret_cast<f32x4*>(ds + co * C::DPLANE + r * WO + 4 * c4) = vd[k];
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
            for (int t = p * C::GP + kpart; t < (p + 1) * C::GP; t += C::KW) {
                const int q = t * 4 + g;                       // this lane group's quarter-row: four adjacent ow of one row
                const int row = q / (WO / 4), ow0 = 4 * (q % (WO / 4));
                const f32x4 a = *reinterpret_cast<const f32x4*>(ds + (cot * 16 + c) * C::DPLANE + row * WO + ow0);
                const float* xb = xs + xplane * C::XPLANE + row * STRIDE * C::WP + ow0 * STRIDE + 3;   // iw = ow0*s - 1
                if constexpr (TAPS == 1) {                     // centre tap at stride 2: iw = 2*(ow0 + tt), ih = 2*row
                    const float* pw = xb + C::WP;
                    const f32x4 m0 = *reinterpret_cast<const f32x4*>(pw + 1), m1 = *reinterpret_cast<const f32x4*>(pw + 5);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], m0.x, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], m0.z, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], m1.x, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], m1.z, acc[0], 0, 0, 0);
                } else {
#pragma unroll
                    for (int kh = 0; kh < 3; ++kh) {
                        const float* pw = xb + kh * C::WP;
                        if constexpr (STRIDE == 1) {
                            const float w0 = pw[0], w5 = pw[5];
                            const f32x4 m = *reinterpret_cast<const f32x4*>(pw + 1);
                            const float win[6] = {w0, m.x, m.y, m.z, m.w, w5};
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                                for (int kw = 0; kw < 3; ++kw)
                                    acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt], win[tt + kw], acc[kh * 3 + kw], 0, 0, 0);
                        } else {
                            const float w0 = pw[0];            // iw = 2*ow0 - 1 .. 2*ow0 + 7: nine floats
                            const f32x4 m0 = *reinterpret_cast<const f32x4*>(pw + 1), m1 = *reinterpret_cast<const f32x4*>(pw + 5);
                            const float win[9] = {w0, m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
                            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                                for (int kw = 0; kw < 3; ++kw)
                                    acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tt], win[2 * tt + kw], acc[kh * 3 + kw], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }

    // partial dW of this K slice, tile order: element ((pair * TAPS + tap) * 4 + reg) * 64 + lane holds
    // dW[co = cot*16 + (lane >> 4)*4 + reg][ci = cit*16 + (lane & 15)][tap]
    const size_t E = (size_t)(Cout / 16) * C::CT * C::TE;
    float* out = partial + (size_t)bx * E;
    if constexpr (C::KW == 1) {
        const int pair_g = (by * C::MT + cot) * C::CT + cit;
#pragma unroll
        for (int j = 0; j < TAPS; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)pair_g * C::TE + (j * 4 + r) * 64 + lane] = acc[j][r];
    } else {
        __syncthreads();                                       // every wave is done with the tiles: reuse them
        float* red = smem;
#pragma unroll
        for (int j = 0; j < TAPS; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave * C::TE + (j * 4 + r) * 64 + lane] = acc[j][r];
        __syncthreads();
        for (int e = tid; e < C::PW * C::TE; e += kThreads) {
            const int p = e / C::TE, rem = e % C::TE;
            float s = red[p * C::TE + rem];                    // waves p, p + PW, p + 2 PW, ...: ascending K part
#pragma unroll
            for (int kp = 1; kp < C::KW; ++kp) s += red[(p + C::PW * kp) * C::TE + rem];
            const int pair_g = (by * C::MT + p / C::CT) * C::CT + p % C::CT;
            out[(size_t)pair_g * C::TE + rem] = s;
        }
    }
}

template <int CIN, int COUT_WG, int WO, int R, int STRIDE, int TAPS, int PH, bool XBN = false>
__global__ __launch_bounds__(kThreads) void k_conv_wgrad(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ partial, int N, int Cout, int ipw,
                                                          const float* __restrict__ xbn) {
    __shared__ __attribute__((aligned(16))) float smem[Wg<CIN, COUT_WG, WO, R, STRIDE, TAPS, PH>::SMEM];
    wgrad_body<CIN, COUT_WG, WO, R, STRIDE, TAPS, PH, XBN>(x, dy, partial, N, Cout, ipw, xbn, smem, blockIdx.x, blockIdx.y);
}

// Second launch, for up to kMaxItems layers at once: dW[co][ci][tap] = sum over the layer's S slices, ascending.
// 16 float4 columns x 16 slice classes per workgroup; a workgroup finds its layer by its first block index.
constexpr int kMaxItems = 48;
struct ReduceItem {
    const float* partial;
    float* dw;
    int S, E4, Cin, taps, blk0, pad;
};
struct ReduceItems {
    int n, pad;
    ReduceItem it[kMaxItems];
};

__global__ __launch_bounds__(kThreads) void k_conv_wgrad_reduce(const ReduceItems items) {
    __shared__ f32x4 red[16][16];
    int li = 0;
    while (li + 1 < items.n && (int)blockIdx.x >= items.it[li + 1].blk0) ++li;
    const ReduceItem& I = items.it[li];
    const int tid = threadIdx.x, col = tid & 15, sp = tid >> 4;
    const int e4 = ((int)blockIdx.x - I.blk0) * 16 + col;
    const int S = I.S, E4 = I.E4;
    const f32x4* p4 = reinterpret_cast<const f32x4*>(I.partial);
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
        for (int kk = 1; kk < 16; ++kk) t += red[kk][col];
        const int e = e4 * 4, lane = e & 63, reg = (e >> 6) & 3, tap = (e >> 8) % I.taps, pr = (e >> 8) / I.taps;
        const int CT = (I.Cin + 15) / 16, cot = pr / CT, cit = pr % CT;
        const int co = cot * 16 + (lane >> 4) * 4 + reg, ci = cit * 16 + (lane & 15);
        float* o = I.dw + ((size_t)co * I.Cin + ci) * I.taps + tap;
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (ci + j < I.Cin) o[(size_t)j * I.taps] = tv[j];  // (channels padded up to 16 exist in the partial tiles only)
    }
}

struct Plan {
    int slices;       // grid.x = K slices = partial copies
    int gy;           // grid.y
    int ipw;          // images per workgroup
    int taps;
    int64_t E;        // floats of one partial copy
    void (*fn)(const float*, const float*, float*, int, int, int, const float*);
};

// Which shapes K7 takes: every convolution of the CIFAR pre-activation ResNets with BasicBlocks
// (URSABench/models/preresnet.py:25-27,100,130-136): the three stages' 3x3 / stride 1 layers, the stem (3 -> 16), the two
// 3x3 / stride 2 layers that open stages 2 and 3 and their 1x1 / stride 2 shortcuts. Anything else -> slices = 0, and the
// caller keeps MIOpen's weight gradient.
}  // namespace
// K12 (ursa_conv1x1.hip): the 1x1 / stride 1 weight gradient's first launch, partial sums in this file's tile order
extern "C" __attribute__((visibility("hidden"))) int ursa_conv1x1_wgrad_plan(int64_t N, int64_t Cin, int64_t Cout, int64_t H,
                                                                             void (**fn)(const float*, const float*, float*, int, int, int, const float*),
                                                                             int* slices, int* ipw, int64_t* E, int xbn);
namespace {

// xbn: the K10 form (x operand normalised + rectified while staged): the 3x3 layers that follow a BatchNorm (all but the stem)
Plan plan_for(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int ksize, int stride, bool xbn = false) {
    Plan p = {0, 0, 0, 0, 0, nullptr};
    if (N < 1 || N > (1 << 20) || H != W) return p;
    int bands = 0;
    const bool k3 = ksize == 3, k1 = ksize == 1;
    if (xbn && ((!k3 && !(k1 && stride == 1)) || Cin == 3)) return p;
    if (k1 && stride == 1) {                                   // K12: the Bottleneck networks' 1x1 layers (xbn: K13)
        if (!ursa_conv1x1_wgrad_plan(N, Cin, Cout, H, &p.fn, &p.slices, &p.ipw, &p.E, xbn)) return Plan{0, 0, 0, 0, 0, nullptr};
        p.gy = 1, p.taps = 1;
        return p;
    }
    if (k3 && stride == 1 && Cin == 16 && Cout == 16 && W == 32) {
        // two images per workgroup: 256 K slices at batch 128 = one per CU; alone the launch takes what 512 slices take
        // (9.1 us), beside the input-gradient workgroups of the paired launch 21.8 -> 19.8 us, and half the partial sums for
        // the second launch to read (tools/k10_bench.py with URSA_CONV_IPW, profiles/r06_k7_ipw_ab.json; 32 channels and the
        // stride-2 layers lose with two: 9.1 -> 13.2, 7.1 -> 9.8 us)
        p.ipw = 2, bands = 4, p.gy = 1, p.fn = xbn ? k_conv_wgrad<16, 16, 32, 8, 1, 9, 4, true> : k_conv_wgrad<16, 16, 32, 8, 1, 9, 4>;
    } else if (k3 && stride == 1 && Cin == 32 && Cout == 32 && W == 16) {
        p.ipw = 1, bands = 2, p.gy = 1, p.fn = xbn ? k_conv_wgrad<32, 32, 16, 8, 1, 9, 4, true> : k_conv_wgrad<32, 32, 16, 8, 1, 9, 4>;
    } else if (k3 && stride == 1 && Cin == 64 && Cout == 64 && W == 8) {
        p.ipw = 2, bands = 1, p.gy = 4, p.fn = xbn ? k_conv_wgrad<64, 16, 8, 8, 1, 9, 4, true> : k_conv_wgrad<64, 16, 8, 8, 1, 9, 4>;
    } else if (k3 && stride == 1 && Cin == 3 && Cout == 16 && W == 32) {
        p.ipw = 1, bands = 4, p.gy = 1, p.fn = k_conv_wgrad<3, 16, 32, 8, 1, 9, 4>;
    } else if (k3 && stride == 2 && Cin == 16 && Cout == 32 && W == 32) {
        p.ipw = 1, bands = 2, p.gy = 1, p.fn = xbn ? k_conv_wgrad<16, 32, 16, 8, 2, 9, 4, true> : k_conv_wgrad<16, 32, 16, 8, 2, 9, 4>;
    } else if (k3 && stride == 2 && Cin == 32 && Cout == 64 && W == 16) {
        p.ipw = 2, bands = 2, p.gy = 2, p.fn = xbn ? k_conv_wgrad<32, 32, 8, 4, 2, 9, 2, true> : k_conv_wgrad<32, 32, 8, 4, 2, 9, 2>;
    } else if (k1 && stride == 2 && Cin == 16 && Cout == 32 && W == 32) {
        p.ipw = 2, bands = 2, p.gy = 1, p.fn = k_conv_wgrad<16, 32, 16, 8, 2, 1, 4>;
    } else if (k1 && stride == 2 && Cin == 32 && Cout == 64 && W == 16) {
        p.ipw = 2, bands = 2, p.gy = 2, p.fn = k_conv_wgrad<32, 32, 8, 4, 2, 1, 2>;
    } else {
        return p;
    }
#ifdef URSA_DEBUG_KNOBS
    if (const char* e = getenv("URSA_CONV_IPW")) {            // images per workgroup: fewer, longer K slices
        const int v = atoi(e);
        if (v >= 1 && v <= 64) p.ipw = v;
    }
#endif
    while ((N + p.ipw - 1) / p.ipw * bands > 768) p.ipw *= 2;   // large batches (HMC's 1,024-row chunks): longer K slices, not more
    p.taps = k3 ? 9 : 1;
    p.slices = (int)((N + p.ipw - 1) / p.ipw) * bands;
    p.E = (Cout / 16) * ((Cin + 15) / 16) * (int64_t)p.taps * 256;
    return p;
}

int launch_reduce(const ReduceItems& items, int blocks, hipStream_t s) {
    hipLaunchKernelGGL(k_conv_wgrad_reduce, dim3((unsigned)blocks), dim3(kThreads), 0, s, items);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? URSA_OK : (int)e;
}

int check_shape(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W) {
    return (N < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) ? URSA_ESIZE : URSA_OK;
}

}  // namespace

extern "C" int64_t ursa_conv_wgrad_ws_floats(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int32_t ksize,
                                             int32_t stride) {
    const Plan p = plan_for(N, Cin, Cout, H, W, ksize, stride);
    return p.slices ? (int64_t)p.slices * p.E : 0;
}

static int wgrad_partial_impl(const float* x, const float* xbn, const float* dy, float* ws, int64_t ws_floats, int64_t N, int64_t Cin,
                              int64_t Cout, int64_t H, int64_t W, int32_t ksize, int32_t stride, ursa_stream_t stream) {
    if (!x || !dy || !ws) return URSA_ENULL;
    if (int rc = check_shape(N, Cin, Cout, H, W)) return rc;
    if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)ws) & 15 || (uintptr_t)xbn & 3) return URSA_EALIGN;
    const Plan p = plan_for(N, Cin, Cout, H, W, ksize, stride, xbn != nullptr);
    if (!p.slices) return URSA_EVALUE;                         // shape not covered: ursa_conv_wgrad_ws_floats() said 0
    if (ws_floats < (int64_t)p.slices * p.E) return URSA_ESIZE;
    hipLaunchKernelGGL(p.fn, dim3(p.slices, p.gy), dim3(kThreads), 0, (hipStream_t)stream, x, dy, ws, (int)N, (int)Cout, p.ipw, xbn);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? URSA_OK : (int)e;
}

extern "C" int ursa_conv_wgrad_partial_f32(const float* x, const float* dy, float* ws, int64_t ws_floats, int64_t N, int64_t Cin,
                                           int64_t Cout, int64_t H, int64_t W, int32_t ksize, int32_t stride,
                                           ursa_stream_t stream) {
    return wgrad_partial_impl(x, nullptr, dy, ws, ws_floats, N, Cin, Cout, H, W, ksize, stride, stream);
}

// K10: the same first launch with x = relu(fma(x, alpha, beta')) taken while the tile is staged (bn_save: the [4][Cin] block
// ursa_preact_conv3x3_f32 saved for the BatchNorm in front of this layer). Same slices, same scratch, same second launch.
extern "C" int ursa_preact_wgrad_partial_f32(const float* x, const float* bn_save, const float* dy, float* ws, int64_t ws_floats,
                                             int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int32_t stride,
                                             ursa_stream_t stream) {
    if (!bn_save) return URSA_ENULL;
    return wgrad_partial_impl(x, bn_save, dy, ws, ws_floats, N, Cin, Cout, H, W, 3, stride, stream);
}

// K13: the same for a 1x1 / stride 1 layer behind a BatchNorm + ReLU (bn_save: what ursa_bn_stats_f32 saved)
extern "C" int ursa_preact_wgrad1x1_partial_f32(const float* x, const float* bn_save, const float* dy, float* ws, int64_t ws_floats,
                                                int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, ursa_stream_t stream) {
    if (!bn_save) return URSA_ENULL;
    return wgrad_partial_impl(x, bn_save, dy, ws, ws_floats, N, Cin, Cout, H, W, 1, 1, stream);
}

extern "C" int ursa_conv_wgrad_reduce_f32(const ursa_conv_pending* items, int32_t n, ursa_stream_t stream) {
    if (n < 0) return URSA_ESIZE;
    if (n == 0) return URSA_OK;
    if (!items) return URSA_ENULL;
    for (int i = 0; i < n; ++i) {                               // everything is checked before anything is launched
        const ursa_conv_pending& q = items[i];
        if (!q.ws || !q.dw) return URSA_ENULL;
        if (int rc = check_shape(q.N, q.Cin, q.Cout, q.H, q.W)) return rc;
        if ((uintptr_t)q.ws & 15 || (uintptr_t)q.dw & 3) return URSA_EALIGN;
        if (!plan_for(q.N, q.Cin, q.Cout, q.H, q.W, q.ksize, q.stride).slices) return URSA_EVALUE;
    }
    ReduceItems r;
    r.n = 0, r.pad = 0;
    int blocks = 0;
    for (int i = 0; i < n; ++i) {
        const ursa_conv_pending& q = items[i];
        const Plan p = plan_for(q.N, q.Cin, q.Cout, q.H, q.W, q.ksize, q.stride);
        ReduceItem& it = r.it[r.n++];
        it.partial = q.ws, it.dw = q.dw, it.S = p.slices, it.E4 = (int)(p.E / 4), it.Cin = (int)q.Cin, it.taps = p.taps;
        it.blk0 = blocks, it.pad = 0;
        blocks += (int)(p.E / 64);
        if (r.n == kMaxItems || i == n - 1) {
            if (int rc = launch_reduce(r, blocks, (hipStream_t)stream)) return rc;
            r.n = 0, blocks = 0;
        }
    }
    return URSA_OK;
}

extern "C" int ursa_conv_wgrad_f32(const float* x, const float* dy, float* dw, float* ws, int64_t ws_floats, int64_t N,
                                   int64_t Cin, int64_t Cout, int64_t H, int64_t W, int32_t ksize, int32_t stride,
                                   ursa_stream_t stream) {
    if (!dw) return URSA_ENULL;
    if ((uintptr_t)dw & 3) return URSA_EALIGN;
    if (int rc = ursa_conv_wgrad_partial_f32(x, dy, ws, ws_floats, N, Cin, Cout, H, W, ksize, stride, stream)) return rc;
    const ursa_conv_pending one = {ws, dw, N, Cin, Cout, H, W, ksize, stride};
    return ursa_conv_wgrad_reduce_f32(&one, 1, stream);
}

namespace {

// =====================================================================================================================
// K8: forward / input gradient of the stride-1 3x3 convolutions, NCHW fp32, gfx950.
//
// What it replaces: `self.conv(x)` (URSABench/models/preresnet.py:42,47) and the input-gradient half of ATen's
// convolution_backward - MIOpen's Winograd F(2,3) assembly kernel on this stack, 18.8 us per layer and direction at
// 16..64 channels (33 of a PreResNet-20 step's launches, 42 % of its kernel time, profiles/r05_step_timeline.json).
//
//     y[n][o][oh][ow] = sum_{i, kh, kw} w[o][i][kh][kw] * x[n][i][oh + kh - 1][ow + kw - 1]                      (forward)
//     dx = the same with w'[o][i][kh][kw] = w[i][o][2 - kh][2 - kw] applied to dy                                (flip)
//
// Form: a GEMM with M = positions, N = output channels, K = Cin * 9 on `v_mfma_f32_16x16x4_f32` (exact fp32 products, fma
// chains). A = x: lane (i = lane & 15, k = lane >> 4) reads ONE float of the staged x tile (position i of a run of 16,
// input channel 4*g + k, shifted by the tap) per MFMA; B = w: every lane keeps its (o = lane & 15, channel k) weights of all
// taps and channel groups in registers for the whole launch (Cin/4 * 9 of them, read once through LDS from a coalesced load).
// D: a lane holds four adjacent positions of one output channel: one float4 store. x is staged exactly as in K7 (phases).
constexpr int pitch64p16(int n) { return ((n - 16 + 63) / 64) * 64 + 16; }

// MODE 0: stride 1 (forward, or the flipped form = input gradient); the position grid (W x W) is the output's = the input's.
// MODE 1: stride 2 forward: positions = the OUTPUT grid (W x W), the staged tensor x is 2W x 2W.
// MODE 2: stride 2 input gradient: positions = the grid of the staged tensor dy (W x W); every position produces the 2 x 2
//         cell of dx above it: dx[2a + ph][2b + pw] = sum over the taps of parity class (ph, pw):
//         ph = 0: kh = 1 (dy row a);  ph = 1: kh = 0 (dy row a + 1) and kh = 2 (dy row a);  the same for pw / kw / columns.
template <int CIN, int COUT_WG, int W, int R, int PH, int MODE>
struct Fw {
    static constexpr int CINP = (CIN + 3) / 4 * 4;
    static constexpr int KG = CINP / 4;                       // channel groups of 4 = MFMA k steps per tap
    static constexpr int WIN = MODE == 1 ? 2 * W : W;         // the staged tensor's width = height
    static constexpr int RI = MODE == 0 ? R + 2 : MODE == 1 ? 2 * R + 1 : R + 1, WP = WIN + 8;
    static constexpr int XPLANE = pitch64p16(RI * WP);        // 4 channels x 16 positions of an A read: 64 distinct banks (stride 1)
    static constexpr int XS = CINP * XPLANE;
    static constexpr int WPITCH = pitch64p4(CINP * 9);        // weights, rows [o][channel * 9 + tap]: staged before x, same LDS
    static constexpr int WL = WPITCH * COUT_WG;
    static constexpr int SMEM = cmax(XS, WL);
    static constexpr int MT = COUT_WG / 16, WPC = 4 / MT;     // waves per output-channel tile (they split the positions)
    static constexpr int PT = R * W / 16;                     // runs of 16 positions per workgroup tile
    static constexpr int BANDS = W / R;
    static constexpr int RP = R / PH, TP = PT / PH;           // rows / runs per phase
    static constexpr int TPW = TP / WPC;                      // runs per wave per phase
    static constexpr int NA = MODE == 2 ? 4 : 9;              // distinct A values per step
    static_assert(MT == 1 || MT == 2 || MT == 4, "output tiles per workgroup");
    static_assert(W % 4 == 0 && W % R == 0 && (R * W) % 16 == 0 && R % PH == 0 && PT % PH == 0 && TP % WPC == 0, "geometry");
    static_assert(TP * 16 == RP * W, "a phase's runs are its rows");
    static_assert(SMEM * 4 <= 64 * 1024, "static LDS");
    // last staged row (relative to the tile's first) that phase p reads
    static constexpr int need(int p) { return MODE == 0 ? RP * (p + 1) + 1 : MODE == 1 ? 2 * RP * (p + 1) : RP * (p + 1); }
};

// K10 forms of the same launch (PRO / EPI; `f` is only read when one of them is set):
//   PRO 1: the staged tensor is relu(bn(x)) with bn's batch statistics merged here from the producer's per-line partial
//          sums (every workgroup the same 16-lane tree: identical scalars); workgroup (0, 0) also stores mean / invstd / alpha /
//          beta' for the backward and updates the running statistics. The zero padding stays zero.
//   PRO 2: the same staging transform in EVALUATION mode: scale / shift from the running statistics (K6's k_bn_eval expressions:
//          invstd = 1 / sqrtf(running_var + eps), alpha = invstd * gamma, beta' = fmaf(-running_mean, alpha, beta)); nothing saved.
//   EPI 4: y + addend stored (`out += residual`), no sums (evaluation).
//   EPI 1: per-channel (sum y, sum y^2) of the output in double -> publish_sums: what the NEXT BatchNorm needs.
//   EPI 2: the same of z = y + addend (`out += residual`, preresnet.py:49-52); z is what is stored.
//   EPI 3: (input-gradient forms) g = the ReLU gate of the BatchNorm in front of the layer applied to the result
//          (gate recomputed from that BatchNorm's input `aux` and saved scalars, as K6's backward does), g is what is stored, and
//          (sum g, sum g * (aux - mean)) in double -> publish_sums: the two sums of native_batch_norm_backward.
// (bx, by, gx): the workgroup's place in the launch's logical grid and that grid's x extent - blockIdx / gridDim.x of k_conv3x3;
// a workgroup of the paired backward launch (k_bwd_pair) plays this role with indices of its own. smem: >= Fw::SMEM floats.
template <int CIN, int COUT_WG, int W, int R, int PH, int MODE, int DBG, int PRO, int EPI>   // DBG != 0: knobs-build experiments only (what bounds the launch)
__device__ __forceinline__ void conv_body(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y, int N, int Cout,
                                          int ipw, int flip_arg, const Fuse& f, float* smem, const int bx, const int by, const int gx) {
    using C = Fw<CIN, COUT_WG, W, R, PH, MODE>;
    static_assert(!PRO || kThreads % (CIN * (C::WIN / 4)) == 0, "PRO: a thread stages one channel");
    static_assert(!EPI || C::TPW == 1, "EPI: one run per wave and phase");
    __shared__ float2 tab[PRO ? CIN : 1];
    float* xs = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int band = bx % C::BANDS, ig = bx / C::BANDS;
    const int co_base = by * COUT_WG;
    const int j = lane & 15, k = lane >> 4;
    const int cot = wave % C::MT, wsub = wave / C::MT;
    const int flip = MODE == 2 ? 1 : MODE == 1 ? 0 : flip_arg;   // which weight layout is read (MODE 2: the transposed one)
    const int row0 = MODE == 0 ? band * R - 1 : MODE == 1 ? band * R * 2 - 1 : band * R;   // the tile's first staged row

    constexpr int XROW = CIN * (C::WIN / 4);
    constexpr int XV = XROW * C::RI;
    constexpr int NX = (XV + kThreads - 1) / kThreads;
    f32x4 vx[NX];
    auto xphase = [](int kk) {
        const int fr = kk * kThreads / XROW;
        int p = 0;
        while (p < PH - 1 && C::need(p) < fr) ++p;
        return p;
    };
    auto load_x = [&](int n, int kk) {
        int idx = tid + kk * kThreads;
        if (XV % kThreads != 0) idx = idx < XV ? idx : XV - 1;
        const int c4 = idx % (C::WIN / 4), ci = (idx / (C::WIN / 4)) % CIN, r = idx / XROW;
        int ih = row0 + r;
        ih = ih < 0 ? 0 : ih >= C::WIN ? C::WIN - 1 : ih;
        vx[kk] = *reinterpret_cast<const f32x4*>(x + (((size_t)n * CIN + ci) * C::WIN + ih) * C::WIN + 4 * c4);
    };
    auto issue = [&](int n) {
#pragma unroll
        for (int p = 0; p < PH; ++p) {
#pragma unroll
            for (int kk = 0; kk < NX; ++kk)
                if (xphase(kk) == p) load_x(n, kk);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    float pro_scale = 0.f, pro_shift = 0.f;                    // PRO: this thread's staged channel is the same for every chunk
    auto stage = [&](int p) {
#pragma unroll
        for (int kk = 0; kk < NX; ++kk) {
            if (xphase(kk) != p) continue;
            const int idx = tid + kk * kThreads;
            const int c4 = idx % (C::WIN / 4), ci = (idx / (C::WIN / 4)) % CIN, r = idx / XROW;
            const int ih = row0 + r;
            const bool in = ih >= 0 && ih < C::WIN;
            if (XV % kThreads == 0 || idx < XV)
                *reinterpret_cast<f32x4*>(xs + ci * C::XPLANE + r * C::WP + 4 + 4 * c4) =
                    in ? (PRO ? bn_relu4(vx[kk], pro_scale, pro_shift) : vx[kk]) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    // EPI 2 / 3: the second operand of the epilogue (addend / BatchNorm input) at this lane's output positions, loaded with the
    // image's rows: NAUX float4 per image (MODE 2 stores a 2 x 8 block per run: four of them)
    constexpr int NAUX = (EPI >= 2) ? PH * (MODE == 2 ? 4 : 1) : 1;
    f32x4 av[NAUX];
    auto aux_off = [&](int n, int p) -> size_t {               // offset of this lane's first output float4 of phase p, image n
        const int t = p * C::TP + wsub;
        const size_t img = ((size_t)n * Cout + co_base + cot * 16 + j) * (MODE == 2 ? 4 * W * W : W * W);
        if constexpr (MODE == 2) {
            const int lin0 = t * 16 + 4 * k, prow = band * R + lin0 / W, pcol = lin0 % W;
            return img + (size_t)(2 * prow) * (2 * W) + 2 * pcol;
        } else {
            return img + band * R * W + t * 16 + 4 * k;
        }
    };
    auto issue_aux = [&](int n) {
        if constexpr (EPI >= 2) {
#pragma unroll
            for (int p = 0; p < PH; ++p) {
                const float* a = f.aux + aux_off(n, p);
                if constexpr (MODE == 2) {
                    av[4 * p] = *reinterpret_cast<const f32x4*>(a);
                    av[4 * p + 1] = *reinterpret_cast<const f32x4*>(a + 4);
                    av[4 * p + 2] = *reinterpret_cast<const f32x4*>(a + 2 * W);
                    av[4 * p + 3] = *reinterpret_cast<const f32x4*>(a + 2 * W + 4);
                } else {
                    av[p] = *reinterpret_cast<const f32x4*>(a);
                }
            }
        }
    };

    const int n0 = ig * ipw;
    const int n1 = n0 + ipw < N ? n0 + ipw : N;
    if (n0 < n1) issue(n0);                                    // the first image's rows are in flight while the weights are read
    if (n0 < n1) issue_aux(n0);
    unsigned ticket = 0;
    if constexpr (EPI == 1 || EPI == 2) ticket = take_ticket(f, bx, by);   // "this workgroup runs": needed at the very end (publish_sums)

    __shared__ double2 dsum[PRO ? CIN : 1];
    if constexpr (PRO == 1) {
        // per-channel sums of x from the producer's partial sums: lane (channel, part) of a 16-lane row loads one partial, the row
        // adds them (K6's tree). The scalars are finished below, between the two barriers of the weight staging, by one thread
        // per channel.
        for (int e = tid; e < CIN * 16; e += kThreads) {       // CIN * 16 is a multiple of 256: every lane takes part in the DPP sums
            const int c = e >> 4, i = e & 15;
            double a = 0.0, b = 0.0;
            if (i < f.in_nl) { const double2 q = f.in_partial[(size_t)c * f.in_nl + i]; a = q.x; b = q.y; }
            a = row_sum16(a);
            b = row_sum16(b);
            if (i == 0) dsum[c] = make_double2(a, b);
        }
    }
    auto pro_finish = [&]() {                                  // after a barrier: torch's CPU BatchNorm rounding, K6's code
        if constexpr (PRO == 2) {
            if (tid < CIN) {
                const float invstd = 1.0f / sqrtf(f.running_var[tid] + f.eps);
                const float alpha = invstd * f.gamma[tid];
                tab[tid] = make_float2(alpha, fmaf(-f.running_mean[tid], alpha, f.beta[tid]));
            }
        }
        if constexpr (PRO == 1) {
            if (tid < CIN) {
                const int c = tid;
                const double cnt = f.in_count;
                const double mean = dsum[c].x / cnt;
                double var = dsum[c].y / cnt - mean * mean;
                if (var < 0.0) var = 0.0;
                const float meanf = (float)mean;
                const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
                const float alpha = invstd * f.gamma[c];
                const float shift = fmaf(-meanf, alpha, f.beta[c]);
                tab[c] = make_float2(alpha, shift);
                if (bx == 0 && by == 0) {
                    f.save[c] = meanf;
                    f.save[CIN + c] = invstd;
                    f.save[2 * CIN + c] = alpha;
                    f.save[3 * CIN + c] = shift;
                    if (f.running_mean) {      // torch: running = momentum * batch + (1 - momentum) * running, unbiased variance
                        f.running_mean[c] = f.momentum * meanf + (1.0f - f.momentum) * f.running_mean[c];
                        f.running_var[c] = f.momentum * (float)(var * (cnt / (cnt - 1.0))) + (1.0f - f.momentum) * f.running_var[c];
                    }
                }
            }
        }
    };

    // weights: coalesced global reads -> LDS rows [o][channel * 9 + tap] at a pitch of 4 mod 64 floats (lane (o = j, channel k)
    // reads bank 4 j + 9 k + const: 64 distinct banks) -> this lane's Cin/4 * 9 registers
    float wr[C::KG][9];
    {
        float* wl = smem;
        constexpr int ROW = CIN * 9;
        if constexpr (C::CINP > CIN)
            for (int i = tid; i < COUT_WG * (C::CINP - CIN) * 9; i += kThreads)
                wl[(i / ((C::CINP - CIN) * 9)) * C::WPITCH + ROW + i % ((C::CINP - CIN) * 9)] = 0.f;
        // every thread's loads are issued together (unrolled, unconditional: a tail index re-loads the last element), then
        // written: one L2 round trip for the whole tile instead of one per loop iteration
        if constexpr (ROW % 4 == 0) {
            constexpr int TOT4 = COUT_WG * ROW / 4, NW = (TOT4 + kThreads - 1) / kThreads;
            f32x4 tw[NW];
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                int idx = tid + i * kThreads;
                idx = idx < TOT4 ? idx : TOT4 - 1;
                if (!flip) {
                    tw[i] = reinterpret_cast<const f32x4*>(w + (size_t)co_base * ROW)[idx];   // (co_base * ROW * 4 bytes: 16-byte multiple)
                } else {                                       // per input channel: COUT_WG * 9 contiguous floats (16-byte aligned runs)
                    constexpr int RUN4 = COUT_WG * 9 / 4;
                    tw[i] = *reinterpret_cast<const f32x4*>(w + ((size_t)(idx / RUN4) * Cout + co_base) * 9 + 4 * (idx % RUN4));
                }
            }
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const int idx = tid + i * kThreads;
                if (idx >= TOT4) continue;
                if (!flip) {
                    *reinterpret_cast<f32x4*>(wl + (idx / (ROW / 4)) * C::WPITCH + 4 * (idx % (ROW / 4))) = tw[i];
                } else {                                       // w'[o][ci][tap] = w[ci][o][8 - tap], w: [CIN][Cout][3][3]
                    constexpr int RUN4 = COUT_WG * 9 / 4;
                    const int ci = idx / RUN4, r4 = 4 * (idx % RUN4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) wl[((r4 + e) / 9) * C::WPITCH + ci * 9 + 8 - (r4 + e) % 9] = tw[i][e];
                }
            }
        } else {                                               // the stem (27 floats per row): scalar loads, forward only
            for (int idx = tid; idx < COUT_WG * ROW; idx += kThreads)
                wl[(idx / ROW) * C::WPITCH + idx % ROW] = w[(size_t)co_base * ROW + idx];
        }
        __syncthreads();
        pro_finish();
#pragma unroll
        for (int g = 0; g < C::KG; ++g)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) wr[g][tap] = wl[(cot * 16 + j) * C::WPITCH + (g * 4 + k) * 9 + tap];
        __syncthreads();                                       // every wave has its weights: the region becomes the x tile
    }
    if constexpr (PRO != 0) {
        const float2 ss = tab[(tid / (C::WIN / 4)) % CIN];
        pro_scale = ss.x, pro_shift = ss.y;
    }
    // EPI: this lane's output channel is the same for the whole launch
    double s1 = 0.0, s2 = 0.0;
    float g_scale = 0.f, g_shift = 0.f, g_mean = 0.f;
    if constexpr (EPI == 3) {
        const int cj = co_base + cot * 16 + j;
        g_mean = f.bsave[cj], g_scale = f.bsave[2 * Cout + cj], g_shift = f.bsave[3 * Cout + cj];
    }
    const double g_meand = (double)g_mean;
    // what is stored for an accumulated float4 `v` whose second operand is `a`; the sums of the stored / gated values
    auto finish4 = [&](f32x4 v, const f32x4& a) -> f32x4 {
        if constexpr (EPI == 2 || EPI == 4) v = v + a;         // z = y + addend, one fp32 add as torch's
        if constexpr (EPI == 1 || EPI == 2) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double d = (double)v[e]; s1 += d; s2 = fma(d, d, s2); }
        }
        if constexpr (EPI == 3) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool open = fmaf(a[e], g_scale, g_shift) > 0.f;
                const float ge = open ? v[e] : 0.f;
                v[e] = ge;
                s1 += (double)ge;
                s2 = fma((double)ge, (double)a[e] - g_meand, s2);
            }
        }
        return v;
    };
    for (int i = tid; i < C::CINP * C::RI; i += kThreads) {    // halo columns (and, for a padded channel count, whole zero planes)
        float* row = xs + (i / C::RI) * C::XPLANE + (i % C::RI) * C::WP;
        row[3] = 0.f;
        row[4 + C::WIN] = 0.f;
    }
    if constexpr (C::CINP > CIN)
        for (int i = tid; i < (C::CINP - CIN) * C::XPLANE; i += kThreads) xs[CIN * C::XPLANE + i] = 0.f;

    for (int n = n0; n < n1; ++n) {
        if (n > n0) __syncthreads();
#pragma unroll
        for (int p = 0; p < PH; ++p) {
            stage(p);
            __syncthreads();
            if (p == PH - 1 && n + 1 < n1) issue(n + 1);
            // steps of 9 MFMAs (one channel group, nine taps); the A reads of step s + 1 are issued before the MFMAs of
            // step s, so the LDS latency hides behind 288 cycles of matrix work even with one wave per SIMD
            auto xbase = [&](int u) {
                const int lin = (p * C::TP + wsub + C::WPC * u) * 16 + j;          // A: position j of this wave's run u ...
                const int prow = lin / W, pcol = lin % W;                          // ... channel k of each group, tap (0, 0)
                if constexpr (MODE == 0) return xs + k * C::XPLANE + prow * C::WP + pcol + 3;
                else if constexpr (MODE == 1) return xs + k * C::XPLANE + 2 * prow * C::WP + 2 * pcol + 3;
                else return xs + k * C::XPLANE + prow * C::WP + pcol + 4;
            };
            auto reada = [&](float (&a)[C::NA], int s) {
                const float* xb = xbase(s / C::KG) + (s % C::KG) * 4 * C::XPLANE;
                if constexpr (MODE == 2) {                     // dy rows a, a + 1 x columns b, b + 1
                    a[0] = xb[0], a[1] = xb[1], a[2] = xb[C::WP], a[3] = xb[C::WP + 1];
                } else {
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) a[tap] = xb[(tap / 3) * C::WP + tap % 3];
                }
            };
            constexpr int STEPS = C::TPW * C::KG;
            float a[2][C::NA];
            constexpr int NACC = 4;                            // MODE 2: one per parity class; else four interleaved chains
            f32x4 acc[NACC];
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
            reada(a[0], 0);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                const int g = s % C::KG;
                if (s + 1 < STEPS) reada(a[(s + 1) & 1], s + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    if constexpr (DBG == 2) {                  // no matrix work: one VALU add per step keeps the reads alive
                        if (tap == 0) acc[0].x += a[s & 1][0];
                        continue;
                    }
                    if constexpr (MODE == 2) {
                        const int kh = tap / 3, kw = tap % 3;
                        const int cls = (kh != 1) * 2 + (kw != 1), src = (kh == 0) * 2 + (kw == 0);
                        acc[cls] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1][src], wr[g][8 - tap], acc[cls], 0, 0, 0);   // (wr holds w[..][8 - tap])
                    } else {
                        acc[(g * 9 + tap) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s & 1][tap], wr[g][tap], acc[(g * 9 + tap) & 3], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (g == C::KG - 1) {                          // D: positions 4*k .. 4*k + 3 of the run, output channel j
                    const int t = p * C::TP + wsub + C::WPC * (s / C::KG);
                    float* yo = y + ((size_t)n * Cout + co_base + cot * 16 + j) * (MODE == 2 ? 4 * W * W : W * W);
                    constexpr int AP = EPI >= 2 ? 1 : 0;        // (EPI: one run per wave and phase, so the phase indexes av)
                    if constexpr (MODE == 2) {
                        const int lin0 = t * 16 + 4 * k, prow = band * R + lin0 / W, pcol = lin0 % W;
                        float* o0 = yo + (size_t)(2 * prow) * (2 * W) + 2 * pcol;
                        *reinterpret_cast<f32x4*>(o0) = finish4(f32x4{acc[0].x, acc[1].x, acc[0].y, acc[1].y}, av[AP * 4 * p]);
                        *reinterpret_cast<f32x4*>(o0 + 4) = finish4(f32x4{acc[0].z, acc[1].z, acc[0].w, acc[1].w}, av[AP * (4 * p + 1)]);
                        *reinterpret_cast<f32x4*>(o0 + 2 * W) = finish4(f32x4{acc[2].x, acc[3].x, acc[2].y, acc[3].y}, av[AP * (4 * p + 2)]);
                        *reinterpret_cast<f32x4*>(o0 + 2 * W + 4) = finish4(f32x4{acc[2].z, acc[3].z, acc[2].w, acc[3].w}, av[AP * (4 * p + 3)]);
                    } else {
                        const f32x4 out = finish4((acc[0] + acc[1]) + (acc[2] + acc[3]), av[AP * p]);
                        if (DBG != 1 || out.x == 12345.678f)   // DBG 1: no stores
                            *reinterpret_cast<f32x4*>(yo + band * R * W + t * 16 + 4 * k) = out;
                    }
#pragma unroll
                    for (int c = 0; c < NACC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if (EPI >= 2 && n + 1 < n1) issue_aux(n + 1);          // after this image's stores: av is free again
    }
    if constexpr ((EPI == 1 || EPI == 2) && DBG == 3) {                      // knobs: no hand-over at all (what the accumulation alone costs)
        if (s1 + s2 == 12345.678) y[0] = 0.f;
    } else if constexpr ((EPI == 1 || EPI == 2) && DBG == 4) {               // knobs: slots stored, nobody adds (what the final poll costs)
        publish_sums<COUT_WG, C::MT, C::WPC>(f, 0xffffffffu, s1, s2, smem, co_base, bx, by, gx);
    } else if constexpr (EPI == 3) {
        store_sums<COUT_WG, C::MT, C::WPC>(f, s1, s2, smem, co_base, bx, gx);
    } else if constexpr (EPI == 1 || EPI == 2) {
        publish_sums<COUT_WG, C::MT, C::WPC>(f, ticket, s1, s2, smem, co_base, bx, by, gx);
    }
}

template <int CIN, int COUT_WG, int W, int R, int PH, int MODE, int DBG = 0, int PRO = 0, int EPI = 0>
__global__ __launch_bounds__(kThreads, minw_for(CIN, MODE, EPI)) void k_conv3x3(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ y, int N, int Cout, int ipw, int flip_arg, const Fuse f) {
    __shared__ __attribute__((aligned(16))) float smem[Fw<CIN, COUT_WG, W, R, PH, MODE>::SMEM];
    conv_body<CIN, COUT_WG, W, R, PH, MODE, DBG, PRO, EPI>(x, w, y, N, Cout, ipw, flip_arg, f, smem, blockIdx.x, blockIdx.y, gridDim.x);
}

// The two convolutions of a unit's backward pass - the input gradient (conv_body, EPI 3) and the weight gradient (wgrad_body,
// XBN) - read the same output gradient and depend on nothing of each other: ONE launch whose workgroups alternate between the
// two roles (even linear index: input gradient, odd: weight gradient; what one role has more of comes last). Both alone are
// bound by latency and launch cost at two workgroups per CU; side by side on the same CUs they fill each other's gaps:
// one launch's fixed cost instead of two, and no cross-queue edge as a parallel graph branch would need.
template <int CIN_A, int COUT_WG_A, int W_A, int R_A, int PH_A, int MODE_A, int CIN_B, int COUT_WG_B, int WO_B, int R_B, int STRIDE_B, int PH_B, int MINW = URSA_MINW>
__global__ __launch_bounds__(kThreads, MINW) void k_bwd_pair(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ g,
                                                        const float* __restrict__ x, float* __restrict__ partial, int N, int Cd, int Cx,
                                                        int ipw_a, int ipw_b, int gx_a, int gy_a, int gx_b, int gy_b, const Fuse f) {
    using A = Fw<CIN_A, COUT_WG_A, W_A, R_A, PH_A, MODE_A>;
    using B = Wg<CIN_B, COUT_WG_B, WO_B, R_B, STRIDE_B, 9, PH_B>;
    __shared__ __attribute__((aligned(16))) float smem[cmax(A::SMEM, B::SMEM)];
    const int na = gx_a * gy_a, nb = gx_b * gy_b, m = na < nb ? na : nb;
    const int id = blockIdx.x;
    int role, idx;
    if (id < 2 * m) role = id & 1, idx = id >> 1;
    else role = na > nb ? 0 : 1, idx = id - m;
    if (role == 0)      // dx' = gate(conv_flip(dy, w)): staged tensor dy [N, Cd, ..], result g [N, Cx, ..]
        conv_body<CIN_A, COUT_WG_A, W_A, R_A, PH_A, MODE_A, 0, 0, 3>(dy, w, g, N, Cx, ipw_a, 1, f, smem, idx % gx_a, idx / gx_a, gx_a);
    else                // dW partial sums: x operand = relu(bn(x)) rebuilt from f.bsave while staged
        wgrad_body<CIN_B, COUT_WG_B, WO_B, R_B, STRIDE_B, 9, PH_B, true>(x, dy, partial, N, Cd, ipw_b, f.bsave, smem, idx % gx_b, idx / gx_b);
}

struct FwPlan {
    int gx_per_image, gy, ipw;
    void (*fn)(const float*, const float*, float*, int, int, int, int, Fuse);
};

// Large batches (an ensemble member's 4,096-row evaluation forward, HMC's 1,024-row chunks): several images per workgroup - the
// weights are staged once and the next image's rows are loaded under this one's matrix work (the kernel's own prefetch) - as
// long as >= 2,048 workgroups remain (8 per CU). Measured at 4,096 rows (tools/k10_eval_bench.py, profiles/r06_k10_eval_bench.json):
// 16 ch 198 -> 173 us (0.62 -> 0.71 of the fp32 matrix peak), 32 ch 188 -> 157 (0.78), 64 ch 259 -> 191, stride 2 118 -> 106 /
// 128 -> 100. Which image a workgroup takes changes nothing in the arithmetic: the same bits. At the training batch (128): no change.
inline void widen_for_large_batches(FwPlan& p, int64_t N) {
    if (!p.fn) return;
    const int64_t wgs1 = N * p.gx_per_image * p.gy;            // workgroups at one image each
    int ipw = 1;
    while (ipw < 16 && wgs1 / (2 * ipw) >= 2048) ipw *= 2;
    if (ipw > p.ipw) p.ipw = ipw;
}

// the 3x3 layers of the CIFAR pre-activation ResNets: stride 1 - the stem and the three stages' equal-width layers (forward and,
// flipped, their input gradient); stride 2 - the two layers that open stages 2 and 3, forward (Cin, Cout, H of x) and input
// gradient (flipped: Cin = dy's channels, Cout = dx's channels, H = dy's size)
FwPlan fw_plan_for(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, uint32_t flags) {
    FwPlan p = {0, 0, 0, nullptr};
    if (N < 1 || N > (1 << 20) || H != W) return p;
    const bool flip = flags & URSA_CONV_FLIP;
    if (flags & URSA_CONV_STRIDE2) {
        if (!flip && Cin == 16 && Cout == 32 && W == 32) p = {2, 1, 1, k_conv3x3<16, 32, 16, 8, 4, 1>};
        else if (!flip && Cin == 32 && Cout == 64 && W == 16) p = {1, 2, 1, k_conv3x3<32, 32, 8, 8, 2, 1>};
        else if (flip && Cin == 32 && Cout == 16 && W == 16) p = {2, 1, 1, k_conv3x3<32, 16, 16, 8, 2, 2>};
        else if (flip && Cin == 64 && Cout == 32 && W == 8) p = {1, 2, 1, k_conv3x3<64, 16, 8, 8, 1, 2>};
        widen_for_large_batches(p, N);
        return p;
    }
    if (flip && Cin != Cout) return p;                         // flipped stride 1: the equal-width layers only
    if (Cin == 16 && Cout == 16 && W == 32) p = {4, 1, 1, k_conv3x3<16, 16, 32, 8, 4, 0>};
    else if (Cin == 3 && Cout == 16 && W == 32) p = {4, 1, 1, k_conv3x3<3, 16, 32, 8, 4, 0>};
    else if (Cin == 32 && Cout == 32 && W == 16) p = {2, 1, 1, k_conv3x3<32, 32, 16, 8, 4, 0>};
    else if (Cin == 64 && Cout == 64 && W == 8) p = {1, 4, 2, k_conv3x3<64, 16, 8, 8, 1, 0>};
    widen_for_large_batches(p, N);
#ifdef URSA_DEBUG_KNOBS
    if (const char* e = getenv("URSA_CONV_FWD_DBG")) {        // what bounds the launch: 1 = no stores, 2 = no matrix work (wrong results)
        if (atoi(e) == 1 && Cin == 16 && Cout == 16 && W == 32) p.fn = k_conv3x3<16, 16, 32, 8, 4, 0, 1>;
        if (atoi(e) == 2 && Cin == 16 && Cout == 16 && W == 32) p.fn = k_conv3x3<16, 16, 32, 8, 4, 0, 2>;
        if (atoi(e) == 1 && Cin == 32 && Cout == 32 && W == 16) p.fn = k_conv3x3<32, 32, 16, 8, 4, 0, 1>;
        if (atoi(e) == 2 && Cin == 32 && Cout == 32 && W == 16) p.fn = k_conv3x3<32, 32, 16, 8, 4, 0, 2>;
        if (atoi(e) == 3 && Cin == 16 && Cout == 16 && W == 32) p = {8, 1, 1, k_conv3x3<16, 16, 32, 4, 2, 0>};    // 1,024 workgroups of 4 rows
        if (atoi(e) == 4 && Cin == 16 && Cout == 16 && W == 32) p = {16, 1, 1, k_conv3x3<16, 16, 32, 2, 1, 0>};   // 2,048 workgroups of 2 rows
    }
#endif
    return p;
}

}  // namespace

extern "C" int ursa_conv3x3_supported(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, uint32_t flags) {
    return !(flags & ~(URSA_CONV_FLIP | URSA_CONV_STRIDE2)) && fw_plan_for(N, Cin, Cout, H, W, flags).fn != nullptr;
}

extern "C" int ursa_conv3x3_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H,
                                int64_t W, uint32_t flags, ursa_stream_t stream) {
    if (flags & ~(URSA_CONV_FLIP | URSA_CONV_STRIDE2)) return URSA_EFLAGS;
    if (!x || !w || !y) return URSA_ENULL;
    if (N < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return URSA_ESIZE;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w) & 15) return URSA_EALIGN;      // w is read through float4 loads too
    const FwPlan p = fw_plan_for(N, Cin, Cout, H, W, flags);
    if (!p.fn) return URSA_EVALUE;
    const int groups = (int)((N + p.ipw - 1) / p.ipw);
    hipLaunchKernelGGL(p.fn, dim3(groups * p.gx_per_image, p.gy), dim3(kThreads), 0, (hipStream_t)stream, x, w, y, (int)N, (int)Cout,
                       p.ipw, (int)(flags & URSA_CONV_FLIP), Fuse{});
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? URSA_OK : (int)e;
}

// =====================================================================================================================
// K10: the pre-activation unit `conv(relu(bn(x)))` (+ `out += residual`) of the BasicBlock ResNets as ONE launch each way
// (URSABench/models/preresnet.py:33-52: bn1 -> relu -> conv1 -> bn2 -> relu -> conv2 -> += residual). K8's launch with the
// BatchNorm that precedes the convolution applied while its tile is staged (PRO) and the per-channel sums the next BatchNorm -
// or this one's backward - needs taken from the accumulators (EPI): the normalised activation is never stored, the separate
// statistics / normalise / reduce launches of K6 (64 of a PreResNet-20 step's 146 launches) disappear.
namespace {

FwPlan fuse_plan_for(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, uint32_t flags) {
    FwPlan p = {0, 0, 0, nullptr};
    if (N < 1 || N > (1 << 20) || H != W) return p;
    const bool flip = flags & URSA_CONV_FLIP, s2 = flags & URSA_CONV_STRIDE2, bn = flags & URSA_PREACT_BN;
    const bool stats = flags & URSA_PREACT_STATS, add = flags & URSA_PREACT_ADD, bwd = flags & URSA_PREACT_BNBWD;
    if (flip) {                                                // input gradient + the sums of the BatchNorm backward in front of the layer
        if (!bwd || bn || stats || add) return p;
        if (s2) {
            if (Cin == 32 && Cout == 16 && W == 16) p = {2, 1, 1, k_conv3x3<32, 16, 16, 8, 2, 2, 0, 0, 3>};
            else if (Cin == 64 && Cout == 32 && W == 8) p = {1, 2, 1, k_conv3x3<64, 16, 8, 8, 1, 2, 0, 0, 3>};
        } else if (Cin == Cout) {
            if (Cin == 16 && W == 32) p = {4, 1, 1, k_conv3x3<16, 16, 32, 8, 4, 0, 0, 0, 3>};
            else if (Cin == 32 && W == 16) p = {2, 1, 1, k_conv3x3<32, 32, 16, 8, 4, 0, 0, 0, 3>};
            else if (Cin == 64 && W == 8) p = {1, 4, 2, k_conv3x3<64, 16, 8, 8, 1, 0, 0, 0, 3>};
        }
#ifdef URSA_DEBUG_KNOBS
        if (const char* e = getenv("URSA_BWD_IPW")) {            // images per workgroup of the input-gradient role
            const int v = atoi(e);
            if (v >= 1 && v <= 8 && p.fn) p.ipw = v;
        }
#endif
        return p;
    }
    if (flags & URSA_PREACT_EVAL) {                            // evaluation: running statistics in the prologue, optional residual add, no sums
        if (bwd || stats || !bn) return p;
        if (s2) {
            if (add) return p;
            if (Cin == 16 && Cout == 32 && W == 32) p = {2, 1, 1, k_conv3x3<16, 32, 16, 8, 4, 1, 0, 2, 0>};
            else if (Cin == 32 && Cout == 64 && W == 16) p = {1, 2, 1, k_conv3x3<32, 32, 8, 8, 2, 1, 0, 2, 0>};
            widen_for_large_batches(p, N);
#ifdef URSA_DEBUG_KNOBS
            if (const char* e = getenv("URSA_K8_EVAL_IPW")) {
                const int v = atoi(e);
                if (v >= 1 && v <= 64 && p.fn) p.ipw = v;
            }
#endif
            return p;
        }
        if (Cin == 16 && Cout == 16 && W == 32) {
            if (add) p = {4, 1, 1, k_conv3x3<16, 16, 32, 8, 4, 0, 0, 2, 4>};
            else p = {4, 1, 1, k_conv3x3<16, 16, 32, 8, 4, 0, 0, 2, 0>};
        } else if (Cin == 32 && Cout == 32 && W == 16) {
            if (add) p = {2, 1, 1, k_conv3x3<32, 32, 16, 8, 4, 0, 0, 2, 4>};
            else p = {2, 1, 1, k_conv3x3<32, 32, 16, 8, 4, 0, 0, 2, 0>};
        } else if (Cin == 64 && Cout == 64 && W == 8) {
            if (add) p = {1, 4, 2, k_conv3x3<64, 16, 8, 8, 1, 0, 0, 2, 4>};
            else p = {1, 4, 2, k_conv3x3<64, 16, 8, 8, 1, 0, 0, 2, 0>};
        }
        widen_for_large_batches(p, N);
#ifdef URSA_DEBUG_KNOBS
        if (const char* e = getenv("URSA_K8_EVAL_IPW")) {         // images per workgroup of the evaluation forms (large batches)
            const int v = atoi(e);
            if (v >= 1 && v <= 64 && p.fn) p.ipw = v;
        }
#endif
        return p;
    }
    if (bwd || !stats) return p;                               // forward forms always leave the next BatchNorm's sums
    if (s2) {
        if (!bn || add) return p;
        if (Cin == 16 && Cout == 32 && W == 32) p = {2, 1, 1, k_conv3x3<16, 32, 16, 8, 4, 1, 0, 1, 1>};
        else if (Cin == 32 && Cout == 64 && W == 16) p = {1, 2, 1, k_conv3x3<32, 32, 8, 8, 2, 1, 0, 1, 1>};
        return p;
    }
    if (Cin == 3 && Cout == 16 && W == 32) {                   // the stem: no BatchNorm in front, no residual
        if (!bn && !add) p = {4, 1, 1, k_conv3x3<3, 16, 32, 8, 4, 0, 0, 0, 1>};
        return p;
    }
    if (!bn) return p;
    if (Cin == 16 && Cout == 16 && W == 32) {
        if (add) p = {4, 1, 1, k_conv3x3<16, 16, 32, 8, 4, 0, 0, 1, 2>};
        else p = {4, 1, 1, k_conv3x3<16, 16, 32, 8, 4, 0, 0, 1, 1>};
#ifdef URSA_DEBUG_KNOBS
        if (const char* e = getenv("URSA_K10_DBG")) {            // what the fused launch pays for: 1 = prologue only (no sums), 2 = sums only (x taken as is)
            if (atoi(e) == 1) p.fn = k_conv3x3<16, 16, 32, 8, 4, 0, 0, 1, 0>;
            if (atoi(e) == 2) p.fn = k_conv3x3<16, 16, 32, 8, 4, 0, 0, 0, 1>;
            if (atoi(e) == 3) p.fn = k_conv3x3<16, 16, 32, 8, 4, 0, 3, 0, 1>;   // sums accumulated, not handed over
            if (atoi(e) == 4) p.fn = k_conv3x3<16, 16, 32, 8, 4, 0, 4, 0, 1>;   // slots stored, no final add (leaves the scratch dirty)
        }
#endif
    } else if (Cin == 32 && Cout == 32 && W == 16) {
        if (add) p = {2, 1, 1, k_conv3x3<32, 32, 16, 8, 4, 0, 0, 1, 2>};
        else p = {2, 1, 1, k_conv3x3<32, 32, 16, 8, 4, 0, 0, 1, 1>};
    } else if (Cin == 64 && Cout == 64 && W == 8) {
        if (add) p = {1, 4, 2, k_conv3x3<64, 16, 8, 8, 1, 0, 0, 1, 2>};
        else p = {1, 4, 2, k_conv3x3<64, 16, 8, 8, 1, 0, 0, 1, 1>};
    }
    return p;
}

struct FuseGeom {
    int S, line_sz, nl;
    int64_t tickets_bytes, scratch_bytes;
};

FuseGeom fuse_geom(const FwPlan& p, int64_t N, int64_t Cout, bool flip) {
    FuseGeom g;
    g.S = (int)((N + p.ipw - 1) / p.ipw) * p.gx_per_image;
    if (flip) {                                                // one partial sum per workgroup, added by the dx launch: no scratch
        g.line_sz = 1, g.nl = g.S, g.tickets_bytes = 0, g.scratch_bytes = 0;
        return g;
    }
    g.line_sz = (g.S + kLines - 1) / kLines;
    g.nl = (g.S + g.line_sz - 1) / g.line_sz;
    g.tickets_bytes = (int64_t)p.gy * kLines * 128;
    g.scratch_bytes = g.tickets_bytes + 128 + Cout * (int64_t)g.S * 16;
    return g;
}

}  // namespace

extern "C" int ursa_preact_geometry(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, uint32_t flags, int64_t* out) {
    if (!out) return URSA_ENULL;
    if (flags & ~URSA_PREACT_ALLFLAGS) return URSA_EFLAGS;
    const FwPlan p = fuse_plan_for(N, Cin, Cout, H, W, flags);
    if (!p.fn) return URSA_EVALUE;
    if (flags & URSA_PREACT_EVAL) {                            // covered; nothing to size
        out[0] = 0, out[1] = 0, out[2] = (int64_t)((N + p.ipw - 1) / p.ipw) * p.gx_per_image, out[3] = 0;
        return URSA_OK;
    }
    const FuseGeom g = fuse_geom(p, N, Cout, flags & URSA_CONV_FLIP);
    out[0] = g.nl, out[1] = g.scratch_bytes, out[2] = g.S, out[3] = g.tickets_bytes;
    return URSA_OK;
}

extern "C" int ursa_preact_conv3x3_f32(const float* x, const float* w, float* y, const double* in_partial, int32_t in_nl,
                                       const float* gamma, const float* beta, float* running_mean, float* running_var,
                                       float* bn_save, float eps, float momentum, const float* aux, const float* aux_bn_save,
                                       double* out_partial, void* scratch, int64_t scratch_bytes, int64_t N, int64_t Cin,
                                       int64_t Cout, int64_t H, int64_t W, uint32_t flags, ursa_stream_t stream) {
    if (flags & ~URSA_PREACT_ALLFLAGS) return URSA_EFLAGS;
    if (flags & URSA_PREACT_EVAL) {                            // evaluation form: no sums, no scratch, nothing saved
        if (!x || !w || !y || !gamma || !beta || !running_mean || !running_var) return URSA_ENULL;
        if ((flags & URSA_PREACT_ADD) && !aux) return URSA_ENULL;
        if (N < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return URSA_ESIZE;
        if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)aux) & 15) return URSA_EALIGN;
        if (((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)running_mean | (uintptr_t)running_var) & 3) return URSA_EALIGN;
        const FwPlan pe = fuse_plan_for(N, Cin, Cout, H, W, flags);
        if (!pe.fn) return URSA_EVALUE;
        Fuse fe = {};
        fe.gamma = gamma, fe.beta = beta, fe.running_mean = running_mean, fe.running_var = running_var, fe.eps = eps, fe.aux = aux;
        const int groups = (int)((N + pe.ipw - 1) / pe.ipw);
        hipLaunchKernelGGL(pe.fn, dim3(groups * pe.gx_per_image, pe.gy), dim3(kThreads), 0, (hipStream_t)stream, x, w, y, (int)N, (int)Cout,
                           pe.ipw, 0, fe);
        const hipError_t ee = hipGetLastError();
        return ee == hipSuccess ? URSA_OK : (int)ee;
    }
    if (!x || !w || !y || !out_partial) return URSA_ENULL;
    if (N < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return URSA_ESIZE;
    const bool bn = flags & URSA_PREACT_BN, add = flags & URSA_PREACT_ADD, bwd = flags & URSA_PREACT_BNBWD;
    if (bn && (!in_partial || !gamma || !beta || !bn_save || in_nl < 1 || in_nl > kLines)) return in_partial && gamma && beta && bn_save ? URSA_ESIZE : URSA_ENULL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return URSA_ENULL;
    if ((add || bwd) && !aux) return URSA_ENULL;
    if (bwd && !aux_bn_save) return URSA_ENULL;
    if (((uintptr_t)x | (uintptr_t)y | (uintptr_t)w | (uintptr_t)aux | (uintptr_t)in_partial | (uintptr_t)out_partial) & 15) return URSA_EALIGN;
    if ((uintptr_t)scratch & 127) return URSA_EALIGN;
    if (((uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)running_mean | (uintptr_t)running_var | (uintptr_t)bn_save | (uintptr_t)aux_bn_save) & 3) return URSA_EALIGN;
    const FwPlan p = fuse_plan_for(N, Cin, Cout, H, W, flags);
    if (!p.fn) return URSA_EVALUE;
    const FuseGeom g = fuse_geom(p, N, Cout, flags & URSA_CONV_FLIP);
    if (g.scratch_bytes && !scratch) return URSA_ENULL;
    if (scratch_bytes < g.scratch_bytes) return URSA_ESIZE;
    if (bn && N * H * W < 2) return URSA_EVALUE;               // torch: "Expected more than 1 value per channel"
    Fuse f = {};
    f.in_partial = reinterpret_cast<const double2*>(in_partial), f.gamma = gamma, f.beta = beta;
    f.running_mean = running_mean, f.running_var = running_var, f.save = bn_save;
    f.in_count = (double)(N * H * W), f.eps = eps, f.momentum = momentum, f.in_nl = in_nl;
    f.nl = g.nl, f.line_sz = g.line_sz, f.aux = aux, f.bsave = aux_bn_save;
    f.tickets = reinterpret_cast<unsigned int*>(scratch);
    f.err = reinterpret_cast<unsigned int*>((char*)scratch + g.tickets_bytes);
    f.slots = reinterpret_cast<u64*>((char*)scratch + g.tickets_bytes + 128);
    f.out_partial = reinterpret_cast<double2*>(out_partial);
    hipLaunchKernelGGL(p.fn, dim3(g.S, p.gy), dim3(kThreads), 0, (hipStream_t)stream, x, w, y, (int)N, (int)Cout, p.ipw,
                       (int)(flags & URSA_CONV_FLIP), f);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? URSA_OK : (int)e;
}

// The paired backward launch: ursa_preact_conv3x3_f32(FLIP | BNBWD) and ursa_preact_wgrad_partial_f32 of one unit in one launch
// (k_bwd_pair). Same workgroup programs, same operands, same results bit for bit; grid = the two launches' workgroups interleaved.
typedef void (*PairFn)(const float*, const float*, float*, const float*, float*, int, int, int, int, int, int, int, int, int, Fuse);

static PairFn pair_fn_for(int64_t Cd, int64_t Cx, int64_t W, bool s2) {
    // (Cd, W) = channels / size of dy; Cx = channels of the layer's input
    if (!s2 && Cd == 16 && Cx == 16 && W == 32) return k_bwd_pair<16, 16, 32, 8, 4, 0, 16, 16, 32, 8, 1, 4>;
    if (!s2 && Cd == 32 && Cx == 32 && W == 16) return k_bwd_pair<32, 32, 16, 8, 4, 0, 32, 32, 16, 8, 1, 4>;
    if (!s2 && Cd == 64 && Cx == 64 && W == 8) return k_bwd_pair<64, 16, 8, 8, 1, 0, 64, 16, 8, 8, 1, 4>;
    if (s2 && Cd == 32 && Cx == 16 && W == 16) return k_bwd_pair<32, 16, 16, 8, 2, 2, 16, 32, 16, 8, 2, 4>;
    if (s2 && Cd == 64 && Cx == 32 && W == 8) return k_bwd_pair<64, 16, 8, 8, 1, 2, 32, 32, 8, 4, 2, 2>;
    return nullptr;
}

extern "C" int ursa_preact_bwd_pair_f32(const float* dy, const float* w, float* g, const float* x, const float* bn_save,
                                        double* out_partial, float* ws, int64_t ws_floats, int64_t N, int64_t Cd, int64_t Cx,
                                        int64_t H, int64_t W, uint32_t flags, ursa_stream_t stream) {
    if (flags & ~URSA_CONV_STRIDE2) return URSA_EFLAGS;
    if (!dy || !w || !g || !x || !bn_save || !out_partial || !ws) return URSA_ENULL;
    if (N < 1 || Cd < 1 || Cx < 1 || H < 1 || W < 1) return URSA_ESIZE;
    if (((uintptr_t)dy | (uintptr_t)w | (uintptr_t)g | (uintptr_t)x | (uintptr_t)out_partial | (uintptr_t)ws) & 15 || (uintptr_t)bn_save & 3) return URSA_EALIGN;
    const bool s2 = flags & URSA_CONV_STRIDE2;
    const int st = s2 ? 2 : 1;
    const FwPlan a = fuse_plan_for(N, Cd, Cx, H, W, URSA_CONV_FLIP | URSA_PREACT_BNBWD | (flags & URSA_CONV_STRIDE2));
    const Plan b = plan_for(N, Cx, Cd, H * st, W * st, 3, st, true);
    const PairFn fn = pair_fn_for(Cd, Cx, W, s2);
    if (!a.fn || !b.slices || !fn || H != W) return URSA_EVALUE;
    if (ws_floats < (int64_t)b.slices * b.E) return URSA_ESIZE;
    const FuseGeom ga = fuse_geom(a, N, Cx, true);
    Fuse f = {};
    f.nl = ga.nl, f.line_sz = ga.line_sz, f.aux = x, f.bsave = bn_save;
    f.out_partial = reinterpret_cast<double2*>(out_partial);
    const int na = ga.S * a.gy, nb = b.slices * b.gy;
    hipLaunchKernelGGL(fn, dim3(na + nb), dim3(kThreads), 0, (hipStream_t)stream, dy, w, g, x, ws, (int)N, (int)Cd, (int)Cx, a.ipw, b.ipw,
                       ga.S, a.gy, b.slices, b.gy, f);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? URSA_OK : (int)e;
}

// =====================================================================================================================
// K9: the 1x1 / stride 2 shortcut convolutions (URSABench/models/preresnet.py:130-136 `downsample`), forward and input
// gradient, NCHW fp32. 33 MFLOP per call at the workload's sizes: memory- and launch-bound, plain FMAs (MIOpen runs these as an
// NHWC implicit GEMM between layout transposes and a zero fill: 12 launches, 70 us per step for the two layers both ways).
//
//     forward:  y[n][o][oh][ow]  = sum_i w[o][i] * x[n][i][2 oh][2 ow]                   (fma chain over i, ascending)
//     flipped:  dx[n][i][ih][iw] = ih, iw even ? sum_o w[o][i] * dy[n][o][ih/2][iw/2] : 0   (fma chain over o, ascending)
//
// One thread: two adjacent output columns (one float4 of an even input row) x CG channels of the result; the weights are
// uniform over a workgroup's threads (scalar loads).
namespace {

template <int CIN, int COUT, int CG, bool FLIP>
__global__ __launch_bounds__(256) void k_conv1x1s2(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                    int N, int OW) {
    // forward: x [N, CIN, 2 OW, 2 OW] -> y [N, COUT, OW, OW]; flipped: x = dy [N, CIN(= the layer's outputs), OW, OW] ->
    // y = dx [N, COUT(= the layer's inputs), 2 OW, 2 OW], w = the layer's [CIN, COUT] tensor
    const int half = OW / 2;                                  // column pairs per output row
    const int per_image = OW * half;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int cg = blockIdx.y * CG;                           // first result channel of this workgroup
    if (idx >= N * per_image) return;
    const int n = idx / per_image, r = idx % per_image, oh = r / half, j = r % half;
    float a0[CG], a1[CG];
#pragma unroll
    for (int c = 0; c < CG; ++c) a0[c] = 0.f, a1[c] = 0.f;
    if constexpr (!FLIP) {
        const int WI = 2 * OW;
        const float* xp = x + (((size_t)n * CIN) * WI + 2 * oh) * WI + 4 * j;
#pragma unroll 4
        for (int i = 0; i < CIN; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xp + (size_t)i * WI * WI);
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                const float wv = w[(cg + c) * CIN + i];
                a0[c] = fmaf(wv, v.x, a0[c]);
                a1[c] = fmaf(wv, v.z, a1[c]);
            }
        }
        float* yp = y + (((size_t)n * COUT + cg) * OW + oh) * OW + 2 * j;
#pragma unroll
        for (int c = 0; c < CG; ++c) *reinterpret_cast<float2*>(yp + (size_t)c * OW * OW) = make_float2(a0[c], a1[c]);
    } else {
        const float* xp = x + (((size_t)n * CIN) * OW + oh) * OW + 2 * j;
#pragma unroll 4
        for (int o = 0; o < CIN; ++o) {
            const float2 v = *reinterpret_cast<const float2*>(xp + (size_t)o * OW * OW);
#pragma unroll
            for (int c = 0; c < CG; ++c) {
                const float wv = w[o * COUT + cg + c];
                a0[c] = fmaf(wv, v.x, a0[c]);
                a1[c] = fmaf(wv, v.y, a1[c]);
            }
        }
        const int WI = 2 * OW;
        float* yp = y + (((size_t)n * COUT + cg) * WI + 2 * oh) * WI + 4 * j;
#pragma unroll
        for (int c = 0; c < CG; ++c) {
            *reinterpret_cast<f32x4*>(yp + (size_t)c * WI * WI) = f32x4{a0[c], 0.f, a1[c], 0.f};
            *reinterpret_cast<f32x4*>(yp + (size_t)c * WI * WI + WI) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
}

struct S2Plan {
    int gy;
    void (*fn)(const float*, const float*, float*, int, int);
};

// (Cin, Cout, H) of the layer; flipped launches take (Cout, Cin) as their (input, result) channel counts
S2Plan s2_plan_for(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, bool flip) {
    S2Plan p = {0, nullptr};
    if (N < 1 || N > (1 << 20) || H != W) return p;
    if (!flip && Cin == 16 && Cout == 32 && W == 32) p = {2, k_conv1x1s2<16, 32, 16, false>};
    else if (!flip && Cin == 32 && Cout == 64 && W == 16) p = {8, k_conv1x1s2<32, 64, 8, false>};
    else if (flip && Cin == 32 && Cout == 16 && W == 16) p = {2, k_conv1x1s2<32, 16, 8, true>};    // dy [N, 32, 16, 16] -> dx [N, 16, 32, 32]
    else if (flip && Cin == 64 && Cout == 32 && W == 8) p = {4, k_conv1x1s2<64, 32, 8, true>};      // dy [N, 64, 8, 8] -> dx [N, 32, 16, 16]
    return p;
}

}  // namespace

extern "C" int ursa_conv1x1s2_supported(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, uint32_t flags) {
    return !(flags & ~URSA_CONV_FLIP) && s2_plan_for(N, Cin, Cout, H, W, flags & URSA_CONV_FLIP).fn != nullptr;
}

extern "C" int ursa_conv1x1s2_f32(const float* x, const float* w, float* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H,
                                  int64_t W, uint32_t flags, ursa_stream_t stream) {
    if (flags & ~URSA_CONV_FLIP) return URSA_EFLAGS;
    if (!x || !w || !y) return URSA_ENULL;
    if (N < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1) return URSA_ESIZE;
    if (((uintptr_t)x | (uintptr_t)y) & 15 || (uintptr_t)w & 3) return URSA_EALIGN;
    const bool flip = flags & URSA_CONV_FLIP;
    const S2Plan p = s2_plan_for(N, Cin, Cout, H, W, flip);
    if (!p.fn) return URSA_EVALUE;
    const int OW = flip ? (int)W : (int)W / 2;                 // the low-resolution side's width
    const int64_t threads = N * OW * (OW / 2);
    hipLaunchKernelGGL(p.fn, dim3((unsigned)((threads + 255) / 256), p.gy), dim3(256), 0, (hipStream_t)stream, x, w, y, (int)N, OW);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? URSA_OK : (int)e;
}
