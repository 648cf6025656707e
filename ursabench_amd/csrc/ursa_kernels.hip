// ursa_kernels.hip — gfx950 (MI355X / CDNA4) kernels behind include/ursa_hip.h.
//
// Every kernel here is HBM-bound elementwise or row-reduction work: the
// design rules are 16-byte-per-lane coalesced access, Philox noise generated in registers
// (zero bytes), wave64 shuffles for row reductions. MFMA is deliberately unused: nothing
// here is GEMM-shaped.
//
// Launch shape of the streaming kernels (A/B in one process, tools/exp/k1_variants.hip, round 1): ONE float4 per
// thread, 512-thread blocks, grid = ceil(n4/512), no grid-stride loop — a third faster at 2^26 elements than the
// textbook "2048 blocks + grid-stride" shape: many short blocks keep every HBM channel busy and let the dispatcher
// balance the 8 XCDs. What the shipped kernels reach is in profiles/r02_kbench.json (K1 at 2^26 elements: 210-231 us
// = 5.8-6.4 TB/s across boxes). Non-temporal loads/stores are used only when the state exceeds the 256 MiB Infinity
// Cache (they cost 3-15 % on cache-resident state; nt_bytes()).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt
// (the rounding sequence of each update is part of the contract, see the header).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/ursa_hip.h"
#include "ursa_rng.h"

namespace {

constexpr int kBlock = 256;          // reductions / row kernels: 4 waves, one per SIMD
constexpr int kMaxGrid = 256 * 8;    // grid-stride kernels (reductions): 256 CUs x 8 blocks/CU
constexpr int kSBlock = 512;         // streaming kernels: one float4 per thread
constexpr int64_t kMaxElems = 1ll << 40;    // keeps ceil(n/2048) blocks inside a 31-bit grid
// State larger than the 256 MiB Infinity Cache streams past it with non-temporal loads/stores. Measured
// (tools/exp/nt_threshold.py, profiles/r02_nt_threshold.json): at the WideResNet-28-10 arena (36.5 M elements,
// 438-731 MB per launch) NT is +9..15 % (K1 5.49 -> 6.13, K2 5.46 -> 6.27, K3 5.06 -> 5.51 TB/s); at 2^24
// elements (201-335 MB, mostly cache-resident) it costs 12 %; round 1's 512 MiB threshold missed the first case.
constexpr int64_t kNtBytesDefault = 256ll << 20;

// Kernel-selection knobs for experiments and A/B tests (tools/k1_ctl_bench.py, tools/k5_bench.py, tools/exp/*): compiled
// into libursa_hip_knobs.so only (-DURSA_DEBUG_KNOBS, `make libursa_hip_knobs.so`). The shipped library reads no
// environment: every knob() below is a constant nullptr there and the branches fold away.
#ifdef URSA_DEBUG_KNOBS
inline const char* knob(const char* name) { return getenv(name); }
#else
inline const char* knob(const char*) { return nullptr; }
#endif

// Knob: the non-temporal threshold in MiB (tools/exp/nt_threshold.py); read once.
inline int64_t nt_bytes()
{
    static const int64_t v = [] {
        const char* e = knob("URSA_NT_MIB");
        return e && e[0] ? (int64_t)atoll(e) << 20 : kNtBytesDefault;
    }();
    return v;
}
#define kNtBytes nt_bytes()

inline int sgrid(int64_t items)      // one item per thread
{
    int64_t g = (items + kSBlock - 1) / kSBlock;
    return (int)(g < 1 ? 1 : g);
}

template <bool NT>
__device__ __forceinline__ float4 ld4(const float4* p)
{
    if (NT) {
        float4 r;
        r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y);
        r.z = __builtin_nontemporal_load(&p->z); r.w = __builtin_nontemporal_load(&p->w);
        return r;
    }
    return *p;
}
template <bool NT>
__device__ __forceinline__ void st4(float4* p, const float4& v)
{
    if (NT) {
        __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
        __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
    } else {
        *p = v;
    }
}

inline int grid_for(int64_t work_items, int per_block)
{
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > kMaxGrid) g = kMaxGrid;
    return (int)g;
}

inline int bma_grid(int64_t rows, int rows_per_block)      // one block per row tile, no cap below 2^20 blocks
{
    int64_t g = (rows + rows_per_block - 1) / rows_per_block;
    return (int)(g < 1 ? 1 : g > (1 << 20) ? (1 << 20) : g);
}

// Knobs that disable a fast path (tools/exp/k5_sweep.sh, tests): set to anything but "0" / empty. Never needed for
// correctness; absent from the shipped library (knob() above).
inline bool getenv_flag(const char* name)
{
    const char* v = knob(name);
    return v && v[0] && !(v[0] == '0' && !v[1]);
}

// Knob (A/B, tools/k5_bench.py): URSA_BMA_PREFETCH=0/1 overrides whether the C > 64 float4 kernel software-pipelines its loads.
constexpr bool kBmaPrefetchDefault = false;
// Form of the float4 lane-group kernel (k_bma_accumulate, no cost matrix): waves per block = the number of contiguous
// ranges the S members of a row group are split into, and whether wave 0 fetches the row's accumulators at the top of a
// round instead of in its epilogue. Knobs (tools/exp/k5_waves_ab.py): URSA_BMA_WAVES=1|2|4|8, URSA_BMA_EARLY=0|1.
struct BmaForm { int waves; bool early; };
inline BmaForm bma_form(int64_t row_groups, int S, int epl)
{
    // Measured (tools/exp/k5_waves_ab.py, profiles/r04_k5_waves_ab.json): the early fetch gains everywhere (2-20 %); test-set
    // sized calls with 8 / 16 classes per lane run best with 2 / 1 member ranges per row group (fewer, longer-lived waves:
    // 25.9 -> 24.5 us at (30, 10^4, 100), 63.5 -> 56.6 at C = 256), a few rows with 8 (6.7 -> 5.4 us at B = 128).
    BmaForm f{4, true};
    if (row_groups >= 2048 && S >= 4 && epl >= 8) f.waves = epl >= 16 ? 1 : 2;
    else if (row_groups < 256 && S >= 16) f.waves = 8;
    if (const char* v = knob("URSA_BMA_WAVES")) {
        const int w = atoi(v);
        if (w == 1 || w == 2 || w == 4 || w == 8) f.waves = w;
    }
    if (const char* v = knob("URSA_BMA_EARLY")) f.early = v[0] && v[0] != '0';
    return f;
}
inline bool bma_prefetch()
{
    const char* v = knob("URSA_BMA_PREFETCH");
    return v && v[0] ? v[0] != '0' : kBmaPrefetchDefault;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }

enum NoiseSrc { kNoiseOff = 0, kNoisePtr = 1, kNoisePhilox = 2 };

struct StepScalars {
    float lr, mu, c_wd, c_noise, n_train;
    uint32_t flags;
    uint64_t seed, step;
};

// ---------------------------------------------------------------------------------------
// K1: optimSGHMC.step (URSABench/inference/optim_sghmc.py:43-67), one element.
template <bool MOM>
__device__ __forceinline__ void step_elem(float& th, float g, float& v, float e, const StepScalars& s,
                                          bool noise)
{
    if (s.flags & URSA_STEP_WD) g = __builtin_fmaf(s.c_wd, th, g);      // :48
    if (s.flags & URSA_STEP_SGD) {                                        // torch.optim.SGD (swa.py:41-42)
        float b = g;
        if (MOM) {
            b = (s.flags & URSA_STEP_FIRST) ? g : v * s.mu + g;
            v = b;
        }
        th = __builtin_fmaf(-s.lr, b, th);
        return;
    }
    float d;
    if (MOM) {
        float b = (s.flags & URSA_STEP_FIRST) ? g : v;                    // :52
        b = b * s.mu;                                                     // :53/:56
        d = __builtin_fmaf(-s.lr, g, b);
    } else {
        d = g * (-s.lr);                                                  // :62
    }
    if (noise) d = d + (e * s.c_noise) / s.n_train;                       // :64
    th = th + d;                                                          // :65
    v = d;                                                                // :67
}

struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `after_loads` runs once per thread right after the thread's vector loads (before the Philox arithmetic, the update
// and the stores): the control-block launch takes its ticket there.
template <bool MOM, int NOISE, bool NT, class Hook = NoHook, class Hook2 = NoHook>
__device__ __forceinline__ void step_body(float* __restrict__ theta, float* __restrict__ grad,
                                          float* __restrict__ mom, const float* __restrict__ eps,
                                          float* __restrict__ snapshot, int64_t n, const StepScalars& s,
                                          Hook after_loads = Hook(), Hook2 before_stores = Hook2())
{
    const int64_t n4 = n >> 2;
    const bool zero_grad = s.flags & URSA_STEP_ZERO_GRAD;
    float4* __restrict__ th4 = reinterpret_cast<float4*>(theta);
    float4* __restrict__ g4 = reinterpret_cast<float4*>(grad);
    float4* __restrict__ m4 = reinterpret_cast<float4*>(mom);
    const float4* __restrict__ e4 = reinterpret_cast<const float4*>(eps);
    float4* __restrict__ s4 = reinterpret_cast<float4*>(snapshot);

    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one float4 per thread
    if (i < n4) {
        float4 t = ld4<NT>(th4 + i);
        const float4 g = ld4<NT>(g4 + i);
        float4 v = MOM ? ld4<NT>(m4 + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
        if (NOISE == kNoisePtr) e = ld4<NT>(e4 + i);
        after_loads();
        if (NOISE == kNoisePhilox) e = ursa::normal4(s.seed, s.step, (uint64_t)i);
        step_elem<MOM>(t.x, g.x, v.x, e.x, s, NOISE != kNoiseOff);
        step_elem<MOM>(t.y, g.y, v.y, e.y, s, NOISE != kNoiseOff);
        step_elem<MOM>(t.z, g.z, v.z, e.z, s, NOISE != kNoiseOff);
        step_elem<MOM>(t.w, g.w, v.w, e.w, s, NOISE != kNoiseOff);
        before_stores();
        st4<NT>(th4 + i, t);
        if (MOM) st4<NT>(m4 + i, v);
        if (zero_grad) st4<NT>(g4 + i, make_float4(0.f, 0.f, 0.f, 0.f));
        if (snapshot) st4<NT>(s4 + i, t);
    } else {
        after_loads();
        before_stores();
    }
    // scalar tail (n % 4 elements): handled by the first lanes of block 0
    const int64_t tail0 = n4 << 2;
    if (blockIdx.x == 0 && threadIdx.x < (n - tail0)) {
        const int64_t j = tail0 + threadIdx.x;
        float t = theta[j];
        const float g = grad[j];
        float v = MOM ? mom[j] : 0.f;
        float e = 0.f;
        if (NOISE == kNoisePtr) e = eps[j];
        if (NOISE == kNoisePhilox) {
            const float4 z = ursa::normal4(s.seed, s.step, (uint64_t)n4);
            e = threadIdx.x == 0 ? z.x : threadIdx.x == 1 ? z.y : z.z;
        }
        step_elem<MOM>(t, g, v, e, s, NOISE != kNoiseOff);
        theta[j] = t;
        if (MOM) mom[j] = v;
        if (zero_grad) grad[j] = 0.f;
        if (snapshot) snapshot[j] = t;
    }
}

template <bool MOM, int NOISE, bool NT>
__global__ __launch_bounds__(kSBlock) void k_sgmcmc_step(float* theta, float* grad, float* mom,
                                                         const float* eps, float* snapshot, int64_t n,
                                                         StepScalars s)
{
    step_body<MOM, NOISE, NT>(theta, grad, mom, eps, snapshot, n, s);
}

// Advance a control block: the Philox call index, the first-step flag, the per-iteration schedule row.
__device__ __forceinline__ void ctl_advance(ursa_step_ctl* ctl)
{
    const uint64_t step = ctl->step + 1;
    ctl->step = step;
    const uint32_t flags = ctl->flags & ~URSA_STEP_FIRST;
    ctl->flags = flags;
    const float* sched = ctl->sched;
    const uint32_t sched_len = ctl->sched_len;
    if (sched != nullptr && sched_len != 0) {
        const uint64_t k = (step - ctl->sched_base) % sched_len;
        ctl->lr = sched[2 * k];
        if (flags & URSA_STEP_SGD) ctl->mu = sched[2 * k + 1];      // SGD mode has no noise: the column is the momentum
        else ctl->c_noise = sched[2 * k + 1];
    }
}

// Scalars from a device control block (graph-replayable launch), K chains per launch: blockIdx.y is the
// chain, its vectors start at blockIdx.y * chain_stride, its scalars are ctl[blockIdx.y]. One kernel covers
// every (mu, noise) combination with wave-uniform branches: the replayed graph must keep working when the
// host flips NOISE between replays.
//
// URSA_STEP_ADVANCE: the launch advances the chain's block itself (round 2: a second, 1-thread launch per step).
// What has to be ordered is "every workgroup has READ *ctl" before "somebody WRITES *ctl". Each workgroup copies the
// block into registers, passes a workgroup barrier (all of its waves hold their copy), and its thread 0 takes a ticket
// (relaxed agent-scope fetch-add) right after its vector loads, with the next schedule row prefetched. Tickets form a
// two-level tree: workgroup b counts on tickets[(b % 16) * 32], the workgroup that draws the last ticket of such a
// counter re-arms it and counts on tickets[16 * 32], and the one that draws the last ticket THERE knows every workgroup
// of the chain holds its copy: it writes the next step's scalars and re-arms the top counter. Nobody waits on anybody
// else: no fence, no spinning. Why a tree on separate 128-byte lines: fetch-adds to one line serialise at ~11.5 ns
// each, whichever word of the line they hit (tools/exp/ticket_probe.hip, us per launch of an otherwise empty kernel,
// 136 / 1,024 workgroups: empty 2.87 / 2.77; one address 3.46 / 13.5; 8 counters in ONE line 3.97 / 14.1; 8 counters
// on 8 lines 2.84 / 4.18; 16 on 16 lines 2.78 / 3.37). History of this launch at the PreResNet-20 arena (134
// workgroups of 512, tools/k1_ctl_bench.py, us per launch inside a graph replay): no advance 3.2; one-address ticket
// + advance at the END of the workgroup behind __threadfence() 6.5 (a release fence is an L2 write-back per
// workgroup); same without fences 5.4; one-address ticket right after the loads 4.06; a dedicated ticket wave firing at
// kernel start 5.4 (3x worse for 16 chains: all tickets at once, 25-47 ns each); the tree 3.72; the tree with the top
// ticket taken before the stores (this) 3.63 — 0.45 us above the launch that does not advance at all, and the same kernel
// serves roofline-sized chains (WideResNet-28-10: 17,846 workgroups, ~1,100 tickets per line spread over a 108 us launch).
// MULTI: the launch carries several chains (blockIdx.y); a separate instantiation so that a profile tells the
// single-chain launches of a sampler from the multi-chain launches of a ChainGroup by name.
template <bool NT, bool MULTI>
__global__ __launch_bounds__(1024) void k_sgmcmc_step_ctl(float* theta, float* grad, float* mom,
                                                          const float* eps, float* snapshot, int64_t n,
                                                          int64_t chain_stride, ursa_step_ctl* ctl_base)
{
    ursa_step_ctl* ctl = ctl_base + (MULTI ? blockIdx.y : 0);
    if (MULTI) {
        const int64_t off = (int64_t)blockIdx.y * chain_stride;
        theta += off; grad += off; mom += off;
        if (eps) eps += off;
        if (snapshot) snapshot += off;
    }
    StepScalars s;
    s.lr = ctl->lr; s.mu = ctl->mu; s.c_wd = ctl->c_wd; s.c_noise = ctl->c_noise;
    s.n_train = ctl->n_train; s.flags = ctl->flags; s.seed = ctl->seed; s.step = ctl->step;
    const bool advance = s.flags & URSA_STEP_ADVANCE;    // uniform over the chain's workgroups
    uint32_t ticket = 0;
    float next_lr = 0.f, next_col = 0.f;
    bool walk = false;
    if (advance) {
        // "Holds its copy" made explicit rather than left to what __syncthreads() happens to lower to: the wait retires
        // this wave's loads of *ctl (scalar and vector) before it arrives at the barrier, and the memory clobber forbids
        // the compiler to sink those loads below this point or to re-load a field after it (a re-load after the ticket
        // could see the NEXT step's scalars). The ticket below is an atomic behind the barrier: it cannot move up.
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();                                 // every wave of this workgroup holds its copy of *ctl
        if (threadIdx.x == 0) {
            const float* sched = ctl->sched;             // (sched, sched_len, sched_base are never written by the device)
            const uint32_t sched_len = ctl->sched_len;
            if (sched != nullptr && sched_len != 0) {
                const uint64_t k = (s.step + 1 - ctl->sched_base) % sched_len;
                next_lr = sched[2 * k];
                next_col = sched[2 * k + 1];
                walk = true;
            }
        }
    }
    // The address goes through a VGPR the compiler must treat as lane-varying (with a provably uniform address its
    // atomic optimizer rewrites the add into a wave-aggregated one whose v_readfirstlane waits for the result at once)
    // and stays a GLOBAL pointer (a flat atomic forces vmcnt(0)/lgkmcnt(0) waits).
    constexpr uint32_t kLines = URSA_CTL_TICKET_LINES;
    const uint32_t line = blockIdx.x % kLines;          // this workgroup's first-level counter
    auto take_ticket = [&]() {
        if (advance && threadIdx.x == 0) {
            typedef __attribute__((address_space(1))) uint32_t gu32;
            gu32* tp = (gu32*)&ctl->tickets[line * 32];
            asm volatile("" : "+v"(tp));
            ticket = __hip_atomic_fetch_add(tp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    // Second level, taken between the arithmetic and the stores (the first-level ticket has been in flight since the
    // loads: it is back by now), so that ITS round trip runs under the stores instead of after them.
    uint32_t top = 0xFFFFFFFFu;
    auto take_top_ticket = [&]() {
        // workgroups counting on this line: those with blockIdx.x = line, line + 16, ... < gridDim.x
        if (advance && threadIdx.x == 0 && ticket == (gridDim.x - line + kLines - 1) / kLines - 1) {
            __hip_atomic_store(&ctl->tickets[line * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            typedef __attribute__((address_space(1))) uint32_t gu32;
            gu32* tp = (gu32*)&ctl->tickets[kLines * 32];
            asm volatile("" : "+v"(tp));
            top = __hip_atomic_fetch_add(tp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    const bool noise = s.flags & URSA_STEP_NOISE;
    if (s.mu != 0.0f) {
        if (!noise) step_body<true, kNoiseOff, NT>(theta, grad, mom, eps, snapshot, n, s, take_ticket, take_top_ticket);
        else if (eps) step_body<true, kNoisePtr, NT>(theta, grad, mom, eps, snapshot, n, s, take_ticket, take_top_ticket);
        else step_body<true, kNoisePhilox, NT>(theta, grad, mom, eps, snapshot, n, s, take_ticket, take_top_ticket);
    } else {
        if (!noise) step_body<false, kNoiseOff, NT>(theta, grad, mom, eps, snapshot, n, s, take_ticket, take_top_ticket);
        else if (eps) step_body<false, kNoisePtr, NT>(theta, grad, mom, eps, snapshot, n, s, take_ticket, take_top_ticket);
        else step_body<false, kNoisePhilox, NT>(theta, grad, mom, eps, snapshot, n, s, take_ticket, take_top_ticket);
    }
    if (advance && threadIdx.x == 0) {
        const uint32_t lines_in_use = gridDim.x < kLines ? gridDim.x : kLines;
        if (top == lines_in_use - 1) {                   // every workgroup of the chain holds its copy of the block
            ctl->step = s.step + 1;                      // same arithmetic as ctl_advance()
            ctl->flags = s.flags & ~URSA_STEP_FIRST;
            if (walk) {
                ctl->lr = next_lr;
                if (s.flags & URSA_STEP_SGD) ctl->mu = next_col;
                else ctl->c_noise = next_col;
            }
            __hip_atomic_store(&ctl->tickets[kLines * 32], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void k_step_ctl_advance(ursa_step_ctl* ctl, int n_ctl)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_ctl) ctl_advance(ctl + k);
}

// Unaligned fallback: same arithmetic, 4 B per lane. Element i still takes lane (i & 3) of
// Philox block (i >> 2), so results do not depend on which path ran.
template <bool MOM, int NOISE>
__global__ __launch_bounds__(kBlock) void k_sgmcmc_step_scalar(float* theta, float* grad, float* mom,
                                                               const float* eps, float* snapshot,
                                                               int64_t n, StepScalars s)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float t = theta[i];
        const float g = grad[i];
        float v = MOM ? mom[i] : 0.f;
        float e = 0.f;
        if (NOISE == kNoisePtr) e = eps[i];
        if (NOISE == kNoisePhilox) {
            const float4 z = ursa::normal4(s.seed, s.step, (uint64_t)(i >> 2));
            const int l = (int)(i & 3);
            e = l == 0 ? z.x : l == 1 ? z.y : l == 2 ? z.z : z.w;
        }
        step_elem<MOM>(t, g, v, e, s, NOISE != kNoiseOff);
        theta[i] = t;
        if (MOM) mom[i] = v;
        if (s.flags & URSA_STEP_ZERO_GRAD) grad[i] = 0.f;
        if (snapshot) snapshot[i] = t;
    }
}

__global__ __launch_bounds__(kSBlock) void k_philox_normal(float* out, int64_t n, uint64_t seed, uint64_t call)
{
    const int64_t n4 = (n + 3) >> 2;
    const int64_t i = (int64_t)blockIdx.x * kSBlock + threadIdx.x;
    if (i >= n4) return;
    const float4 z = ursa::normal4(seed, call, (uint64_t)i);
    const int64_t b = i << 2;
    if (b + 3 < n && ((reinterpret_cast<uintptr_t>(out) & 15u) == 0)) {
        reinterpret_cast<float4*>(out)[i] = z;
    } else {
        if (b < n) out[b] = z.x;
        if (b + 1 < n) out[b + 1] = z.y;
        if (b + 2 < n) out[b + 2] = z.z;
        if (b + 3 < n) out[b + 3] = z.w;
    }
}

// Self-test of the generator's fast division / square root (csrc/ursa_rng.h): for EVERY 32-bit Philox word the
// Box-Muller radius computed with them must equal, bit for bit, the one computed with the compiler's IEEE division and
// square root (what the scalar C oracle uses). counts[0] += mismatching radii, counts[1] += mismatching logarithms.
__global__ __launch_bounds__(256) void k_selftest_rng(unsigned long long* counts)
{
    const uint32_t base = ((uint32_t)blockIdx.x * 256u + threadIdx.x) << 8;      // 2^24 threads x 256 words
    unsigned bad_r = 0, bad_l = 0;
    for (uint32_t k = 0; k < 256u; ++k) {
        const uint32_t w = base + k;
        const float u1 = __builtin_fmaf((float)w, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
        bad_l += __float_as_uint(ursa::det_logf<false>(u1)) != __float_as_uint(ursa::det_logf<true>(u1));
        bad_r += __float_as_uint(ursa::radius_of<false>(w)) != __float_as_uint(ursa::radius_of<true>(w));
    }
    if (bad_r) atomicAdd(&counts[0], (unsigned long long)bad_r);
    if (bad_l) atomicAdd(&counts[1], (unsigned long long)bad_l);
}

// ---------------------------------------------------------------------------------------
// K2: SWA._collect_model (URSABench/inference/swa.py:81-88)
__device__ __forceinline__ void collect_elem(float& m, float& q, float w, float decay, float denom)
{
    m = m * decay + w / denom;
    q = q * decay + (w * w) / denom;
}

template <bool NT>
__global__ __launch_bounds__(kSBlock) void k_swag_collect_v(float* __restrict__ mean, float* __restrict__ sq,
                                                            const float* __restrict__ w, int64_t n, float decay,
                                                            float denom)
{
    const int64_t n4 = n >> 2;
    const int64_t i = (int64_t)blockIdx.x * kSBlock + threadIdx.x;
    if (i < n4) {
        float4 m = ld4<NT>(reinterpret_cast<const float4*>(mean) + i);
        float4 q = ld4<NT>(reinterpret_cast<const float4*>(sq) + i);
        const float4 x = ld4<NT>(reinterpret_cast<const float4*>(w) + i);
        collect_elem(m.x, q.x, x.x, decay, denom);
        collect_elem(m.y, q.y, x.y, decay, denom);
        collect_elem(m.z, q.z, x.z, decay, denom);
        collect_elem(m.w, q.w, x.w, decay, denom);
        st4<NT>(reinterpret_cast<float4*>(mean) + i, m);
        st4<NT>(reinterpret_cast<float4*>(sq) + i, q);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t j = (n4 << 2) + threadIdx.x;
        collect_elem(mean[j], sq[j], w[j], decay, denom);
    }
}

__global__ __launch_bounds__(kBlock) void k_swag_collect_s(float* __restrict__ mean, float* __restrict__ sq,
                                                           const float* __restrict__ w, int64_t n, float decay,
                                                           float denom)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride)
        collect_elem(mean[i], sq[i], w[i], decay, denom);
}

// K3: diagonal SWAG draw (swa.py:106-108, swag.py:84-86)
__device__ __forceinline__ float draw_elem(float m, float q, float e, float var_clamp, float scale)
{
    float var = q - m * m;
    var = var < var_clamp ? var_clamp : var;
    return e * (__builtin_sqrtf(var) * scale) + m;
}

template <bool PHILOX, bool NT>
__global__ __launch_bounds__(kSBlock) void k_swag_draw_v(float* __restrict__ out, const float* __restrict__ mean,
                                                         const float* __restrict__ sq, const float* __restrict__ eps,
                                                         int64_t n, float var_clamp, float scale, uint64_t seed,
                                                         uint64_t draw)
{
    const int64_t n4 = n >> 2;
    const int64_t i = (int64_t)blockIdx.x * kSBlock + threadIdx.x;
    if (i < n4) {
        const float4 m = ld4<NT>(reinterpret_cast<const float4*>(mean) + i);
        const float4 q = ld4<NT>(reinterpret_cast<const float4*>(sq) + i);
        const float4 e = PHILOX ? ursa::normal4(seed, draw, (uint64_t)i)
                                : ld4<NT>(reinterpret_cast<const float4*>(eps) + i);
        float4 t;
        t.x = draw_elem(m.x, q.x, e.x, var_clamp, scale);
        t.y = draw_elem(m.y, q.y, e.y, var_clamp, scale);
        t.z = draw_elem(m.z, q.z, e.z, var_clamp, scale);
        t.w = draw_elem(m.w, q.w, e.w, var_clamp, scale);
        st4<NT>(reinterpret_cast<float4*>(out) + i, t);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t j = (n4 << 2) + threadIdx.x;
        float e;
        if (PHILOX) {
            const float4 z = ursa::normal4(seed, draw, (uint64_t)n4);
            e = threadIdx.x == 0 ? z.x : threadIdx.x == 1 ? z.y : z.z;
        } else {
            e = eps[j];
        }
        out[j] = draw_elem(mean[j], sq[j], e, var_clamp, scale);
    }
}

template <bool PHILOX>
__global__ __launch_bounds__(kBlock) void k_swag_draw_s(float* __restrict__ out, const float* __restrict__ mean,
                                                        const float* __restrict__ sq, const float* __restrict__ eps,
                                                        int64_t n, float var_clamp, float scale, uint64_t seed,
                                                        uint64_t draw)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float e;
        if (PHILOX) {
            const float4 z = ursa::normal4(seed, draw, (uint64_t)(i >> 2));
            const int l = (int)(i & 3);
            e = l == 0 ? z.x : l == 1 ? z.y : l == 2 ? z.z : z.w;
        } else {
            e = eps[i];
        }
        out[i] = draw_elem(mean[i], sq[i], e, var_clamp, scale);
    }
}

// K3 split for ensembles: the standard deviation sqrt(max(sq - mean^2, clamp)) * scale is the same for every member
// drawn from one pair of moment vectors (30 members in BASELINE configs[3]); computing it inside every draw costs four
// correctly rounded square roots per float4 — a fifth of the draw's VALU work, and the draw is VALU-co-limited (Philox +
// Box-Muller in registers against only 12 B/param): under the 1,400 W package power cap the clock drops to ~1.8 GHz
// and the fused draw reads 0.57 of the HBM peak (tools/exp/k3_spread.py). k_swag_std_v stores it once; k_swag_draw_std_v
// is the per-member draw theta = eps * std + mean on the same 12 B/param, bit-identical to the fused kernel.
template <bool NT>
__global__ __launch_bounds__(kSBlock) void k_swag_std_v(float* __restrict__ out, const float* __restrict__ mean,
                                                        const float* __restrict__ sq, int64_t n, float var_clamp, float scale)
{
    const int64_t n4 = n >> 2;
    const int64_t i = (int64_t)blockIdx.x * kSBlock + threadIdx.x;
    auto sd = [&](float m, float q) {
        float var = q - m * m;
        var = var < var_clamp ? var_clamp : var;
        return __builtin_sqrtf(var) * scale;
    };
    if (i < n4) {
        const float4 m = ld4<NT>(reinterpret_cast<const float4*>(mean) + i);
        const float4 q = ld4<NT>(reinterpret_cast<const float4*>(sq) + i);
        st4<NT>(reinterpret_cast<float4*>(out) + i, make_float4(sd(m.x, q.x), sd(m.y, q.y), sd(m.z, q.z), sd(m.w, q.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t j = (n4 << 2) + threadIdx.x;
        out[j] = sd(mean[j], sq[j]);
    }
}

__global__ __launch_bounds__(kBlock) void k_swag_std_s(float* __restrict__ out, const float* __restrict__ mean,
                                                       const float* __restrict__ sq, int64_t n, float var_clamp, float scale)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float var = sq[i] - mean[i] * mean[i];
        var = var < var_clamp ? var_clamp : var;
        out[i] = __builtin_sqrtf(var) * scale;
    }
}

template <bool PHILOX, bool NT>
__global__ __launch_bounds__(kSBlock) void k_swag_draw_std_v(float* __restrict__ out, const float* __restrict__ mean,
                                                             const float* __restrict__ sd, const float* __restrict__ eps,
                                                             int64_t n, uint64_t seed, uint64_t draw)
{
    const int64_t n4 = n >> 2;
    const int64_t i = (int64_t)blockIdx.x * kSBlock + threadIdx.x;
    if (i < n4) {
        const float4 m = ld4<NT>(reinterpret_cast<const float4*>(mean) + i);
        const float4 s = ld4<NT>(reinterpret_cast<const float4*>(sd) + i);
        const float4 e = PHILOX ? ursa::normal4(seed, draw, (uint64_t)i)
                                : ld4<NT>(reinterpret_cast<const float4*>(eps) + i);
        st4<NT>(reinterpret_cast<float4*>(out) + i, make_float4(e.x * s.x + m.x, e.y * s.y + m.y, e.z * s.z + m.z, e.w * s.w + m.w));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t j = (n4 << 2) + threadIdx.x;
        float e;
        if (PHILOX) {
            const float4 z = ursa::normal4(seed, draw, (uint64_t)n4);
            e = threadIdx.x == 0 ? z.x : threadIdx.x == 1 ? z.y : z.z;
        } else {
            e = eps[j];
        }
        out[j] = e * sd[j] + mean[j];
    }
}

template <bool PHILOX>
__global__ __launch_bounds__(kBlock) void k_swag_draw_std_s(float* __restrict__ out, const float* __restrict__ mean,
                                                            const float* __restrict__ sd, const float* __restrict__ eps,
                                                            int64_t n, uint64_t seed, uint64_t draw)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        float e;
        if (PHILOX) {
            const float4 z = ursa::normal4(seed, draw, (uint64_t)(i >> 2));
            const int l = (int)(i & 3);
            e = l == 0 ? z.x : l == 1 ? z.y : l == 2 ? z.z : z.w;
        } else {
            e = eps[i];
        }
        out[i] = e * sd[i] + mean[i];
    }
}

// ---------------------------------------------------------------------------------------
// K5: ensemble softmax-mean / entropy / risk accumulation (tasks/prediction.py:57-63,
// ood_detection.py:59-65, decision_making.py:124-129, util.py:126-144).
//
// A row of C logits is owned by a group of G lanes (G = power of two <= 64, G >= C when
// C <= 64), lane l of the group holds classes l, l+G, ... (EPL per lane). Row reductions
// are xor-butterflies over the G lanes (ds_swizzle/dpp, no LDS round trip). The S members
// are walked in order with the accumulators in registers: one read-modify-write of
// proba_sum / ent_sum / risk_sum per row per launch.
// Softmax arithmetic of K5 (both kernels): e^(x - max) = 2^(x log2e - max log2e) as ONE fma + v_exp_f32 per class.
// The fma's rounding (half an ulp of |x - max| log2e <= ~2^-19 for gaps < 44) gives <= 7e-7 relative error in e —
// the size of the error the reference's own log_softmax().exp() carries from rounding x - logsumexp(x) — against the
// 1e-5 bar; the rounding of max*log2e is common to a row and cancels in e / sum. (This round first used a hi/lo
// split of log2e that folded the product's error back in, 1-2 ulp, 6 instructions: at C = 100 the kernel is
// VALU-bound and that cost 2 us of 30; measured worst case of the plain form over 8 members: 1.6e-6 on p, 2.5e-6
// on the entropy.) 2^-inf = 0 exactly: a masked class (-inf logit) needs no special case.

// Lane-group reductions on the VALU only (no LDS round trip, no s_waitcnt): xor butterflies inside
// a quad (DPP quad_perm), then row_half_mirror / row_mirror inside a 16-lane row; across rows
// v_readlane of one lane per row. Every lane of the group ends with the same bits (the combining
// order is a fixed tree). ds_bpermute-based __shfl_xor spent
// 63 % of this kernel's wave-cycles in s_waitcnt (rocprofv3 SQ_WAIT_ANY, profiles/).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
struct OpMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct OpSum { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };

template <int G, class Op>
__device__ __forceinline__ float group_reduce(float v, Op op)
{
    if (G >= 2) v = op(v, dpp_mov<0xB1>(v));        // quad_perm [1,0,3,2]
    if (G >= 4) v = op(v, dpp_mov<0x4E>(v));        // quad_perm [2,3,0,1]
    if (G >= 8) v = op(v, dpp_mov<0x141>(v));       // row_half_mirror
    if (G >= 16) v = op(v, dpp_mov<0x140>(v));      // row_mirror
    // Across 16-lane rows: every lane of a row now holds its row's result, so four v_readlane pull
    // the row results into SGPRs and two/three more ops combine them. (gfx950's v_permlane16/32_swap
    // would do it in fewer instructions, but the builtin's second result is mis-lowered by ROCm 7.2's
    // hipcc — both halves alias one register: tools/exp/dpp_probe.hip.)
    if (G >= 32) {
        const int iv = __builtin_bit_cast(int, v);
        const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
        const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
        const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
        const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
        const float lo = op(r0, r1), hi = op(r2, r3);
        if (G >= 64) v = op(lo, hi);
        else v = (threadIdx.x & 32) ? hi : lo;
    }
    return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) { return group_reduce<G>(v, OpMax()); }
template <int G>
__device__ __forceinline__ float group_sum(float v) { return group_reduce<G>(v, OpSum()); }

// Class owned by register e of lane `lane`. Scalar mapping: lane + e*G (4-byte loads, any C / alignment).
// V4 mapping: 4*(lane + (e/4)*G) + e%4 — four consecutive classes per float4 load (rows must be 16-byte
// aligned: C % 4 == 0 and aligned base pointers); EPL is then a multiple of 4.
template <int G, bool V4>
__device__ __forceinline__ int bma_class(int lane, int e) { return V4 ? 4 * (lane + (e >> 2) * G) + (e & 3) : lane + e * G; }

// U consecutive members of one row, for the lane group that owns the row: the U x EPL loads are issued before
// any arithmetic; each member is folded into the wave's partial sums as soon as it is computed (member order).
// K5 is a 1e-5-relative kernel (ATen's softmax is not bit-reproducible in scalar code anyway), so unlike K1-K4
// its arithmetic fuses explicitly: e = exp(x - max) through v_exp_f32 with a compensated exponent (1-2 ulp),
// p = e / sum (one IEEE reciprocal per row instead of the reference's second exp), q = fma(p, 1-g, g/C),
// entropy accumulated as sum q log2 q on v_log_f32 (ln 2 applied once per row at the end).
// Loads carry NO class masks: a lane whose classes lie beyond C reads (clamped address) logits that really exist,
// which cannot change the row maximum; the mask enters once, as the -inf bias of the exponent fma in lg_compute.
template <int G, int EPL, bool V4, int U>
__device__ __forceinline__ void lg_load(const float* __restrict__ z0, int64_t member_stride, int lane, int C, float (&xs)[U][EPL])
{
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float* z = z0 + u * member_stride;
        if (V4) {
#pragma unroll
            for (int v4 = 0; v4 < EPL / 4; ++v4) {
                const int c0 = 4 * (lane + v4 * G);         // C % 4 == 0: a float4 is all inside or all outside
                const float4 v = *reinterpret_cast<const float4*>(z + (c0 < C ? c0 : C - 4));
                xs[u][4 * v4 + 0] = v.x; xs[u][4 * v4 + 1] = v.y; xs[u][4 * v4 + 2] = v.z; xs[u][4 * v4 + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int c = lane + e * G;
                xs[u][e] = z[c < C ? c : C - 1];
            }
        }
    }
}

template <int G, int EPL, bool RISK, bool V4, int U>
__device__ __forceinline__ void lg_compute(float (&xs)[U][EPL], int lane, int C, float omg, float goc, bool want_ent,
                                           const float* __restrict__ cost, float (&acc_p)[EPL],
                                           float (&acc_r)[RISK ? EPL : 1], float& acc_e2)
{
    constexpr float kLog2e = 1.44269502162933349609375f;
    // every class slot beyond C adds exactly q = gamma/C to the entropy sum: taken out analytically, per row and member
    const float ent_fake = (float)(G * EPL - C) * (goc * __builtin_amdgcn_logf(goc));
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float mx = xs[u][0];
#pragma unroll
        for (int e = 1; e < EPL; ++e) mx = fmaxf(mx, xs[u][e]);
        mx = group_max<G>(mx);
        // e^(x - max) = 2^(x log2e - max log2e): ONE fma per class; its bias is -inf for class slots beyond C (e = 0
        // there). The rounding of max*log2e is common to the row and cancels in e / sum; a -inf logit gives 0.
        const float nb = -(mx * kLog2e);
        float nbv[V4 ? EPL / 4 : EPL];                      // one bias per float4 (V4) or per class
#pragma unroll
        for (int k = 0; k < (V4 ? EPL / 4 : EPL); ++k) nbv[k] = bma_class<G, V4>(lane, V4 ? 4 * k : k) < C ? nb : -INFINITY;
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            xs[u][e] = __builtin_amdgcn_exp2f(__builtin_fmaf(xs[u][e], kLog2e, nbv[V4 ? e / 4 : e]));
            sum += xs[u][e];
        }
        const float inv = 1.0f / group_sum<G>(sum);
        const float inv_omg = inv * omg;
        float ent2 = 0.f;
        float qv[RISK ? EPL : 1];
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            // p = e / sum enters both uses through one fma each: q = (1-g) p + g/C and acc += p
            const float q = __builtin_fmaf(xs[u][e], inv_omg, goc);   // goc > 0 (the launcher floors it): q > 0, log finite
            ent2 = __builtin_fmaf(q, __builtin_amdgcn_logf(q), ent2);
            acc_p[e] = __builtin_fmaf(xs[u][e], inv, acc_p[e]);       // raw p always; smoothing of the SUM is applied once per row
            if (RISK) qv[e] = q;                             // (slots beyond C: junk, never stored / never read)
        }
        if (want_ent) acc_e2 += group_sum<G>(ent2) - ent_fake;
        if (RISK) {
            // risk[b, j] += sum_c ps[c] * cost[c, j]; ps[c] broadcast from its owner lane
            float r[RISK ? EPL : 1];
#pragma unroll
            for (int e = 0; e < (RISK ? EPL : 1); ++e) r[e] = 0.f;
#pragma unroll
            for (int es = 0; es < (RISK ? EPL : 1); ++es) {
                for (int src = 0; src < G; ++src) {
                    const int c = bma_class<G, V4>(src, es);
                    if (c >= C) break;                      // uniform across the group
                    const float pc = __shfl(qv[es], src, G);
#pragma unroll
                    for (int e = 0; e < (RISK ? EPL : 1); ++e) {
                        const int j = bma_class<G, V4>(lane, e);
                        if (j < C) r[e] = __builtin_fmaf(pc, cost[c * C + j], r[e]);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < (RISK ? EPL : 1); ++e) acc_r[e] += r[e];
        }
    }
}

template <int G, int EPL, bool RISK, bool V4, int U>
__device__ __forceinline__ void lg_members(const float* __restrict__ z0, int64_t member_stride, int lane, int C,
                                           float omg, float goc, bool want_ent,
                                           const float* __restrict__ cost, float (&acc_p)[EPL],
                                           float (&acc_r)[RISK ? EPL : 1], float& acc_e2)
{
    float xs[U][EPL];
    lg_load<G, EPL, V4, U>(z0, member_stride, lane, C, xs);
    lg_compute<G, EPL, RISK, V4, U>(xs, lane, C, omg, goc, want_ent, cost, acc_p, acc_r, acc_e2);
}

template <int G, int EPL, bool RISK, bool V4, bool PF = false, int W = kBlock / 64, bool EARLY = false>
__global__ __launch_bounds__(64 * W) void k_bma_accumulate(const float* __restrict__ logits,
                                                           float* __restrict__ proba_sum,
                                                           float* __restrict__ ent_sum,
                                                           float* __restrict__ risk_sum,
                                                           const float* __restrict__ cost, int S, int64_t B,
                                                           int C, float omg, float goc, uint32_t flags)
{
    // A block owns 64/G rows. Its 4 waves split the S members into 4 contiguous ranges (4x the
    // parallelism of one-wave-per-row: B = 10^4 rows are only 2,500 waves otherwise, 2.4 per SIMD), each
    // wave sums its range in member order in registers, and wave 0 folds the partial sums in wave order
    // into the global accumulators: ((acc + P0) + P1) + P2) + P3 — a fixed, reproducible order.
    // RISK (Decision's expected-cost accumulator) is a template flag so Prediction/OOD do not carry its
    // registers: <16 lanes, 8 classes per lane> needs 122 VGPRs with it (4 waves/SIMD), half without.
    constexpr int kRows = 64 / G;
    constexpr int kWaves = W;
    constexpr int kVals = (RISK ? 2 : 1) * EPL + 1;          // proba[EPL], (risk[EPL]), entropy
    __shared__ float part[kWaves > 1 ? kWaves - 1 : 1][kVals][64];
    const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63;
    const int lane = wl % G, grp = wl / G;
    const bool smoothed = flags & URSA_BMA_SMOOTHED;
    const int s_per = (S + kWaves - 1) / kWaves;
    const int s_lo = wave * s_per < S ? wave * s_per : S;
    const int s_hi = s_lo + s_per < S ? s_lo + s_per : S;
    const int64_t row_stride = (int64_t)gridDim.x * kRows;
    const int64_t nrounds = (B + row_stride - 1) / row_stride;   // same trip count for every wave (barriers)
    const int64_t BC = B * (int64_t)C;

    for (int64_t it = 0; it < nrounds; ++it) {
        const int64_t b = it * row_stride + (int64_t)blockIdx.x * kRows + grp;
        const bool row_ok = b < B;
        float acc_p[EPL], acc_r[RISK ? EPL : 1];
        float acc_e2 = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc_p[e] = 0.f;
#pragma unroll
        for (int e = 0; e < (RISK ? EPL : 1); ++e) acc_r[e] = 0.f;

        // EARLY: wave 0 fetches the row's accumulators now, so that the read-modify-write at the end of the round does not
        // start with a cold load while every other block of the launch is in its own epilogue too
        float old_p[EARLY ? EPL : 1], old_e = 0.f;
        if (EARLY && wave == 0 && row_ok) {
#pragma unroll
            for (int e = 0; e < (EARLY ? EPL : 1); ++e) {
                const int c = bma_class<G, V4>(lane, e);
                old_p[e] = proba_sum[b * C + (c < C ? c : C - 1)];
            }
            if (ent_sum) old_e = ent_sum[b];
        }
        // full chunks of U members, then the remainder one at a time: no per-member liveness masks anywhere
#ifndef URSA_BMA_U_EPL8
#define URSA_BMA_U_EPL8 2
#endif
#ifndef URSA_BMA_U_EPL16
#define URSA_BMA_U_EPL16 1
#endif
#ifndef URSA_BMA_U_EPL8_W2
#define URSA_BMA_U_EPL8_W2 1
#endif
#ifndef URSA_BMA_U_EPL4
#define URSA_BMA_U_EPL4 4
#endif
        // members whose loads are in flight per lane
        constexpr int U = EPL <= 2 ? 8 : EPL <= 4 ? URSA_BMA_U_EPL4 : EPL <= 8 ? (W <= 2 ? URSA_BMA_U_EPL8_W2 : URSA_BMA_U_EPL8) : URSA_BMA_U_EPL16;
        const float* zrow = logits + (row_ok ? b : 0) * (int64_t)C;          // rows past B recompute row 0 (never stored)
        int s0 = s_lo;
        if (PF) {
            // software pipeline: the next chunk's loads are issued before this chunk's arithmetic (two register buffers)
            float xa[U][EPL], xb[U][EPL];
            if (s0 + U <= s_hi) lg_load<G, EPL, V4, U>(zrow + s0 * BC, BC, lane, C, xa);
#pragma unroll 1
            for (; s0 + 2 * U <= s_hi; s0 += 2 * U) {
                lg_load<G, EPL, V4, U>(zrow + (s0 + U) * BC, BC, lane, C, xb);
                lg_compute<G, EPL, RISK, V4, U>(xa, lane, C, omg, goc, ent_sum != nullptr, cost, acc_p, acc_r, acc_e2);
                if (s0 + 3 * U <= s_hi) lg_load<G, EPL, V4, U>(zrow + (s0 + 2 * U) * BC, BC, lane, C, xa);
                lg_compute<G, EPL, RISK, V4, U>(xb, lane, C, omg, goc, ent_sum != nullptr, cost, acc_p, acc_r, acc_e2);
            }
            if (s0 + U <= s_hi) {            // one full chunk left: it is in xa already
                lg_compute<G, EPL, RISK, V4, U>(xa, lane, C, omg, goc, ent_sum != nullptr, cost, acc_p, acc_r, acc_e2);
                s0 += U;
            }
        } else {
#pragma unroll 1
            for (; s0 + U <= s_hi; s0 += U)
                lg_members<G, EPL, RISK, V4, U>(zrow + s0 * BC, BC, lane, C, omg, goc, ent_sum != nullptr, cost,
                                                acc_p, acc_r, acc_e2);
        }
        if (U > 1)
#pragma unroll 1
            for (; s0 < s_hi; ++s0)
                lg_members<G, EPL, RISK, V4, 1>(zrow + s0 * BC, BC, lane, C, omg, goc, ent_sum != nullptr,
                                                cost, acc_p, acc_r, acc_e2);
        const float acc_e = -0.693147182464599609375f * acc_e2;
        if (smoothed) {                                      // sum_s ((1-g) p_s + g/C) = (1-g) sum_s p_s + n g/C
            const float ng = (float)(s_hi - s_lo) * goc;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc_p[e] = __builtin_fmaf(acc_p[e], omg, ng);
        }

        if (wave > 0) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                part[wave - 1][e][wl] = acc_p[e];
                if (RISK) part[wave - 1][EPL + e][wl] = acc_r[e];
            }
            part[wave - 1][kVals - 1][wl] = acc_e;
        }
        __syncthreads();
        if (wave == 0 && row_ok) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const int c = bma_class<G, V4>(lane, e);
                if (c < C) {
                    float p = (EARLY ? old_p[EARLY ? e : 0] : proba_sum[b * C + c]) + acc_p[e];
#pragma unroll
                    for (int w = 0; w < kWaves - 1; ++w) p += part[w][e][wl];
                    proba_sum[b * C + c] = p;
                    if (RISK) {
                        float r = risk_sum[b * C + c] + acc_r[e];
#pragma unroll
                        for (int w = 0; w < kWaves - 1; ++w) r += part[w][EPL + e][wl];
                        risk_sum[b * C + c] = r;
                    }
                }
            }
            if (ent_sum && lane == 0) {
                float en = (EARLY ? old_e : ent_sum[b]) + acc_e;
#pragma unroll
                for (int w = 0; w < kWaves - 1; ++w) en += part[w][kVals - 1][wl];
                ent_sum[b] = en;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// K5, few classes (C <= 16): ONE LANE OWNS ONE (row, member-slot) — all C classes in registers, zero
// cross-lane reductions, no idle lanes. A block owns a tile of 16 consecutive rows; its W waves x 4
// quarter-waves are 4W member slots, slot q walks a contiguous range of members in order. For one member
// the tile's 16 x C logits are CONTIGUOUS in the [S, B, C] slab, so a wave fetches the four tiles of a
// round (one per quarter-wave) with coalesced float4 loads — every round of a chunk issued before any
// arithmetic — and passes them through a wave-private LDS stage to turn "float4 i of the tile" into "row r
// of the tile" (the LDS-staged transpose north_star names; 10 floats per row make direct row loads 40-byte
// strided). Partial sums of the 4W slots are folded in slot order through LDS into ONE read-modify-write
// of the global accumulators per row: a fixed, reproducible order. 16-row tiles give ceil(B/16) blocks
// (625 for the 10,000-row test set: every CU streams; with 64-row tiles only 157 of 256 CUs would).
// Needs 16-byte aligned member tiles: aligned `logits` and B*C % 4 == 0; otherwise the lane-group kernel runs.
constexpr int kRlRows = 16;      // rows per tile
constexpr int kRlChunk = 4;      // rounds whose loads are in flight together

// K5 tolerates 1e-5 relative (ATen's own softmax is not bit-reproducible in scalar code), so unlike
// K1-K4 its arithmetic may fuse: explicit fmaf where a multiply feeds an add.
// One (row, member): C logits from the LDS stage -> softmax -> smoothed entropy -> fold into the slot's sums.
// EXACT: C == CP (no class masks). MASKED: the slot may have no member in this round (last round only).
template <int CP, bool EXACT, bool RISK, bool MASKED>
__device__ __forceinline__ void rl_member(const float* __restrict__ row, int C, bool live, float omg, float goc,
                                          const float* __restrict__ cost, float (&acc_p)[CP],
                                          float (&acc_r)[RISK ? CP : 1], float& acc_e2)
{
    float x[CP];
    if (EXACT && CP % 2 == 0) {                          // rows are 8-byte aligned in the stage: ds_read_b64
#pragma unroll
        for (int c = 0; c < CP; c += 2) {
            const float2 v = *reinterpret_cast<const float2*>(row + c);
            x[c] = v.x; x[c + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int c = 0; c < CP; ++c) x[c] = (EXACT || c < C) ? row[(EXACT || c < C) ? c : 0] : -INFINITY;
    }
    float mx = x[0];
#pragma unroll
    for (int c = 1; c < CP; ++c) mx = fmaxf(mx, x[c]);
    constexpr float kLog2e = 1.44269502162933349609375f;
    const float nb = -(mx * kLog2e);                     // see lg_members: one fma per class, common rounding cancels
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        x[c] = __builtin_amdgcn_exp2f(__builtin_fmaf(x[c], kLog2e, nb));      // masked classes hold -inf: exp = 0
        sum += x[c];
    }
    const float inv = 1.0f / sum;
    const float w = (MASKED && !live) ? 0.f : 1.f;       // an idle slot computes on zeros and adds nothing
    float ent2 = 0.f;                                    // sum q log2 q (ln 2 applied once per slot at the end)
    float qv[RISK ? CP : 1];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
        const float p = x[c] * inv;
        float q = __builtin_fmaf(p, omg, goc);           // goc > 0 (the launcher floors it): q > 0, log finite
        const float ql = q * __builtin_amdgcn_logf(q);
        ent2 += (EXACT || c < C) ? ql : 0.f;
        if (!EXACT) q = c < C ? q : 0.f;
        acc_p[c] = MASKED ? __builtin_fmaf(w, p, acc_p[c]) : acc_p[c] + p;    // raw p; the SUM is smoothed once per slot
        if (RISK) qv[c] = q;
    }
    acc_e2 = MASKED ? __builtin_fmaf(w, ent2, acc_e2) : acc_e2 + ent2;
    if (RISK) {
#pragma unroll
        for (int jc = 0; jc < (RISK ? CP : 1); ++jc) {
            float rr = 0.f;
#pragma unroll
            for (int c = 0; c < (RISK ? CP : 1); ++c)
                if (EXACT || (c < C && jc < C)) rr = __builtin_fmaf(qv[c], cost[c * C + jc], rr);   // uniform address: scalar loads
            acc_r[jc] = MASKED ? __builtin_fmaf(w, rr, acc_r[jc]) : acc_r[jc] + rr;
        }
    }
}

template <int CP, bool EXACT, bool RISK>
__global__ __launch_bounds__(512) void k_bma_rowlane(const float* __restrict__ logits, float* __restrict__ proba_sum,
                                                      float* __restrict__ ent_sum, float* __restrict__ risk_sum,
                                                      const float* __restrict__ cost, int S, int64_t B, int C,
                                                      float omg, float goc, uint32_t flags)
{
    extern __shared__ float4 smem4[];
    float* smem = reinterpret_cast<float*>(smem4);
    constexpr int NL = (CP + 3) / 4;                     // float4 loads per lane per round (16*C float4 over 64 lanes)
    const int W = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, wl = threadIdx.x & 63;
    const int r = wl & 15, k = wl >> 4;
    const int nslots = 4 * W;
    const int base = S / nslots, extra = S % nslots;     // every slot has `base` members, the first `extra` one more
    const int tile4 = 4 * C;                             // float4 per (member, tile): 16 rows x C floats
    const int64_t r0 = (int64_t)blockIdx.x * kRlRows;
    const int64_t BC = B * (int64_t)C;
    const bool smoothed = flags & URSA_BMA_SMOOTHED;
    float* stage = smem + wave * (4 * kRlRows * C);      // wave-private: 4 tiles
    const int out_per_slot = kRlRows * C * (RISK ? 2 : 1) + kRlRows;
    float* part = smem + W * (4 * kRlRows * C);          // [nslots][out_per_slot]

    // global address of output o of this tile ([16 x C] probabilities, [16] entropies, [16 x C] risks), or null
    auto out_ptr = [&](int o) -> float* {
        if (o < kRlRows * C) return r0 * C + o < BC ? proba_sum + r0 * C + o : nullptr;
        if (o < kRlRows * C + kRlRows) return (ent_sum && r0 + (o - kRlRows * C) < B) ? ent_sum + r0 + (o - kRlRows * C) : nullptr;
        const int64_t i = r0 * C + (o - kRlRows * C - kRlRows);
        return i < BC ? risk_sum + i : nullptr;
    };
    // the first accumulator this thread will fold into is fetched now and consumed after the member loop
    float* const dst0 = (int)threadIdx.x < out_per_slot ? out_ptr(threadIdx.x) : nullptr;
    const float prev0 = dst0 ? *dst0 : 0.f;

    // which (quarter, float4) each of this lane's loads fetches, and the member range of that quarter's slot
    int ld_lo[NL], ld_cnt[NL];
    int64_t ld_base[NL];
#pragma unroll
    for (int t = 0; t < NL; ++t) {
        const int i = wl + 64 * t;
        const int kk = i / tile4;                        // quarter-wave whose tile this float4 belongs to
        const int q = wave * 4 + (kk < 4 ? kk : 3);
        const int off = i - kk * tile4;
        ld_lo[t] = q * base + (q < extra ? q : extra);
        const bool in_tile = kk < 4 && (r0 * C + 4 * (int64_t)off < BC);          // B*C % 4 == 0: no straddling
        ld_cnt[t] = in_tile ? base + (q < extra ? 1 : 0) : 0;
        ld_base[t] = r0 * C + 4 * (int64_t)off;
    }
    const int slot = wave * 4 + k;
    const bool tail_live = slot < extra;
    const int rounds = base + (extra ? 1 : 0);           // uniform across the block

    float acc_p[CP], acc_r[RISK ? CP : 1];
    float acc_e2 = 0.f;
#pragma unroll
    for (int c = 0; c < CP; ++c) acc_p[c] = 0.f;
#pragma unroll
    for (int c = 0; c < (RISK ? CP : 1); ++c) acc_r[c] = 0.f;
    const float* row = stage + (k * kRlRows + r) * C;

    for (int j0 = 0; j0 < rounds; j0 += kRlChunk) {
        float4 buf[kRlChunk][NL];
#pragma unroll
        for (int u = 0; u < kRlChunk; ++u) {
#pragma unroll
            for (int t = 0; t < NL; ++t) {
                const int j = j0 + u;
                const bool ok = j < ld_cnt[t];
                const float4 v = *reinterpret_cast<const float4*>(logits + (ok ? (ld_lo[t] + j) * BC + ld_base[t] : 0));
                buf[u][t] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);   // unconditional load from a valid address
            }
        }
#pragma unroll
        for (int u = 0; u < kRlChunk; ++u) {
            const int j = j0 + u;
            if (j >= rounds) break;                      // uniform
#pragma unroll
            for (int t = 0; t < NL; ++t) {
                const int i = wl + 64 * t;
                if ((EXACT && NL * 64 <= 16 * CP) || i < 4 * tile4) reinterpret_cast<float4*>(stage)[i] = buf[u][t];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (j < base) rl_member<CP, EXACT, RISK, false>(row, C, true, omg, goc, cost, acc_p, acc_r, acc_e2);
            else rl_member<CP, EXACT, RISK, true>(row, C, tail_live, omg, goc, cost, acc_p, acc_r, acc_e2);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();             // the next round's stage writes must not pass these reads
        }
    }

    if (smoothed) {                                      // sum_s ((1-g) p_s + g/C) = (1-g) sum_s p_s + n g/C
        const float ng = (float)(base + (tail_live ? 1 : 0)) * goc;
#pragma unroll
        for (int c = 0; c < CP; ++c) acc_p[c] = __builtin_fmaf(acc_p[c], omg, ng);
    }
    // partial sums -> LDS, then every thread folds whole outputs over the slots in slot order
    float* mine = part + slot * out_per_slot;
#pragma unroll
    for (int c = 0; c < CP; ++c)
        if (EXACT || c < C) {
            mine[r * C + c] = acc_p[c];
            if (RISK) mine[kRlRows * C + kRlRows + r * C + c] = acc_r[c];
        }
    mine[kRlRows * C + r] = -0.693147182464599609375f * acc_e2;
    __syncthreads();
    for (int o = threadIdx.x; o < out_per_slot; o += blockDim.x) {
        float* const dst = o == (int)threadIdx.x ? dst0 : out_ptr(o);
        if (!dst) continue;
        float a = o == (int)threadIdx.x ? prev0 : *dst;
        for (int q = 0; q < nslots; ++q) a += part[q * out_per_slot + o];
        *dst = a;
    }
}

// ---------------------------------------------------------------------------------------
// K4: HMC leapfrog sub-steps (call site URSABench/inference/hmc.py:71-75; hamiltorch
// arithmetic, parity unpinned) and sum-of-squares reductions.
__device__ __forceinline__ float block_sum(float v)
{
    __shared__ float part[kBlock / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = v + __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) part[wave] = v;
    __syncthreads();
    float t = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) t += part[w];
    }
    __syncthreads();
    return t;   // valid in thread 0
}

// Streaming form (no energy wanted): one float4 per thread like K1 (non-temporal past the Infinity Cache).
template <bool NT>
__global__ __launch_bounds__(kSBlock) void k_leapfrog_v(float* __restrict__ theta, float* __restrict__ mom,
                                                        const float* __restrict__ grad, int64_t n, float kick,
                                                        float drift, uint32_t flags)
{
    const bool do_kick = flags & URSA_LEAP_KICK, do_drift = flags & URSA_LEAP_DRIFT;
    const int64_t n4 = n >> 2;
    const int64_t i = (int64_t)blockIdx.x * kSBlock + threadIdx.x;
    if (i < n4) {
        float4 p = ld4<NT>(reinterpret_cast<const float4*>(mom) + i);
        if (do_kick) {
            const float4 g = ld4<NT>(reinterpret_cast<const float4*>(grad) + i);
            p.x = p.x + kick * g.x; p.y = p.y + kick * g.y; p.z = p.z + kick * g.z; p.w = p.w + kick * g.w;
            st4<NT>(reinterpret_cast<float4*>(mom) + i, p);
        }
        if (do_drift) {
            float4 t = ld4<NT>(reinterpret_cast<const float4*>(theta) + i);
            t.x = t.x + drift * p.x; t.y = t.y + drift * p.y; t.z = t.z + drift * p.z; t.w = t.w + drift * p.w;
            st4<NT>(reinterpret_cast<float4*>(theta) + i, t);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const int64_t j = (n4 << 2) + threadIdx.x;
        float p = mom[j];
        if (do_kick) { p = p + kick * grad[j]; mom[j] = p; }
        if (do_drift) theta[j] = theta[j] + drift * p;
    }
}

// Grid-stride form: also used when the kinetic energy is wanted (bounded number of block partials). NT: non-temporal
// accesses when the vectors touched exceed the Infinity Cache, as in the streaming form (round 2 left this form on
// plain accesses: kick + kinetic read 0.63 of the HBM peak at 2^26 elements against 0.78 for the streaming kick).
#ifndef URSA_LEAP_U
#define URSA_LEAP_U 4
#endif
constexpr int kLeapU = URSA_LEAP_U;
template <bool VEC, bool NT>
__global__ __launch_bounds__(kBlock) void k_leapfrog(float* __restrict__ theta, float* __restrict__ mom,
                                                     const float* __restrict__ grad, int64_t n, float kick,
                                                     float drift, uint32_t flags, float* __restrict__ ws)
{
    const bool do_kick = flags & URSA_LEAP_KICK, do_drift = flags & URSA_LEAP_DRIFT;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    float ke = 0.f;
    if (VEC) {
        const int64_t n4 = n >> 2;
        const float4* mv = reinterpret_cast<const float4*>(mom);
        const float4* gv = reinterpret_cast<const float4*>(grad);
        const float4* tv = reinterpret_cast<const float4*>(theta);
        // Grid-stride over BATCHES of kLeapU adjacent 4 KB tiles: all loads of a batch in flight before the first use, and the
        // workgroups of the launch together still sweep ONE contiguous window of the vectors (round 4: one float4 per vector in
        // flight per thread, 0.70 of the HBM peak for kick + kinetic against 0.80 for the streaming kick. Round 5 A/B, 2^26
        // elements: batches whose members sit a whole grid stride - 8 MB - apart 0.66; one contiguous span per workgroup, i.e.
        // 2,048 separate streams, 0.66; this form: see DESIGN.md).
        const int64_t hi = n4;
        for (int64_t i = (int64_t)blockIdx.x * (kLeapU * kBlock) + threadIdx.x; i < hi; i += (int64_t)gridDim.x * (kLeapU * kBlock)) {
            float4 p[kLeapU], g[kLeapU], t[kLeapU];
#pragma unroll
            for (int u = 0; u < kLeapU; ++u) if (i + u * kBlock < hi) p[u] = ld4<NT>(mv + i + u * kBlock);
            if (do_kick) {
#pragma unroll
                for (int u = 0; u < kLeapU; ++u) if (i + u * kBlock < hi) g[u] = ld4<NT>(gv + i + u * kBlock);
            }
            if (do_drift) {
#pragma unroll
                for (int u = 0; u < kLeapU; ++u) if (i + u * kBlock < hi) t[u] = ld4<NT>(tv + i + u * kBlock);
            }
#pragma unroll
            for (int u = 0; u < kLeapU; ++u) {
                if (i + u * kBlock < hi) {
                    if (do_kick) {
                        p[u].x = p[u].x + kick * g[u].x; p[u].y = p[u].y + kick * g[u].y; p[u].z = p[u].z + kick * g[u].z; p[u].w = p[u].w + kick * g[u].w;
                        st4<NT>(reinterpret_cast<float4*>(mom) + i + u * kBlock, p[u]);
                    }
                    if (do_drift) {
                        t[u].x = t[u].x + drift * p[u].x; t[u].y = t[u].y + drift * p[u].y; t[u].z = t[u].z + drift * p[u].z; t[u].w = t[u].w + drift * p[u].w;
                        st4<NT>(reinterpret_cast<float4*>(theta) + i + u * kBlock, t[u]);
                    }
                    ke += (p[u].x * p[u].x + p[u].y * p[u].y) + (p[u].z * p[u].z + p[u].w * p[u].w);
                }
            }
        }
        if (tid < (n & 3)) {
            const int64_t i = (n4 << 2) + tid;
            float p = mom[i];
            if (do_kick) { p = p + kick * grad[i]; mom[i] = p; }
            if (do_drift) theta[i] = theta[i] + drift * p;
            ke += p * p;
        }
    } else {
        for (int64_t i = tid; i < n; i += stride) {
            float p = mom[i];
            if (do_kick) { p = p + kick * grad[i]; mom[i] = p; }
            if (do_drift) theta[i] = theta[i] + drift * p;
            ke += p * p;
        }
    }
    if (ws) {
        const float t = block_sum(ke);
        if (threadIdx.x == 0) ws[blockIdx.x] = t;
    }
}

template <bool VEC>
__global__ __launch_bounds__(kBlock) void k_sumsq(const float* __restrict__ x, int64_t n, float* __restrict__ ws)
{
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    float acc = 0.f;
    if (VEC) {
        const int64_t n4 = n >> 2;
        const float4* __restrict__ xv = reinterpret_cast<const float4*>(x);
        const int64_t hi = n4;                                                  // grid-stride over batches of kLeapU adjacent tiles (see k_leapfrog)
        for (int64_t i = (int64_t)blockIdx.x * (kLeapU * kBlock) + threadIdx.x; i < hi; i += (int64_t)gridDim.x * (kLeapU * kBlock)) {
            float4 v[kLeapU];
#pragma unroll
            for (int u = 0; u < kLeapU; ++u) if (i + u * kBlock < hi) v[u] = xv[i + u * kBlock];
#pragma unroll
            for (int u = 0; u < kLeapU; ++u)
                if (i + u * kBlock < hi) acc += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
        }
        if (tid < (n & 3)) { const float v = x[(n4 << 2) + tid]; acc += v * v; }
    } else {
        for (int64_t i = tid; i < n; i += stride) { const float v = x[i]; acc += v * v; }
    }
    const float t = block_sum(acc);
    if (threadIdx.x == 0) ws[blockIdx.x] = t;
}

// out[0] += scale * sum_{b < nblocks} ws[b], summed in a fixed order by one block.
__global__ __launch_bounds__(kBlock) void k_finish_sum(const float* __restrict__ ws, int nblocks, float scale,
                                                       float* __restrict__ out)
{
    float acc = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += kBlock) acc += ws[b];
    const float t = block_sum(acc);
    if (threadIdx.x == 0) out[0] = out[0] + scale * t;
}

inline int launch_status() { return (int)hipGetLastError(); }

template <bool MOM>
int launch_step(NoiseSrc ns, bool vec, bool nt, hipStream_t st, float* theta, float* grad, float* mom,
                const float* eps, float* snapshot, int64_t n, const StepScalars& s)
{
#define URSA_LAUNCH(K, G, B) hipLaunchKernelGGL(K, dim3(G), dim3(B), 0, st, theta, grad, mom, eps, snapshot, n, s)
    if (vec) {
        const int grid = sgrid(n >> 2);
        if (nt) {
            if (ns == kNoiseOff) URSA_LAUNCH((k_sgmcmc_step<MOM, kNoiseOff, true>), grid, kSBlock);
            else if (ns == kNoisePtr) URSA_LAUNCH((k_sgmcmc_step<MOM, kNoisePtr, true>), grid, kSBlock);
            else URSA_LAUNCH((k_sgmcmc_step<MOM, kNoisePhilox, true>), grid, kSBlock);
        } else {
            if (ns == kNoiseOff) URSA_LAUNCH((k_sgmcmc_step<MOM, kNoiseOff, false>), grid, kSBlock);
            else if (ns == kNoisePtr) URSA_LAUNCH((k_sgmcmc_step<MOM, kNoisePtr, false>), grid, kSBlock);
            else URSA_LAUNCH((k_sgmcmc_step<MOM, kNoisePhilox, false>), grid, kSBlock);
        }
    } else {
        const int grid = grid_for(n, kBlock);
        if (ns == kNoiseOff) URSA_LAUNCH((k_sgmcmc_step_scalar<MOM, kNoiseOff>), grid, kBlock);
        else if (ns == kNoisePtr) URSA_LAUNCH((k_sgmcmc_step_scalar<MOM, kNoisePtr>), grid, kBlock);
        else URSA_LAUNCH((k_sgmcmc_step_scalar<MOM, kNoisePhilox>), grid, kBlock);
    }
#undef URSA_LAUNCH
    return launch_status();
}

}  // namespace

// =======================================================================================
// C ABI
extern "C" {

int ursa_abi_version(void) { return URSA_ABI_VERSION; }

const char* ursa_strerror(int code)
{
    switch (code) {
    case URSA_OK: return "ok";
    case URSA_ENULL: return "required pointer is NULL";
    case URSA_ESIZE: return "invalid size";
    case URSA_EALIGN: return "pointer is not 4-byte aligned";
    case URSA_EFLAGS: return "invalid flags or flag/pointer combination";
    case URSA_EVALUE: return "scalar argument out of range";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown ursa error";
    }
}

int ursa_sgmcmc_step_f32(float* theta, float* grad, float* mom, const float* eps, float* snapshot, int64_t n,
                         float lr, float mu, float c_wd, float c_noise, float n_train, uint64_t seed,
                         uint64_t step, uint32_t flags, ursa_stream_t stream)
{
    if (n < 0 || n > kMaxElems) return URSA_ESIZE;
    if (flags & ~URSA_STEP_ALLFLAGS) return URSA_EFLAGS;
    if ((flags & URSA_STEP_SGD) && (flags & URSA_STEP_NOISE)) return URSA_EFLAGS;
    if (n == 0) return URSA_OK;
    if (!theta || !grad) return URSA_ENULL;
    if (mu != 0.0f && !mom) return URSA_ENULL;
    if (!aligned4(theta) || !aligned4(grad) || !aligned4(mom) || !aligned4(eps) || !aligned4(snapshot))
        return URSA_EALIGN;
    const bool noise = flags & URSA_STEP_NOISE;
    const NoiseSrc ns = !noise ? kNoiseOff : (eps ? kNoisePtr : kNoisePhilox);
    const bool vec = aligned16(theta) && aligned16(grad) && aligned16(mom) && aligned16(eps) && aligned16(snapshot);
    const StepScalars s{lr, mu, c_wd, c_noise, n_train, flags, seed, step};
    const bool nt = n * (int64_t)(mu != 0.0f ? 12 : 8) > kNtBytes;       // theta + grad (+ mom) past the cache
    hipStream_t st = (hipStream_t)stream;
    return mu != 0.0f ? launch_step<true>(ns, vec, nt, st, theta, grad, mom, eps, snapshot, n, s)
                      : launch_step<false>(ns, vec, nt, st, theta, grad, mom, eps, snapshot, n, s);
}

// Threads per workgroup of a control-block launch: 512, one float4 per thread (134 workgroups per PreResNet-20 chain).
// With one-address tickets multi-chain launches were better off with 1,024 (half the tickets); on the ticket tree the
// ticket count no longer matters (tools/k1_ctl_bench.py sweeps it through the debug override URSA_CTL_BLOCK).
inline int ctl_block(int64_t n4, int n_chains)
{
    static const int forced = [] {
        const char* e = knob("URSA_CTL_BLOCK");
        const int v = e && e[0] ? atoi(e) : 0;
        return (v == 64 || v == 128 || v == 256 || v == 512 || v == 1024) ? v : 0;
    }();
    if (forced) return forced;
    (void)n4; (void)n_chains;
    return kSBlock;
}

int ursa_sgmcmc_step_multi_f32(float* theta, float* grad, float* mom, const float* eps, float* snapshot,
                               int64_t n_per_chain, int32_t n_chains, int64_t chain_stride, ursa_step_ctl* ctl,
                               ursa_stream_t stream)
{
    const int64_t n = n_per_chain;
    if (n < 0 || n > kMaxElems || n_chains < 0 || n_chains > 65535) return URSA_ESIZE;
    if (n == 0 || n_chains == 0) return URSA_OK;
    if (n_chains > 1 && (chain_stride < n || (chain_stride & 3))) return URSA_ESIZE;
    if (!theta || !grad || !mom || !ctl) return URSA_ENULL;   // mom always required: mu lives on the device
    if (!(aligned16(theta) && aligned16(grad) && aligned16(mom) && aligned16(eps) && aligned16(snapshot)))
        return URSA_EALIGN;                                    // the replayable form is float4-only
    if (reinterpret_cast<uintptr_t>(ctl) & 127u) return URSA_EALIGN;   // ticket counters must sit on separate 128-byte lines
    const int block = ctl_block(n >> 2, n_chains);
    int64_t gx = ((n >> 2) + block - 1) / block;
    if (gx < 1) gx = 1;
    const dim3 grid((unsigned)gx, (unsigned)n_chains);
    const bool nt = n * 12ll * n_chains > kNtBytes;
#define URSA_LAUNCH(K) hipLaunchKernelGGL(K, grid, dim3(block), 0, (hipStream_t)stream, theta, grad, mom, eps, snapshot, n, chain_stride, ctl)
    if (n_chains > 1) { if (nt) URSA_LAUNCH((k_sgmcmc_step_ctl<true, true>)); else URSA_LAUNCH((k_sgmcmc_step_ctl<false, true>)); }
    else              { if (nt) URSA_LAUNCH((k_sgmcmc_step_ctl<true, false>)); else URSA_LAUNCH((k_sgmcmc_step_ctl<false, false>)); }
#undef URSA_LAUNCH
    return launch_status();
}

int ursa_sgmcmc_step_ctl_f32(float* theta, float* grad, float* mom, const float* eps, float* snapshot,
                             int64_t n, ursa_step_ctl* ctl, ursa_stream_t stream)
{
    return ursa_sgmcmc_step_multi_f32(theta, grad, mom, eps, snapshot, n, 1, 0, ctl, stream);
}

int ursa_step_ctl_advance(ursa_step_ctl* ctl, int32_t n_ctl, ursa_stream_t stream)
{
    if (!ctl) return URSA_ENULL;
    if (reinterpret_cast<uintptr_t>(ctl) & 7u) return URSA_EALIGN;
    if (n_ctl < 0) return URSA_ESIZE;
    if (n_ctl == 0) return URSA_OK;
    hipLaunchKernelGGL(k_step_ctl_advance, dim3((n_ctl + 63) / 64), dim3(64), 0, (hipStream_t)stream, ctl, (int)n_ctl);
    return launch_status();
}

int ursa_philox_normal_f32(float* out, int64_t n, uint64_t seed, uint64_t step, ursa_stream_t stream)
{
    if (n < 0 || n > kMaxElems) return URSA_ESIZE;
    if (n == 0) return URSA_OK;
    if (!out) return URSA_ENULL;
    if (!aligned4(out)) return URSA_EALIGN;
    hipLaunchKernelGGL(k_philox_normal, dim3(sgrid((n + 3) >> 2)), dim3(kSBlock), 0, (hipStream_t)stream, out, n,
                       seed, step);
    return launch_status();
}

int ursa_selftest_rng_f32(uint64_t* mismatches, ursa_stream_t stream)
{
    if (!mismatches) return URSA_ENULL;
    if (reinterpret_cast<uintptr_t>(mismatches) & 7u) return URSA_EALIGN;
    hipLaunchKernelGGL(k_selftest_rng, dim3(1u << 16), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<unsigned long long*>(mismatches));
    return launch_status();
}

int ursa_swag_collect_f32(float* mean, float* sq, const float* w, int64_t n, float decay, float denom,
                          ursa_stream_t stream)
{
    if (n < 0 || n > kMaxElems) return URSA_ESIZE;
    if (n == 0) return URSA_OK;
    if (!mean || !sq || !w) return URSA_ENULL;
    if (!aligned4(mean) || !aligned4(sq) || !aligned4(w)) return URSA_EALIGN;
    const bool vec = aligned16(mean) && aligned16(sq) && aligned16(w);
    hipStream_t st = (hipStream_t)stream;
    if (vec && n * 12ll > kNtBytes)
        hipLaunchKernelGGL(k_swag_collect_v<true>, dim3(sgrid(n >> 2)), dim3(kSBlock), 0, st, mean, sq, w, n, decay,
                           denom);
    else if (vec)
        hipLaunchKernelGGL(k_swag_collect_v<false>, dim3(sgrid(n >> 2)), dim3(kSBlock), 0, st, mean, sq, w, n, decay,
                           denom);
    else
        hipLaunchKernelGGL(k_swag_collect_s, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, st, mean, sq, w, n, decay,
                           denom);
    return launch_status();
}

int ursa_swag_draw_f32(float* theta_out, const float* mean, const float* sq, const float* eps, int64_t n,
                       float var_clamp, float scale, uint64_t seed, uint64_t draw, ursa_stream_t stream)
{
    if (n < 0 || n > kMaxElems) return URSA_ESIZE;
    if (n == 0) return URSA_OK;
    if (!theta_out || !mean || !sq) return URSA_ENULL;
    if (!aligned4(theta_out) || !aligned4(mean) || !aligned4(sq) || !aligned4(eps)) return URSA_EALIGN;
    const bool vec = aligned16(theta_out) && aligned16(mean) && aligned16(sq) && aligned16(eps);
    hipStream_t st = (hipStream_t)stream;
    const bool nt = n * 12ll > kNtBytes;
    const dim3 grid(vec ? sgrid(n >> 2) : grid_for(n, kBlock)), block(vec ? kSBlock : kBlock);
#define URSA_LAUNCH(K) hipLaunchKernelGGL(K, grid, block, 0, st, theta_out, mean, sq, eps, n, var_clamp, scale, seed, draw)
    if (vec && nt) { if (eps) URSA_LAUNCH((k_swag_draw_v<false, true>)); else URSA_LAUNCH((k_swag_draw_v<true, true>)); }
    else if (vec)  { if (eps) URSA_LAUNCH((k_swag_draw_v<false, false>)); else URSA_LAUNCH((k_swag_draw_v<true, false>)); }
    else           { if (eps) URSA_LAUNCH((k_swag_draw_s<false>)); else URSA_LAUNCH((k_swag_draw_s<true>)); }
#undef URSA_LAUNCH
    return launch_status();
}

int ursa_swag_std_f32(float* std_out, const float* mean, const float* sq, int64_t n, float var_clamp, float scale,
                      ursa_stream_t stream)
{
    if (n < 0 || n > kMaxElems) return URSA_ESIZE;
    if (n == 0) return URSA_OK;
    if (!std_out || !mean || !sq) return URSA_ENULL;
    if (!aligned4(std_out) || !aligned4(mean) || !aligned4(sq)) return URSA_EALIGN;
    const bool vec = aligned16(std_out) && aligned16(mean) && aligned16(sq);
    hipStream_t st = (hipStream_t)stream;
    if (vec && n * 12ll > kNtBytes)
        hipLaunchKernelGGL(k_swag_std_v<true>, dim3(sgrid(n >> 2)), dim3(kSBlock), 0, st, std_out, mean, sq, n, var_clamp, scale);
    else if (vec)
        hipLaunchKernelGGL(k_swag_std_v<false>, dim3(sgrid(n >> 2)), dim3(kSBlock), 0, st, std_out, mean, sq, n, var_clamp, scale);
    else
        hipLaunchKernelGGL(k_swag_std_s, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, st, std_out, mean, sq, n, var_clamp, scale);
    return launch_status();
}

int ursa_swag_draw_std_f32(float* theta_out, const float* mean, const float* std, const float* eps, int64_t n,
                           uint64_t seed, uint64_t draw, ursa_stream_t stream)
{
    if (n < 0 || n > kMaxElems) return URSA_ESIZE;
    if (n == 0) return URSA_OK;
    if (!theta_out || !mean || !std) return URSA_ENULL;
    if (!aligned4(theta_out) || !aligned4(mean) || !aligned4(std) || !aligned4(eps)) return URSA_EALIGN;
    const bool vec = aligned16(theta_out) && aligned16(mean) && aligned16(std) && aligned16(eps);
    hipStream_t st = (hipStream_t)stream;
    const bool nt = n * 12ll > kNtBytes;
    const dim3 grid(vec ? sgrid(n >> 2) : grid_for(n, kBlock)), block(vec ? kSBlock : kBlock);
#define URSA_LAUNCH(K) hipLaunchKernelGGL(K, grid, block, 0, st, theta_out, mean, std, eps, n, seed, draw)
    if (vec && nt) { if (eps) URSA_LAUNCH((k_swag_draw_std_v<false, true>)); else URSA_LAUNCH((k_swag_draw_std_v<true, true>)); }
    else if (vec)  { if (eps) URSA_LAUNCH((k_swag_draw_std_v<false, false>)); else URSA_LAUNCH((k_swag_draw_std_v<true, false>)); }
    else           { if (eps) URSA_LAUNCH((k_swag_draw_std_s<false>)); else URSA_LAUNCH((k_swag_draw_std_s<true>)); }
#undef URSA_LAUNCH
    return launch_status();
}

int ursa_bma_accumulate_f32(const float* logits, float* proba_sum, float* ent_sum, float* risk_sum,
                            const float* cost, int32_t S, int64_t B, int32_t C, float one_minus_gamma,
                            float gamma_over_c, uint32_t flags, ursa_stream_t stream)
{
    if (S < 0 || B < 0) return URSA_ESIZE;
    if (C < 1 || C > URSA_BMA_MAX_CLASSES) return URSA_EVALUE;
    if (flags & ~URSA_BMA_SMOOTHED) return URSA_EFLAGS;
    if (S == 0 || B == 0) return URSA_OK;
    if (!logits || !proba_sum) return URSA_ENULL;
    if ((risk_sum == nullptr) != (cost == nullptr)) return URSA_EFLAGS;
    if (!aligned4(logits) || !aligned4(proba_sum) || !aligned4(ent_sum) || !aligned4(risk_sum) || !aligned4(cost))
        return URSA_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    // gamma = 0 (allowed): an underflowing probability must add 0 ln 0 = 0 to the entropy. With the smallest normal
    // number in place of 0 every q is positive, log2 q is finite and q log2 q < 1e-35: no per-class guard in the kernels.
    if (!(gamma_over_c > 0.0f)) gamma_over_c = 1.17549435e-38f;
    // Few classes and 16-byte aligned member tiles: the row-per-lane kernel (see k_bma_rowlane).
    const bool tiles_aligned = aligned16(logits) && ((B * (int64_t)C) % 4 == 0);
    // (11 <= C <= 15 with a cost matrix spills registers in the masked 16-class body: lane-group kernel instead)
    if (C <= 16 && tiles_aligned && !(risk_sum && C > 10 && C < 16) && !getenv_flag("URSA_BMA_NO_ROWLANE")) {
        // 4 waves (16 member slots) from 12 members up: measured best at S = 20 (5.7 vs 6.2 us with 2 waves) and at
        // S = 50 (8.0 vs 9.0 us with 8 waves); small ensembles keep small blocks (tools/k5_bench.py rowlane_wavesN)
        int W = S >= 12 ? 4 : S >= 5 ? 2 : 1;
        if (const char* w = knob("URSA_BMA_RL_WAVES")) {          // knob: tools/k5_bench.py sweeps
            const int v = atoi(w);
            if (v >= 1 && v <= 8) W = v;
        }
        const int cp = C <= 4 ? 4 : C <= 8 ? 8 : C == 10 ? 10 : 16;
        const bool exact = C == cp;
        auto lds_for = [&](int w) {
            return sizeof(float) * ((size_t)w * 4 * kRlRows * C + (size_t)4 * w * (kRlRows * C * (risk_sum ? 2 : 1) + kRlRows));
        };
        while (W > 1 && lds_for(W) > 64 * 1024) W >>= 1;      // the default dynamic-LDS limit (a debug override could exceed it: ADVICE r2)
        const size_t lds = lds_for(W);
        const dim3 grid((unsigned)bma_grid(B, kRlRows)), block(64 * W);
#define URSA_RL2(CPV, EX, RK)                                                                                   \
        hipLaunchKernelGGL((k_bma_rowlane<CPV, EX, RK>), grid, block, lds, st, logits, proba_sum, ent_sum,      \
                           risk_sum, cost, (int)S, B, (int)C, one_minus_gamma, gamma_over_c, flags)
#define URSA_RL(CPV)                                                                                            \
        do {                                                                                                    \
            if (risk_sum) { if (exact) URSA_RL2(CPV, true, true); else URSA_RL2(CPV, false, true); }            \
            else { if (exact) URSA_RL2(CPV, true, false); else URSA_RL2(CPV, false, false); }                   \
        } while (0)
        if (cp == 4) URSA_RL(4);
        else if (cp == 8) URSA_RL(8);
        else if (cp == 10) URSA_RL(10);
        else URSA_RL(16);
#undef URSA_RL2
#undef URSA_RL
        return launch_status();
    }
#define URSA_LAUNCH_WE(G, EPL, V4, W, EARLY)                                                                    \
    hipLaunchKernelGGL((k_bma_accumulate<G, EPL, false, V4, false, W, EARLY>), dim3(bma_grid(B, 64 / G)), dim3(64 * W), 0, st, \
                       logits, proba_sum, ent_sum, risk_sum, cost, (int)S, B, (int)C, one_minus_gamma,          \
                       gamma_over_c, flags)
#define URSA_LAUNCH_V(G, EPL, V4)                                                                               \
    do {                                                                                                        \
        if (risk_sum)                                                                                           \
            hipLaunchKernelGGL((k_bma_accumulate<G, EPL, true, V4>), dim3(bma_grid(B, 64 / G)), dim3(kBlock), 0, st, \
                               logits, proba_sum, ent_sum, risk_sum, cost, (int)S, B, (int)C, one_minus_gamma,  \
                               gamma_over_c, flags);                                                            \
        else if (V4 && EPL >= 8 && bma_prefetch())                                                              \
            hipLaunchKernelGGL((k_bma_accumulate<G, EPL, false, V4, true>), dim3(bma_grid(B, 64 / G)), dim3(kBlock), 0, st, \
                               logits, proba_sum, ent_sum, risk_sum, cost, (int)S, B, (int)C, one_minus_gamma,  \
                               gamma_over_c, flags);                                                            \
        else if (V4) {                                                                                          \
            const BmaForm f = bma_form((B + 64 / G - 1) / (64 / G), (int)S, EPL);                               \
            switch (f.waves * 2 + (f.early ? 1 : 0)) {                                                          \
            case 2: URSA_LAUNCH_WE(G, EPL, V4, 1, false); break;                                                \
            case 3: URSA_LAUNCH_WE(G, EPL, V4, 1, true); break;                                                 \
            case 4: URSA_LAUNCH_WE(G, EPL, V4, 2, false); break;                                                \
            case 5: URSA_LAUNCH_WE(G, EPL, V4, 2, true); break;                                                 \
            case 9: URSA_LAUNCH_WE(G, EPL, V4, 4, true); break;                                                 \
            case 16: URSA_LAUNCH_WE(G, EPL, V4, 8, false); break;                                               \
            case 17: URSA_LAUNCH_WE(G, EPL, V4, 8, true); break;                                                \
            default: URSA_LAUNCH_WE(G, EPL, V4, 4, false); break;                                               \
            }                                                                                                   \
        }                                                                                                       \
        else                                                                                                    \
            hipLaunchKernelGGL((k_bma_accumulate<G, EPL, false, V4>), dim3(bma_grid(B, 64 / G)), dim3(kBlock), 0, st, \
                               logits, proba_sum, ent_sum, risk_sum, cost, (int)S, B, (int)C, one_minus_gamma,  \
                               gamma_over_c, flags);                                                            \
    } while (0)
#define URSA_LAUNCH(G, EPL) URSA_LAUNCH_V(G, EPL, false)
    // Rows are owned by 16-lane groups whenever C <= 256: the three row reductions are then pure DPP
    // (no cross-row step) and one DPP instruction serves the 4 rows of the wave at once; more classes per
    // lane also means more independent exp/log work per lane. With 16-byte aligned rows (C % 4 == 0) every
    // lane fetches its classes as float4 (four consecutive classes per load) instead of 4-byte strided loads.
    const bool rows_aligned = aligned16(logits) && (C % 4 == 0) && !getenv_flag("URSA_BMA_NO_V4");
    if (rows_aligned && C > 16 && C <= 32) URSA_LAUNCH_V(8, 4, true);          // 8 float4 per row: 8 lanes, 8 rows per wave
    else if (rows_aligned && C > 32 && C <= 64) URSA_LAUNCH_V(16, 4, true);
    else if (rows_aligned && C > 64 && C <= 128) URSA_LAUNCH_V(16, 8, true);
    else if (rows_aligned && C > 128 && C <= 256) URSA_LAUNCH_V(16, 16, true);
    else if (C <= 4) URSA_LAUNCH(4, 1);
    else if (C <= 8) URSA_LAUNCH(8, 1);
    else if (C <= 16) URSA_LAUNCH(16, 1);
    else if (C <= 32) URSA_LAUNCH(16, 2);
    else if (C <= 64) URSA_LAUNCH(16, 4);
    else if (C <= 128) URSA_LAUNCH(16, 8);
    else if (C <= 256) URSA_LAUNCH(16, 16);
    else if (C <= 512) URSA_LAUNCH(64, 8);
    else URSA_LAUNCH(64, 16);
#undef URSA_LAUNCH_V
#undef URSA_LAUNCH_WE
#undef URSA_LAUNCH
    return launch_status();
}

int ursa_leapfrog_f32(float* theta, float* mom, const float* grad, int64_t n, float kick_coef, float step_size,
                      float inv_mass, uint32_t flags, float* kinetic_out, float* ws, ursa_stream_t stream)
{
    if (n < 0 || n > kMaxElems) return URSA_ESIZE;
    if (flags & ~(URSA_LEAP_KICK | URSA_LEAP_DRIFT)) return URSA_EFLAGS;
    if (n == 0) return URSA_OK;
    if (!mom) return URSA_ENULL;
    if ((flags & URSA_LEAP_KICK) && !grad) return URSA_ENULL;
    if ((flags & URSA_LEAP_DRIFT) && !theta) return URSA_ENULL;
    if (kinetic_out && !ws) return URSA_ENULL;
    if (!aligned4(theta) || !aligned4(mom) || !aligned4(grad) || !aligned4(kinetic_out) || !aligned4(ws))
        return URSA_EALIGN;
    const bool vec = aligned16(theta) && aligned16(mom) && aligned16(grad);
    int grid = vec ? grid_for(n >> 2, kBlock) : grid_for(n, kBlock);
    if (grid > URSA_REDUCE_WS_FLOATS) grid = URSA_REDUCE_WS_FLOATS;
    hipStream_t st = (hipStream_t)stream;
    float* wsp = kinetic_out ? ws : nullptr;
    const float drift = step_size * inv_mass;
    if (vec && !kinetic_out && n * 12ll > kNtBytes)
        hipLaunchKernelGGL(k_leapfrog_v<true>, dim3(sgrid(n >> 2)), dim3(kSBlock), 0, st, theta, mom, grad, n,
                           kick_coef, drift, flags);
    else if (vec && !kinetic_out)
        hipLaunchKernelGGL(k_leapfrog_v<false>, dim3(sgrid(n >> 2)), dim3(kSBlock), 0, st, theta, mom, grad, n,
                           kick_coef, drift, flags);
    else if (vec && n * 4ll * (1 + ((flags & URSA_LEAP_KICK) ? 1 : 0) + ((flags & URSA_LEAP_DRIFT) ? 1 : 0)) > kNtBytes)
        hipLaunchKernelGGL((k_leapfrog<true, true>), dim3(grid), dim3(kBlock), 0, st, theta, mom, grad, n, kick_coef, drift,
                           flags, wsp);
    else if (vec)
        hipLaunchKernelGGL((k_leapfrog<true, false>), dim3(grid), dim3(kBlock), 0, st, theta, mom, grad, n, kick_coef, drift,
                           flags, wsp);
    else
        hipLaunchKernelGGL((k_leapfrog<false, false>), dim3(grid), dim3(kBlock), 0, st, theta, mom, grad, n, kick_coef,
                           drift, flags, wsp);
    if (kinetic_out)
        hipLaunchKernelGGL(k_finish_sum, dim3(1), dim3(kBlock), 0, st, ws, grid, 0.5f * inv_mass, kinetic_out);
    return launch_status();
}

int ursa_sumsq_f32(const float* x, int64_t n, float* out, float* ws, ursa_stream_t stream)
{
    if (n < 0 || n > kMaxElems) return URSA_ESIZE;
    if (!out || !ws) return URSA_ENULL;
    if (n == 0) return URSA_OK;
    if (!x) return URSA_ENULL;
    if (!aligned4(x) || !aligned4(out) || !aligned4(ws)) return URSA_EALIGN;
    const bool vec = aligned16(x);
    int grid = vec ? grid_for(n >> 2, kBlock) : grid_for(n, kBlock);
    if (grid > URSA_REDUCE_WS_FLOATS) grid = URSA_REDUCE_WS_FLOATS;
    hipStream_t st = (hipStream_t)stream;
    if (vec) hipLaunchKernelGGL(k_sumsq<true>, dim3(grid), dim3(kBlock), 0, st, x, n, ws);
    else hipLaunchKernelGGL(k_sumsq<false>, dim3(grid), dim3(kBlock), 0, st, x, n, ws);
    hipLaunchKernelGGL(k_finish_sum, dim3(1), dim3(kBlock), 0, st, ws, grid, 1.0f, out);
    return launch_status();
}

}  // extern "C"
