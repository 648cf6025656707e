// Experiment: where a workgroup of K6's held forward spends its time (the chunk sits in registers from the first load to
// the last store). Includes csrc/ursa_bn.hip itself (its kernels live in an anonymous namespace) and re-times a copy of
// k_bn_fwd_held<true,false,EPT,0,true> (registers only; TL_BLOCK_THREADS x TL_EPT) with wall_clock64 stamps between the phases; argv: [stagger groups] [stagger sleep 20|40|80]
// (delaying the first residency's channels to de-phase loads and stores: measured, no gain).
// Measured at [1024,64,32,32] (r04): ticket 3.3 us, load + reduce 9.4, publish 2.0, wait 8.4, merge 2.6, apply 3.9 = 29.5 us per
// workgroup, 4 residencies of 1,024 workgroups -> 139 us; the chunk is held in registers all that time.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I. -o /tmp/bn_tl tools/exp/bn_held_timeline.hip && /tmp/bn_tl
#include "../../ursabench_amd/csrc/ursa_bn.hip"
#include <cstdio>
#include <vector>
#include <algorithm>

#ifndef TL_BLOCK_THREADS
#define TL_BLOCK_THREADS 512       // the shipped forward shape's workgroup (round 4's first shape: -DTL_BLOCK_THREADS=256 -DTL_EPT=16)
#endif
#ifndef TL_EPT
#define TL_EPT 32
#endif
namespace {
constexpr int TL_BLOCK = TL_BLOCK_THREADS;
constexpr int EPT = TL_EPT;
template <int stagger_sleep>
__global__ __launch_bounds__(TL_BLOCK) void k_tl(const float* __restrict__ x, float* __restrict__ y, bn_u64* slots_base, BnSync* sync,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, BnGeom g, int S,
                                                     long long* stamps /* [grid][8] */, int stagger_groups)
{
    __shared__ double sh[2 * TL_BLOCK / 64];
    __shared__ float shf[2];
    __shared__ int sh_item[2];
    long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        int c0 = -1, sp0 = 0;
        bn_take_item(sync, g.C, S, c0, sp0);
        sh_item[0] = c0; sh_item[1] = sp0;
    }
    __syncthreads();
    const int c = sh_item[0], sp = sh_item[1];
    if (c < 0) return;
    const uint32_t tk = (uint32_t)(c * S + sp);
    if (c < stagger_groups) for (int k = 0; k < c; ++k) __builtin_amdgcn_s_sleep(stagger_sleep);     // de-phase the first residency
    const float4* __restrict__ xv = reinterpret_cast<const float4*>(x);
    float4* __restrict__ yv = reinterpret_cast<float4*>(y);
    const int lo = sp * g.chunk;
    const int hi = lo + g.chunk < (int)g.per_ch ? lo + g.chunk : (int)g.per_ch;
    long long t1 = wall_clock64();
    float4 v[EPT];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = lo + threadIdx.x + u * TL_BLOCK;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < hi) v[u] = bn_ld<true>(xv + bn_off32(g, c, i));
    }
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int u = 0; u < EPT; ++u)
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double d = (double)comp(v[u], k); s1 += d; s2 = fma(d, d, s2); }
    bn_block_sum2_n<TL_BLOCK>(s1, s2, sh);
    long long t2 = wall_clock64();
    bn_u64* slots = slots_base + (int64_t)c * kBnMaxSplit * 2;
    if (threadIdx.x == 0) bn_publish(slots + 2 * sp, s1, s2);
    long long t3 = wall_clock64(), t4 = 0;
    uint32_t left = 0;
    if (threadIdx.x < 64) {
        double a, b;
        bn_gather(slots, S, &sync->err, a, b);
        t4 = wall_clock64();
        left = bn_leave_issue(sync, c);
        if (threadIdx.x == 0) {
            const double n = (double)g.per_ch * 4.0;
            const double mean = a / n;
            double var = b / n - mean * mean;
            const float invstd = (float)(1.0 / sqrt(var + 1e-5));
            const float alpha = invstd * gamma[c];
            shf[0] = alpha;
            shf[1] = fmaf(-(float)mean, alpha, beta[c]);
        }
    }
    __syncthreads();
    long long t5 = wall_clock64();
    const float scale = shf[0], shift = shf[1];
#pragma unroll
    for (int u = 0; u < EPT; ++u) {
        const int i = lo + threadIdx.x + u * TL_BLOCK;
        if (i < hi) {
            float4 r;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float t = fmaf(comp(v[u], k), scale, shift); setc(r, k, bn_relu_fwd(t)); }
            bn_st<true>(yv + bn_off32(g, c, i), r);
        }
    }
    if (threadIdx.x < 64) bn_leave_finish(sync, slots, c, g.C, S, left);
    long long t6 = wall_clock64();
    if (threadIdx.x == 0) {
        long long* s = stamps + (int64_t)tk * 8;
        s[0] = t0; s[1] = t1; s[2] = t2; s[3] = t3; s[4] = t4; s[5] = t5; s[6] = t6;
    }
}
}  // namespace

int main(int argc, char** argv)
{
    const int64_t N = 1024, C = 64, HW = 1024;
    const int64_t e = N * C * HW;
    float *x, *y, *gam, *bet, *ws;
    hipMalloc((void**)&x, e * 4); hipMalloc((void**)&y, e * 4); hipMalloc((void**)&gam, C * 4); hipMalloc((void**)&bet, C * 4);
    hipMalloc((void**)&ws, URSA_BN_WS_FLOATS(C) * 4);
    hipMemset(ws, 0, URSA_BN_WS_FLOATS(C) * 4);
    std::vector<float> h(e);
    for (int64_t i = 0; i < e; ++i) h[i] = (float)((i * 2654435761u) % 1000) / 500.f - 1.f;
    hipMemcpy(x, h.data(), e * 4, hipMemcpyHostToDevice);
    std::vector<float> ones(C, 1.f), zeros(C, 0.f);
    hipMemcpy(gam, ones.data(), C * 4, hipMemcpyHostToDevice); hipMemcpy(bet, zeros.data(), C * 4, hipMemcpyHostToDevice);
    BnPlan p;
    bn_plan(N, C, HW, true, &p);
    BnHeld hd;
    const BnHeldShape shapes[] = {{EPT, 0}};
    if (!bn_held_plan(p, URSA_BN_HELD, TL_BLOCK, shapes, false, kBnMaxSplit, 0, 0, &hd)) { printf("not eligible\n"); return 1; }
    BnGeom gh = p.g; gh.chunk = hd.chunk;
    const int grid = hd.S * (int)C;
    long long* st;
    hipMalloc((void**)&st, (int64_t)grid * 8 * 8);
    BnSync* sync = reinterpret_cast<BnSync*>(ws + C * kBnMaxSplit * 8);
    const int sg = argc > 1 ? atoi(argv[1]) : 0;
    for (int rep = 0; rep < 3; ++rep) {
        if (argc > 2 && atoi(argv[2]) == 20) hipLaunchKernelGGL(k_tl<20>, dim3(grid), dim3(TL_BLOCK), 0, 0, x, y, reinterpret_cast<bn_u64*>(ws + C * kBnMaxSplit * 4), sync, gam, bet, gh, hd.S, st, sg);
        else if (argc > 2 && atoi(argv[2]) == 80) hipLaunchKernelGGL(k_tl<80>, dim3(grid), dim3(TL_BLOCK), 0, 0, x, y, reinterpret_cast<bn_u64*>(ws + C * kBnMaxSplit * 4), sync, gam, bet, gh, hd.S, st, sg);
        else hipLaunchKernelGGL(k_tl<40>, dim3(grid), dim3(TL_BLOCK), 0, 0, x, y, reinterpret_cast<bn_u64*>(ws + C * kBnMaxSplit * 4), sync, gam, bet, gh, hd.S, st, sg);
    }
    hipDeviceSynchronize();
    std::vector<long long> hs((int64_t)grid * 8);
    hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost);
    long long first = hs[0], last = 0;
    for (int t = 0; t < grid; ++t) { first = std::min(first, hs[t * 8]); last = std::max(last, hs[t * 8 + 6]); }
    double sum[6] = {0};
    for (int t = 0; t < grid; ++t) for (int k = 0; k < 6; ++k) sum[k] += (double)(hs[t * 8 + k + 1] - hs[t * 8 + k]);
    const char* names[6] = {"ticket", "load+reduce", "publish", "gather (wait + read)", "scalars + barrier", "apply + store issue + leave"};
    printf("%d threads per workgroup, %d float4 per thread in registers: S=%d chunk=%d grid=%d  kernel span %lld cycles (wall_clock64: 100 MHz => %.1f us)\n", TL_BLOCK, EPT, hd.S, hd.chunk, grid, last - first, (last - first) / 100.0);
    for (int k = 0; k < 6; ++k) printf("  %-18s mean %.2f us\n", names[k], sum[k] / grid / 100.0);
    // start-time histogram: how many workgroups start in each 10 us slice
    int hist[64] = {0};
    for (int t = 0; t < grid; ++t) { int b = (int)((hs[t * 8] - first) / 1000); if (b < 64) hist[b]++; }
    printf("  starts per 10 us:");
    for (int b = 0; b < 20; ++b) printf(" %d", hist[b]);
    int hist2[64] = {0};
    for (int t = 0; t < grid; ++t) { int b = (int)((hs[t * 8 + 6] - first) / 1000); if (b < 64) hist2[b]++; }
    printf("\n  ends per 10 us:  ");
    for (int b = 0; b < 20; ++b) printf(" %d", hist2[b]);
    printf("\n");
    return 0;
}
