// Interleaved A/B of K1 (SGHMC + Philox + fused zero-grad, 24 B/elem) launch/loop variants at a
// roofline-sized arena. Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o k1_variants k1_variants.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../ursabench_amd/csrc/ursa_rng.h"

struct S { float lr, mu, c_wd, c_noise, n_train; uint64_t seed, step; };

__device__ __forceinline__ void elem(float& th, float g, float& v, float e, const S& s) {
    g = __builtin_fmaf(s.c_wd, th, g);
    float b = v * s.mu;
    float d = __builtin_fmaf(-s.lr, g, b);
    d = d + (e * s.c_noise) / s.n_train;
    th = th + d; v = d;
}
__device__ __forceinline__ void upd(float4& t, const float4& g, float4& v, const float4& e, const S& s) {
    elem(t.x, g.x, v.x, e.x, s); elem(t.y, g.y, v.y, e.y, s); elem(t.z, g.z, v.z, e.z, s); elem(t.w, g.w, v.w, e.w, s);
}

template <int NT>
__device__ __forceinline__ float4 ld(const float4* p) {
    if (NT) { float4 r; r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y);
              r.z = __builtin_nontemporal_load(&p->z); r.w = __builtin_nontemporal_load(&p->w); return r; }
    return *p;
}
template <int NT>
__device__ __forceinline__ void st(float4* p, const float4& v) {
    if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
              __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w); }
    else *p = v;
}

// UNROLL float4 per thread per iteration, all loads issued before compute
template <int BLOCK, int UNROLL, int NT>
__global__ __launch_bounds__(BLOCK) void k(float4* __restrict__ th, float4* __restrict__ g, float4* __restrict__ m,
                                           int64_t n4, S s) {
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 t[UNROLL], gg[UNROLL], v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { t[u] = ld<NT>(th + i + u * stride); gg[u] = ld<NT>(g + i + u * stride); v[u] = ld<NT>(m + i + u * stride); }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const float4 e = ursa::normal4(s.seed, s.step, (uint64_t)(i + u * stride));
            upd(t[u], gg[u], v[u], e, s);
            st<NT>(th + i + u * stride, t[u]); st<NT>(m + i + u * stride, v[u]);
            st<NT>(g + i + u * stride, make_float4(0, 0, 0, 0));
        }
    }
    for (; i < n4; i += stride) {
        float4 t = th[i], gg = g[i], v = m[i];
        upd(t, gg, v, ursa::normal4(s.seed, s.step, (uint64_t)i), s);
        th[i] = t; m[i] = v; g[i] = make_float4(0, 0, 0, 0);
    }
}

// contiguous chunk per block instead of grid stride (each block owns n4/grid consecutive float4)
template <int BLOCK, int NT>
__global__ __launch_bounds__(BLOCK) void kc(float4* __restrict__ th, float4* __restrict__ g, float4* __restrict__ m,
                                            int64_t n4, S s) {
    const int64_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < n4 ? b0 + per : n4;
    for (int64_t i = b0 + threadIdx.x; i < b1; i += BLOCK) {
        float4 t = ld<NT>(th + i), gg = ld<NT>(g + i), v = ld<NT>(m + i);
        upd(t, gg, v, ursa::normal4(s.seed, s.step, (uint64_t)i), s);
        st<NT>(th + i, t); st<NT>(m + i, v); st<NT>(g + i, make_float4(0, 0, 0, 0));
    }
}

template <int BLOCK, int CH, int NT>
__global__ __launch_bounds__(BLOCK) void kf(float4* __restrict__ th, float4* __restrict__ g, float4* __restrict__ m,
                                            int64_t n4, S s) {
    const int64_t b0 = (int64_t)blockIdx.x * (CH * BLOCK);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int64_t i = b0 + c * BLOCK + threadIdx.x;
        if (i < n4) {
            float4 t = ld<NT>(th + i), gg = ld<NT>(g + i), v = ld<NT>(m + i);
            upd(t, gg, v, ursa::normal4(s.seed, s.step, (uint64_t)i), s);
            st<NT>(th + i, t); st<NT>(m + i, v); st<NT>(g + i, make_float4(0, 0, 0, 0));
        }
    }
}

__global__ void copyk(const float4* __restrict__ a, float4* __restrict__ b, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) b[i] = a[i];
}

__global__ void fillk(float4* a, int64_t n4, uint64_t seed) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) a[i] = ursa::normal4(seed, 0, (uint64_t)i);
}

struct V { const char* name; int bytes; void (*run)(float4*, float4*, float4*, int64_t, S, hipStream_t); };
#define RUN(NAME, KERN, GRID, BLOCK) {NAME, 24, [](float4* a, float4* b, float4* c, int64_t n4, S s, hipStream_t st_) { \
    hipLaunchKernelGGL(KERN, dim3(GRID), dim3(BLOCK), 0, st_, a, b, c, n4, s); }}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : (1ll << 26);
    const int64_t n4 = n / 4;
    float4 *th, *g, *m;
    hipMalloc(&th, n * 4); hipMalloc(&g, n * 4); hipMalloc(&m, n * 4);
    hipLaunchKernelGGL(fillk, dim3(4096), dim3(256), 0, 0, th, n4, 11ull);
    hipLaunchKernelGGL(fillk, dim3(4096), dim3(256), 0, 0, g, n4, 12ull);
    hipLaunchKernelGGL(fillk, dim3(4096), dim3(256), 0, 0, m, n4, 13ull);
    hipDeviceSynchronize();
    S s{1e-3f, 0.5f, 8e-5f, 0.03f, 50000.f, 1, 3};
#define RUNF(NAME, BLOCK, CH, NT) {NAME, 24, [](float4* a, float4* b, float4* c, int64_t n4, S s, hipStream_t st_) { \
    const int64_t grid = (n4 + (int64_t)BLOCK * CH - 1) / ((int64_t)BLOCK * CH); \
    hipLaunchKernelGGL((kf<BLOCK, CH, NT>), dim3((unsigned)grid), dim3(BLOCK), 0, st_, a, b, c, n4, s); }}
    std::vector<V> vs = {
        RUN("stride b256 g2048 (current)", (k<256, 1, 0>), 2048, 256),
        RUNF("fixed b256 ch1", 256, 1, 0), RUNF("fixed b256 ch1 nt", 256, 1, 1),
        RUNF("fixed b256 ch2", 256, 2, 0), RUNF("fixed b256 ch2 nt", 256, 2, 1),
        RUNF("fixed b256 ch4", 256, 4, 0), RUNF("fixed b256 ch4 nt", 256, 4, 1),
        RUNF("fixed b256 ch8", 256, 8, 0), RUNF("fixed b256 ch8 nt", 256, 8, 1),
        RUNF("fixed b512 ch1", 512, 1, 0), RUNF("fixed b512 ch1 nt", 512, 1, 1),
        RUNF("fixed b512 ch2", 512, 2, 0), RUNF("fixed b512 ch2 nt", 512, 2, 1),
        RUNF("fixed b1024 ch1", 1024, 1, 0), RUNF("fixed b1024 ch1 nt", 1024, 1, 1),
        RUNF("fixed b128 ch2", 128, 2, 0), RUNF("fixed b128 ch4 nt", 128, 4, 1),
        {"copy (8 B/elem) g2048", 8, [](float4* a, float4* b, float4*, int64_t n4, S, hipStream_t st_) {
             hipLaunchKernelGGL(copyk, dim3(2048), dim3(256), 0, st_, a, b, n4); }},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<std::vector<float>> t(vs.size());
    for (int round = 0; round < 12; ++round)
        for (size_t v = 0; v < vs.size(); ++v) {
            hipEventRecord(e0, 0);
            for (int r = 0; r < 3; ++r) vs[v].run(th, g, m, n4, s, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (round >= 2) t[v].push_back(ms / 3);
        }
    for (size_t v = 0; v < vs.size(); ++v) {
        std::sort(t[v].begin(), t[v].end());
        const float med = t[v][t[v].size() / 2], best = t[v][0];
        printf("%-28s median %8.1f us  %7.1f GB/s   best %8.1f us %7.1f GB/s\n", vs[v].name, med * 1e3,
               vs[v].bytes * (double)n / (med * 1e-3) / 1e9, best * 1e3, vs[v].bytes * (double)n / (best * 1e-3) / 1e9);
    }
    return 0;
}
