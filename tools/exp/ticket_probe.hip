// How do agent-scope fetch-add tickets cost on gfx950 as a function of how many workgroups take one and of where the
// counters live? One workgroup = 512 threads; thread 0 takes a ticket from counter[(blockIdx.x % spread) * stride_words]
// and the last ticket of a counter bumps a second-level counter (two-level tree). Compared: spread = 1 (one address),
// 8 counters in ONE 64-byte line, 8 counters on 8 different 128-byte lines. Timed as 64-launch hipGraph-free loops with
// events (the kernel does nothing else, so this is the ticket cost on top of an empty launch).
//   hipcc --offload-arch=gfx950 -O3 -o ticket_probe ticket_probe.hip && ./ticket_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__global__ void k_tickets(uint32_t* c, int spread, int stride_words, uint32_t per_counter, int two_level, uint32_t* done)
{
    if (threadIdx.x != 0) return;
    if (spread == 0) return;                                  // empty launch: the baseline
    const int g = blockIdx.x % spread;
    uint32_t* p = c + (size_t)g * stride_words;
    const uint32_t t = __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == per_counter - 1) {
        __hip_atomic_store(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (two_level) {
            uint32_t* top = c + 4096;
            const uint32_t t2 = __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t2 == (uint32_t)spread - 1) { __hip_atomic_store(top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); done[0] += 1; }
        } else {
            done[0] += 1;
        }
    }
}

int main()
{
    uint32_t *c, *done;
    hipMalloc(&c, 8192 * 4); hipMemset(c, 0, 8192 * 4);
    hipMalloc(&done, 4); hipMemset(done, 0, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int reps = 200;
    struct { const char* name; int spread, stride_words, two; } cfgs[] = {
        {"empty launch", 0, 0, 0}, {"one address", 1, 0, 0}, {"8 counters, one 64-B line, two-level", 8, 2, 1},
        {"8 counters, 8 x 128-B lines, two-level", 8, 32, 1}, {"16 counters, 16 x 128-B lines, two-level", 16, 32, 1}};
    for (int blocks : {128, 136, 256, 512, 1024}) {
        for (auto& cf : cfgs) {
            if (cf.spread && blocks % cf.spread) continue;
            const uint32_t per = cf.spread ? blocks / cf.spread : 0;
            for (int w = 0; w < 5; ++w) hipLaunchKernelGGL(k_tickets, dim3(blocks), dim3(512), 0, 0, c, cf.spread, cf.stride_words, per, cf.two, done);
            hipDeviceSynchronize();
            hipEventRecord(a, 0);
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_tickets, dim3(blocks), dim3(512), 0, 0, c, cf.spread, cf.stride_words, per, cf.two, done);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("{\"blocks\": %d, \"config\": \"%s\", \"us_per_launch\": %.3f}\n", blocks, cf.name, ms * 1e3 / reps);
        }
    }
    uint32_t h; hipMemcpy(&h, done, 4, hipMemcpyDeviceToHost);
    printf("{\"advances\": %u}\n", h);
    return 0;
}
