// Probe: what do DPP row_mirror / permlane16_swap / permlane32_swap actually return on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__global__ void k(float* out) {
    const float v = (float)threadIdx.x;
    out[threadIdx.x] = dpp<0xB1>(v);
    out[64 + threadIdx.x] = dpp<0x4E>(v);
    out[128 + threadIdx.x] = dpp<0x141>(v);
    out[192 + threadIdx.x] = dpp<0x140>(v);
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    out[256 + threadIdx.x] = __builtin_bit_cast(float, r[0]);
    out[320 + threadIdx.x] = __builtin_bit_cast(float, r[1]);
    auto q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    out[384 + threadIdx.x] = __builtin_bit_cast(float, q[0]);
    out[448 + threadIdx.x] = __builtin_bit_cast(float, q[1]);
}
__global__ void k2(float* out) {
    float v = (float)threadIdx.x;
    v = v + dpp<0xB1>(v); v = v + dpp<0x4E>(v); v = v + dpp<0x141>(v); v = v + dpp<0x140>(v);
    out[threadIdx.x] = v;
    float w;
    asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "v"(v));
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, w), false, false);
    out[64 + threadIdx.x] = __builtin_bit_cast(float, r[0]);
    out[128 + threadIdx.x] = __builtin_bit_cast(float, r[1]);
    v = __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
    out[192 + threadIdx.x] = v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(w) : "v"(v));
    auto q = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, w), false, false);
    out[256 + threadIdx.x] = __builtin_bit_cast(float, q[0]);
    out[320 + threadIdx.x] = __builtin_bit_cast(float, q[1]);
    out[384 + threadIdx.x] = __builtin_bit_cast(float, q[0]) + __builtin_bit_cast(float, q[1]);
}
int main() {
    float* d; hipMalloc(&d, 512 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[512]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[] = {"quad[1,0,3,2]", "quad[2,3,0,1]", "row_half_mirror", "row_mirror", "p16swap[0]", "p16swap[1]", "p32swap[0]", "p32swap[1]"};
    for (int r = 0; r < 8; ++r) { printf("%-16s", names[r]); for (int i = 0; i < 64; ++i) printf("%2.0f ", h[r * 64 + i]); printf("\n"); }
    hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* n2[] = {"row sums", "p16[0]", "p16[1]", "pair sums", "p32[0]", "p32[1]", "total"};
    for (int r = 0; r < 7; ++r) { printf("%-16s", n2[r]); for (int i = 0; i < 64; ++i) printf("%4.0f ", h[r * 64 + i]); printf("\n"); }
    return 0;
}
