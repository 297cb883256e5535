// gfx950 v_permlane16_swap / v_permlane32_swap: what the clang builtins return vs the instruction written as inline assembly.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/dbg_permlane.hip -o tools/dbg_permlane
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_builtin(const float* A, float* out) {
    float v = A[threadIdx.x], w = A[threadIdx.x + 64];
    auto p = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, w), false, false);
    out[threadIdx.x] = __builtin_bit_cast(float, p[0]);
    out[threadIdx.x + 64] = __builtin_bit_cast(float, p[1]);
}
__global__ void k_asm(const float* A, float* out) {
    float v = A[threadIdx.x], w = A[threadIdx.x + 64];
    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v), "+v"(w));
    out[threadIdx.x] = v;
    out[threadIdx.x + 64] = w;
}
int main() {
    float h[128], o[128], *dA, *dO;
    for (int i = 0; i < 64; ++i) { h[i] = (float)i; h[64 + i] = 100.0f + i; }
    hipMalloc(&dA, sizeof(h)); hipMalloc(&dO, sizeof(o));
    hipMemcpy(dA, h, sizeof(h), hipMemcpyHostToDevice);
    for (int which = 0; which < 2; ++which) {
        if (which == 0) hipLaunchKernelGGL(k_builtin, dim3(1), dim3(64), 0, 0, dA, dO);
        else hipLaunchKernelGGL(k_asm, dim3(1), dim3(64), 0, 0, dA, dO);
        hipMemcpy(o, dO, sizeof(o), hipMemcpyDeviceToHost);
        printf("%s\n first : ", which == 0 ? "builtin" : "inline asm");
        for (int i = 0; i < 64; i += 8) printf("%g ", o[i]);
        printf("\n second: ");
        for (int i = 0; i < 64; i += 8) printf("%g ", o[64 + i]);
        printf("\n");
    }
    return 0;
}
