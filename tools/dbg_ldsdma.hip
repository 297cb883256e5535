// semantics check of the LDS-DMA load builtin on gfx950: each lane's 4 / 16 bytes land at lds_base + lane * size
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
__global__ void k(const float* __restrict__ src, float* __restrict__ out, const int* __restrict__ perm) {
    __shared__ __attribute__((aligned(16))) float buf[2][256];
    const int lane = threadIdx.x;
    // dword: lane reads src[perm[lane]]
    __builtin_amdgcn_global_load_lds((gptr_t*)(src + perm[lane]), (lptr_t*)&buf[0][0], 4, 0, 0);
    // dwordx4: lane reads 4 floats at src + 4 * perm[lane]
    __builtin_amdgcn_global_load_lds((gptr_t*)(src + 4 * perm[lane]), (lptr_t*)&buf[1][0], 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);           // vmcnt(0) lgkmcnt(0) expcnt(0)
    __syncthreads();
    out[lane] = buf[0][lane];
    for (int q = 0; q < 4; ++q) out[64 + lane * 4 + q] = buf[1][lane * 4 + q];
}
int main() {
    std::vector<float> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (float)i;
    std::vector<int> p(64); for (int i = 0; i < 64; ++i) p[i] = (i * 37) % 64;
    float *d, *o; int* dp;
    (void)hipMalloc(&d, 4096); (void)hipMalloc(&o, 320 * 4); (void)hipMalloc(&dp, 256);
    (void)hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice); (void)hipMemcpy(dp, p.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, dp);
    std::vector<float> r(320); (void)hipMemcpy(r.data(), o, 320 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) { if (r[i] != (float)p[i]) ++bad; for (int q = 0; q < 4; ++q) if (r[64 + i * 4 + q] != (float)(4 * p[i] + q)) ++bad; }
    printf("lds dma mismatches: %d (first values %g %g %g | %g %g %g %g)\n", bad, r[0], r[1], r[2], r[64], r[65], r[66], r[67]);
    return 0;
}
