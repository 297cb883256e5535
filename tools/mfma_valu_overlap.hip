// Do vector instructions issue under a running MFMA on gfx950?  Each loop iteration issues 4 independent
// v_mfma_f32_16x16x4_f32 (own accumulators) with K independent v_fma_f32 after each of them, from W waves per SIMD.
// Reports shader cycles per MFMA (s_memtime at 100 MHz is too coarse: HIP events and the 2.4 GHz clock are used instead).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap && tools/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int K, bool DEP>
__global__ void __launch_bounds__(64) k_overlap(float* out, int iters, float seed) {
    f32x4 acc[4];
    float x[12];
    const float a = seed + (threadIdx.x & 63) * 1e-3f, b = 1.0f - seed * (threadIdx.x & 15) * 1e-3f;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f32x4{(float)q, 0.f, 1.f, 0.f};
#pragma unroll
    for (int q = 0; q < 12; ++q) x[q] = seed * q;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (DEP) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < K; ++k) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[k % 12]) : "v"(a), "v"(b));
        }
    }
    float s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
#pragma unroll
    for (int q = 0; q < 12; ++q) s += x[q];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int K, bool DEP>
static void run(int waves_per_simd) {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 4 * waves_per_simd, iters = 8192;
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_overlap<K, DEP>), dim3(blocks), dim3(64), 0, 0, out, iters, 0.37f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_overlap<K, DEP>), dim3(blocks), dim3(64), 0, 0, out, iters, 0.37f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms / 5 * 1e-3 * 2.4e9 / ((double)iters * 4 * waves_per_simd);      // SIMD cycles per MFMA (+ its K fmas)
    printf("%s MFMAs, %2d v_fma per MFMA, %d waves/SIMD: %6.1f cycles per MFMA group  (MFMA alone 32, the fmas alone %d)\n",
           DEP ? "dependent  " : "independent", K, waves_per_simd, cyc, 4 * K);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; ++w) {
        run<0, false>(w); run<2, false>(w); run<4, false>(w); run<6, false>(w); run<8, false>(w); run<12, false>(w); run<16, false>(w);
        run<0, true>(w); run<4, true>(w); run<8, true>(w); run<12, true>(w);
    }
    return 0;
}
