// Measured peak of the matrix-core instructions the PACOH kernels use: v_mfma_f64_16x16x4_f64 (the large-context fp64 path) and
// v_mfma_f32_16x16x4_f32 (everything else).  Back-to-back issue on independent accumulators, operands in registers, every SIMD of
// the chip busy (4 waves per SIMD), random non-trivial operands; reports TFLOP/s over the timed launches (HIP events).
// MI355X_MICROARCH.md lists 157.3 TFLOP/s for fp32-input MFMA and gives no fp64 figure: this fills it in.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak && tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>

using f64x4 = __attribute__((ext_vector_type(4))) double;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <typename T> struct M;
template <> struct M<double> { using acc = f64x4; static __device__ acc mma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); } };
template <> struct M<float> { using acc = f32x4; static __device__ acc mma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); } };

template <typename T, int NACC>
__global__ void __launch_bounds__(256) peak_kernel(T* out, int iters, T seed) {
    typename M<T>::acc acc[NACC];
    const T a = seed + (T)(threadIdx.x & 63) * (T)1e-3, b = (T)1 - seed * (T)(threadIdx.x & 15) * (T)1e-3;
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = typename M<T>::acc{(T)q, (T)0, (T)1, (T)0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NACC; ++q) acc[q] = M<T>::mma(a, b, acc[q]);
    }
    T s = 0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename T, int NACC>
static double run(const char* name) {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 4, iters = 4096;                 // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    T* out;
    hipMalloc(&out, sizeof(T) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((peak_kernel<T, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, (T)0.37);
    hipDeviceSynchronize();
    const int reps = 10;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((peak_kernel<T, NACC>), dim3(blocks), dim3(256), 0, 0, out, iters, (T)0.37);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 16 * 16 * 4 * (double)NACC * iters * (double)blocks * 4 * reps;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-28s %d accumulators: %8.2f TFLOP/s  (%.3f ms per launch)\n", name, NACC, tf, ms / reps);
    hipFree(out);
    return tf;
}

int main() {
    run<double, 4>("v_mfma_f64_16x16x4_f64");
    run<double, 8>("v_mfma_f64_16x16x4_f64");
    run<float, 4>("v_mfma_f32_16x16x4_f32");
    run<float, 8>("v_mfma_f32_16x16x4_f32");
    return 0;
}
