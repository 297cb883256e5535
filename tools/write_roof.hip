// micro-benchmark: what does a pure 16-B-per-lane streaming write reach on this GPU?  (context for the Gram roofline)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int NT, int QPT>
__global__ void __launch_bounds__(256) fill_kernel(float4* __restrict__ out, long nquads, float v) {
    long base = (long)blockIdx.x * 256 * QPT;
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        long q = base + u * 256 + threadIdx.x;
        if (q < nquads) {
            float4 o = make_float4(v + u, v, v, v);
            if (NT) { v4f w = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(w, reinterpret_cast<v4f*>(out + q)); } else out[q] = o;
        }
    }
}

template <int NT, int QPT>
float run(float4* buf, long nquads, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    unsigned blocks = (unsigned)((nquads + 256 * QPT - 1) / (256 * QPT));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((fill_kernel<NT, QPT>), dim3(blocks), dim3(256), 0, 0, buf, nquads, 1.0f);
    hipEventRecord(a);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((fill_kernel<NT, QPT>), dim3(blocks), dim3(256), 0, 0, buf, nquads, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    for (long bytes : {346030080L, 1L << 30, 4L << 30}) {
        float4* buf; if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
        long nq = bytes / 16;
        float t0 = run<0, 4>(buf, nq, 20), t1 = run<1, 4>(buf, nq, 20), t2 = run<0, 1>(buf, nq, 20), t3 = run<0, 16>(buf, nq, 20);
        printf("bytes %ld: plain qpt4 %.1f GB/s | nt qpt4 %.1f GB/s | plain qpt1 %.1f GB/s | plain qpt16 %.1f GB/s\n", bytes,
               bytes / t0 * 1e-6, bytes / t1 * 1e-6, bytes / t2 * 1e-6, bytes / t3 * 1e-6);
        float t100 = run<0, 4>(buf, nq, 100), t400 = run<0, 4>(buf, nq, 400);
        printf("   sustained, plain qpt4: %.1f GB/s over 100 launches, %.1f GB/s over 400\n", bytes / t100 * 1e-6, bytes / t400 * 1e-6);
        hipFree(buf);
    }
    return 0;
}
