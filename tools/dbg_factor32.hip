// timing harness for factor_invert_diag32<T> (dense_mfma.hip): one wavefront, 32x32 SPD block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#define PACOH_FACT_DEBUG 1
__device__ long long g_tdbg[4];
#include "../meta_learning_pacoh_amd/csrc/dense_mfma.hip"

template <typename T>
__global__ void __launch_bounds__(256) k(const T* __restrict__ Ain, T* __restrict__ Lout, T* __restrict__ Xout, long long* t) {
    __shared__ T Ds[pacoh::DNB][pacoh::DNB + 1];
    __shared__ T Li[pacoh::DNB * pacoh::DLP];
    __shared__ T scr[128];
    for (int q = threadIdx.x; q < 1024; q += 256) Ds[q / 32][q % 32] = Ain[q];
    if (threadIdx.x == 0) scr[16] = 0;
    __syncthreads();
    long long t0 = wall_clock64();
    if (threadIdx.x < 64) pacoh::factor_invert_diag32<T>(Ds, Li, scr + 32, scr + 96, scr + 16, threadIdx.x);
    __syncthreads();
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = g_tdbg[0]; t[2] = g_tdbg[1]; }
    for (int q = threadIdx.x; q < 1024; q += 256) { Lout[q] = (q % 32 <= q / 32) ? Ds[q / 32][q % 32] : T(0); Xout[q] = Li[(q / 32) * pacoh::DLP + q % 32]; }
}

template <typename T> void run(const char* name) {
    std::vector<T> A(1024), L(1024), X(1024);
    std::vector<double> M(1024);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) M[i * 32 + j] = std::sin(0.37 * i * j + i) * 0.3;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = (i == j) ? 2.0 : 0.0; for (int q = 0; q < 32; ++q) s += M[i * 32 + q] * M[j * 32 + q]; A[i * 32 + j] = (T)s; }
    T *dA, *dL, *dX; long long* dt;
    (void)hipMalloc(&dA, 1024 * sizeof(T)); (void)hipMalloc(&dL, 1024 * sizeof(T)); (void)hipMalloc(&dX, 1024 * sizeof(T)); (void)hipMalloc(&dt, 32);
    (void)hipMemcpy(dA, A.data(), 1024 * sizeof(T), hipMemcpyHostToDevice);
    long long tt[3] = {0, 0, 0};
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<T>, dim3(1), dim3(256), 0, 0, dA, dL, dX, dt);
        (void)hipMemcpy(tt, dt, 24, hipMemcpyDeviceToHost);
    }
    (void)hipMemcpy(L.data(), dL, 1024 * sizeof(T), hipMemcpyDeviceToHost); (void)hipMemcpy(X.data(), dX, 1024 * sizeof(T), hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int q = 0; q <= j; ++q) s += (double)L[i * 32 + q] * L[j * 32 + q]; e1 = std::fmax(e1, std::fabs(s - A[i * 32 + j])); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double s = 0; for (int q = 0; q < 32; ++q) s += (double)L[i * 32 + q] * X[q * 32 + j]; e2 = std::fmax(e2, std::fabs(s - (i == j))); }
    printf("%s: %lld ticks (x10 ns) per block (factor %lld, invert %lld), |LL^T-A| %.2e |L X - I| %.2e\n", name, tt[0], tt[1], tt[2], e1, e2);
}

int main() { run<float>("fp32"); run<double>("fp64"); return 0; }
