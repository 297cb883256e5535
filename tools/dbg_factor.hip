// isolation harness for factor_diag_block (gp_mfma.hip): one 16x16 SPD block, one wavefront; prints the error of L11^-1
#include <cstdio>
#include <cmath>
#include <vector>
#include "../meta_learning_pacoh_amd/csrc/gp_mfma.hip"
using namespace pacoh;
__global__ void dbg_kernel(const float* Kin, float* Xout, float* invd_out, int* ok_out) {
    __shared__ __attribute__((aligned(16))) float A[16 * 20];
    __shared__ __attribute__((aligned(16))) float invd[16];
    __shared__ __attribute__((aligned(16))) float scr[64];
    int lane = threadIdx.x;
    for (int e = lane; e < 256; e += 64) A[(e / 16) * 20 + (e % 16)] = Kin[e];
    if (lane == 0) scr[40] = 0.0f;
    __syncthreads();
    factor_diag_block<1>(A, 20, 0, invd, scr, lane & 15, true);
    __syncthreads();
    for (int e = lane; e < 256; e += 64) Xout[e] = A[(e / 16) * 20 + (e % 16)];
    if (lane < 16) invd_out[lane] = invd[lane];
    if (lane == 0) *ok_out = scr[40] == 0.0f;
}
int main() {
    std::vector<float> K(256), X(256), iv(16);
    std::vector<double> M(256), L(256, 0.0), Xr(256, 0.0);
    for (int i = 0; i < 256; ++i) M[i] = std::sin(0.91 * i + 0.3 * (i % 7));
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = (i == j) ? 4.0 : 0.0; for (int q = 0; q < 16; ++q) s += M[i * 16 + q] * M[j * 16 + q]; K[i * 16 + j] = (float)s; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j <= i; ++j) { double s = K[i * 16 + j]; for (int q = 0; q < j; ++q) s -= L[i * 16 + q] * L[j * 16 + q]; L[i * 16 + j] = (i == j) ? std::sqrt(s) : s / L[j * 16 + j]; }
    for (int c = 0; c < 16; ++c) for (int i = 0; i < 16; ++i) { double s = (i == c) ? 1.0 : 0.0; for (int j = 0; j < i; ++j) s -= L[i * 16 + j] * Xr[j * 16 + c]; Xr[i * 16 + c] = s / L[i * 16 + i]; }
    float *dK, *dX, *dI; int* dok; int ok = 0;
    (void)hipMalloc(&dK, 1024); (void)hipMalloc(&dX, 1024); (void)hipMalloc(&dI, 64); (void)hipMalloc(&dok, 4);
    (void)hipMemcpy(dK, K.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(dbg_kernel, dim3(1), dim3(64), 0, 0, dK, dX, dI, dok);
    (void)hipMemcpy(X.data(), dX, 1024, hipMemcpyDeviceToHost); (void)hipMemcpy(iv.data(), dI, 64, hipMemcpyDeviceToHost); (void)hipMemcpy(&ok, dok, 4, hipMemcpyDeviceToHost);
    double ex = 0, ei = 0;
    for (int i = 0; i < 256; ++i) ex = std::fmax(ex, std::fabs(X[i] - Xr[i]));
    for (int i = 0; i < 16; ++i) ei = std::fmax(ei, std::fabs(iv[i] - 1.0 / L[i * 16 + i]));
    printf("ok %d  |X - L^-1| %.3e  |invd - 1/diag| %.3e\n", ok, ex, ei);
    return 0;
}
