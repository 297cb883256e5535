#include "../meta_learning_pacoh_amd/csrc/gp_mfma.hip"
using namespace pacoh;
__global__ void dbg_kernel(const float* Kin, float* Xout, float* invd_out, int* ok_out) {
    __shared__ __attribute__((aligned(16))) float A[16 * 20];
    __shared__ float invd[16];
    int lane = threadIdx.x;
    for (int e = lane; e < 256; e += 64) A[(e / 16) * 20 + (e % 16)] = Kin[e];
    __syncthreads();
    bool ok = factor_diag_block(A, 20, 0, invd, lane & 15);
    __syncthreads();
    for (int e = lane; e < 256; e += 64) Xout[e] = A[(e / 16) * 20 + (e % 16)];
    if (lane < 16) invd_out[lane] = invd[lane];
    if (lane == 0) *ok_out = ok;
}
extern "C" int dbg_run(const float* K, float* X, float* invd, int* ok) {
    hipLaunchKernelGGL(dbg_kernel, dim3(1), dim3(64), 0, 0, K, X, invd, ok);
    return (int)hipDeviceSynchronize();
}
