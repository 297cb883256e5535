// Standalone (materialised) ARD-RBF Gram build: the HBM-write-bound kernel of the path.
// K[b,i,j] = os_p * exp(-0.5 * sum_k ((z1[b,i,k]-z2[b,j,k])/l_pk)^2) (+ noise_p on the diagonal).
// Replaces SEKernelLight.forward (meta_learn/models.py:428-446) / ScaleKernel(RBFKernel(ard))
// (GPR_meta_mll.py:218,223) when K is needed in memory (large-n dense path, K_xs, K_ss).
//
// Algorithmic bytes per Gram: n*f*s (+ m*f*s) read + n*m*s written.  Each lane produces
// VW consecutive columns of one row and stores them with ONE 16-byte store, so a wave writes 1 KiB
// contiguous; the few input rows it needs come out of L1/L2.
#include "common.h"

namespace pacoh {

template <typename T, int FP>
__global__ void __launch_bounds__(256) gram_kernel(const T* __restrict__ z1, int z1_div, const T* __restrict__ z2, int z2_div,
                                                   const T* __restrict__ ls, const T* __restrict__ os,
                                                   const T* __restrict__ noise, int add_noise, T* __restrict__ K,
                                                   int B, int P, int n, int m, int f, int mq /* = ceil(m/VW) */) {
    using V = typename VecOf<T>::type;
    constexpr int VW = VecOf<T>::W;
    const long total = (long)B * n * mq;
    const bool vec_ok = (m % VW) == 0;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
        const int jq = (int)(q % mq);
        const long bi = q / mq;
        const int i = (int)(bi % n);
        const long b = bi / n;
        const int p = (int)(b % P);
        const int j0 = jq * VW;
        T inv[FP], a[FP];
        const T* lp = ls + (long)p * f;
        const T* ap = z1 + ((b / z1_div) * n + i) * (long)f;
#pragma unroll
        for (int c = 0; c < FP; ++c) {
            inv[c] = (c < f) ? T(1) / lp[c] : T(0);
            a[c] = (c < f) ? ap[c] * inv[c] : T(0);
        }
        const T osv = os ? os[p] : T(1);
        T out[VW];
#pragma unroll
        for (int v = 0; v < VW; ++v) {
            const int j = j0 + v;
            T s = 0;
            if (j < m) {
                const T* bp = z2 + ((b / z2_div) * m + j) * (long)f;
#pragma unroll
                for (int c = 0; c < FP; ++c) if (c < f) { T d = a[c] - bp[c] * inv[c]; s = fma(d, d, s); }
            }
            T k = osv * t_exp<T>(T(-0.5) * s);
            if (add_noise && i == j) k += noise[p];
            out[v] = k;
        }
        T* kp = K + (b * n + i) * (long)m + j0;
        if (vec_ok) {
            V o;
            if constexpr (VW == 4) { o.x = out[0]; o.y = out[1]; o.z = out[2]; o.w = out[3]; } else { o.x = out[0]; o.y = out[1]; }
            *reinterpret_cast<V*>(kp) = o;
        } else {
#pragma unroll
            for (int v = 0; v < VW; ++v) if (j0 + v < m) kp[v] = out[v];
        }
    }
}

template <typename T>
static int launch_gram(const void* z1, int z1_div, const void* z2, int z2_div, const void* ls, const void* os,
                       const void* noise, int add_noise, void* K, int B, int P, int n, int m, int f, hipStream_t s) {
    constexpr int VW = VecOf<T>::W;
    const int mq = (m + VW - 1) / VW;
    const long total = (long)B * n * mq;
    long blocks = (total + 255) / 256;
    const long cap = 256L * 32;          // 32 resident-ish workgroups per CU worth of grid, grid-stride beyond
    if (blocks > cap) blocks = cap;
    const int FP = f <= 2 ? 2 : (f <= 4 ? 4 : (f <= 8 ? 8 : 16));
#define PACOH_GRAM_CASE(fp) case fp: hipLaunchKernelGGL((gram_kernel<T, fp>), dim3((unsigned)blocks), dim3(256), 0, s, \
        (const T*)z1, z1_div, (const T*)z2, z2_div, (const T*)ls, (const T*)os, (const T*)noise, add_noise, (T*)K, B, P, n, m, f, mq); break;
    switch (FP) { PACOH_GRAM_CASE(2) PACOH_GRAM_CASE(4) PACOH_GRAM_CASE(8) default: PACOH_GRAM_CASE(16) }
#undef PACOH_GRAM_CASE
    return launch_status();
}

}  // namespace pacoh

using namespace pacoh;

extern "C" int pacoh_gram_rbf_ard(const void* z1, int z1_div, const void* z2, int z2_div,
                                  const void* lengthscale, const void* outputscale, const void* noise,
                                  int add_noise_diag, void* K,
                                  int B, int P, int n, int m, int f, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!z1 || !z2 || !lengthscale || !K || B <= 0 || P <= 0 || n <= 0 || m <= 0 || f <= 0 || z1_div <= 0 || z2_div <= 0)
        return PACOH_EINVAL;
    if (add_noise_diag && !noise) return PACOH_EINVAL;
    if (f > PACOH_MAX_FEATURES) return PACOH_ELIMIT;
    if (dtype == PACOH_F32)
        return launch_gram<float>(z1, z1_div, z2, z2_div, lengthscale, outputscale, noise, add_noise_diag, K, B, P, n, m, f, (hipStream_t)stream);
    return launch_gram<double>(z1, z1_div, z2, z2_div, lengthscale, outputscale, noise, add_noise_diag, K, B, P, n, m, f, (hipStream_t)stream);
}
