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

// One workgroup = one problem b and one tile of TI rows x TJ columns (1024 output quads, QPT per
// thread).  The tile's input rows are staged once into LDS already divided by the lengthscale, so
// the inner loop is branch-free LDS reads + FMA + exp + one 16-byte store per quad; b, its
// hyper-parameters and all 64-bit index arithmetic are wave-uniform scalars.
constexpr int QPT = 4;

template <typename T, int FP>
__global__ void __launch_bounds__(256) gram_kernel(const T* __restrict__ z1, int z1_div, const T* __restrict__ z2, int z2_div,
                                                   const T* __restrict__ ls, const T* __restrict__ os,
                                                   const T* __restrict__ noise, int add_noise, T* __restrict__ K,
                                                   int P, int n, int m, int f, int tjq_shift, int tiles_i, int tiles_j) {
    using V = typename VecOf<T>::type;
    constexpr int VW = VecOf<T>::W;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* z1s = reinterpret_cast<T*>(smem_raw);
    const int TJQ = 1 << tjq_shift;                  // quads per tile row
    const int TJ = TJQ * VW;                         // tile columns
    const int TI = (QPT * 256) >> tjq_shift;         // tile rows
    T* z2s = z1s + TI * FP;
    const int tj = blockIdx.x % tiles_j;
    const int rest = blockIdx.x / tiles_j;
    const int ti = rest % tiles_i;
    const int b = rest / tiles_i;
    const int p = b % P;
    const int i0 = ti * TI, j00 = tj * TJ;
    const T* lp = ls + (long)p * f;
    const T osv = os ? os[p] : T(1);
    const T nz = add_noise ? noise[p] : T(0);
    const T* z1b = z1 + (long)(b / z1_div) * n * f;
    const T* z2b = z2 + (long)(b / z2_div) * m * f;
    for (int e = threadIdx.x; e < TI * FP; e += 256) {
        const int r = e / FP, c = e - r * FP;
        z1s[e] = (i0 + r < n && c < f) ? z1b[(long)(i0 + r) * f + c] / lp[c] : T(0);
    }
    for (int e = threadIdx.x; e < TJ * FP; e += 256) {
        const int r = e / FP, c = e - r * FP;
        z2s[e] = (j00 + r < m && c < f) ? z2b[(long)(j00 + r) * f + c] / lp[c] : T(0);
    }
    __syncthreads();
    T* Kb = K + (long)b * n * m;
    const bool vec_ok = (m % VW) == 0;
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const int q = u * 256 + threadIdx.x;
        const int il = q >> tjq_shift, jq = q & (TJQ - 1);
        const int i = i0 + il, j0 = j00 + jq * VW;
        T a[FP];
#pragma unroll
        for (int c = 0; c < FP; ++c) a[c] = z1s[il * FP + c];
        T out[VW];
#pragma unroll
        for (int v = 0; v < VW; ++v) {
            const T* bp = z2s + (jq * VW + v) * FP;
            T s = 0;
#pragma unroll
            for (int c = 0; c < FP; ++c) { T d = a[c] - bp[c]; s = fma(d, d, s); }
            T k = osv * rbf_exp<T>(T(-0.5) * s);
            if (i == j0 + v) k += nz;
            out[v] = k;
        }
        if (i < n && j0 < m) {
            T* kp = Kb + (long)i * m + j0;
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
}

template <typename T>
static int launch_gram(const void* z1, int z1_div, const void* z2, int z2_div, const void* ls, const void* os,
                       const void* noise, int add_noise, void* K, int B, int P, int n, int m, int f, hipStream_t s) {
    constexpr int VW = VecOf<T>::W;
    const int mq = (m + VW - 1) / VW;
    int tjq_shift = 0;
    while ((1 << tjq_shift) < mq && tjq_shift < 6) ++tjq_shift;       // tile: up to 64 quads wide
    const int TJ = (1 << tjq_shift) * VW, TI = (QPT * 256) >> tjq_shift;
    const int tiles_i = (n + TI - 1) / TI, tiles_j = (m + TJ - 1) / TJ;
    const long blocks = (long)B * tiles_i * tiles_j;
    if (blocks > 0x7fffffffL) return PACOH_ELIMIT;
    const int FP = f <= 2 ? 2 : (f <= 4 ? 4 : (f <= 8 ? 8 : 16));
    const size_t lds = (size_t)(TI + TJ) * FP * sizeof(T);
#define PACOH_GRAM_CASE(fp) case fp: hipLaunchKernelGGL((gram_kernel<T, fp>), dim3((unsigned)blocks), dim3(256), lds, s, \
        (const T*)z1, z1_div, (const T*)z2, z2_div, (const T*)ls, (const T*)os, (const T*)noise, add_noise, (T*)K, P, n, m, f, \
        tjq_shift, tiles_i, tiles_j); break;
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
