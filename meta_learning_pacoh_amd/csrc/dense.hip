// Dense (large-n) path: blocked right-looking Cholesky of materialised covariance matrices that live
// in HBM/L2, one workgroup per matrix, followed by the triangular solves and the Gaussian log-density.
// Used for the large-context configuration (n = 512, fp64: K comes from pacoh_gram_rbf_ard) and for
// the joint test log-likelihood of RegressionModelMetaLearned.eval (meta_learn/abstract.py:134-163),
// i.e. torch/gpytorch's MultivariateNormal.log_prob -> potrf/potrs on the reference's CPU path.
#include "common.h"
#include <stdlib.h>

namespace pacoh {

constexpr int NB = 32;     // panel width
constexpr int TT = 64;     // trailing-update tile

template <typename T>
__global__ void __launch_bounds__(256) chol_dense_kernel(T* __restrict__ A, const T* __restrict__ resid,
                                                         T* __restrict__ logp, T* __restrict__ alpha_out,
                                                         int32_t* __restrict__ info, T scale, int n, int attempt) {
    // jitter-ladder retries (attempt > 0) only touch the problems that have not succeeded yet
    if (attempt > 0 && info && info[blockIdx.x] >= 0) return;
    // all LDS in ONE dynamic array (cdna_hip_programming.md Guideline 17: statics in front of the
    // dynamic region can shift its base off its natural alignment)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sm = reinterpret_cast<T*>(smem_raw);
    T (*Ds)[NB + 1] = reinterpret_cast<T (*)[NB + 1]>(sm);
    T (*Pa)[NB + 1] = reinterpret_cast<T (*)[NB + 1]>(sm + NB * (NB + 1));
    T (*Pb)[NB + 1] = reinterpret_cast<T (*)[NB + 1]>(sm + NB * (NB + 1) + TT * (NB + 1));
    T* invd_s = sm + NB * (NB + 1) + 2 * TT * (NB + 1);
    T* red = invd_s + NB;
    int* fail_p = reinterpret_cast<int*>(red + 4);
    T* rv = red + 8;                                   // [n] residual -> u -> alpha
#define fail_s (*fail_p)

    const int tid = threadIdx.x;
    T* Ab = A + (size_t)blockIdx.x * n * n;
    if (tid == 0) fail_s = 0;
    for (int q = tid; q < n; q += 256) rv[q] = resid[(size_t)blockIdx.x * n + q];
    T logdet_part = 0;
    __syncthreads();

    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kb = (n - k0 < NB) ? (n - k0) : NB;
        // 1. diagonal block -> LDS (identity padded)
        for (int q = tid; q < NB * NB; q += 256) {
            int r = q / NB, c = q - r * NB;
            T v = (r == c) ? T(1) : T(0);
            if (r < kb && c <= r) v = Ab[(size_t)(k0 + r) * n + k0 + c];
            Ds[r][c] = v;
        }
        __syncthreads();
        // 2. factor it with one wavefront (lane = row), left-looking, pivot by shuffle
        if (tid < 64) {
            const int r = tid & 31;
            for (int j = 0; j < NB; ++j) {
                T s = Ds[r][j];
                for (int c = 0; c < j; ++c) s = fma(-Ds[r][c], Ds[j][c], s);
                T piv = __shfl(s, j, 64);
                if (!(piv > T(0))) { if (tid == 0) fail_s = 1; piv = 1; }
                T d = t_sqrt<T>(piv);
                if (tid < 32) {
                    if (r == j) { Ds[j][j] = d; invd_s[j] = T(1) / d; }
                    else if (r > j) Ds[r][j] = s / d;
                }
            }
        }
        __syncthreads();
        for (int q = tid; q < NB * NB; q += 256) {
            int r = q / NB, c = q - r * NB;
            if (r < kb && c <= r) Ab[(size_t)(k0 + r) * n + k0 + c] = Ds[r][c];
        }
        if (tid < kb) logdet_part += t_log<T>(Ds[tid][tid]);
        // 3. panel below the block: x L11^T = a, one row per thread
        for (int r = k0 + kb + tid; r < n; r += 256) {
            T a[NB];
            T* ap = Ab + (size_t)r * n + k0;
#pragma unroll
            for (int c = 0; c < NB; ++c) a[c] = (c < kb) ? ap[c] : T(0);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                T s = a[j];
#pragma unroll
                for (int c = 0; c < j; ++c) s = fma(-a[c], Ds[j][c], s);
                a[j] = s * invd_s[j];
            }
#pragma unroll
            for (int c = 0; c < NB; ++c) if (c < kb) ap[c] = a[c];
        }
        __syncthreads();
        // 4. trailing update of the lower triangle: A22 -= L21 L21^T, 64x64 tiles, 4x4 per thread
        const int t0 = k0 + kb;
        const int tx = tid & 15, ty = tid >> 4;
        for (int ti = t0; ti < n; ti += TT) {
            for (int tj = t0; tj <= ti; tj += TT) {
                for (int q = tid; q < TT * NB; q += 256) {
                    int r = q / NB, c = q - r * NB;
                    Pa[r][c] = (ti + r < n && c < kb) ? Ab[(size_t)(ti + r) * n + k0 + c] : T(0);
                    Pb[r][c] = (tj + r < n && c < kb) ? Ab[(size_t)(tj + r) * n + k0 + c] : T(0);
                }
                __syncthreads();
                T acc[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int v = 0; v < 4; ++v) acc[u][v] = 0;
                for (int c = 0; c < NB; ++c) {
                    T av[4], bv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { av[u] = Pa[ty + 16 * u][c]; bv[u] = Pb[tx + 16 * u][c]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v) acc[u][v] = fma(av[u], bv[v], acc[u][v]);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    int gi = ti + ty + 16 * u;
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        int gj = tj + tx + 16 * v;
                        if (gi < n && gj <= gi) Ab[(size_t)gi * n + gj] -= acc[u][v];
                    }
                }
                __syncthreads();
            }
        }
    }

    // ---- forward solve L u = r, blocked --------------------------------------------------------
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int kb = (n - k0 < NB) ? (n - k0) : NB;
        for (int q = tid; q < NB * NB; q += 256) {
            int r = q / NB, c = q - r * NB;
            Ds[r][c] = (r < kb && c <= r) ? Ab[(size_t)(k0 + r) * n + k0 + c] : ((r == c) ? T(1) : T(0));
        }
        __syncthreads();
        if (tid < 64) {
            const int r = tid & 31;
            T v = (r < kb) ? rv[k0 + r] : T(0);
            for (int c = 0; c < NB; ++c) {
                T uc = __shfl(v, c, 64) / Ds[c][c];
                if (r == c) v = uc;
                else if (r > c) v = fma(-Ds[r][c], uc, v);
            }
            if (tid < kb) rv[k0 + tid] = v;
        }
        __syncthreads();
        for (int r = k0 + kb + tid; r < n; r += 256) {
            const T* ap = Ab + (size_t)r * n + k0;
            T s = rv[r];
            for (int c = 0; c < kb; ++c) s = fma(-ap[c], rv[k0 + c], s);
            rv[r] = s;
        }
        __syncthreads();
    }
    T quad_part = 0;
    for (int q = tid; q < n; q += 256) quad_part = fma(rv[q], rv[q], quad_part);
    quad_part = subwave_sum<T>(quad_part, 64);
    logdet_part = subwave_sum<T>(logdet_part, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = quad_part;
    __syncthreads();
    T quad = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = logdet_part;
    __syncthreads();
    T logdet = red[0] + red[1] + red[2] + red[3];
    const bool ok = fail_s == 0;
    if (tid == 0) {
        const T LOG2PI = T(1.8378770664093453);
        T lp = T(-0.5) * (quad + T(2) * logdet + T(n) * LOG2PI) * scale;
        logp[blockIdx.x] = ok ? lp : T(NAN);
        if (info) info[blockIdx.x] = ok ? attempt : -1;
    }
    if (!alpha_out) return;
    // ---- backward solve L^T alpha = u, blocked from the bottom ---------------------------------
    const int nblk = (n + NB - 1) / NB;
    for (int kbk = nblk - 1; kbk >= 0; --kbk) {
        const int k0 = kbk * NB;
        const int kb = (n - k0 < NB) ? (n - k0) : NB;
        __syncthreads();
        for (int q = tid; q < NB * NB; q += 256) {
            int r = q / NB, c = q - r * NB;
            Ds[r][c] = (r < kb && c <= r) ? Ab[(size_t)(k0 + r) * n + k0 + c] : ((r == c) ? T(1) : T(0));
        }
        __syncthreads();
        if (tid < 64) {
            const int r = tid & 31;
            T v = (r < kb) ? rv[k0 + r] : T(0);
            for (int c = NB - 1; c >= 0; --c) {
                T ac = __shfl(v, c, 64) / Ds[c][c];
                if (r == c) v = ac;
                else if (r < c) v = fma(-Ds[c][r], ac, v);
            }
            if (tid < kb) rv[k0 + tid] = v;
        }
        __syncthreads();
        for (int i = tid; i < k0; i += 256) {
            T s = rv[i];
            for (int c = 0; c < kb; ++c) s = fma(-Ab[(size_t)(k0 + c) * n + i], rv[k0 + c], s);
            rv[i] = s;
        }
    }
    __syncthreads();
    for (int q = tid; q < n; q += 256) alpha_out[(size_t)blockIdx.x * n + q] = ok ? rv[q] : T(NAN);
}

#undef fail_s
}  // namespace pacoh

using namespace pacoh;

namespace pacoh {
bool dense_mfma_fits(int n, int dtype);                    // dense_mfma.hip
int dense_mfma_try(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n,
                   int dtype, int attempt, int u_only, hipStream_t s);       // dense_mfma.hip; returns 1 if the panel does not fit in LDS

// Cholesky + solves + log-density of B materialised matrices; attempt > 0 re-runs only problems with info[b] < 0
// (and then writes info[b] = attempt on success): the psd_safe_cholesky ladder of the dense path.
// true when dense_chol_launch() takes the MFMA kernel, which leaves the inverses of the diagonal blocks in the upper triangle
bool dense_ll_fits(int n, int dtype);                      // dense_ll.hip
int dense_ll_retry_try(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n, int dtype,
                       int att_lo, int att_hi, int u_only, const void* z, int z_div, const void* ls, const void* os, const void* noise,
                       const int32_t* n_valid, int y_div, double jitter_base, int P, int f, int kind, hipStream_t s);   // dense_ll.hip
int dense_ll_try(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n, int dtype,
                 int attempt, int u_only, hipStream_t s);                    // dense_ll.hip; returns 1 if n is outside its plan
constexpr int LL_MIN_N = 97;                               // below: the right-looking kernel with fewer waves
static bool ll_enabled() { return g_sw.chol_ll; }
bool dense_chol_saves_inverse(int n, int dtype) {
    if (!g_sw.mfma) return false;
    return (ll_enabled() && n >= LL_MIN_N && dense_ll_fits(n, dtype)) || dense_mfma_fits(n, dtype);
}

// rungs 1 .. 3 of the jitter ladder in one launch where the left-looking kernel factors this size (dense_ll.hip); 1: not here --
// the caller issues the re-Gram and factorisation launches rung by rung
int dense_chol_retry_fused(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n, int dtype,
                           int u_only, const void* z, int z_div, const void* ls, const void* os, const void* noise, const int32_t* n_valid,
                           int y_div, double jitter_base, int P, int f, int kind, hipStream_t stream) {
    const bool mfma_on = g_sw.mfma, fused_on = g_sw.retry_fused;
    if (!(mfma_on && fused_on && ll_enabled() && n >= LL_MIN_N)) return 1;
    return dense_ll_retry_try(A, resid, logp, alpha_out, info, scale, B, n, dtype, 1, 3, u_only, z, z_div, ls, os, noise, n_valid, y_div,
                              jitter_base, P, f, kind, stream);
}

// u_only (only honoured on the MFMA path, i.e. when dense_chol_saves_inverse()): alpha_out receives u = L^-1 r instead of alpha
int dense_chol_launch(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n,
                      int dtype, int attempt, hipStream_t stream, int u_only) {
    const bool mfma_on = g_sw.mfma;
    if (mfma_on && ll_enabled() && n >= LL_MIN_N) {
        int rc = dense_ll_try(A, resid, logp, alpha_out, info, scale, B, n, dtype, attempt, u_only, stream);
        if (rc != 1) return rc;
    }
    if (mfma_on) {
        int rc = dense_mfma_try(A, resid, logp, alpha_out, info, scale, B, n, dtype, attempt, u_only, stream);
        if (rc != 1) return rc;
    }
    size_t lds = ((size_t)n + NB * (NB + 1) + 2 * TT * (NB + 1) + NB + 8) * (dtype == PACOH_F64 ? 8 : 4);
    if (lds > 64u * 1024u) return PACOH_ELIMIT;
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(chol_dense_kernel<float>, dim3(B), dim3(256), lds, stream, (float*)A,
                           (const float*)resid, (float*)logp, (float*)alpha_out, info, (float)scale, n, attempt);
    else
        hipLaunchKernelGGL(chol_dense_kernel<double>, dim3(B), dim3(256), lds, stream, (double*)A,
                           (const double*)resid, (double*)logp, (double*)alpha_out, info, scale, n, attempt);
    return launch_status();
}
}  // namespace pacoh

extern "C" int pacoh_mvn_logprob_dense(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info,
                                       double scale, int B, int n, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!A || !resid || !logp || B <= 0 || n <= 0) return PACOH_EINVAL;
    return dense_chol_launch(A, resid, logp, alpha_out, info, scale, B, n, dtype, 0, (hipStream_t)stream, 0);
}
