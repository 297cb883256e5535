// Dense (large-n) GP path: LML forward + backward and the posterior predictive for context sets that do not fit the
// LDS-resident kernels (gp_mfma.hip / gp_small.hip).  Same mathematics and outputs as pacoh_gp_lml_fwdbwd /
// pacoh_gp_predict (reference call sites: random_gp.py:54-89,204-222, GPR_meta_mll.py:104-117,174-181 and the gpytorch
// ExactGP / ExactMarginalLogLikelihood / psd_safe_cholesky code behind them), with every n x n matrix materialised in HBM:
//
//   A  = os K(z,z) + (noise + jitter) I          gram.hip               (retry of failed problems: regram_failed_kernel)
//   L  = chol(A), alpha = A^-1 r, log-density    dense_mfma.hip / dense.hip (one workgroup per matrix, MFMA panels)
//   Z  = L^-1 in place                           trtri_dense_kernel     (one workgroup per matrix, MFMA panels)
//   W  = Z^T Z = A^-1                            bgemm_kernel           (batched MFMA GEMM from L2, triangular k-ranges)
//   G  = (alpha alpha^T - W) / 2n -> d_z, d_mean, d_lengthscale, d_outputscale, d_noise
//                                                dense_grad_rows_kernel (one wave per row) + dense_finish_kernel
//   predictive: V = Z K_xs, cov = K_ss + noise I - V^T V, mu = m_s + K_xs^T alpha        (bgemm_kernel, small kernels)
//
// fp32 and fp64 (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64).  Limits: the 32-column panel of the Cholesky /
// inverse must fit in LDS (n <= ~1000 fp32, ~520 fp64).
#include "common.h"
#include <stdlib.h>

namespace pacoh {

int dense_chol_launch(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n,
                      int dtype, int attempt, hipStream_t stream, int u_only = 0);                  // dense.hip
bool dense_chol_saves_inverse(int n, int dtype);                                                     // dense.hip
int dense_chol_retry_fused(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n, int dtype,
                           int u_only, const void* z, int z_div, const void* ls, const void* os, const void* noise, const int32_t* n_valid,
                           int y_div, double jitter_base, int P, int f, int kind, hipStream_t stream);           // dense.hip (1: not fused)
int trtri_ll_try(void* A, const int32_t* info, int B, int n, int dtype, hipStream_t s, const void* u, void* alpha);
bool trtri_ll_fits(int n, int dtype);
bool dense_ll_fits(int n, int dtype);                                                                // dense_ll.hip
int dense_ll_try(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n, int dtype,
                 int attempt, int u_only, hipStream_t s);                                            // dense_ll.hip (1: not in its plan)
int dense_grad_mfma_try(const void* zs, const void* ls, const void* os, const int32_t* n_valid, int y_div, const void* g_lml,
                        const void* alpha, const void* Wm, const int32_t* info, void* d_z, void* d_mean, int mean_mode, void* rowpart,
                        void* scratch, size_t scratch_bytes, int B, int P, int n, int f, int kind, int dtype, hipStream_t s);   // dense_grad_mfma.hip            // dense_trtri_ll.hip (1: not in its plan)
bool dense_grad_mfma_plan(int B, int n, int f, int kind, int dtype, size_t scratch_bytes);             // dense_grad_mfma.hip
int gram_rbf_for_chol(const void* z, int z_div, const void* ls, const void* os, const void* noise, void* K, int B, int P, int n, int f,
                      int dtype, hipStream_t s, int lower);                                          // gram.hip
int dense_gram_mfma_try(const void* z, int z_div, const void* ls, const void* os, const void* noise, void* K, int B, int P, int n, int f,
                        int dtype, hipStream_t s);                                                   // dense_grad_mfma.hip (1: not in its plan)
// A = os K + noise I for the factorisation (lower block triangle): fp64 distances on the matrix cores where that kernel applies
static int gram_for_chol(const void* z, int z_div, const void* ls, const void* os, const void* noise, void* K, int B, int P, int n, int f,
                         int dtype, hipStream_t s) {
    const int rc = dense_gram_mfma_try(z, z_div, ls, os, noise, K, B, P, n, f, dtype, s);
    return rc == 1 ? gram_rbf_for_chol(z, z_div, ls, os, noise, K, B, P, n, f, dtype, s, 1) : rc;
}

namespace {

using f32x4_t = __attribute__((ext_vector_type(4))) float;
using f64x4_t = __attribute__((ext_vector_type(4))) double;

template <typename T> struct Mf;
template <> struct Mf<float> {
    using acc = f32x4_t;
    static __device__ __forceinline__ acc mma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int g, int q) { return 4 * g + q; }
};
template <> struct Mf<double> {
    using acc = f64x4_t;
    static __device__ __forceinline__ acc mma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int g, int q) { return g + 4 * q; }
};

constexpr int DNB = 32;            // panel width
constexpr int DLP = 36;            // leading dimension of LDS panels

__device__ __forceinline__ int clamp_nv(const int32_t* n_valid, long ty, int n) {
    int nv = n_valid ? n_valid[ty] : n;
    nv = nv < n ? nv : n;
    return nv < 0 ? 0 : nv;
}

// ---- residual r = y - mean (zero on padded rows) ------------------------------------------------------------------------
template <typename T>
__global__ void dense_resid_kernel(const T* __restrict__ y, int y_div, const T* __restrict__ mean, int mean_mode,
                                   const int32_t* __restrict__ n_valid, T* __restrict__ resid, int P, int n, long total) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total) return;
    const long b = q / n;
    const int i = (int)(q - b * n);
    const long ty = b / y_div;
    const int nv = clamp_nv(n_valid, ty, n);
    T m = 0;
    if (mean_mode == PACOH_MEAN_VECTOR) m = mean[q];
    else if (mean_mode == PACOH_MEAN_CONST) m = mean[b % P];
    resid[q] = i < nv ? y[ty * n + i] - m : T(0);
}

// ---- zs[b,i,c] = z[b / z_div, i, c] / lengthscale[p,c]: the gradient contractions read pre-scaled coordinates ----------------
template <typename T>
__global__ void dense_scale_kernel(const T* __restrict__ z, int z_div, const T* __restrict__ ls, T* __restrict__ zs, int P, int n,
                                   int f, long total) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total) return;
    const int c = (int)(q % f);
    const long bi = q / f;
    const long b = bi / n;
    const int i = (int)(bi - b * n);
    zs[q] = z[((b / z_div) * n + i) * (long)f + c] / ls[(b % P) * (long)f + c];
}

// ---- ragged tasks: rows / columns >= n_valid become an identity block -----------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) dense_mask_kernel(T* __restrict__ A, const int32_t* __restrict__ n_valid, int y_div,
                                                         const int32_t* __restrict__ info, int attempt, int n) {
    const long b = blockIdx.y;
    if (attempt > 0 && info[b] >= 0) return;
    const int nv = clamp_nv(n_valid, b / y_div, n);
    if (nv >= n) return;
    const int i = blockIdx.x;
    T* row = A + (b * n + i) * (long)n;
    for (int j = threadIdx.x; j < n; j += 256)
        if (i >= nv || j >= nv) row[j] = (i == j) ? T(1) : T(0);
}

// rows >= n_valid of a [B, n, m] matrix := 0
template <typename T>
__global__ void __launch_bounds__(256) dense_zero_rows_kernel(T* __restrict__ X, const int32_t* __restrict__ n_valid, int y_div,
                                                              int n, int m) {
    const long b = blockIdx.y;
    const int i = blockIdx.x;
    if (i < clamp_nv(n_valid, b / y_div, n)) return;
    T* row = X + (b * n + i) * (long)m;
    for (int j = threadIdx.x; j < m; j += 256) row[j] = T(0);
}

// ---- jitter-ladder retry: rebuild A = os K + (noise + jitter) I for the problems whose factorisation failed ---------------
template <typename T>
__global__ void __launch_bounds__(256) regram_failed_kernel(const T* __restrict__ z, int z_div, const T* __restrict__ ls,
                                                            const T* __restrict__ os, const T* __restrict__ noise,
                                                            const int32_t* __restrict__ info, T jitter, T* __restrict__ A,
                                                            int P, int n, int f, int kind, int B) {
    // a SMALL grid that walks over the problems (the usual case is "nothing failed": 2048 workgroups that only exit cost 5 us per
    // rung of the ladder, three rungs per evaluation)
    for (long b = blockIdx.y; b < B; b += gridDim.y) {
    if (info[b] >= 0) continue;
    const int p = (int)(b % P);
    const T* zb = z + (b / z_div) * (long)n * f;
    const T osv = os ? os[p] : T(1);
    const T dg = noise[p] + jitter;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {          // few workgroups per problem: the usual case is the early exit above
        T* row = A + (b * n + i) * (long)n;
        for (int j = threadIdx.x; j < n; j += 256) {
            T s = 0;
            for (int c = 0; c < f; ++c) { const T d = zb[(long)i * f + c] / ls[(long)p * f + c] - zb[(long)j * f + c] / ls[(long)p * f + c]; s = fma(d, d, s); }
            row[j] = osv * kern_val<T>(kind, s) + (i == j ? dg : T(0));
        }
    }
    }
}

// ---- Z = L^-1 in place (lower triangle), right-to-left over 32-column panels ------------------------------------------------
//   Z11 = L11^-1 (one wavefront, LDS);  Q = L21 Z11 (MFMA, LDS panel);  Z21 = -Z22 Q (MFMA: Z22 blocks from L2, Q from LDS)
template <typename T, int NT>
__global__ void __launch_bounds__(NT) trtri_dense_kernel(T* __restrict__ A, const int32_t* __restrict__ info, int n, int mpad, int saved_inv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sm = reinterpret_cast<T*>(smem_raw);
    T (*Ds)[DNB + 1] = reinterpret_cast<T (*)[DNB + 1]>(sm);          // L11
    T* Li = sm + DNB * (DNB + 1);                                      // Z11 = L11^-1, [32][DLP]
    T* Pn = Li + DNB * DLP;                                            // panel L21 -> Q, [mpad][DLP]
    using Acc = typename Mf<T>::acc;
    constexpr int NW = NT / 64;
    if (info && info[blockIdx.x] < 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    T* Ab = A + (size_t)blockIdx.x * n * n;
    const int nblk = (n + DNB - 1) / DNB;
    for (int kbk = nblk - 1; kbk >= 0; --kbk) {
        const int k0 = kbk * DNB;
        const int kb = (n - k0 < DNB) ? (n - k0) : DNB;
        const int t0 = k0 + kb, m = n - t0;
        if (saved_inv) {
            // the MFMA Cholesky (dense_mfma.hip) left Z11 behind: strictly lower part transposed in the strictly upper part of the
            // diagonal block, and the diagonal of L11 itself (re-deriving it here: 496 dependent fp64 fmas + 32 divisions per lane
            // by one wave while fifteen wait -- a third of this kernel's time at n = 512)
            for (int q = tid; q < DNB * DNB; q += NT) {
                const int rr = q / DNB, c = q - rr * DNB;
                T v = T(0);
                if (rr < kb && c < rr) v = Ab[(size_t)(k0 + c) * n + k0 + rr];
                if (rr == c) v = rr < kb ? T(1) / Ab[(size_t)(k0 + rr) * n + k0 + rr] : T(1);
                Li[rr * DLP + c] = v;
            }
        } else {
        for (int q = tid; q < DNB * DNB; q += NT) {
            const int rr = q / DNB, c = q - rr * DNB;
            T v = (rr == c) ? T(1) : T(0);
            if (rr < kb && c <= rr) v = Ab[(size_t)(k0 + rr) * n + k0 + c];
            Ds[rr][c] = v;
        }
        __syncthreads();
        if (tid < 64) {                              // lane c owns column c of the inverse (both half-waves compute it)
            const int rr = tid & 31;
            T x[DNB];
#pragma unroll
            for (int i = 0; i < DNB; ++i) {
                T s = (i == rr) ? T(1) : T(0);
#pragma unroll
                for (int j = 0; j < i; ++j) s = fma(-Ds[i][j], x[j], s);
                x[i] = s / Ds[i][i];
            }
            if (tid < 32) {
#pragma unroll
                for (int i = 0; i < DNB; ++i) Li[i * DLP + rr] = x[i];
            }
        }
        }
        __syncthreads();
        for (int q = tid; q < DNB * DNB; q += NT) {
            const int rr = q / DNB, c = q - rr * DNB;
            if (rr < kb && c <= rr) Ab[(size_t)(k0 + rr) * n + k0 + c] = Li[rr * DLP + c];
        }
        if (m > 0) {
            const int mb = (m + 15) / 16;
            for (int q = tid; q < mb * 16 * DNB; q += NT) {
                const int rr = q / DNB, c = q - rr * DNB;
                Pn[(size_t)rr * DLP + c] = (rr < m && c < kb) ? Ab[(size_t)(t0 + rr) * n + k0 + c] : T(0);
            }
            __syncthreads();
            // Q = L21 Z11, in place (a wave owns whole row blocks and reads them completely before writing)
            for (int ib = wave; ib < mb; ib += NW) {
                Acc acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < DNB / 4; ++c) {
                    const T a = Pn[(size_t)(ib * 16 + r) * DLP + 4 * c + g];
                    acc0 = Mf<T>::mma(a, Li[(4 * c + g) * DLP + r], acc0);
                    acc1 = Mf<T>::mma(a, Li[(4 * c + g) * DLP + 16 + r], acc1);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = ib * 16 + Mf<T>::row(g, q);
                    Pn[(size_t)row * DLP + r] = acc0[q];
                    Pn[(size_t)row * DLP + 16 + r] = acc1[q];
                }
            }
            __syncthreads();
            // Z21 = -Z22 Q: row block ib needs the (already inverted) blocks Z22[ib][0..ib]; heavy blocks first
            for (int ib = mb - 1 - wave; ib >= 0; ib -= NW) {
                Acc acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
                const int rowi = ib * 16 + r;
                const T* arow = Ab + (size_t)(t0 + rowi) * n + t0;
                // (the operands of the NEXT 16 columns are requested before this block's MFMAs: the loop otherwise alternates
                //  between an L2 round trip and eight MFMAs)
                T an[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) { const int kc = 4 * c + g; an[c] = (rowi < m && kc <= rowi) ? arow[kc] : T(0); }
                for (int kb2 = 0; kb2 <= ib; ++kb2) {
                    T ac[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) ac[c] = an[c];
                    if (kb2 < ib) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) { const int kc = (kb2 + 1) * 16 + 4 * c + g; an[c] = (rowi < m && kc <= rowi) ? arow[kc] : T(0); }
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int kc = kb2 * 16 + 4 * c + g;
                        acc0 = Mf<T>::mma(ac[c], Pn[(size_t)kc * DLP + r], acc0);
                        acc1 = Mf<T>::mma(ac[c], Pn[(size_t)kc * DLP + 16 + r], acc1);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = ib * 16 + Mf<T>::row(g, q);
                    if (row < m) {
                        if (r < kb) Ab[(size_t)(t0 + row) * n + k0 + r] = -acc0[q];
                        if (16 + r < kb) Ab[(size_t)(t0 + row) * n + k0 + 16 + r] = -acc1[q];
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- batched GEMM on the matrix cores, operands straight from HBM/L2 --------------------------------------------------------
//   C[b] (M x N) = alpha op(A[b]) op(B[b]) + beta C[b];  op = identity or transpose of the STORED matrix;
//   lowerA / lowerB: the stored matrix is lower triangular (entries above its diagonal are ignored and the k-range of every
//   output tile is clipped accordingly).  One wave per 32x32 output tile, k consumed 16 at a time with the permuted
//   k index (k = 4g + s) so that both operands of the four MFMAs of a chunk come from consecutive addresses per lane.
struct GemmArgs {
    const void* A; const void* B; void* C;
    long sA, sB, sC;
    int lda, ldb, ldc, M, N, K;
    int transA, transB, lowerA, lowerB;
    double alpha, beta;
    const int32_t* info;         // optional: skip problems with info[b] < 0
    int symC;                    // C is symmetric: only tiles on / below the diagonal are computed, then mirrored
};

template <typename T>
__device__ __forceinline__ T gemm_ld(const T* __restrict__ X, int ld, int row, int col, int R, int Cn, bool lower) {
    if (row >= R || col >= Cn || (lower && col > row)) return T(0);
    return X[(long)row * ld + col];
}

template <typename T>
__global__ void __launch_bounds__(256) bgemm_kernel(GemmArgs ga) {
    using Acc = typename Mf<T>::acc;
    const int b = blockIdx.y;
    if (ga.info && ga.info[b] < 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int tiles_n = (ga.N + 63) / 64;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    if (ga.symC && tn > tm) return;
    const int i0 = tm * 64 + (wave >> 1) * 32, j0 = tn * 64 + (wave & 1) * 32;
    if (i0 >= ga.M || j0 >= ga.N) return;
    const T* A = (const T*)ga.A + (long)b * ga.sA;
    const T* Bm = (const T*)ga.B + (long)b * ga.sB;
    T* C = (T*)ga.C + (long)b * ga.sC;
    // stored shapes
    const int Ar = ga.transA ? ga.K : ga.M, Ac = ga.transA ? ga.M : ga.K;
    const int Br = ga.transB ? ga.N : ga.K, Bc = ga.transB ? ga.K : ga.N;
    int klo = 0, khi = ga.K;
    if (ga.lowerA) { if (ga.transA) klo = max(klo, i0); else khi = min(khi, i0 + 32); }
    if (ga.lowerB) { if (ga.transB) khi = min(khi, j0 + 32); else klo = max(klo, j0); }
    klo &= ~15;
    Acc acc[2][2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) acc[ib][jb] = Acc{0, 0, 0, 0};
    const bool edge_free = i0 + 32 <= ga.M && j0 + 32 <= ga.N;
    for (int kk = klo; kk < khi; kk += 16) {
        T av[2][4], bv[2][4];
        // chunk entirely inside the matrices and inside the stored triangles: plain loads, no per-element tests
        const bool ina = !ga.lowerA || (ga.transA ? kk >= i0 + 31 : kk + 15 <= i0);
        const bool inb = !ga.lowerB || (ga.transB ? kk + 15 <= j0 : kk >= j0 + 31);
        if (edge_free && kk + 16 <= ga.K && ina && inb) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = kk + 4 * g + s;
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int i = i0 + 16 * ib + r;
                    av[ib][s] = ga.transA ? A[(long)k * ga.lda + i] : A[(long)i * ga.lda + k];
                }
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const int j = j0 + 16 * jb + r;
                    bv[jb][s] = ga.transB ? Bm[(long)j * ga.ldb + k] : Bm[(long)k * ga.ldb + j];
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = kk + 4 * g + s;
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int i = i0 + 16 * ib + r;
                    av[ib][s] = ga.transA ? gemm_ld<T>(A, ga.lda, k, i, Ar, Ac, ga.lowerA) : gemm_ld<T>(A, ga.lda, i, k, Ar, Ac, ga.lowerA);
                }
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    const int j = j0 + 16 * jb + r;
                    bv[jb][s] = ga.transB ? gemm_ld<T>(Bm, ga.ldb, j, k, Br, Bc, ga.lowerB) : gemm_ld<T>(Bm, ga.ldb, k, j, Br, Bc, ga.lowerB);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) acc[ib][jb] = Mf<T>::mma(av[ib][s], bv[jb][s], acc[ib][jb]);
    }
    const T alpha = (T)ga.alpha, beta = (T)ga.beta;
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + 16 * ib + Mf<T>::row(g, q), j = j0 + 16 * jb + r;
                if (i < ga.M && j < ga.N) {
                    T* cp = C + (long)i * ga.ldc + j;
                    const T v = beta == T(0) ? alpha * acc[ib][jb][q] : fma(alpha, acc[ib][jb][q], beta * *cp);
                    *cp = v;
                    if (ga.symC && tm != tn) C[(long)j * ga.ldc + i] = v;
                }
            }
}

// ---- W = Z^T Z for lower-triangular Z (the inverse Cholesky factor): LDS-tiled ------------------------------------------------
// One workgroup (4 waves) per 128 x 128 tile of the lower block triangle of W, k in slabs of 16 rows of Z: both operands of a
// slab are ROW segments of Z (Z[k][i0..i0+128), Z[k][j0..j0+128)), loaded once per workgroup with coalesced 128-element rows
// into a double-buffered LDS image (entries above Z's diagonal as zeros) and read from there by the four waves, each of which
// holds a 64 x 64 sub-tile as 4 x 4 accumulator blocks.  Against the direct-from-L2 bgemm_kernel above (one wave per 32 x 32
// tile, every wave fetching its own operands: ~55 B/clk/CU of L2 traffic, the kernel's bound) this moves 4x fewer bytes per MFMA.
// k runs from the tile's first row (Z[k][i] = 0 for k < i) to n.  Tiles above the diagonal are mirrored from below.
// Round 4: (a) the blocks a wave multiplies in a slab are a compile-time mask (ztz_slab<MASK>: the rows the slab has reached x the
// wave's pattern in a diagonal tile) instead of sixteen tested bits around sixteen MFMAs; (b) slabs below the tile's own diagonal
// range are staged by 16-byte loads without the triangle test; (c) the tiles of one matrix run on ONE XCD (workgroup ids are dealt
// round-robin to the eight XCDs, each with its own L2: with a plain (tile, matrix) grid the ten tiles of a matrix fetched the same
// rows of Z from HBM up to eight times).
template <typename T, unsigned MASK, int LDT>
__device__ __forceinline__ void ztz_slab(const T (*__restrict__ Asb)[LDT], const T (*__restrict__ Bsb)[LDT], typename Mf<T>::acc (&acc)[4][4],
                                         int ca, int cb, int g) {
    constexpr unsigned rows = ((MASK & 0xFu) ? 1u : 0u) | ((MASK & 0xF0u) ? 2u : 0u) | ((MASK & 0xF00u) ? 4u : 0u) | ((MASK & 0xF000u) ? 8u : 0u);
    constexpr unsigned cols = (MASK | MASK >> 4 | MASK >> 8 | MASK >> 12) & 0xFu;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        T av[4], bv[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) if (rows >> ib & 1u) av[ib] = Asb[4 * g + s][ca + 32 * ib];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) if (cols >> jb & 1u) bv[jb] = Bsb[4 * g + s][cb + 32 * jb];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int jb = 0; jb < 4; ++jb)
                if (MASK >> (4 * ib + jb) & 1u) acc[ib][jb] = Mf<T>::mma(av[ib], bv[jb], acc[ib][jb]);
    }
}

template <typename T>
__global__ void __launch_bounds__(256, 2) ztz_kernel(const T* __restrict__ Zall, T* __restrict__ Wall, int n, const int32_t* __restrict__ info,
                                                     int nt, int Bn, int mirror) {
    using Acc = typename Mf<T>::acc;
    constexpr int TS = 128, KS = 16, LDT = TS + 4;
    constexpr int VE = 16 / (int)sizeof(T);                        // elements per 16-byte chunk
    typedef T VT __attribute__((ext_vector_type(VE)));
    __shared__ __attribute__((aligned(16))) T As[2][KS][LDT];
    __shared__ __attribute__((aligned(16))) T Bs[2][KS][LDT];
    // workgroup id -> (matrix, tile): ids L, L + 8, L + 16, ... (one XCD) walk the tiles of one matrix
    int b, tile;
    {
        const int L = blockIdx.x, B8 = Bn & ~7;
        if (L < nt * B8) { const int slot = L >> 3; b = (L & 7) + 8 * (slot / nt); tile = slot % nt; }
        else { const int Lr = L - nt * B8; b = B8 + Lr / nt; tile = Lr % nt; }
    }
    if (info && info[b] < 0) return;
    // lower-triangle tile index -> (tm, tn), tn <= tm
    int tm = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
    while (tm * (tm + 1) / 2 > tile) --tm;
    while ((tm + 1) * (tm + 2) / 2 <= tile) ++tm;
    const int tn = tile - tm * (tm + 1) / 2;
    const int i0 = tm * TS, j0 = tn * TS;
    const T* Z = Zall + (long)b * n * n;
    T* W = Wall + (long)b * n * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    // a wave holds 4 x 4 accumulator blocks of 16 x 16: every second row block and every second column block of the tile
    // (row blocks wr, wr+2, wr+4, wr+6; column blocks wc, wc+2, ...).  Interleaved, not a 64 x 64 quadrant: Z is lower triangular,
    // a row block contributes only from the slab that reaches its first row on, and in a diagonal tile the blocks above the
    // diagonal are never stored -- with quadrants the wave holding the top-left one did every MFMA and paced the workgroup
    // (0.61 -> 0.55 ms by skipping in the others); interleaved, all four skip alike.  In a diagonal tile block (ib, jb) is on or
    // below the diagonal iff wc + 2 jb <= wr + 2 ib: jb <= ib for three of the waves, jb < ib for wave (0, 1).
    const int wr = wave >> 1, wc = wave & 1;
    const int pat = tm != tn ? 0 : (wr == 0 && wc == 1 ? 2 : 1);
    Acc acc[4][4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = Acc{0, 0, 0, 0};
    // staging: a slab operand is KS rows of TS elements = KS * CPR 16-byte chunks; thread t moves chunks t, t + 256, ...: a wave
    // reads and writes one whole row segment per instruction (1 KiB contiguous in HBM, conflict-free in LDS; the first version gave
    // each thread eight consecutive elements of a row -- its LDS writes were 4-way bank-conflicted: SQ_LDS_BANK_CONFLICT three
    // times SQ_ACTIVE_INST_LDS)
    constexpr int CPR = TS / VE, NCH = KS * CPR / 256;
    // Loads are unconditional, from clamped addresses, and nothing touches the loaded values before store_slab (a select or a
    // conditional load makes the compiler wait for the data where it is requested, i.e. before the slab's MFMAs instead of
    // after them: 418 -> 390 us); the triangle / edge mask is applied when the slab is written to LDS, and only in the slabs that
    // need one.  (Staging a diagonal tile's one operand once -- B fragments read from the A image -- was slower: 426 us.  With
    // the staging compiled out the kernel takes 311 us, without its barriers 297: the loads cost 55 us, the LDS writes 24.)
    const bool vec_ok = ((long)n * sizeof(T)) % 16 == 0 && i0 + TS <= n;     // (j0 + TS <= i0 + TS)
    auto load_slab = [&](int k0, VT (&ra)[NCH], VT (&rb)[NCH]) __attribute__((always_inline)) {
        if (vec_ok) {
#pragma unroll
            for (int v = 0; v < NCH; ++v) {
                const int c = (int)threadIdx.x + 256 * v, k = k0 + c / CPR, col = (c % CPR) * VE;
                const int kc = k < n ? k : n - 1;
                ra[v] = *(const VT*)(Z + (long)kc * n + i0 + col);
                rb[v] = *(const VT*)(Z + (long)kc * n + j0 + col);
            }
        } else {
#pragma unroll
            for (int v = 0; v < NCH; ++v) {
                const int c = (int)threadIdx.x + 256 * v, k = k0 + c / CPR, col = (c % CPR) * VE;
                const int kc = k < n ? k : n - 1;
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const int ci = i0 + col + e, cj = j0 + col + e;
                    ra[v][e] = Z[(long)kc * n + (ci < n ? ci : n - 1)];
                    rb[v][e] = Z[(long)kc * n + (cj < n ? cj : n - 1)];
                }
            }
        }
    };
    auto store_slab = [&](int buf, int k0, VT (&ra)[NCH], VT (&rb)[NCH]) __attribute__((always_inline)) {
        if (!vec_ok || k0 < i0 + TS || k0 + KS > n) {
#pragma unroll
            for (int v = 0; v < NCH; ++v) {
                const int c = (int)threadIdx.x + 256 * v, k = k0 + c / CPR, col = (c % CPR) * VE;
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const int ci = i0 + col + e, cj = j0 + col + e;
                    ra[v][e] = (k < n && ci < n && ci <= k) ? ra[v][e] : T(0);
                    rb[v][e] = (k < n && cj < n && cj <= k) ? rb[v][e] : T(0);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < NCH; ++v) {
            const int c = (int)threadIdx.x + 256 * v, row = c / CPR, col = (c % CPR) * VE;
            *(VT*)&As[buf][row][col] = ra[v];
            *(VT*)&Bs[buf][row][col] = rb[v];
        }
    };
    const int kbeg = i0;                                           // (i0 >= j0: rows above the tile's first row contribute nothing)
    const int ca = 16 * wr + r, cb = 16 * wc + r;
    VT ra[NCH], rb[NCH];
    load_slab(kbeg, ra, rb);
    store_slab(0, kbeg, ra, rb);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < n; k0 += KS) {
        const bool more = k0 + KS < n;
        if (more) load_slab(k0 + KS, ra, rb);                      // global loads of the next slab fly under this slab's MFMAs
        // row blocks this slab reaches (slab rows k0 .. k0+15 against the block's first row i0 + 16 (wr + 2 ib)): a prefix of nl
        const int reach = (k0 - i0) / 16 - wr;
        const int nl = reach < 0 ? 0 : (reach >= 6 ? 4 : reach / 2 + 1);
#define PACOH_ZTZ(M) ztz_slab<T, M, LDT>(As[buf], Bs[buf], acc, ca, cb, g)
#define PACOH_ZTZ_NL(F, TRI, STR) do { if (pat == 0) PACOH_ZTZ(F); else if (pat == 1) PACOH_ZTZ(TRI); else PACOH_ZTZ(STR); } while (0)
        switch (nl) {
            case 4: PACOH_ZTZ_NL(0xFFFFu, 0xF731u, 0x7310u); break;
            case 3: PACOH_ZTZ_NL(0x0FFFu, 0x0731u, 0x0310u); break;
            case 2: PACOH_ZTZ_NL(0x00FFu, 0x0031u, 0x0010u); break;
            case 1: if (pat == 0) PACOH_ZTZ(0x000Fu); else if (pat == 1) PACOH_ZTZ(0x0001u); break;
            default: break;
        }
#undef PACOH_ZTZ_NL
#undef PACOH_ZTZ
        if (more) store_slab(buf ^ 1, k0 + KS, ra, rb);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + 16 * (wr + 2 * ib) + Mf<T>::row(g, q), j = j0 + 16 * (wc + 2 * jb) + r;
                if (i < n && j < n && (tm != tn || j <= i)) {
                    W[(long)i * n + j] = acc[ib][jb][q];
                    if (mirror || tm == tn) W[(long)j * n + i] = acc[ib][jb][q];      // (mirror == 0: the reader takes the lower block triangle only)
                }
            }
}

template <typename T>
void launch_ztz(const T* Z, T* W, int n, const int32_t* info, int Bn, hipStream_t s, int mirror) {
    const int t = (n + 127) / 128, nt = t * (t + 1) / 2;
    hipLaunchKernelGGL(ztz_kernel<T>, dim3((unsigned)(nt * Bn)), dim3(256), 0, s, Z, W, n, info, nt, Bn, mirror);
}

// ---- the general batched GEMM, LDS-tiled (round 5) ---------------------------------------------------------------------------
// ztz_kernel's tile scheme for ANY op(A) op(B) of GemmArgs: one workgroup per 128 x 128 tile of C, k in slabs of 16, both operands of
// a slab through a double-buffered LDS image [k][128] read by four waves of 4 x 4 accumulator blocks each (ztz_slab), the next
// slab's global loads in flight under this slab's MFMAs.  An operand stored k-major (transA / !transB: a slab row is contiguous) is
// staged by 16-byte loads and 16-byte LDS stores; stored i-major (!transA / transB: 16 bytes = 4 / 2 consecutive k of one row) it
// is transposed on the way into LDS (consecutive lanes = consecutive rows: conflict-free scalar stores).  Stored-lower operands
// clip the tile's k range and are masked where a slab crosses their diagonal.  (Tried on the n = 784 fp32 two-level call: loads two
// slabs ahead through a second register set -- 256 registers, two workgroups per CU: 2.34 -> 2.48 ms at 128 problems, 1.20 -> 1.19 at
// 16; edge tiles that multiply only the row blocks inside the matrix, as a switch over seven ztz_slab masks -- 206 registers in
// fp32, 347 spilled in fp64: not measured.)  bgemm_kernel above -- one wave per 32 x 32 tile,
// operands straight from L2 -- ran the predictive's V = Z K_xs at 18 TFLOP/s in fp32; it stays for outputs narrower than a tile.
template <typename T>
__global__ void __launch_bounds__(256, 2) gemm_tile_kernel(GemmArgs ga, int tiles_n, int nt, int Bn) {
    using Acc = typename Mf<T>::acc;
    constexpr int TS = 128, KS = 16, LDT = TS + 4;
    constexpr int VE = 16 / (int)sizeof(T);
    typedef T VT __attribute__((ext_vector_type(VE)));
    __shared__ __attribute__((aligned(16))) T As[2][KS][LDT];
    __shared__ __attribute__((aligned(16))) T Bs[2][KS][LDT];
    int b, tile;                                                   // ids L, L + 8, ... (one XCD) walk the tiles of one matrix
    {
        const int L = blockIdx.x, B8 = Bn & ~7;
        if (L < nt * B8) { const int slot = L >> 3; b = (L & 7) + 8 * (slot / nt); tile = slot % nt; }
        else { const int Lr = L - nt * B8; b = B8 + Lr / nt; tile = Lr % nt; }
    }
    if (ga.info && ga.info[b] < 0) return;
    int tm, tn;
    if (ga.symC) {
        tm = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
        while (tm * (tm + 1) / 2 > tile) --tm;
        while ((tm + 1) * (tm + 2) / 2 <= tile) ++tm;
        tn = tile - tm * (tm + 1) / 2;
    } else { tm = tile / tiles_n; tn = tile - tm * tiles_n; }
    const int i0 = tm * TS, j0 = tn * TS;
    const T* A = (const T*)ga.A + (long)b * ga.sA;
    const T* Bm = (const T*)ga.B + (long)b * ga.sB;
    T* C = (T*)ga.C + (long)b * ga.sC;
    const int M = ga.M, N = ga.N, K = ga.K, lda = ga.lda, ldb = ga.ldb;
    const bool tA = ga.transA != 0, tB = ga.transB != 0, lA = ga.lowerA != 0, lB = ga.lowerB != 0;
    int klo = 0, khi = K;
    if (lA) { if (tA) klo = max(klo, i0); else khi = min(khi, i0 + TS); }
    if (lB) { if (tB) khi = min(khi, j0 + TS); else klo = max(klo, j0); }
    klo &= ~(KS - 1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    Acc acc[4][4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) acc[ib][jb] = Acc{0, 0, 0, 0};
    constexpr int NCH = KS * TS / VE / 256;                          // 16-byte chunks per thread, slab and operand
    // chunk c of a slab: k-major storage -> (k = c / CPR, columns (c % CPR) VE ..); i-major -> (row c % TS, k = (c / TS) VE ..)
    constexpr int CPR = TS / VE;
    const bool vecA = ((long)lda * sizeof(T)) % 16 == 0 && ((size_t)A % 16) == 0 && (tA ? i0 + TS <= M : true);
    const bool vecB = ((long)ldb * sizeof(T)) % 16 == 0 && ((size_t)Bm % 16) == 0 && (tB ? true : j0 + TS <= N);
    // unconditional loads from clamped addresses (nothing tests the loaded values before the slab is written to LDS: ztz_kernel's note)
    auto load_op = [&](const T* X, int ld, bool kmajor, int dim, int o0, bool vec, int k0, VT (&rg)[NCH]) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NCH; ++v) {
            const int c = (int)threadIdx.x + 256 * v;
            if (kmajor) {
                const int k = k0 + c / CPR, col = o0 + (c % CPR) * VE;
                const int kc = k < K ? k : K - 1;
                if (vec) rg[v] = *(const VT*)(X + (long)kc * ld + col);
                else {
#pragma unroll
                    for (int e = 0; e < VE; ++e) rg[v][e] = X[(long)kc * ld + (col + e < dim ? col + e : dim - 1)];
                }
            } else {
                const int row = o0 + c % TS, k = k0 + (c / TS) * VE;
                const int rc = row < dim ? row : dim - 1;
                if (vec && k + VE <= K) rg[v] = *(const VT*)(X + (long)rc * ld + k);
                else {
#pragma unroll
                    for (int e = 0; e < VE; ++e) rg[v][e] = X[(long)rc * ld + (k + e < K ? k + e : K - 1)];
                }
            }
        }
    };
    // lower: (idx, k) is stored iff  kmajor ? idx <= k : k <= idx
    auto store_op = [&](T (*Xs)[LDT], bool kmajor, int dim, int o0, bool lower, int k0, VT (&rg)[NCH]) __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < NCH; ++v) {
            const int c = (int)threadIdx.x + 256 * v;
            if (kmajor) {
                const int kl = c / CPR, cl = (c % CPR) * VE, k = k0 + kl;
                VT x = rg[v];
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const int idx = o0 + cl + e;
                    if (k >= K || idx >= dim || (lower && idx > k)) x[e] = T(0);
                }
                *(VT*)&Xs[kl][cl] = x;
            } else {
                const int rl = c % TS, kl = (c / TS) * VE, idx = o0 + rl;
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const int k = k0 + kl + e;
                    Xs[kl + e][rl] = (k >= K || idx >= dim || (lower && k > idx)) ? T(0) : rg[v][e];
                }
            }
        }
    };
    const int ca = 16 * wr + r, cb = 16 * wc + r;
    VT ra[NCH], rb[NCH];
    if (klo < khi) {
        load_op(A, lda, tA, M, i0, vecA, klo, ra);
        load_op(Bm, ldb, !tB, N, j0, vecB, klo, rb);
        store_op(As[0], tA, M, i0, lA, klo, ra);
        store_op(Bs[0], !tB, N, j0, lB, klo, rb);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = klo; k0 < khi; k0 += KS) {
        const bool more = k0 + KS < khi;
        if (more) { load_op(A, lda, tA, M, i0, vecA, k0 + KS, ra); load_op(Bm, ldb, !tB, N, j0, vecB, k0 + KS, rb); }
        ztz_slab<T, 0xFFFFu, LDT>(As[buf], Bs[buf], acc, ca, cb, g);
        if (more) { store_op(As[buf ^ 1], tA, M, i0, lA, k0 + KS, ra); store_op(Bs[buf ^ 1], !tB, N, j0, lB, k0 + KS, rb); }
        __syncthreads();
        buf ^= 1;
    }
    const T alpha = (T)ga.alpha, beta = (T)ga.beta;
    const int ldc = ga.ldc;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib)
#pragma unroll
        for (int jb = 0; jb < 4; ++jb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + 16 * (wr + 2 * ib) + Mf<T>::row(g, q), j = j0 + 16 * (wc + 2 * jb) + r;
                if (i < M && j < N) {
                    T* cp = C + (long)i * ldc + j;
                    const T v = beta == T(0) ? alpha * acc[ib][jb][q] : fma(alpha, acc[ib][jb][q], beta * *cp);
                    *cp = v;
                    if (ga.symC && tm != tn) C[(long)j * ldc + i] = v;
                }
            }
}

template <typename T>
void launch_bgemm(const GemmArgs& ga, int Bn, hipStream_t s, int mirror = 1) {
    if (ga.A == ga.B && ga.transA && !ga.transB && ga.lowerA && ga.lowerB && ga.symC && ga.M == ga.N && ga.M == ga.K &&
        ga.lda == ga.M && ga.ldb == ga.M && ga.ldc == ga.M && ga.alpha == 1.0 && ga.beta == 0.0 && ga.M >= 128) {
        launch_ztz<T>((const T*)ga.A, (T*)ga.C, ga.M, ga.info, Bn, s, mirror);
        return;
    }
    const bool tile_on = g_sw.gemm_tile;
    if (tile_on && ga.M >= 96 && ga.N >= 96 && ga.K >= 32 && (const void*)ga.C != ga.A && (const void*)ga.C != ga.B) {
        const int tm = (ga.M + 127) / 128, tn = (ga.N + 127) / 128;
        const int nt = ga.symC ? tm * (tm + 1) / 2 : tm * tn;
        hipLaunchKernelGGL(gemm_tile_kernel<T>, dim3((unsigned)(nt * Bn)), dim3(256), 0, s, ga, tn, nt, Bn);
        return;
    }
    const int tiles = ((ga.M + 63) / 64) * ((ga.N + 63) / 64);
    hipLaunchKernelGGL(bgemm_kernel<T>, dim3(tiles, Bn), dim3(256), 0, s, ga);
}

// ---- gradient sums, one wave per matrix row -----------------------------------------------------------------------------------
//   rowpart[b, i, 0..f) = sum_j M_ij df_c^2,  [f] = sum_j G_ij e_ij,  [f+1] = G_ii,  [f+2] = alpha_i
// A workgroup (4 waves) owns ROWS consecutive rows of one problem.  With LDS = true the problem's pre-scaled coordinates
// (feature-major, so that consecutive lanes read consecutive words) and alpha are staged in LDS once per workgroup; without
// it every row re-read them through L1/L2 (n*f*8 bytes per row: 4.3 GB of cache traffic at n = 512, d = 8, 256 problems --
// that, not the arithmetic, bounded the kernel).
template <typename T, int FP, bool LDS>
__global__ void __launch_bounds__(256) dense_grad_rows_kernel(const T* __restrict__ zs, const T* __restrict__ lsp,
                                                              const T* __restrict__ osp, const int32_t* __restrict__ n_valid,
                                                              int y_div, const T* __restrict__ g_lml, const T* __restrict__ alpha,
                                                              const T* __restrict__ Wm, const int32_t* __restrict__ info,
                                                              T* __restrict__ d_z, T* __restrict__ d_mean, int mean_mode,
                                                              T* __restrict__ rowpart, int P, int n, int f, int rows, int kind) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* zt = reinterpret_cast<T*>(smem_raw);          // [f][n]  (LDS only)
    T* al = zt + (size_t)f * n;                       // [n]
    const long b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = (int)(b % P);
    const int nv = clamp_nv(n_valid, b / y_div, n);
    const bool failed = info[b] < 0;
    const T gup = g_lml ? g_lml[b] : T(1);
    const int W3 = f + 3;
    const T* zb = zs + b * (long)n * f;
    const T* ab = alpha + b * (long)n;
    if (LDS) {
        for (int e = threadIdx.x; e < n * f; e += 256) { const int j = e / f, c = e - j * f; zt[c * n + j] = zb[e]; }
        for (int j = threadIdx.x; j < n; j += 256) al[j] = ab[j];
        __syncthreads();
    }
    T ls[FP];
#pragma unroll
    for (int c = 0; c < FP; ++c) ls[c] = c < f ? lsp[(long)p * f + c] : T(1);
    const T os = osp ? osp[p] : T(1);
    const T inv2n = nv > 0 ? T(0.5) / T(nv) : T(0);
    const int i_end = min(n, (int)(blockIdx.x + 1) * rows);
    for (int i = blockIdx.x * rows + wave; i < i_end; i += 4) {
        T* rp = rowpart + (b * n + i) * (long)W3;
        if (failed || i >= nv) {
            const T v = failed ? T(NAN) : T(0);
            if (lane == 0) {
                if (d_z) for (int c = 0; c < f; ++c) d_z[(b * n + i) * (long)f + c] = v;
                if (d_mean && mean_mode == PACOH_MEAN_VECTOR) d_mean[b * n + i] = v;
                for (int c = 0; c < W3; ++c) rp[c] = v;
            }
            continue;
        }
        T zi[FP];
#pragma unroll
        for (int c = 0; c < FP; ++c) zi[c] = c < f ? (LDS ? zt[c * n + i] : zb[(long)i * f + c]) : T(0);
        const T ai = LDS ? al[i] : ab[i];
        const T* wrow = Wm + (b * n + i) * (long)n;
        T dz[FP], dls[FP];
#pragma unroll
        for (int c = 0; c < FP; ++c) { dz[c] = 0; dls[c] = 0; }
        T dos = 0, dnz = 0;
        for (int j = lane; j < nv; j += 64) {
            const T Gij = (ai * (LDS ? al[j] : ab[j]) - wrow[j]) * inv2n;
            T s = 0, df[FP];
#pragma unroll
            for (int c = 0; c < FP; ++c) {
                df[c] = c < f ? (LDS ? zt[c * n + j] : zb[(long)j * f + c]) - zi[c] : T(0);
                s = fma(df[c], df[c], s);
            }
            T e, ed;                                   // k / os and the weight of (z_j - z_i) in its derivative (common.h: kern_eval)
            kern_eval<T>(kind, s, e, ed);
            dos = fma(Gij, e, dos);
            const T M = Gij * os * ed;
#pragma unroll
            for (int c = 0; c < FP; ++c) { const T md = M * df[c]; dz[c] += md; dls[c] = fma(md, df[c], dls[c]); }
            if (j == i) dnz = Gij;
        }
#pragma unroll
        for (int c = 0; c < FP; ++c) { dz[c] = subwave_sum<T>(dz[c], 64); dls[c] = subwave_sum<T>(dls[c], 64); }
        dos = subwave_sum<T>(dos, 64);
        dnz = subwave_sum<T>(dnz, 64);
        if (lane == 0) {
#pragma unroll
            for (int c = 0; c < FP; ++c) {
                if (c < f) {
                    if (d_z) d_z[(b * n + i) * (long)f + c] = T(2) * gup * dz[c] / ls[c];
                    rp[c] = dls[c];
                }
            }
            rp[f] = dos; rp[f + 1] = dnz; rp[f + 2] = ai;
            if (d_mean && mean_mode == PACOH_MEAN_VECTOR) d_mean[b * n + i] = gup * ai / T(nv);
        }
    }
}

// Same sums with the roles turned: a LANE owns row i (64 rows per workgroup), the four waves split the columns j, and W[i][j] is
// read as W[j][i] (W = K^-1 is symmetric: consecutive lanes, consecutive addresses).  Everything a row accumulates stays in its
// lane -- the row-per-wave kernel above spends as long in its 2 f + 2 cross-lane reductions per row (216 ds_bpermute at f = 8 in
// fp64) as in the 8 entries per lane they follow -- and z_j, alpha_j are LDS broadcasts.  Needs the coordinates in LDS.
template <typename T, int FP>
__global__ void __launch_bounds__(256) dense_grad_cols_kernel(const T* __restrict__ zs, const T* __restrict__ lsp,
                                                              const T* __restrict__ osp, const int32_t* __restrict__ n_valid,
                                                              int y_div, const T* __restrict__ g_lml, const T* __restrict__ alpha,
                                                              const T* __restrict__ Wm, const int32_t* __restrict__ info,
                                                              T* __restrict__ d_z, T* __restrict__ d_mean, int mean_mode,
                                                              T* __restrict__ rowpart, int P, int n, int f, int kind) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* zt = reinterpret_cast<T*>(smem_raw);          // [f][n]
    T* al = zt + (size_t)f * n;                       // [n]
    T* red = al + n;                                  // [3][64][2 FP + 2]: partial sums of waves 1..3
    const long b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = (int)(b % P);
    const int nv = clamp_nv(n_valid, b / y_div, n);
    const bool failed = info[b] < 0;
    const T gup = g_lml ? g_lml[b] : T(1);
    const int W3 = f + 3;
    const T* zb = zs + b * (long)n * f;
    const T* ab = alpha + b * (long)n;
    for (int e = threadIdx.x; e < n * f; e += 256) { const int j = e / f, c = e - j * f; zt[c * n + j] = zb[e]; }
    for (int j = threadIdx.x; j < n; j += 256) al[j] = ab[j];
    __syncthreads();
    const T os = osp ? osp[p] : T(1);
    const T inv2n = nv > 0 ? T(0.5) / T(nv) : T(0);
    const int i = blockIdx.x * 64 + lane;
    const bool live = i < nv && !failed;
    const int ic = i < n ? i : n - 1;
    T zi[FP];
#pragma unroll
    for (int c = 0; c < FP; ++c) zi[c] = c < f ? zt[c * n + ic] : T(0);
    const T ai = al[ic];
    const T* wcol = Wm + b * (long)n * n + ic;        // W[j][i] at wcol[j * n]
    T dz[FP], dls[FP];
#pragma unroll
    for (int c = 0; c < FP; ++c) { dz[c] = 0; dls[c] = 0; }
    T dos = 0, dnz = 0;
    if (!failed) {
#pragma unroll 2
        for (int j = wave; j < nv; j += 4) {
            const T Gij = (ai * al[j] - wcol[(long)j * n]) * inv2n;
            T s = 0, df[FP];
#pragma unroll
            for (int c = 0; c < FP; ++c) {
                df[c] = c < f ? zt[c * n + j] - zi[c] : T(0);
                s = fma(df[c], df[c], s);
            }
            T e, ed;
            kern_eval<T>(kind, s, e, ed);
            dos = fma(Gij, e, dos);
            const T M = Gij * os * ed;
            // (the lengthscale sums sum_j M_ij (z_j - z_i)_c^2 are not accumulated: M being symmetric, their total over i equals
            //  -2 sum_i (z_i - z_0)_c dz_i[c] -- one multiply per row on the finished d_z sums below; round 3, as in gp_reg.hip)
#pragma unroll
            for (int c = 0; c < FP; ++c) dz[c] = fma(M, df[c], dz[c]);
            if (j == i) dnz = Gij;
        }
    }
    constexpr int NR = 2 * FP + 2;
    if (wave > 0) {
        T* r = red + ((size_t)(wave - 1) * 64 + lane) * NR;
#pragma unroll
        for (int c = 0; c < FP; ++c) { r[c] = dz[c]; r[FP + c] = dls[c]; }
        r[2 * FP] = dos; r[2 * FP + 1] = dnz;
    }
    __syncthreads();
    if (wave == 0 && i < n) {
#pragma unroll
        for (int w = 0; w < 3; ++w) {                 // fixed order: deterministic
            const T* r = red + ((size_t)w * 64 + lane) * NR;
#pragma unroll
            for (int c = 0; c < FP; ++c) dz[c] += r[c];
            dos += r[2 * FP]; dnz += r[2 * FP + 1];
        }
#pragma unroll
        for (int c = 0; c < FP; ++c) dls[c] = c < f ? T(-2) * (zi[c] - zt[c * n]) * dz[c] : T(0);
        T* rp = rowpart + (b * n + i) * (long)W3;
        if (!live) {
            const T v = failed ? T(NAN) : T(0);
            if (d_z) for (int c = 0; c < f; ++c) d_z[(b * n + i) * (long)f + c] = v;
            if (d_mean && mean_mode == PACOH_MEAN_VECTOR) d_mean[b * n + i] = v;
            for (int c = 0; c < W3; ++c) rp[c] = v;
        } else {
#pragma unroll
            for (int c = 0; c < FP; ++c) {
                if (c < f) {
                    if (d_z) d_z[(b * n + i) * (long)f + c] = T(2) * gup * dz[c] / lsp[(long)p * f + c];
                    rp[c] = dls[c];
                }
            }
            rp[f] = dos; rp[f + 1] = dnz; rp[f + 2] = ai;
            if (d_mean && mean_mode == PACOH_MEAN_VECTOR) d_mean[b * n + i] = gup * ai / T(nv);
        }
    }
}

// ---- per-problem results: lml and the reduced hyper-parameter gradients; one 256-thread workgroup per problem, a thread per row (its
// f + 3 values are contiguous), wave sums then the four waves in order (fixed: deterministic).  One wave reading column by column
// with an 88-byte stride took 19 us per 256 x 512 launch -- as long as the 256 x 512^2 Gram kernel's tail.
template <typename T>
__global__ void __launch_bounds__(256) dense_finish_kernel(const T* __restrict__ logp, const T* __restrict__ rowpart,
                                                          const T* __restrict__ lsp, const int32_t* __restrict__ n_valid, int y_div,
                                                          const T* __restrict__ g_lml, const int32_t* __restrict__ info,
                                                          T* __restrict__ lml, T* __restrict__ d_mean, int mean_mode,
                                                          T* __restrict__ d_ls, T* __restrict__ d_os, T* __restrict__ d_noise,
                                                          int P, int n, int f, int bwd) {
    __shared__ T red[4][PACOH_MAX_FEATURES + 3];
    const long b = blockIdx.x;
    const int lane = threadIdx.x;
    const int p = (int)(b % P);
    const int nv = clamp_nv(n_valid, b / y_div, n);
    const bool failed = info[b] < 0;
    const T LOG2PI = T(1.8378770664093453);
    if (lane == 0) {
        // logp counts log(2 pi) for all n rows; the identity rows of a ragged task contribute nothing else
        T v = nv > 0 ? (logp[b] + T(0.5) * T(n - nv) * LOG2PI) / T(nv) : T(0);
        lml[b] = failed ? T(NAN) : v;
    }
    if (!bwd) return;
    const T gup = g_lml ? g_lml[b] : T(1);
    const int W3 = f + 3;
    T part[PACOH_MAX_FEATURES + 3];
#pragma unroll
    for (int c = 0; c < PACOH_MAX_FEATURES + 3; ++c) part[c] = 0;
    for (int i = lane; i < n; i += 256) {
        const T* rp = rowpart + (b * n + i) * (long)W3;
#pragma unroll
        for (int c = 0; c < PACOH_MAX_FEATURES + 3; ++c) if (c < W3) part[c] += rp[c];
    }
#pragma unroll
    for (int c = 0; c < PACOH_MAX_FEATURES + 3; ++c) {
        if (c < W3) {
            const T s = subwave_sum<T>(part[c], 64);
            if ((lane & 63) == 0) red[lane >> 6][c] = s;
        }
    }
    __syncthreads();
    if (lane < W3) {
        const int c = lane;
        T s = ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
        {
            if (failed) s = T(NAN);
            if (c < f) d_ls[b * f + c] = gup * s / lsp[(long)p * f + c];
            else if (c == f) { if (d_os) d_os[b] = gup * s; }
            else if (c == f + 1) d_noise[b] = gup * s;
            else if (d_mean && mean_mode == PACOH_MEAN_CONST) d_mean[b] = nv > 0 ? gup * s / T(nv) : T(0);
        }
    }
}

// ---- predictive pieces ---------------------------------------------------------------------------------------------------------
// mu[b,s] = mean_tst + sum_k Kxs[b,k,s] alpha[b,k];  var[b,s] = os + noise - sum_k V[b,k,s]^2.
// One workgroup per (problem, 64 test points): 64 lanes = 64 consecutive s (coalesced rows of K_xs / V), the four waves take every
// fourth k and their partial sums meet in LDS in wave order (round 5; one thread per (b, s) walking all n rows was 130 us of a
// 0.83 ms call at n = 512, m = 128, 128 problems: 16 k threads, two dependent-latency loads per step).
template <typename T>
__global__ void __launch_bounds__(256) dense_predict_finish_kernel(const T* __restrict__ Kxs, const T* __restrict__ V, const T* __restrict__ alpha,
                                            const T* __restrict__ mean_tst, int mean_mode, const T* __restrict__ osp,
                                            const T* __restrict__ noise, const int32_t* __restrict__ info,
                                            T* __restrict__ mu, T* __restrict__ var, T* __restrict__ cov, int P, int n, int m) {
    __shared__ T pa[4][64], pv[4][64];
    const long b = blockIdx.y;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int s = blockIdx.x * 64 + lane;
    const int p = (int)(b % P);
    const bool failed = info[b] < 0;
    T acc = 0, vv = 0;
    if (s < m) {
        const T* kx = Kxs + b * (long)n * m + s;
        const T* vx = V + b * (long)n * m + s;
        const T* al = alpha + b * (long)n;
#pragma unroll 4
        for (int k = w; k < n; k += 4) {
            acc = fma(kx[(long)k * m], al[k], acc);
            const T v = vx[(long)k * m];
            vv = fma(v, v, vv);
        }
    }
    pa[w][lane] = acc; pv[w][lane] = vv;
    __syncthreads();
    if (w == 0 && s < m) {
        acc = (pa[0][lane] + pa[1][lane]) + (pa[2][lane] + pa[3][lane]);
        vv = (pv[0][lane] + pv[1][lane]) + (pv[2][lane] + pv[3][lane]);
        const long q = b * m + s;
        T mt = 0;
        if (mean_mode == PACOH_MEAN_VECTOR) mt = mean_tst[q];
        else if (mean_mode == PACOH_MEAN_CONST) mt = mean_tst[p];
        const T bad = failed ? T(NAN) : T(0);
        mu[q] = mt + acc + bad;
        var[q] = (osp ? osp[p] : T(1)) + noise[p] - vv + bad;
        if (cov && failed) for (int t = 0; t < m; ++t) cov[(b * m + s) * (long)m + t] = T(NAN);
    }
}

// alpha = Z^T u for the lower-triangular Z = L^-1 (after trtri_dense_kernel): alpha[i] = sum_{j >= i} Z[j][i] u[j].  The blocked
// backward solve inside the Cholesky kernel did the same with one dependent L2/HBM round trip per 32-row panel (0.2 ms at
// n = 512, one workgroup per matrix); here every load is independent: one workgroup per (matrix, 64 columns), four row groups.
template <typename T>
__global__ void __launch_bounds__(256) dense_alpha_kernel(const T* __restrict__ Zall, const T* __restrict__ u, T* __restrict__ alpha,
                                                          const int32_t* __restrict__ info, int n) {
    __shared__ T part[4][64];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    const T* Z = Zall + (size_t)b * n * n;
    const T* ub = u + (size_t)b * n;
    T acc = 0;
    if (i < n) {
        const int j0 = blockIdx.x * 64;               // rows below the chunk's first column; entries above the diagonal are not Z
#pragma unroll 8
        for (int j = j0 + rg; j < n; j += 4) acc = fma((j >= i) ? Z[(size_t)j * n + i] : T(0), ub[j], acc);
    }
    part[rg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rg == 0 && i < n) {
        const T tot = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        alpha[(size_t)b * n + i] = (info && info[b] < 0) ? T(NAN) : tot;
    }
}

// the split of the two-level path: n1 (a multiple of 64, <= 512) + n2 (<= 512).  (Measured at n = 784 fp32 with n1 = 320 / 384 / 448 /
// 512: 2.33 / 2.35 / 2.34 / 2.43 ms per call for 128 problems, 1.19 / 1.18 / 1.20 / 1.26 for 16 -- the split does not matter.)
static inline int blocked_n1(int n) { return 64 * ((n + 127) / 128); }

// rows x cols sub-matrices of a batch, 16 bytes per thread where both sides allow it
template <typename T>
__global__ void __launch_bounds__(256) sub_copy_kernel(const T* __restrict__ src, long ss, int lds_, T* __restrict__ dst, long sd, int ldd,
                                                       int rows, int cols, int vec, const int32_t* __restrict__ mask) {
    constexpr int VE = 16 / (int)sizeof(T);
    typedef T VT __attribute__((ext_vector_type(VE)));
    const long b = blockIdx.y;
    if (mask && mask[b] < 0) return;
    const T* S = src + b * ss;
    T* D = dst + b * sd;
    if (vec) {
        const int cpr = cols / VE;
        const long total = (long)rows * cpr;
        for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
            const int i = (int)(q / cpr), j = (int)(q - (long)i * cpr) * VE;
            *(VT*)(D + (long)i * ldd + j) = *(const VT*)(S + (long)i * lds_ + j);
        }
    } else {
        const long total = (long)rows * cols;
        for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total; q += (long)gridDim.x * 256) {
            const int i = (int)(q / cols), j = (int)(q - (long)i * cols);
            D[(long)i * ldd + j] = S[(long)i * lds_ + j];
        }
    }
}
template <typename T>
void launch_sub_copy(const T* src, long ss, int lds_, T* dst, long sd, int ldd, int rows, int cols, int B, hipStream_t s,
                     const int32_t* mask = nullptr) {
    constexpr int VE = 16 / (int)sizeof(T);
    const bool vec = cols % VE == 0 && lds_ % VE == 0 && ldd % VE == 0 && ss % VE == 0 && sd % VE == 0 && ((size_t)src % 16) == 0 && ((size_t)dst % 16) == 0;
    long chunks = ((long)rows * (vec ? cols / VE : cols) + 255) / 256;
    if (chunks > 64) chunks = 64;
    hipLaunchKernelGGL(sub_copy_kernel<T>, dim3((unsigned)chunks, B), dim3(256), 0, s, src, ss, lds_, dst, sd, ldd, rows, cols, vec ? 1 : 0, mask);
}

// Z = L^-1 for 512 < n <= 1024 (round 5), two-level: Z = [Z11 0; -Z22 L21 Z11, Z22].  The diagonal sub-blocks (n1 = a multiple of 64,
// both <= 512) go through the left-looking kernel as compact copies -- it takes the leading dimension = the size -- and the
// off-diagonal block is two products on the LDS-tiled GEMM; the right-looking kernel this replaces there runs ONE workgroup per
// matrix at 16-17 TFLOP/s in fp32 (profiles/r05_dense_big_n.txt: 1.22 ms of a 3.34 ms call at n = 784, 128 problems).
// scratch: B (n1^2 + n2^2 + n1 n2) elements.  1: outside the plan.
template <typename T>
int trtri_blocked(T* A, const int32_t* info, int B, int n, T* scratch, size_t scratch_elems, hipStream_t s) {
    const bool on = g_sw.trtri_blocked;
    const int dtype = sizeof(T) == 4 ? PACOH_F32 : PACOH_F64;
    if (!on || !scratch || n <= 512 || n > 1024) return 1;
    const int n1 = blocked_n1(n), n2 = n - n1;
    if (n2 < 1 || !trtri_ll_fits(n1, dtype) || !trtri_ll_fits(n2, dtype)) return 1;
    const size_t e1 = (size_t)B * n1 * n1, e2 = (size_t)B * n2 * n2, ex = (size_t)B * n2 * n1;
    if (e1 + e2 + ex > scratch_elems) return 1;
    T* C1 = scratch; T* C2 = C1 + e1; T* X = C2 + e2;
    const long sA = (long)n * n;
    launch_sub_copy<T>(A, sA, n, C1, (long)n1 * n1, n1, n1, n1, B, s);
    launch_sub_copy<T>(A + (long)n1 * n + n1, sA, n, C2, (long)n2 * n2, n2, n2, n2, B, s);
    int rc = trtri_ll_try(C1, info, B, n1, dtype, s, nullptr, nullptr);
    if (rc) return rc == 1 ? PACOH_ELIMIT : rc;
    rc = trtri_ll_try(C2, info, B, n2, dtype, s, nullptr, nullptr);
    if (rc) return rc == 1 ? PACOH_ELIMIT : rc;
    // X = L21 Z11;  Z21 = -Z22 X (into A's off-diagonal block)
    GemmArgs g1 = {A + (long)n1 * n, C1, X, sA, (long)n1 * n1, (long)n2 * n1, n, n1, n1, n2, n1, n1, 0, 0, 0, 1, 1.0, 0.0, info, 0};
    launch_bgemm<T>(g1, B, s, 1);
    GemmArgs g2 = {C2, X, A + (long)n1 * n, (long)n2 * n2, (long)n2 * n1, sA, n2, n1, n, n2, n1, n2, 0, 0, 1, 0, -1.0, 0.0, info, 0};
    launch_bgemm<T>(g2, B, s, 1);
    launch_sub_copy<T>(C1, (long)n1 * n1, n1, A, sA, n, n1, n1, B, s);
    launch_sub_copy<T>(C2, (long)n2 * n2, n2, A + (long)n1 * n + n1, sA, n, n2, n2, B, s);
    return launch_status();
}

// r2[b][i] = r[b][n1 + i] - sum_k L21[b][i][k] u1[b][k]: one wave per row of the second sub-block
template <typename T>
__global__ void __launch_bounds__(256) blocked_resid2_kernel(const T* __restrict__ resid, const T* __restrict__ L21, const T* __restrict__ u1,
                                                             T* __restrict__ r2, int n, int n1, int n2) {
    const long b = blockIdx.y;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n2) return;
    const T* Lr = L21 + (b * n2 + i) * (long)n1;
    const T* ub = u1 + b * (long)n1;
    T acc = 0;
    for (int k = lane; k < n1; k += 64) acc = fma(Lr[k], ub[k], acc);
    acc = subwave_sum<T>(acc, 64);
    if (lane == 0) r2[b * (long)n2 + i] = resid[b * (long)n + n1 + i] - acc;
}

// the two sub-factorisations' verdicts, log-densities and u = L^-1 r as ONE problem's, at rung `rung` of the jitter ladder (rung > 0:
// only the problems still marked failed take part -- info1 / info2 read `rung` where a sub-factorisation of this rung succeeded)
template <typename T>
__global__ void __launch_bounds__(256) blocked_combine_kernel(const int32_t* __restrict__ info1, const int32_t* __restrict__ info2,
                                                              const T* __restrict__ logp1, const T* __restrict__ logp2,
                                                              const T* __restrict__ u1, const T* __restrict__ u2, int32_t* __restrict__ info,
                                                              T* __restrict__ logp, T* __restrict__ u, int n, int n1, int rung,
                                                              int32_t* __restrict__ act) {
    const long b = blockIdx.x;
    if (rung > 0 && info[b] >= 0) { if (threadIdx.x == 0 && act) act[b] = -1; return; }     // solved at an earlier rung
    const bool ok = info1[b] == rung && info2[b] == rung;
    __syncthreads();
    if (threadIdx.x == 0) { info[b] = ok ? rung : -1; logp[b] = ok ? logp1[b] + logp2[b] : T(NAN); if (act) act[b] = ok ? 0 : -1; }
    if (u) {
        const int n2 = n - n1;
        for (int q = threadIdx.x; q < n; q += 256) u[b * (long)n + q] = !ok ? T(NAN) : (q < n1 ? u1[b * (long)n1 + q] : u2[b * (long)n2 + (q - n1)]);
    }
}

// rung > 0: which problems run (info < 0).  run[b] = 0 / -1 (the "skip if negative" convention of the copies, products and inverses),
// pre1 / pre2[b] = -1 / 0 (the factorisation kernels' own: at a rung > 0 they skip what reads >= 0)
__global__ void blocked_need_kernel(const int32_t* __restrict__ info, int32_t* __restrict__ run, int32_t* __restrict__ pre1,
                                    int32_t* __restrict__ pre2, int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) { const bool need = info[b] < 0; run[b] = need ? 0 : -1; pre1[b] = need ? -1 : 0; pre2[b] = need ? -1 : 0; }
}
// run1[b] = 0 where the first sub-factorisation of this rung succeeded for a problem that runs
__global__ void blocked_stage_kernel(const int32_t* __restrict__ run, const int32_t* __restrict__ i1, int32_t* __restrict__ run1, int rung, int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) run1[b] = (run[b] == 0 && i1[b] == rung) ? 0 : -1;
}

// Cholesky + log-density + u = L^-1 r for 512 < n <= 1024 (round 5), rung 0 of the jitter ladder, two-level:
//   L11 = chol(A11), Z11 = L11^-1;  L21 = A21 Z11^T;  L22 = chol(A22 - L21 L21^T);  u1 = L11^-1 r1, u2 = L22^-1 (r2 - L21 u1)
// with the diagonal sub-blocks (n1 a multiple of 64, both <= 512) as compact copies through the left-looking kernels and the rest on
// the LDS-tiled GEMM.  The right-looking kernel this replaces at these sizes runs one workgroup per matrix for ~1.3 ms at n = 784
// however few matrices there are -- and the reference's MNIST tasks come a handful at a time.  A is left in the layout every
// Cholesky kernel of the path leaves (L below, the inverse 32-blocks above the diagonal); a problem that fails either
// sub-factorisation gets info = -1 and goes through the ladder's later rungs on the right-looking kernel, like everybody else's failures.
// scratch: B (n1^2 + n2^2 + n1 n2 + 2 n + 4) elements + 2 B int32.  1: outside the plan.
template <typename T>
size_t chol_blocked_scratch(int B, int n) {
    const int n1 = blocked_n1(n), n2 = n - n1;
    return (size_t)B * ((size_t)n1 * n1 + (size_t)n2 * n2 + (size_t)n1 * n2 + 2 * (size_t)n + 8) + 64;
}
template <typename T>
bool chol_blocked_plan(int B, int n, size_t scratch_elems) {
    const bool on = g_sw.chol_blocked;
    const bool mfma_on = g_sw.mfma;
    const bool ll_on = g_sw.chol_ll;
    const int dtype = sizeof(T) == 4 ? PACOH_F32 : PACOH_F64;
    if (!on || !mfma_on || !ll_on || n <= 512 || n > 1024) return false;
    const int n1 = blocked_n1(n), n2 = n - n1;
    return n2 >= 97 && dense_ll_fits(n1, dtype) && dense_ll_fits(n2, dtype) && trtri_ll_fits(n1, dtype) && chol_blocked_scratch<T>(B, n) <= scratch_elems;
}
// want_inv: go on to Z = L^-1 for the problems both sub-factorisations solved -- Z11 exists already, Z22 = L22^-1 in place,
// Z21 = -Z22 (L21 Z11) -- and leave Z in A instead of L (no factor is written back: four of the block copies and the second
// inversion of L11 that a separate inverse stage would need).  Without it (forward-only call) nothing reads A afterwards.
// rung > 0: the same on the problems still marked failed (their A rebuilt with the rung's jitter by the caller), everybody else's
// blocks, vectors and verdicts untouched: the copies, products and inverses take a run mask, the factorisation kernels their own
// "skip what is solved" convention.  Used where no right-looking kernel exists for the later rungs (fp64 above n ~ 520).
template <typename T>
int chol_blocked(T* A, const T* resid, T* logp, T* u_out, int32_t* info, int B, int n, T* scratch, bool want_inv, int rung, hipStream_t s) {
    const int dtype = sizeof(T) == 4 ? PACOH_F32 : PACOH_F64;
    const int n1 = blocked_n1(n), n2 = n - n1;
    const size_t e1 = (size_t)B * n1 * n1, e2 = (size_t)B * n2 * n2, ex = (size_t)B * n2 * n1;
    // scratch: C1 | C2 | T21 | r1 | r2 | u1 | u2 | logp1 | logp2 | info1, info2 | run, run1, act
    T* C1 = scratch; T* C2 = C1 + e1; T* T21 = C2 + e2;
    T* r1 = T21 + ex; T* r2 = r1 + (size_t)B * n1;
    T* const u1v = r2 + (size_t)B * n2; T* const u2v = u1v + (size_t)B * n1;
    T* lp1 = u2v + (size_t)B * n2; T* lp2 = lp1 + B;
    int32_t* i1 = reinterpret_cast<int32_t*>(lp2 + B); int32_t* i2 = i1 + B;
    int32_t* run = i2 + B; int32_t* run1 = run + B; int32_t* act = run1 + B;
    const long sA = (long)n * n, s1 = (long)n1 * n1, s2 = (long)n2 * n2, sx = (long)n2 * n1;
    T* const A21 = A + (long)n1 * n; T* const A22 = A21 + n1;
    const unsigned gb = (unsigned)((B + 255) / 256);
    const int32_t* m0 = nullptr;                      // who runs at all
    if (rung > 0) { hipLaunchKernelGGL(blocked_need_kernel, dim3(gb), dim3(256), 0, s, (const int32_t*)info, run, i1, i2, B); m0 = run; }
    launch_sub_copy<T>(A, sA, n, C1, s1, n1, n1, n1, B, s, m0);
    launch_sub_copy<T>(resid, n, n, r1, n1, n1, 1, n1, B, s, m0);
    int rc = dense_ll_try(C1, r1, lp1, u1v, i1, 1.0, B, n1, dtype, rung, 1, s);
    if (rc) return rc == 1 ? PACOH_ELIMIT : rc;
    const int32_t* m1 = i1;                           // ... and got through the first sub-factorisation (rung 0: info1 = 0 / -1 is that mask)
    if (rung > 0) { hipLaunchKernelGGL(blocked_stage_kernel, dim3(gb), dim3(256), 0, s, (const int32_t*)run, (const int32_t*)i1, run1, rung, B); m1 = run1; }
    rc = trtri_ll_try(C1, m1, B, n1, dtype, s, nullptr, nullptr);                            // C1 = Z11
    if (rc) return rc == 1 ? PACOH_ELIMIT : rc;
    GemmArgs g1 = {A21, C1, T21, sA, s1, sx, n, n1, n1, n2, n1, n1, 0, 1, 0, 1, 1.0, 0.0, m1, 0};      // L21 = A21 Z11^T
    launch_bgemm<T>(g1, B, s, 1);
    launch_sub_copy<T>(A22, sA, n, C2, s2, n2, n2, n2, B, s, m0);
    GemmArgs g2 = {T21, T21, C2, sx, sx, s2, n1, n1, n2, n2, n2, n1, 0, 1, 0, 0, -1.0, 1.0, m1, 1};    // C2 = A22 - L21 L21^T
    launch_bgemm<T>(g2, B, s, 1);
    hipLaunchKernelGGL(blocked_resid2_kernel<T>, dim3((n2 + 3) / 4, B), dim3(256), 0, s, resid, (const T*)T21, (const T*)u1v, r2, n, n1, n2);
    rc = dense_ll_try(C2, r2, lp2, u2v, i2, 1.0, B, n2, dtype, rung, 1, s);
    if (rc) return rc == 1 ? PACOH_ELIMIT : rc;
    hipLaunchKernelGGL(blocked_combine_kernel<T>, dim3(B), dim3(256), 0, s, (const int32_t*)i1, (const int32_t*)i2, (const T*)lp1, (const T*)lp2,
                       (const T*)u1v, (const T*)u2v, info, logp, u_out, n, n1, rung, rung > 0 ? act : nullptr);
    if (want_inv) {
        const int32_t* m2 = rung > 0 ? act : info;    // ... and were solved at this rung (rung 0: info = 0 / -1 is that mask)
        rc = trtri_ll_try(C2, m2, B, n2, dtype, s, nullptr, nullptr);                        // C2 = Z22
        if (rc) return rc == 1 ? PACOH_ELIMIT : rc;
        GemmArgs g3 = {T21, C1, A21, sx, s1, sA, n1, n1, n, n2, n1, n1, 0, 0, 0, 1, 1.0, 0.0, m2, 0};          // X = L21 Z11 (into A's dead A21)
        launch_bgemm<T>(g3, B, s, 1);
        GemmArgs g4 = {C2, A21, T21, s2, sA, sx, n2, n, n1, n2, n1, n2, 0, 0, 1, 0, -1.0, 0.0, m2, 0};         // Z21 = -Z22 X
        launch_bgemm<T>(g4, B, s, 1);
        launch_sub_copy<T>(T21, sx, n1, A21, sA, n, n2, n1, B, s, m2);
        launch_sub_copy<T>(C1, s1, n1, A, sA, n, n1, n1, B, s, m2);
        launch_sub_copy<T>(C2, s2, n2, A22, sA, n, n2, n2, B, s, m2);
    }
    return launch_status();
}

// late[b] = the problems a LATER rung of the ladder solved on the right-looking kernel (info > 0): what is still a factor after
// chol_blocked(want_inv) at rung 0
__global__ void late_mask_kernel(const int32_t* __restrict__ info, int32_t* __restrict__ late, int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) late[b] = info[b] > 0 ? info[b] : -1;
}

template <typename T>
int launch_trtri(T* A, const int32_t* info, int B, int n, int mpad, size_t lds, int saved_inv, hipStream_t s, const T* u = nullptr,
                 T* alpha = nullptr, int* did_alpha = nullptr, T* scratch = nullptr, size_t scratch_elems = 0) {
    // the left-looking kernel (dense_trtri_ll.hip) needs the inverse diagonal blocks the MFMA Cholesky kernels leave behind
    const bool ll_on = g_sw.trtri_ll;
    if (ll_on && saved_inv && n >= 97) {
        const int rc = trtri_ll_try(A, info, B, n, sizeof(T) == 4 ? PACOH_F32 : PACOH_F64, s, u, alpha);
        if (rc != 1) { if (did_alpha && u && alpha) *did_alpha = 1; return rc; }
        const int rb = trtri_blocked<T>(A, info, B, n, scratch, scratch_elems, s);
        if (rb != 1) return rb;
    }
#define PACOH_TRTRI_LAUNCH(nt) do { auto kern = trtri_dense_kernel<T, nt>; \
        if (lds > 64u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return PACOH_ELIMIT; \
        hipLaunchKernelGGL(kern, dim3(B), dim3(nt), lds, s, A, info, n, mpad, saved_inv); } while (0)
    if (n >= 256) PACOH_TRTRI_LAUNCH(1024); else if (n >= 96) PACOH_TRTRI_LAUNCH(512); else PACOH_TRTRI_LAUNCH(256);
#undef PACOH_TRTRI_LAUNCH
    return 0;
}

template <typename T>
size_t trtri_lds(int n, int* mpad_out) {
    const int mpad = n > DNB ? (n - DNB + 15) / 16 * 16 : 16;
    *mpad_out = mpad;
    return ((size_t)DNB * (DNB + 1) + (size_t)DNB * DLP + (size_t)mpad * DLP) * sizeof(T);
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

template <typename T>
int lml_dense_impl(const void* z, int z_div, const void* mean, int mean_mode, const void* y, int y_div, const void* ls,
                   const void* os, const void* noise, const int32_t* n_valid, const void* g_lml, void* lml, void* d_z,
                   void* d_mean, void* d_ls, void* d_os, void* d_noise, int32_t* info, void* workspace, int B, int P, int n,
                   int f_arg, int dtype, hipStream_t s) {
    const int f = features_of(f_arg), kind = kernel_of(f_arg);
    const bool bwd = d_ls != nullptr;
    const size_t nn = (size_t)B * n * n;
    unsigned char* w = (unsigned char*)workspace;
    T* A = (T*)w;                 w += align256(nn * sizeof(T));
    T* Wm = (T*)w;                w += bwd ? align256(nn * sizeof(T)) : 0;
    T* resid = (T*)w;             w += align256((size_t)B * n * sizeof(T));
    T* alpha = (T*)w;             w += align256((size_t)B * n * sizeof(T));
    T* logp = (T*)w;              w += align256((size_t)B * sizeof(T));
    T* rowpart = (T*)w;           w += align256((size_t)B * n * (f + 3) * sizeof(T));
    T* zsc = (T*)w;
    int mpad = 0;
    const size_t lds = trtri_lds<T>(n, &mpad);
    const long total = (long)B * n;
    hipLaunchKernelGGL(dense_resid_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T*)y, y_div,
                       (const T*)mean, mean_mode, n_valid, resid, P, n, total);
    const double jitter_base = dtype == PACOH_F32 ? 1e-6 : 1e-8;            // psd_safe_cholesky [gpytorch-upstream]
    // 512 < n <= 1024: rung 0 of the ladder two-level on the left-looking kernels + the tiled GEMM (chol_blocked; its scratch is W's
    // buffer in a call with gradients).  Failures take the later rungs on the right-looking kernel where one exists for the size
    // (two early-exit launches per rung when nothing failed); where none does -- fp64 above n ~ 520, which returned PACOH_ELIMIT for
    // a call with gradients until round 5 -- the later rungs are two-level as well (ladder_blocked).
    T* const cb_scratch = bwd ? Wm : (T*)((unsigned char*)zsc + align256((size_t)B * n * f * sizeof(T)));    // (forward only: its own region, see the workspace query)
    const bool blocked = chol_blocked_plan<T>(B, n, bwd ? nn : chol_blocked_scratch<T>(B, n));
    const bool ladder_blocked = blocked && !dense_chol_saves_inverse(n, dtype);
    if (bwd && !ladder_blocked && lds > 160u * 1024u) return PACOH_ELIMIT;
    const bool u_only = bwd && (blocked || dense_chol_saves_inverse(n, dtype));     // alpha comes from Z afterwards (dense_alpha_kernel)
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (attempt == 0) {
            // (RBF family: the tiles above the diagonal are skipped -- nothing downstream reads A's upper triangle)
            int rc = kind == PACOH_KERNEL_RBF ? gram_for_chol(z, z_div, ls, os, noise, A, B, P, n, f, dtype, s)
                                              : pacoh_gram_rbf_ard(z, z_div, z, z_div, ls, os, noise, 1, A, B, P, n, n, f_arg, dtype, s);
            if (rc) return rc;
        } else {
            if (attempt == 1) {                        // the whole ladder in one launch where the left-looking kernel factors this size
                const int rf = dense_chol_retry_fused(A, resid, logp, bwd ? alpha : nullptr, info, 1.0, B, n, dtype, u_only ? 1 : 0, z, z_div, ls, os,
                                                      noise, n_valid, y_div, jitter_base, P, f, kind, s);
                if (rf == 0) break;
                if (rf != 1) return rf;
            }
            double jit = jitter_base;
            for (int q = 1; q < attempt; ++q) jit *= 10.0;
            hipLaunchKernelGGL(regram_failed_kernel<T>, dim3(n < 8 ? n : 8, B < 32 ? B : 32), dim3(256), 0, s, (const T*)z, z_div, (const T*)ls, (const T*)os,
                               (const T*)noise, (const int32_t*)info, (T)jit, A, P, n, f, kind, B);
        }
        if (n_valid) hipLaunchKernelGGL(dense_mask_kernel<T>, dim3(n, B), dim3(256), 0, s, A, n_valid, y_div, (const int32_t*)info, attempt, n);
        int rc = (blocked && (attempt == 0 || ladder_blocked)) ? chol_blocked<T>(A, resid, logp, bwd ? alpha : nullptr, info, B, n, cb_scratch, bwd, attempt, s)
                                           : dense_chol_launch(A, resid, logp, bwd ? alpha : nullptr, info, 1.0, B, n, dtype, attempt, s, u_only ? 1 : 0);
        if (rc) return rc;
    }
    if (bwd) {
        int did_alpha = 0;                             // (the left-looking inverse produces alpha = Z^T u on the way)
        int rc;
        if (ladder_blocked) rc = 0;                    // (every rung left Z)
        else if (blocked) {
            // the two-level factorisation left Z for what rung 0 solved; only what a later rung solved is still a factor (rare: the
            // right-looking inverse, which exits at once for everybody else).  The mask lives at the head of rowpart, which the
            // contraction kernels write later.
            int32_t* late = reinterpret_cast<int32_t*>(rowpart);
            hipLaunchKernelGGL(late_mask_kernel, dim3((B + 255) / 256), dim3(256), 0, s, (const int32_t*)info, late, B);
            rc = launch_trtri<T>(A, late, B, n, mpad, lds, 1, s);
        } else
            rc = launch_trtri<T>(A, info, B, n, mpad, lds, dense_chol_saves_inverse(n, dtype) ? 1 : 0, s, u_only ? alpha : nullptr,
                                 u_only ? resid : nullptr, &did_alpha, Wm, nn);                 // (W's buffer: free until W = Z^T Z)
        if (rc) return rc;
        if (u_only) {                                  // alpha = Z^T u into the residual buffer (free after the last factorisation attempt)
            if (!did_alpha)
                hipLaunchKernelGGL(dense_alpha_kernel<T>, dim3((n + 63) / 64, B), dim3(256), 0, s, (const T*)A, (const T*)alpha, resid,
                                   (const int32_t*)info, n);
            T* t = alpha; alpha = resid; resid = t;
        }
        GemmArgs ga = {A, A, Wm, (long)n * n, (long)n * n, (long)n * n, n, n, n, n, n, n, 1, 0, 1, 1, 1.0, 0.0, info, 1};
        // fp64 ARD-RBF: the symmetric MFMA contraction (dense_grad_mfma.hip) reads the lower 64-tiles of W only -- the 128-tiles of
        // Z^T Z below the block diagonal are then not mirrored (268 MB of 32-byte-segment stores less per 256 x 512^2 launch)
        const bool gm_on = g_sw.grad_mfma;
        const bool gm_plan = gm_on && dense_grad_mfma_plan(B, n, f, kind, dtype, nn * sizeof(T));
        launch_bgemm<T>(ga, B, s, gm_plan ? 0 : 1);                         // W = Z^T Z
        const long tz = (long)B * n * f;
        hipLaunchKernelGGL(dense_scale_kernel<T>, dim3((unsigned)((tz + 255) / 256)), dim3(256), 0, s, (const T*)z, z_div, (const T*)ls, zsc,
                           P, n, f, tz);
        // (its scratch is the factor's buffer, free once W = Z^T Z exists)
        const int grc = gm_plan ? dense_grad_mfma_try(zsc, ls, os, n_valid, y_div, g_lml, alpha, Wm, (const int32_t*)info, d_z, d_mean, mean_mode,
                                                    rowpart, A, nn * sizeof(T), B, P, n, f, kind, dtype, s) : 1;
        if (grc != 0 && grc != 1) return grc;
        const int FP = f <= 2 ? 2 : (f <= 4 ? 4 : (f <= 8 ? 8 : 16));
        // 16 rows per workgroup with the coordinates staged in LDS when they fit, else 4 rows per workgroup from L1/L2
        const size_t glds = ((size_t)n * f + n) * sizeof(T);
        const bool use_lds = glds <= 60u * 1024u;
        const int rows = use_lds ? 16 : 4;
#define PACOH_DG_CASE(fp) case fp: \
        if (use_lds && glds + 3 * 64 * (2 * fp + 2) * sizeof(T) <= 64u * 1024u) \
            hipLaunchKernelGGL((dense_grad_cols_kernel<T, fp>), dim3((n + 63) / 64, B), dim3(256), glds + 3 * 64 * (2 * fp + 2) * sizeof(T), s, \
            (const T*)zsc, (const T*)ls, (const T*)os, n_valid, y_div, (const T*)g_lml, (const T*)alpha, (const T*)Wm, \
            (const int32_t*)info, (T*)d_z, (T*)d_mean, mean_mode, rowpart, P, n, f, kind); \
        else if (use_lds) hipLaunchKernelGGL((dense_grad_rows_kernel<T, fp, true>), dim3((n + rows - 1) / rows, B), dim3(256), glds, s, \
            (const T*)zsc, (const T*)ls, (const T*)os, n_valid, y_div, (const T*)g_lml, (const T*)alpha, (const T*)Wm, \
            (const int32_t*)info, (T*)d_z, (T*)d_mean, mean_mode, rowpart, P, n, f, rows, kind); \
        else hipLaunchKernelGGL((dense_grad_rows_kernel<T, fp, false>), dim3((n + rows - 1) / rows, B), dim3(256), 0, s, \
            (const T*)zsc, (const T*)ls, (const T*)os, n_valid, y_div, (const T*)g_lml, (const T*)alpha, (const T*)Wm, \
            (const int32_t*)info, (T*)d_z, (T*)d_mean, mean_mode, rowpart, P, n, f, rows, kind); \
        break;
        if (grc == 1) switch (FP) { PACOH_DG_CASE(2) PACOH_DG_CASE(4) PACOH_DG_CASE(8) default: PACOH_DG_CASE(16) }
#undef PACOH_DG_CASE
    }
    hipLaunchKernelGGL(dense_finish_kernel<T>, dim3(B), dim3(256), 0, s, (const T*)logp, (const T*)rowpart, (const T*)ls, n_valid,
                       y_div, (const T*)g_lml, (const int32_t*)info, (T*)lml, (T*)d_mean, mean_mode, (T*)d_ls, (T*)d_os,
                       (T*)d_noise, P, n, f, bwd ? 1 : 0);
    return launch_status();
}

template <typename T>
int predict_dense_impl(const void* z_ctx, int z_div, const void* mean_ctx, int mean_mode, const void* y, int y_div,
                       const void* z_tst, int zt_div, const void* mean_tst, const void* ls, const void* os, const void* noise,
                       const int32_t* n_valid, void* mu, void* var, void* cov, int32_t* info, void* workspace, int B, int P,
                       int n, int m, int f_arg, int dtype, hipStream_t s) {
    const int f = features_of(f_arg), kind = kernel_of(f_arg);
    const size_t nn = (size_t)B * n * n, nm = (size_t)B * n * m;
    unsigned char* w = (unsigned char*)workspace;
    T* A = (T*)w;                 w += align256(nn * sizeof(T));
    T* Kxs = (T*)w;               w += align256(nm * sizeof(T));
    T* V = (T*)w;                 w += align256(nm * sizeof(T));
    T* resid = (T*)w;             w += align256((size_t)B * n * sizeof(T));
    T* alpha = (T*)w;             w += align256((size_t)B * n * sizeof(T));
    T* logp = (T*)w;              w += align256((size_t)B * sizeof(T));
    T* const cb_scratch = (T*)w;                                           // (512 < n <= 1024 only, see the workspace query)
    int mpad = 0;
    const size_t lds = trtri_lds<T>(n, &mpad);
    // 512 < n <= 1024: the two-level factorisation + inverse of lml_dense_impl (u = L^-1 r and Z come out of it, alpha = Z^T u)
    const bool blocked = chol_blocked_plan<T>(B, n, chol_blocked_scratch<T>(B, n));
    const bool ladder_blocked = blocked && !dense_chol_saves_inverse(n, dtype);
    if (!ladder_blocked && lds > 160u * 1024u) return PACOH_ELIMIT;
    const long total = (long)B * n;
    hipLaunchKernelGGL(dense_resid_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const T*)y, y_div,
                       (const T*)mean_ctx, mean_mode, n_valid, resid, P, n, total);
    const double jitter_base = dtype == PACOH_F32 ? 1e-6 : 1e-8;
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (attempt == 0) {
            int rc = kind == PACOH_KERNEL_RBF ? gram_for_chol(z_ctx, z_div, ls, os, noise, A, B, P, n, f, dtype, s)
                                              : pacoh_gram_rbf_ard(z_ctx, z_div, z_ctx, z_div, ls, os, noise, 1, A, B, P, n, n, f_arg, dtype, s);
            if (rc) return rc;
        } else {
            if (attempt == 1 && !blocked) {
                const int rf = dense_chol_retry_fused(A, resid, logp, alpha, info, 1.0, B, n, dtype, 0, z_ctx, z_div, ls, os, noise, n_valid, y_div,
                                                      jitter_base, P, f, kind, s);
                if (rf == 0) break;
                if (rf != 1) return rf;
            }
            double jit = jitter_base;
            for (int q = 1; q < attempt; ++q) jit *= 10.0;
            hipLaunchKernelGGL(regram_failed_kernel<T>, dim3(n < 8 ? n : 8, B < 32 ? B : 32), dim3(256), 0, s, (const T*)z_ctx, z_div, (const T*)ls,
                               (const T*)os, (const T*)noise, (const int32_t*)info, (T)jit, A, P, n, f, kind, B);
        }
        if (n_valid) hipLaunchKernelGGL(dense_mask_kernel<T>, dim3(n, B), dim3(256), 0, s, A, n_valid, y_div, (const int32_t*)info, attempt, n);
        int rc = (blocked && (attempt == 0 || ladder_blocked)) ? chol_blocked<T>(A, resid, logp, alpha, info, B, n, cb_scratch, true, attempt, s)
                                                               : dense_chol_launch(A, resid, logp, alpha, info, 1.0, B, n, dtype, attempt, s, blocked ? 1 : 0);
        if (rc) return rc;
    }
    if (blocked) {
        if (!ladder_blocked) {                         // what a later rung solved on the right-looking kernel is still a factor
            int32_t* late = reinterpret_cast<int32_t*>(V);
            hipLaunchKernelGGL(late_mask_kernel, dim3((B + 255) / 256), dim3(256), 0, s, (const int32_t*)info, late, B);
            int rc = launch_trtri<T>(A, late, B, n, mpad, lds, 1, s);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(dense_alpha_kernel<T>, dim3((n + 63) / 64, B), dim3(256), 0, s, (const T*)A, (const T*)alpha, resid, (const int32_t*)info, n);
        T* t = alpha; alpha = resid; resid = t;         // alpha = Z^T u (the two-level path hands over u)
    } else {
        int rc = launch_trtri<T>(A, info, B, n, mpad, lds, dense_chol_saves_inverse(n, dtype) ? 1 : 0, s, nullptr, nullptr, nullptr, V, nm);
        if (rc) return rc;
    }
    int rc = pacoh_gram_rbf_ard(z_ctx, z_div, z_tst, zt_div, ls, os, nullptr, 0, Kxs, B, P, n, m, f_arg, dtype, s);
    if (rc) return rc;
    // padded context rows carry no information: Z's identity rows would pass those rows of K_xs straight into V
    if (n_valid) hipLaunchKernelGGL(dense_zero_rows_kernel<T>, dim3(n, B), dim3(256), 0, s, Kxs, n_valid, y_div, n, m);
    GemmArgs gv = {A, Kxs, V, (long)n * n, (long)n * m, (long)n * m, n, m, m, n, m, n, 0, 0, 1, 0, 1.0, 0.0, info, 0};
    launch_bgemm<T>(gv, B, s);                                              // V = Z K_xs
    if (cov) {
        rc = pacoh_gram_rbf_ard(z_tst, zt_div, z_tst, zt_div, ls, os, noise, 1, cov, B, P, m, m, f_arg, dtype, s);
        if (rc) return rc;
        GemmArgs gc = {V, V, cov, (long)n * m, (long)n * m, (long)m * m, m, m, m, m, m, n, 1, 0, 0, 0, -1.0, 1.0, info, 1};
        launch_bgemm<T>(gc, B, s);                                          // cov = K_ss + noise I - V^T V
    }
    hipLaunchKernelGGL(dense_predict_finish_kernel<T>, dim3((unsigned)((m + 63) / 64), B), dim3(256), 0, s, (const T*)Kxs, (const T*)V,
                       (const T*)alpha, (const T*)mean_tst, mean_mode, (const T*)os, (const T*)noise, (const int32_t*)info,
                       (T*)mu, (T*)var, (T*)cov, P, n, m);
    return launch_status();
}

}  // namespace
}  // namespace pacoh

using namespace pacoh;

// ---- what the C ABI carries itself since round 6 (it lived in the Python binding): contexts whose rows are not a multiple
//      of 16 bytes reach the left-looking kernels as a RAGGED batch of the next aligned size, and a workspace smaller than the whole
//      batch needs runs the batch in slabs of whole tasks ------------------------------------------------------------------------------
// dst[r][i][c] = i < n_src ? src[r][i][c] : 0 for i < n_dst (pad: n_dst > n_src; crop: n_dst < n_src), rows of w elements
template <typename T>
__global__ void __launch_bounds__(256) dense_rows_copy_kernel(const T* __restrict__ src, T* __restrict__ dst, long rows, int n_src, int n_dst, int w) {
    const long per = (long)n_dst * w, tot = rows * per;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < tot; q += (long)gridDim.x * 256) {
        const long rr = q / per; const int rem = (int)(q - rr * per), i = rem / w, c = rem - i * w;
        dst[q] = i < n_src ? src[(rr * n_src + i) * w + c] : T(0);
    }
}
__global__ void __launch_bounds__(256) dense_nv_kernel(const int32_t* __restrict__ nv_in, int32_t* __restrict__ nv_out, int count, int n) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q < count) { const int v = nv_in ? nv_in[q] : n; nv_out[q] = v < n ? v : n; }
}
template <typename T>
static void rows_copy(const void* src, void* dst, long rows, int n_src, int n_dst, int w, hipStream_t s) {
    const long tot = rows * n_dst * w;
    if (tot <= 0) return;
    long blocks = (tot + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(dense_rows_copy_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, (const T*)src, (T*)dst, rows, n_src, n_dst, w);
}

// the context size a problem of n points is factored at: n itself, or -- 97 <= n < 1024 with rows that are no multiple of 16 bytes,
// where the left-looking kernels (n <= 512) / the two-level path (<= 1024) would otherwise hand the size to the right-looking
// generation (1.022 vs 0.788 ms at n = 509 fp32, profiles/r05_dense_pad_ab.txt) -- the next aligned size: the padded rows are identity
// rows of the matrices, exactly what the kernels do for tasks of unequal length.  PACOH_DENSE_PAD=0 / PACOH_CHOL_LL=0: never.
static int dense_padded_n(int n, int dtype) {
    const int align = dtype == PACOH_F32 ? 4 : 2;
    if (!g_sw.dense_pad || !g_sw.chol_ll || n % align == 0) return n;
    if ((n >= 97 && n < 512) || (n > 512 && n < 1024)) return (n + align - 1) / align * align;
    return n;
}

static size_t lml_core_bytes(int B, int n, int f, int dtype, int want_grad) {
    const size_t e = dtype == PACOH_F64 ? 8 : 4;
    const size_t nn = (size_t)B * n * n;
    // (forward only, 512 < n <= 1024: the two-level Cholesky's scratch, which a call with gradients finds in W's buffer)
    const size_t cb = (!want_grad && n > 512 && n <= 1024) ? align256((dtype == PACOH_F64 ? chol_blocked_scratch<double>(B, n) : chol_blocked_scratch<float>(B, n)) * e) : 0;
    return align256(nn * e) * (want_grad ? 2 : 1) + 2 * align256((size_t)B * n * e) + align256((size_t)B * e) +
           align256((size_t)B * n * (f + 3) * e) + align256((size_t)B * n * f * e) + cb + 256;
}
// the padded copies of a slab of B problems: z, mean, y in; n_valid; d_z, d_mean out (upper bounds: every problem its own inputs)
static size_t lml_pad_bytes(int B, int npad, int f, int dtype) {
    const size_t e = dtype == PACOH_F64 ? 8 : 4;
    return 2 * align256((size_t)B * npad * f * e) + 3 * align256((size_t)B * npad * e) + align256((size_t)B * 4);
}
static size_t lml_slab_bytes(int B, int n, int f, int dtype, int want_grad) {
    const int npad = dense_padded_n(n, dtype);
    return lml_core_bytes(B, npad, f, dtype, want_grad) + (npad != n ? lml_pad_bytes(B, npad, f, dtype) : 0);
}

extern "C" size_t pacoh_gp_lml_dense_workspace_bytes(int B, int n, int f, int dtype, int want_grad) {
    f = features_of(f);
    if (B <= 0 || n <= 0 || f <= 0) return 0;
    return lml_slab_bytes(B, n, f, dtype, want_grad);
}

// one slab of Bc whole problems starting at problem b0 of the call: pointer offsets, the padded copies where the size asks for them
template <typename T>
static int lml_dense_slab(const void* z, int z_div, const void* mean, int mean_mode, const void* y, int y_div, const void* ls,
                          const void* os, const void* noise, const int32_t* n_valid, const void* g_lml, void* lml, void* d_z,
                          void* d_mean, void* d_ls, void* d_os, void* d_noise, int32_t* info, void* workspace, long b0, int Bc, int P, int n,
                          int f_arg, int dtype, hipStream_t s) {
    const int f = features_of(f_arg);
    auto at = [](const void* p, long elems) -> const T* { return p ? (const T*)p + elems : nullptr; };
    auto atw = [](void* p, long elems) -> T* { return p ? (T*)p + elems : nullptr; };
    const long zr0 = b0 / z_div, yr0 = b0 / y_div;
    const long zrows = (b0 + Bc + z_div - 1) / z_div - zr0, yrows = (b0 + Bc + y_div - 1) / y_div - yr0;
    const T* zs = at(z, zr0 * n * f);
    const T* ms = mean_mode == PACOH_MEAN_VECTOR ? at(mean, b0 * n) : (const T*)mean;
    const T* ys = at(y, yr0 * n);
    const int32_t* nvs = n_valid ? n_valid + yr0 : nullptr;
    T* dzs = atw(d_z, b0 * n * f);
    T* dms = mean_mode == PACOH_MEAN_VECTOR ? atw(d_mean, b0 * n) : atw(d_mean, b0);
    const int npad = dense_padded_n(n, dtype);
    unsigned char* w = (unsigned char*)workspace;
    T* dz_p = nullptr; T* dm_p = nullptr;
    if (npad != n) {
        const size_t e = sizeof(T);
        T* z_p = (T*)w;  w += align256((size_t)Bc * npad * f * e);
        T* m_p = (T*)w;  w += align256((size_t)Bc * npad * e);
        T* y_p = (T*)w;  w += align256((size_t)Bc * npad * e);
        int32_t* nv_p = (int32_t*)w; w += align256((size_t)Bc * 4);
        dz_p = (T*)w;    w += align256((size_t)Bc * npad * f * e);
        dm_p = (T*)w;    w += align256((size_t)Bc * npad * e);
        rows_copy<T>(zs, z_p, zrows, n, npad, f, s);
        if (mean_mode == PACOH_MEAN_VECTOR) rows_copy<T>(ms, m_p, Bc, n, npad, 1, s);
        rows_copy<T>(ys, y_p, yrows, n, npad, 1, s);
        hipLaunchKernelGGL(dense_nv_kernel, dim3((unsigned)((yrows + 255) / 256)), dim3(256), 0, s, nvs, nv_p, (int)yrows, n);
        zs = z_p; ys = y_p; nvs = nv_p;
        if (mean_mode == PACOH_MEAN_VECTOR) ms = m_p;
    }
    const bool padded = npad != n;
    int rc = lml_dense_impl<T>(zs, z_div, ms, mean_mode, ys, y_div, ls, os, noise, nvs, at(g_lml, b0), atw(lml, b0),
                               padded && d_z ? dz_p : dzs, (padded && d_mean && mean_mode == PACOH_MEAN_VECTOR) ? dm_p : dms,
                               atw(d_ls, b0 * f), atw(d_os, b0), atw(d_noise, b0), info + b0, w, Bc, P, npad, f_arg, dtype, s);
    if (rc) return rc;
    if (padded && d_ls) {                               // cut the padded gradients back
        if (d_z) rows_copy<T>(dz_p, dzs, Bc, npad, n, f, s);
        if (d_mean && mean_mode == PACOH_MEAN_VECTOR) rows_copy<T>(dm_p, dms, Bc, npad, n, 1, s);
    }
    return launch_status();
}

template <typename T>
static int lml_dense_slabs(const void* z, int z_div, const void* mean, int mean_mode, const void* y, int y_div, const void* ls,
                           const void* os, const void* noise, const int32_t* n_valid, const void* g_lml, void* lml, void* d_z,
                           void* d_mean, void* d_ls, void* d_os, void* d_noise, int32_t* info, void* workspace, size_t workspace_bytes,
                           int B, int P, int n, int f_arg, int dtype, hipStream_t s) {
    const int f = features_of(f_arg), want_grad = d_ls != nullptr;
    int tb = 0;                                          // tasks per slab; 0: the whole call at once
    if (lml_slab_bytes(B, n, f, dtype, want_grad) > workspace_bytes) {
        // slabs of WHOLE tasks (all P problems of a task share its inputs): needs the task-major layout b = t * P + p
        if (B % P != 0 || (z_div != 1 && z_div != P) || (y_div != 1 && y_div != P)) return PACOH_EINVAL;
        const size_t one = lml_slab_bytes(P, n, f, dtype, want_grad);
        if (one == 0 || one > workspace_bytes) return PACOH_EINVAL;
        tb = (int)(workspace_bytes / one);
        while (tb > 1 && lml_slab_bytes(tb * P, n, f, dtype, want_grad) > workspace_bytes) --tb;
    }
    if (tb == 0) return lml_dense_slab<T>(z, z_div, mean, mean_mode, y, y_div, ls, os, noise, n_valid, g_lml, lml, d_z, d_mean, d_ls, d_os,
                                          d_noise, info, workspace, 0, B, P, n, f_arg, dtype, s);
    const int T_ = B / P;
    for (int t0 = 0; t0 < T_; t0 += tb) {
        const int tc = T_ - t0 < tb ? T_ - t0 : tb;
        const int rc = lml_dense_slab<T>(z, z_div, mean, mean_mode, y, y_div, ls, os, noise, n_valid, g_lml, lml, d_z, d_mean, d_ls, d_os,
                                         d_noise, info, workspace, (long)t0 * P, tc * P, P, n, f_arg, dtype, s);
        if (rc) return rc;
    }
    return PACOH_OK;
}

extern "C" int pacoh_gp_lml_dense(const void* z, int z_div, const void* mean, int mean_mode, const void* y, int y_div,
                                  const void* lengthscale, const void* outputscale, const void* noise, const int32_t* n_valid,
                                  const void* g_lml, void* lml, void* d_z, void* d_mean, void* d_lengthscale, void* d_outputscale,
                                  void* d_noise, int32_t* info, void* workspace, size_t workspace_bytes, int B, int P, int n, int f,
                                  int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!z || !y || !lengthscale || !noise || !lml || !info || !workspace || B <= 0 || P <= 0 || n <= 0 || f <= 0 || z_div <= 0 ||
        y_div <= 0)
        return PACOH_EINVAL;
    if (mean_mode != PACOH_MEAN_ZERO && !mean) return PACOH_EINVAL;
    if (d_lengthscale && !d_noise) return PACOH_EINVAL;
    if (features_of(f) > PACOH_MAX_FEATURES || features_of(f) <= 0 || kernel_of(f) > PACOH_KERNEL_COSINE) return PACOH_ELIMIT;
    if (dtype == PACOH_F32)
        return lml_dense_slabs<float>(z, z_div, mean, mean_mode, y, y_div, lengthscale, outputscale, noise, n_valid, g_lml, lml, d_z, d_mean,
                                      d_lengthscale, d_outputscale, d_noise, info, workspace, workspace_bytes, B, P, n, f, dtype, (hipStream_t)stream);
    return lml_dense_slabs<double>(z, z_div, mean, mean_mode, y, y_div, lengthscale, outputscale, noise, n_valid, g_lml, lml, d_z, d_mean,
                                   d_lengthscale, d_outputscale, d_noise, info, workspace, workspace_bytes, B, P, n, f, dtype, (hipStream_t)stream);
}

static size_t predict_core_bytes(int B, int n, int m, int dtype) {
    const size_t e = dtype == PACOH_F64 ? 8 : 4;
    const size_t cb = (n > 512 && n <= 1024) ? align256((dtype == PACOH_F64 ? chol_blocked_scratch<double>(B, n) : chol_blocked_scratch<float>(B, n)) * e) : 0;
    return align256((size_t)B * n * n * e) + 2 * align256((size_t)B * n * m * e) + 2 * align256((size_t)B * n * e) +
           align256((size_t)B * e) + cb + 256;
}
// (the query does not know f: the padded copy of the context inputs is sized for PACOH_MAX_FEATURES -- 16 / n of the matrix buffer)
extern "C" size_t pacoh_gp_predict_dense_workspace_bytes(int B, int n, int m, int dtype) {
    if (B <= 0 || n <= 0 || m <= 0) return 0;
    const size_t e = dtype == PACOH_F64 ? 8 : 4;
    const int npad = dense_padded_n(n, dtype);
    return predict_core_bytes(B, npad, m, dtype) +
           (npad != n ? align256((size_t)B * npad * PACOH_MAX_FEATURES * e) + 2 * align256((size_t)B * npad * e) + align256((size_t)B * 4) : 0);
}
// the context side padded as for the LML (dense_padded_n); the test side and the outputs have m points and are untouched
template <typename T>
static int predict_dense_padded(const void* z_ctx, int z_div, const void* mean_ctx, int mean_mode, const void* y, int y_div,
                                const void* z_tst, int zt_div, const void* mean_tst, const void* ls, const void* os, const void* noise,
                                const int32_t* n_valid, void* mu, void* var, void* cov, int32_t* info, void* workspace, int B, int P,
                                int n, int m, int f_arg, int dtype, hipStream_t s) {
    const int f = features_of(f_arg), npad = dense_padded_n(n, dtype);
    if (npad == n)
        return predict_dense_impl<T>(z_ctx, z_div, mean_ctx, mean_mode, y, y_div, z_tst, zt_div, mean_tst, ls, os, noise, n_valid, mu, var,
                                     cov, info, workspace, B, P, n, m, f_arg, dtype, s);
    const size_t e = sizeof(T);
    unsigned char* w = (unsigned char*)workspace;
    T* z_p = (T*)w;  w += align256((size_t)B * npad * PACOH_MAX_FEATURES * e);
    T* m_p = (T*)w;  w += align256((size_t)B * npad * e);
    T* y_p = (T*)w;  w += align256((size_t)B * npad * e);
    int32_t* nv_p = (int32_t*)w; w += align256((size_t)B * 4);
    const long zrows = (B + z_div - 1) / z_div, yrows = (B + y_div - 1) / y_div;
    rows_copy<T>(z_ctx, z_p, zrows, n, npad, f, s);
    if (mean_mode == PACOH_MEAN_VECTOR) rows_copy<T>(mean_ctx, m_p, B, n, npad, 1, s);
    rows_copy<T>(y, y_p, yrows, n, npad, 1, s);
    hipLaunchKernelGGL(dense_nv_kernel, dim3((unsigned)((yrows + 255) / 256)), dim3(256), 0, s, n_valid, nv_p, (int)yrows, n);
    return predict_dense_impl<T>(z_p, z_div, mean_mode == PACOH_MEAN_VECTOR ? (const void*)m_p : mean_ctx, mean_mode, y_p, y_div, z_tst, zt_div,
                                 mean_tst, ls, os, noise, nv_p, mu, var, cov, info, w, B, P, npad, m, f_arg, dtype, s);
}

extern "C" int pacoh_gp_predict_dense(const void* z_ctx, int z_div, const void* mean_ctx, int mean_mode, const void* y, int y_div,
                                      const void* z_tst, int zt_div, const void* mean_tst, const void* lengthscale,
                                      const void* outputscale, const void* noise, const int32_t* n_valid, void* mu, void* var,
                                      void* cov, int32_t* info, void* workspace, int B, int P, int n, int m, int f, int dtype,
                                      void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!z_ctx || !y || !z_tst || !lengthscale || !noise || !mu || !var || !info || !workspace || B <= 0 || P <= 0 || n <= 0 ||
        m <= 0 || f <= 0 || z_div <= 0 || y_div <= 0 || zt_div <= 0)
        return PACOH_EINVAL;
    if (mean_mode != PACOH_MEAN_ZERO && (!mean_ctx || !mean_tst)) return PACOH_EINVAL;
    if (features_of(f) > PACOH_MAX_FEATURES || features_of(f) <= 0 || kernel_of(f) > PACOH_KERNEL_COSINE) return PACOH_ELIMIT;
    if (dtype == PACOH_F32)
        return predict_dense_padded<float>(z_ctx, z_div, mean_ctx, mean_mode, y, y_div, z_tst, zt_div, mean_tst, lengthscale, outputscale,
                                           noise, n_valid, mu, var, cov, info, workspace, B, P, n, m, f, dtype, (hipStream_t)stream);
    return predict_dense_padded<double>(z_ctx, z_div, mean_ctx, mean_mode, y, y_div, z_tst, zt_div, mean_tst, lengthscale, outputscale,
                                        noise, n_valid, mu, var, cov, info, workspace, B, P, n, m, f, dtype, (hipStream_t)stream);
}
