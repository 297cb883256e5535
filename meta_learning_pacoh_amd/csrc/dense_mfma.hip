// Dense (large-n) Cholesky / Gaussian log-density, MFMA version: one 256-thread workgroup per matrix in
// HBM/L2, right-looking with 32-wide panels.  Per panel: the 32x32 diagonal block is factored and inverted
// by one wavefront in LDS; the panel below it is staged in LDS once (<= 138 KB at n = 512 fp64) and
// L21 = A21 * L11^-T as well as the trailing update A22 -= L21 L21^T run on the matrix cores
// (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, one LDS operand read per lane per MFMA), each 16x16
// block of A22 making exactly one HBM/L2 round trip per panel.  Solves and log-density as in dense.hip,
// which remains the fallback for matrices whose panel does not fit in LDS.
// Replaces torch/gpytorch MultivariateNormal.log_prob -> LAPACK potrf/potrs on the reference's CPU path
// (large-context configuration; joint test log-likelihood of abstract.py:134-163).
#include "common.h"

namespace pacoh {

using f32x4_t = __attribute__((ext_vector_type(4))) float;
using f64x4_t = __attribute__((ext_vector_type(4))) double;

template <typename T> struct Mf;
template <> struct Mf<float> {
    using acc = f32x4_t;
    static __device__ __forceinline__ acc mma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int g, int q) { return 4 * g + q; }          // C/D row of register q
};
template <> struct Mf<double> {
    using acc = f64x4_t;
    static __device__ __forceinline__ acc mma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int g, int q) { return g + 4 * q; }          // f64 uses a different C/D map
};

constexpr int DNB = 32;            // panel width
constexpr int DLP = 36;            // leading dimension of the LDS panel / inverse images

template <typename T>
__global__ void __launch_bounds__(256) chol_dense_mfma_kernel(T* __restrict__ A, const T* __restrict__ resid,
                                                              T* __restrict__ logp, T* __restrict__ alpha_out,
                                                              int32_t* __restrict__ info, T scale, int n, int mpad, int attempt) {
    if (attempt > 0 && info && info[blockIdx.x] >= 0) return;      // jitter-ladder retry: only the failed problems
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sm = reinterpret_cast<T*>(smem_raw);
    T (*Ds)[DNB + 1] = reinterpret_cast<T (*)[DNB + 1]>(sm);          // diagonal block / L11
    T* Li = sm + DNB * (DNB + 1);                                      // L11^-1, [32][DLP]
    T* Pn = Li + DNB * DLP;                                            // panel, [mpad][DLP]
    T* red = Pn + (size_t)mpad * DLP;                                  // [8]: sums, fail flag
    T* rv = red + 8;                                                   // [n]
    using Acc = typename Mf<T>::acc;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    T* Ab = A + (size_t)blockIdx.x * n * n;
    if (tid == 0) red[4] = 0;
    for (int q = tid; q < n; q += 256) rv[q] = resid[(size_t)blockIdx.x * n + q];
    T logdet_part = 0;
    __syncthreads();

    for (int k0 = 0; k0 < n; k0 += DNB) {
        const int kb = (n - k0 < DNB) ? (n - k0) : DNB;
        const int t0 = k0 + kb, m = n - t0;                            // trailing rows
        // 1. diagonal block -> LDS (identity padded)
        for (int q = tid; q < DNB * DNB; q += 256) {
            const int rr = q / DNB, c = q - rr * DNB;
            T v = (rr == c) ? T(1) : T(0);
            if (rr < kb && c <= rr) v = Ab[(size_t)(k0 + rr) * n + k0 + c];
            Ds[rr][c] = v;
        }
        __syncthreads();
        // 2. wave 0: factor (left-looking, pivot by shuffle) and invert (lane c owns column c of L11^-1)
        if (tid < 64) {
            const int rr = tid & 31;
            for (int j = 0; j < DNB; ++j) {
                T s = Ds[rr][j];
                for (int c = 0; c < j; ++c) s = fma(-Ds[rr][c], Ds[j][c], s);
                T piv = __shfl(s, j, 64);
                if (!(piv > T(0))) { if (tid == 0) red[4] = 1; piv = 1; }
                const T d = t_sqrt<T>(piv);
                if (tid < 32) {
                    if (rr == j) Ds[j][j] = d;
                    else if (rr > j) Ds[rr][j] = s / d;
                }
            }
            if (tid < kb) logdet_part += t_log<T>(Ds[tid][tid]);
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
                for (int i = 0; i < DNB; ++i) Li[i * DLP + rr] = x[i];                 // row-major L11^-1 (zeros above)
            }
        }
        __syncthreads();
        for (int q = tid; q < DNB * DNB; q += 256) {
            const int rr = q / DNB, c = q - rr * DNB;
            if (rr < kb && c <= rr) Ab[(size_t)(k0 + rr) * n + k0 + c] = Ds[rr][c];
        }
        if (m > 0) {
            // 3. stage the panel A21 (m x kb, zero padded to 16-row blocks x 32 columns)
            const int mb = (m + 15) / 16;
            for (int q = tid; q < mb * 16 * DNB; q += 256) {
                const int rr = q / DNB, c = q - rr * DNB;
                Pn[(size_t)rr * DLP + c] = (rr < m && c < kb) ? Ab[(size_t)(t0 + rr) * n + k0 + c] : T(0);
            }
            __syncthreads();
            // 4. L21 = A21 * L11^-T on the matrix core, in place in LDS and written back to HBM
            for (int ib = wave; ib < mb; ib += 4) {
                Acc acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
                for (int c = 0; c < DNB / 4; ++c) {
                    const T a = Pn[(size_t)(ib * 16 + r) * DLP + 4 * c + g];
                    acc0 = Mf<T>::mma(a, Li[r * DLP + 4 * c + g], acc0);
                    acc1 = Mf<T>::mma(a, Li[(16 + r) * DLP + 4 * c + g], acc1);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = ib * 16 + Mf<T>::row(g, q);
                    Pn[(size_t)row * DLP + r] = acc0[q];
                    Pn[(size_t)row * DLP + 16 + r] = acc1[q];
                    if (row < m) {
                        if (r < kb) Ab[(size_t)(t0 + row) * n + k0 + r] = acc0[q];
                        if (16 + r < kb) Ab[(size_t)(t0 + row) * n + k0 + 16 + r] = acc1[q];
                    }
                }
            }
            __syncthreads();
            // 5. trailing update A22 -= L21 L21^T, lower 16x16 blocks dealt to the waves
            int cnt = 0;
            for (int ib = 0; ib < mb; ++ib) {
                for (int jb = 0; jb <= ib; ++jb, ++cnt) {
                    if ((cnt & 3) != wave) continue;
                    Acc acc;
                    T* cp = Ab + (size_t)(t0 + ib * 16) * n + t0 + jb * 16 + r;
                    const bool colok = jb * 16 + r < m;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = Mf<T>::row(g, q);
                        acc[q] = (colok && ib * 16 + row < m) ? cp[(size_t)row * n] : T(0);
                    }
#pragma unroll
                    for (int c = 0; c < DNB / 4; ++c) {
                        const T a = -Pn[(size_t)(ib * 16 + r) * DLP + 4 * c + g];
                        const T b = Pn[(size_t)(jb * 16 + r) * DLP + 4 * c + g];
                        acc = Mf<T>::mma(a, b, acc);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = Mf<T>::row(g, q);
                        if (colok && ib * 16 + row < m) cp[(size_t)row * n] = acc[q];
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---- forward solve L u = r, blocked (L read back from HBM/L2) ----------------------------------
    for (int k0 = 0; k0 < n; k0 += DNB) {
        const int kb = (n - k0 < DNB) ? (n - k0) : DNB;
        for (int q = tid; q < DNB * DNB; q += 256) {
            const int rr = q / DNB, c = q - rr * DNB;
            Ds[rr][c] = (rr < kb && c <= rr) ? Ab[(size_t)(k0 + rr) * n + k0 + c] : ((rr == c) ? T(1) : T(0));
        }
        __syncthreads();
        if (tid < 64) {
            const int rr = tid & 31;
            T v = (rr < kb) ? rv[k0 + rr] : T(0);
            for (int c = 0; c < DNB; ++c) {
                const T uc = __shfl(v, c, 64) / Ds[c][c];
                if (rr == c) v = uc;
                else if (rr > c) v = fma(-Ds[rr][c], uc, v);
            }
            if (tid < kb) rv[k0 + tid] = v;
        }
        __syncthreads();
        for (int rr = k0 + kb + tid; rr < n; rr += 256) {
            const T* ap = Ab + (size_t)rr * n + k0;
            T s = rv[rr];
            for (int c = 0; c < kb; ++c) s = fma(-ap[c], rv[k0 + c], s);
            rv[rr] = s;
        }
        __syncthreads();
    }
    T quad_part = 0;
    for (int q = tid; q < n; q += 256) quad_part = fma(rv[q], rv[q], quad_part);
    quad_part = subwave_sum<T>(quad_part, 64);
    logdet_part = subwave_sum<T>(logdet_part, 64);
    __syncthreads();
    if (lane == 0) red[wave] = quad_part;
    __syncthreads();
    const T quad = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    if (lane == 0) red[wave] = logdet_part;
    __syncthreads();
    const T logdet = red[0] + red[1] + red[2] + red[3];
    const bool ok = red[4] == T(0);
    if (tid == 0) {
        const T LOG2PI = T(1.8378770664093453);
        const T lp = T(-0.5) * (quad + T(2) * logdet + T(n) * LOG2PI) * scale;
        logp[blockIdx.x] = ok ? lp : T(NAN);
        if (info) info[blockIdx.x] = ok ? attempt : -1;
    }
    if (!alpha_out) return;
    // ---- backward solve L^T alpha = u, blocked from the bottom ----------------------------------------
    const int nblk = (n + DNB - 1) / DNB;
    for (int kbk = nblk - 1; kbk >= 0; --kbk) {
        const int k0 = kbk * DNB;
        const int kb = (n - k0 < DNB) ? (n - k0) : DNB;
        __syncthreads();
        for (int q = tid; q < DNB * DNB; q += 256) {
            const int rr = q / DNB, c = q - rr * DNB;
            Ds[rr][c] = (rr < kb && c <= rr) ? Ab[(size_t)(k0 + rr) * n + k0 + c] : ((rr == c) ? T(1) : T(0));
        }
        __syncthreads();
        if (tid < 64) {
            const int rr = tid & 31;
            T v = (rr < kb) ? rv[k0 + rr] : T(0);
            for (int c = DNB - 1; c >= 0; --c) {
                const T ac = __shfl(v, c, 64) / Ds[c][c];
                if (rr == c) v = ac;
                else if (rr < c) v = fma(-Ds[c][rr], ac, v);
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

// returns 1 when the panel does not fit in LDS (caller falls back to the VALU kernel of dense.hip)
template <typename T>
static int launch_dense_mfma(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale,
                             int B, int n, int attempt, hipStream_t s) {
    const int mpad = n > DNB ? (n - DNB + 15) / 16 * 16 : 16;        // rows of the largest panel, in 16-row blocks
    const size_t elems = (size_t)DNB * (DNB + 1) + (size_t)DNB * DLP + (size_t)mpad * DLP + 8 + n;
    const size_t lds = elems * sizeof(T);
    if (lds > 160u * 1024u) return 1;
    auto kern = chol_dense_mfma_kernel<T>;
    if (lds > 64u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return 1;
    hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, s, (T*)A, (const T*)resid, (T*)logp, (T*)alpha_out, info, (T)scale, n, mpad, attempt);
    return launch_status();
}

int dense_mfma_try(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n,
                   int dtype, int attempt, hipStream_t s) {
    return dtype == PACOH_F32 ? launch_dense_mfma<float>(A, resid, logp, alpha_out, info, scale, B, n, attempt, s)
                              : launch_dense_mfma<double>(A, resid, logp, alpha_out, info, scale, B, n, attempt, s);
}

}  // namespace pacoh
