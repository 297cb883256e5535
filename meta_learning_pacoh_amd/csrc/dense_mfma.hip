// Dense (large-n) Cholesky / Gaussian log-density, MFMA version: one 256-thread workgroup per matrix in
// HBM/L2, right-looking with 32-wide panels.  Per panel: the 32x32 diagonal block is factored and inverted
// by one wavefront in LDS; the panel below it is staged in LDS once (<= 138 KB at n = 512 fp64) and
// L21 = A21 * L11^-T as well as the trailing update A22 -= L21 L21^T run on the matrix cores
// (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32, one LDS operand read per lane per MFMA), each 16x16
// block of A22 making exactly one HBM/L2 round trip per panel.  The forward solve is fused into the panel loop and the
// backward solve multiplies by the saved inverse diagonal blocks, so neither has a serial substitution chain.  dense.hip
// remains the fallback for matrices whose panel does not fit in LDS.
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

// Cholesky factor and inverse of the 32x32 diagonal block held in Ds (lower part, identity padded), by ONE wavefront.
// Right-looking with row rr in the registers of lane rr: per step the lanes publish column j to LDS, read it back as broadcast
// reads (all reads of a step are issued together), and apply the rank-1 update to their own row.  The left-looking form this
// replaces waited for an LDS round trip per inner iteration (496 of them) and cost 40 us per block in fp64; this one ~4 us.
// On return Ds holds L11 (lower), invd[j] = 1 / L11[j][j], Li = L11^-1 (row major, zeros above the diagonal).
template <typename T>
__device__ __forceinline__ void factor_invert_diag32(T (*Ds)[DNB + 1], T* __restrict__ Li, T* __restrict__ colb /*[64]*/,
                                                     T* __restrict__ invd /*[32]*/, T* __restrict__ fail, int lane) {
#ifdef PACOH_FACT_DEBUG
    long long tf0 = wall_clock64();
#endif
    const int rr = lane & 31, h = lane >> 5;         // lane = (row rr, column half h): 16 columns of the row in registers
    const bool wr = lane < 32;
    T a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = Ds[rr][16 * h + c];
#pragma unroll
    for (int j = 0; j < DNB; ++j) {
        const int jh = j >> 4, jl = j & 15;
        T* cb = colb + (j & 1) * DNB;
        if (h == jh) cb[rr] = a[jl];
        __builtin_amdgcn_wave_barrier();             // LDS ops of one wave execute in order; only the compiler must keep it
        // one batch of reads per step (pivot, own entry, the 16 entries of this lane's column half), issued together
        T piv = cb[j];
        const T own = cb[rr];
        T cv[16];
#pragma unroll
        for (int cl = 0; cl < 16; ++cl) cv[cl] = cb[16 * h + cl];
        if (!(piv > T(0))) { if (lane == 0) *fail = 1; piv = 1; }
        const T d = t_sqrt<T>(piv);
        const T inv = T(1) / d;
        if (lane == 0) invd[j] = inv;
        const T lown = own * inv;                    // L[rr][j] (meaningful for rr > j)
        const T nl2 = -lown * inv;
#pragma unroll
        for (int cl = 0; cl < 16; ++cl) {
            const T upd = fma(nl2, cv[cl], a[cl]);   // a[rr][c] -= L[rr][j] L[c][j]
            a[cl] = (16 * h + cl > j) ? upd : a[cl];
        }
        if (h == jh) a[jl] = (rr == j) ? d : lown;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);           // keep the 32 unrolled steps from being interleaved (register pressure)
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) if (16 * h + c <= rr) Ds[rr][16 * h + c] = a[c];
    __builtin_amdgcn_wave_barrier();
#ifdef PACOH_FACT_DEBUG
    long long tf1 = wall_clock64();
#endif
    // inverse: lane rr owns column rr of X = L11^-1 and keeps it in LDS (Li[i][rr]: lane-private words, conflict-free);
    // L entries and 1/diag come as broadcast reads.  Deliberately NOT fully unrolled: with everything unrolled the compiler
    // hoists all 496 broadcast loads to the top and spills ~1000 registers.
    if (wr) {
#pragma unroll 1
        for (int i = 0; i < DNB; ++i) {
            T sacc = (i == rr) ? T(1) : T(0);
#pragma unroll 8
            for (int j = 0; j < i; ++j) sacc = fma(-Ds[i][j], Li[j * DLP + rr], sacc);
            Li[i * DLP + rr] = (i < rr) ? T(0) : sacc * invd[i];
        }
    }
#ifdef PACOH_FACT_DEBUG
    if (lane == 0) { g_tdbg[0] = tf1 - tf0; g_tdbg[1] = wall_clock64() - tf1; }
#endif
}

template <typename T, int NT>
__global__ void __launch_bounds__(NT) chol_dense_mfma_kernel(T* __restrict__ A, const T* __restrict__ resid,
                                                              T* __restrict__ logp, T* __restrict__ alpha_out,
                                                              int32_t* __restrict__ info, T scale, int n, int mpad, int attempt) {
    if (attempt > 0 && info && info[blockIdx.x] >= 0) return;      // jitter-ladder retry: only the failed problems
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* sm = reinterpret_cast<T*>(smem_raw);
    T (*Ds)[DNB + 1] = reinterpret_cast<T (*)[DNB + 1]>(sm);          // diagonal block / L11
    T* Li = sm + DNB * (DNB + 1);                                      // L11^-1, [32][DLP]
    T* Pn = Li + DNB * DLP;                                            // panel, [mpad][DLP]
    T* red = Pn + (size_t)mpad * DLP;                                  // [128]: per-wave sums [0..16), fail flag [16], column scratch [32..96), 1/diag [96..128)
    T* rv = red + 128;                                                 // [n] residual -> u -> alpha
    constexpr int NW = NT / 64;
    using Acc = typename Mf<T>::acc;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    T* Ab = A + (size_t)blockIdx.x * n * n;
    if (tid == 0) red[16] = 0;
    for (int q = tid; q < n; q += NT) rv[q] = resid[(size_t)blockIdx.x * n + q];
    T logdet_part = 0;
    __syncthreads();

    for (int k0 = 0; k0 < n; k0 += DNB) {
        const int kb = (n - k0 < DNB) ? (n - k0) : DNB;
        const int t0 = k0 + kb, m = n - t0;                            // trailing rows
        // 1. diagonal block -> LDS (identity padded)
        for (int q = tid; q < DNB * DNB; q += NT) {
            const int rr = q / DNB, c = q - rr * DNB;
            T v = (rr == c) ? T(1) : T(0);
            if (rr < kb && c <= rr) v = Ab[(size_t)(k0 + rr) * n + k0 + c];
            Ds[rr][c] = v;
        }
        __syncthreads();
        // 2. wave 0: factor and invert the diagonal block
        if (tid < 64) {
            factor_invert_diag32<T>(Ds, Li, red + 32, red + 96, red + 16, tid);
            if (tid < kb) logdet_part += t_log<T>(Ds[tid][tid]);
        }
        __syncthreads();
        // L11 -> lower triangle; the strictly-lower part of L11^-1 is kept, transposed, in the (otherwise unused) strictly
        // upper part of the diagonal block: the backward solve reads it from there (its diagonal is 1 / L11's diagonal)
        for (int q = tid; q < DNB * DNB; q += NT) {
            const int rr = q / DNB, c = q - rr * DNB;
            if (rr < kb && c <= rr) Ab[(size_t)(k0 + rr) * n + k0 + c] = Ds[rr][c];
            if (rr < kb && c < rr) Ab[(size_t)(k0 + c) * n + k0 + rr] = Li[rr * DLP + c];
        }
        // forward solve, fused: u_k = L11^-1 r_k now, r_rest -= L21 u_k once L21 is in LDS (no serial substitution chain)
        T u_reg = 0;
        if (tid < kb) {
            for (int c = 0; c <= tid; ++c) u_reg = fma(Li[tid * DLP + c], rv[k0 + c], u_reg);
        }
        if (m <= 0) {
            __syncthreads();
            if (tid < kb) rv[k0 + tid] = u_reg;
        }
        if (m > 0) {
            // 3. stage the panel A21 (m x kb, zero padded to 16-row blocks x 32 columns)
            const int mb = (m + 15) / 16;
            for (int q = tid; q < mb * 16 * DNB; q += NT) {
                const int rr = q / DNB, c = q - rr * DNB;
                Pn[(size_t)rr * DLP + c] = (rr < m && c < kb) ? Ab[(size_t)(t0 + rr) * n + k0 + c] : T(0);
            }
            __syncthreads();
            if (tid < kb) rv[k0 + tid] = u_reg;
            // 4. L21 = A21 * L11^-T on the matrix core, in place in LDS and written back to HBM
            for (int ib = wave; ib < mb; ib += NW) {
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
            for (int rr = tid; rr < m; rr += NT) {                          // r_rest -= L21 u_k (L21 rows from the LDS panel)
                T sacc = rv[t0 + rr];
#pragma unroll 8
                for (int c = 0; c < DNB; ++c) sacc = fma(-Pn[(size_t)rr * DLP + c], (c < kb) ? rv[k0 + c] : T(0), sacc);
                rv[t0 + rr] = sacc;
            }
            // 5. trailing update A22 -= L21 L21^T: the lower 16x16 blocks are dealt to the waves, two per step so that the
            //    L2/HBM round trip of one block's accumulator overlaps the other's MFMAs
            const int total = mb * (mb + 1) / 2;
            for (int c0 = wave; c0 < total; c0 += 2 * NW) {
                int ibs[2], jbs[2];
                bool live[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int c = c0 + u * NW;
                    live[u] = c < total;
                    int ib = (int)((sqrtf(8.0f * (float)c + 1.0f) - 1.0f) * 0.5f);
                    while (ib * (ib + 1) / 2 > c) --ib;
                    while ((ib + 1) * (ib + 2) / 2 <= c) ++ib;
                    ibs[u] = ib; jbs[u] = c - ib * (ib + 1) / 2;
                }
                Acc acc[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const T* cp = Ab + (size_t)(t0 + ibs[u] * 16) * n + t0 + jbs[u] * 16 + r;
                    const bool colok = live[u] && jbs[u] * 16 + r < m;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = Mf<T>::row(g, q);
                        acc[u][q] = (colok && ibs[u] * 16 + row < m) ? cp[(size_t)row * n] : T(0);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    if (!live[u]) continue;
#pragma unroll
                    for (int c = 0; c < DNB / 4; ++c) {
                        const T a = -Pn[(size_t)(ibs[u] * 16 + r) * DLP + 4 * c + g];
                        const T b = Pn[(size_t)(jbs[u] * 16 + r) * DLP + 4 * c + g];
                        acc[u] = Mf<T>::mma(a, b, acc[u]);
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    T* cp = Ab + (size_t)(t0 + ibs[u] * 16) * n + t0 + jbs[u] * 16 + r;
                    const bool colok = live[u] && jbs[u] * 16 + r < m;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = Mf<T>::row(g, q);
                        if (colok && ibs[u] * 16 + row < m) cp[(size_t)row * n] = acc[u][q];
                    }
                }
            }
        }
        __syncthreads();
    }

    // (the forward solve L u = r happened panel by panel above: rv holds u)
    T quad_part = 0;
    for (int q = tid; q < n; q += NT) quad_part = fma(rv[q], rv[q], quad_part);
    quad_part = subwave_sum<T>(quad_part, 64);
    logdet_part = subwave_sum<T>(logdet_part, 64);
    __syncthreads();
    if (lane == 0) red[wave] = quad_part;
    __syncthreads();
    T quad = 0;
    for (int w = 0; w < NW; ++w) quad += red[w];
    __syncthreads();
    if (lane == 0) red[wave] = logdet_part;
    __syncthreads();
    T logdet = 0;
    for (int w = 0; w < NW; ++w) logdet += red[w];
    const bool ok = red[16] == T(0);
    if (tid == 0) {
        const T LOG2PI = T(1.8378770664093453);
        const T lp = T(-0.5) * (quad + T(2) * logdet + T(n) * LOG2PI) * scale;
        logp[blockIdx.x] = ok ? lp : T(NAN);
        if (info) info[blockIdx.x] = ok ? attempt : -1;
    }
    if (!alpha_out) return;
    // ---- backward solve L^T alpha = u, blocked from the bottom: alpha_k = L11^-T w_k as a 32x32 product with the inverse
    //      block saved in the upper triangle, then w_i -= sum_c L[k0+c][i] alpha[k0+c] for the rows above
    const int nblk = (n + DNB - 1) / DNB;
    for (int kbk = nblk - 1; kbk >= 0; --kbk) {
        const int k0 = kbk * DNB;
        const int kb = (n - k0 < DNB) ? (n - k0) : DNB;
        __syncthreads();
        T a_reg = 0;
        if (tid < kb) {
            const T* zrow = Ab + (size_t)(k0 + tid) * n + k0;           // [tid] = L diag, [c > tid] = Z11[c][tid]
            a_reg = rv[k0 + tid] / zrow[tid];
            for (int c = tid + 1; c < kb; ++c) a_reg = fma(zrow[c], rv[k0 + c], a_reg);
        }
        __syncthreads();
        if (tid < kb) rv[k0 + tid] = a_reg;
        __syncthreads();
        for (int i = tid; i < k0; i += NT) {
            T sacc = rv[i];
            for (int c = 0; c < kb; ++c) sacc = fma(-Ab[(size_t)(k0 + c) * n + i], rv[k0 + c], sacc);
            rv[i] = sacc;
        }
    }
    __syncthreads();
    for (int q = tid; q < n; q += NT) alpha_out[(size_t)blockIdx.x * n + q] = ok ? rv[q] : T(NAN);
}

// returns 1 when the panel does not fit in LDS (caller falls back to the VALU kernel of dense.hip)
template <typename T>
static int launch_dense_mfma(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale,
                             int B, int n, int attempt, hipStream_t s) {
    const int mpad = n > DNB ? (n - DNB + 15) / 16 * 16 : 16;        // rows of the largest panel, in 16-row blocks
    const size_t elems = (size_t)DNB * (DNB + 1) + (size_t)DNB * DLP + (size_t)mpad * DLP + 128 + n;
    const size_t lds = elems * sizeof(T);
    if (lds > 160u * 1024u) return 1;
    // one workgroup per matrix: large matrices get 16 wavefronts (4 per SIMD) so that the trailing update's independent
    // 16x16 blocks hide each other's L2 and MFMA latency (256 threads = 1 wave per SIMD ran the n = 512 factorisation 2x slower)
#define PACOH_CHOL_LAUNCH(nt) do { auto kern = chol_dense_mfma_kernel<T, nt>; \
        if (lds > 64u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return 1; \
        hipLaunchKernelGGL(kern, dim3(B), dim3(nt), lds, s, (T*)A, (const T*)resid, (T*)logp, (T*)alpha_out, info, (T)scale, n, mpad, attempt); } while (0)
    if (n >= 256) PACOH_CHOL_LAUNCH(1024); else if (n >= 96) PACOH_CHOL_LAUNCH(512); else PACOH_CHOL_LAUNCH(256);
#undef PACOH_CHOL_LAUNCH
    return launch_status();
}

int dense_mfma_try(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n,
                   int dtype, int attempt, hipStream_t s) {
    return dtype == PACOH_F32 ? launch_dense_mfma<float>(A, resid, logp, alpha_out, info, scale, B, n, attempt, s)
                              : launch_dense_mfma<double>(A, resid, logp, alpha_out, info, scale, B, n, attempt, s);
}

}  // namespace pacoh
