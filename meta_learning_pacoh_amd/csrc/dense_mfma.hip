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
#include "dense_diag.h"

namespace pacoh {

// Cholesky factor and inverse of the 32x32 diagonal block held in Ds (lower part, identity padded), by ONE wavefront.
// Lane rr keeps row rr in registers (both half-waves do the same work).  Right-looking: per step the pivot and the entries of
// column j reach the other lanes as v_readlane broadcasts into scalar registers -- no LDS round trip inside the elimination (the
// version this replaces published every column through LDS and waited for it 32 times: 27 us per block in fp64 with fifteen waves
// idle, a third of the whole factorisation at n = 512).  The inverse X = L11^-1 is then built column per lane, X kept in
// registers, L11 entries as LDS broadcast reads.
// On return Ds holds L11 (lower), invd[j] = 1 / L11[j][j], Li = L11^-1 (row major, zeros above the diagonal).
template <typename T>
__device__ __forceinline__ void factor_invert_diag32(T (*Ds)[DNB + 1], T* __restrict__ Li, T* __restrict__ colb /*unused*/,
                                                     T* __restrict__ invd /*[32]*/, T* __restrict__ fail, int lane) {
    int rr = lane & 31;
    asm volatile("" : "+v"(rr));                     // (opaque: or the compiler lifts every lane mask below out of the panel loop
                                                     //  of the caller and spills hundreds of scalar registers to keep them)
    T a[DNB];
#pragma unroll
    for (int c = 0; c < DNB; ++c) a[c] = Ds[rr][c];
    bool bad = false;
#ifdef PACOH_FACT_DEBUG
    const long long tf0 = wall_clock64();
#endif
    ElimSteps<T, 0>::run(a, invd, bad, lane);
#ifdef PACOH_FACT_DEBUG
    const long long tf1 = wall_clock64();
#endif
    if (bad && lane == 0) *fail = 1;
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < DNB; ++c) Ds[rr][c] = a[c];          // (above the diagonal: leftovers nobody reads)
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // inverse, row rr per lane, in the registers that held row rr of L (InvSteps above)
    InvSteps<T, DNB - 1>::run(a, invd, rr);
    if (lane < 32) {
#pragma unroll
        for (int c = 0; c < DNB; ++c) Li[rr * DLP + c] = a[c];          // (zeros above the diagonal: sums of 0 x finite)
    }
#ifdef PACOH_FACT_DEBUG
    if (lane == 0) { g_tdbg[0] = tf1 - tf0; g_tdbg[1] = wall_clock64() - tf1; }
#endif
    __builtin_amdgcn_wave_barrier();
}

template <typename T, int NT>
__global__ void __launch_bounds__(NT) chol_dense_mfma_kernel(T* __restrict__ A, const T* __restrict__ resid,
                                                              T* __restrict__ logp, T* __restrict__ alpha_out,
                                                              int32_t* __restrict__ info, T scale, int n, int mpad, int attempt,
                                                              int u_only) {
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
#ifdef PACOH_CHOL_STAMPS
    __shared__ double stamp_wait[2];                   // (own storage: red[96..128) is the reciprocal-pivot scratch of factor_invert_diag32)
    if (tid == 0) { stamp_wait[0] = 0; stamp_wait[1] = 0; }
#endif
    for (int q = tid; q < n; q += NT) rv[q] = resid[(size_t)blockIdx.x * n + q];
    T logdet_part = 0;
    __syncthreads();
#ifdef PACOH_CHOL_STAMPS      // diagnostic build: python -m meta_learning_pacoh_amd._build --variant cst -DPACOH_CHOL_STAMPS=1 (tools/dense_quick.sh)
    long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tp_ = wall_clock64();
#define CSTAMP(k) do { __syncthreads(); const long long t_ = wall_clock64(); ph_[k] += t_ - tp_; tp_ = t_; } while (0)
#else
#define CSTAMP(k) do {} while (0)
#endif

    // One 16x16 block (ib, jb) of the trailing update A22 -= L21 L21^T, two at a time per wave so that the L2/HBM round trip of one
    // block's accumulator overlaps the other's MFMAs.
    auto update_pair = [&](const int (&ibs)[2], const int (&jbs)[2], const bool (&live)[2], int t0, int m) {
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
    };
    // diagonal block at k0 -> Ds (identity padded), by the threads [0, nthr)
    auto load_diag = [&](int k0, int kb, int t, int nthr) {
        for (int q = t; q < DNB * DNB; q += nthr) {
            const int rr = q / DNB, c = q - rr * DNB;
            T v = (rr == c) ? T(1) : T(0);
            if (rr < kb && c <= rr) v = Ab[(size_t)(k0 + rr) * n + k0 + c];
            Ds[rr][c] = v;
        }
    };

    // Look-ahead (round 3): the 32 x 32 diagonal block of panel k+1 is factored and inverted by wave 0 (10 us in fp64, registers
    // only) WHILE the other waves run the bulk of panel k's trailing update, which is HBM-bound (16 bytes per 64 flops): the
    // blocks of the next panel's own 32 columns are updated first by all waves, then wave 0 leaves.  (Before: one wave busy and
    // fifteen idle for 0.18 of the 0.75 ms at n = 512.)  Per panel: 1. [prologue / previous iteration] diagonal block factored,
    // 2. L11 and L11^-1 written back, 3. panel staged in LDS, 4. L21 = A21 L11^-T on the matrix cores, forward solve,
    // 5a. trailing update of block columns 0-1, 5b. wave 0: next diagonal block | others: the remaining block columns.
    {
        const int kb0 = n < DNB ? n : DNB;
        load_diag(0, kb0, tid, NT);
        __syncthreads();
        if (tid < 64) {
            factor_invert_diag32<T>(Ds, Li, red + 32, red + 96, red + 16, tid);
            if (tid < kb0) logdet_part += t_log<T>(Ds[tid][tid]);
        }
        __syncthreads();
        CSTAMP(1);
    }
    for (int k0 = 0; k0 < n; k0 += DNB) {
        const int kb = (n - k0 < DNB) ? (n - k0) : DNB;
        const int t0 = k0 + kb, m = n - t0;                            // trailing rows
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
#pragma unroll 4
            for (int q = tid; q < mb * 16 * DNB; q += NT) {                 // (unrolled: four loads per thread in flight)
                const int rr = q / DNB, c = q - rr * DNB;
                Pn[(size_t)rr * DLP + c] = (rr < m && c < kb) ? Ab[(size_t)(t0 + rr) * n + k0 + c] : T(0);
            }
            __syncthreads();
            CSTAMP(2);
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
            CSTAMP(3);
            for (int rr = tid; rr < m; rr += NT) {                          // r_rest -= L21 u_k (L21 rows from the LDS panel)
                T sacc = rv[t0 + rr];
#pragma unroll 8
                for (int c = 0; c < DNB; ++c) sacc = fma(-Pn[(size_t)rr * DLP + c], (c < kb) ? rv[k0 + c] : T(0), sacc);
                rv[t0 + rr] = sacc;
            }
            CSTAMP(4);
            // 5a. trailing update of the block columns 0 and 1 (the next panel's own 32 columns), by every wave:
            //     q < mb -> block (q, 0); q >= mb -> block (q - mb + 1, 1)
            {
                const int first = 2 * mb - 1;
                for (int c0 = wave; c0 < first; c0 += 2 * NW) {
                    int ibs[2], jbs[2];
                    bool live[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int c = c0 + u * NW;
                        live[u] = c < first;
                        ibs[u] = c < mb ? c : c - mb + 1;
                        jbs[u] = c < mb ? 0 : 1;
                    }
                    update_pair(ibs, jbs, live, t0, m);
                }
            }
            __syncthreads();                                                 // (the stores above are read back below)
            load_diag(t0, (n - t0 < DNB) ? (n - t0) : DNB, tid, NT);         // next diagonal block -> Ds: one element per thread,
            __syncthreads();                                                 // one round trip (by wave 0 alone: sixteen in a row)
            // 5b. wave 0: the next diagonal block, factored and inverted in registers | the other waves: block columns >= 2
#ifdef PACOH_CHOL_STAMPS
            const long long tb_ = wall_clock64();
#endif
            if (wave == 0) {
                const int kbn = (n - t0 < DNB) ? (n - t0) : DNB;
                factor_invert_diag32<T>(Ds, Li, red + 32, red + 96, red + 16, tid);
                if (tid < kbn) logdet_part += t_log<T>(Ds[tid][tid]);
            } else if (mb > 2) {
                const int mr = mb - 2;                                       // lower triangle of the blocks (ib >= jb >= 2)
                const int total = mr * (mr + 1) / 2;
                for (int c0 = wave - 1; c0 < total; c0 += 2 * (NW - 1)) {
                    int ibs[2], jbs[2];
                    bool live[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int c = c0 + u * (NW - 1);
                        live[u] = c < total;
                        int ib = (int)((sqrtf(8.0f * (float)c + 1.0f) - 1.0f) * 0.5f);
                        while (ib * (ib + 1) / 2 > c) --ib;
                        while ((ib + 1) * (ib + 2) / 2 <= c) ++ib;
                        ibs[u] = ib + 2; jbs[u] = c - ib * (ib + 1) / 2 + 2;
                    }
                    update_pair(ibs, jbs, live, t0, m);
                }
            }
#ifdef PACOH_CHOL_STAMPS
            if (tid == 0) ph_[6] += wall_clock64() - tb_;
            if (tid == 64) stamp_wait[0] += (double)(wall_clock64() - tb_);
            if (tid == 960) stamp_wait[1] += (double)(wall_clock64() - tb_);
#endif
        }
        __syncthreads();
        CSTAMP(5);
    }
#ifdef PACOH_CHOL_STAMPS
    if (tid == 0 && blockIdx.x == 0)
        printf("chol phases (us): load diag %.1f | factor+invert %.1f | write L11, stage panel %.1f | panel solve %.1f | resid %.1f | trailing %.1f (5b: wave 0 %.1f, wave 1 %.1f, wave 15 %.1f)\n",
               ph_[0] * 0.01, ph_[1] * 0.01, ph_[2] * 0.01, ph_[3] * 0.01, ph_[4] * 0.01, ph_[5] * 0.01, ph_[6] * 0.01, stamp_wait[0] * 0.01, stamp_wait[1] * 0.01);
#endif
#undef CSTAMP

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
#ifdef PACOH_CHOL_STAMPS
    long long tq_ = wall_clock64();
#endif
    if (tid == 0) {
        const T LOG2PI = T(1.8378770664093453);
        const T lp = T(-0.5) * (quad + T(2) * logdet + T(n) * LOG2PI) * scale;
        logp[blockIdx.x] = ok ? lp : T(NAN);
        if (info) info[blockIdx.x] = ok ? attempt : -1;
    }
    if (!alpha_out) return;
    if (u_only) {      // the caller goes on to Z = L^-1 and takes alpha = Z^T u from there (dense_alpha_kernel): no backward solve here
        for (int q = tid; q < n; q += NT) alpha_out[(size_t)blockIdx.x * n + q] = ok ? rv[q] : T(NAN);
        return;
    }
    // ---- backward solve L^T alpha = u, blocked from the bottom: alpha_k = L11^-T w_k as a 32x32 product with the inverse
    //      block saved in the upper triangle, then w_i -= sum_c L[k0+c][i] alpha[k0+c] for the rows above
    const int nblk = (n + DNB - 1) / DNB;
    // The diagonal block (L11's diagonal, Z11^T above it) goes through LDS, double-buffered (Ds and the Li area): the block of
    // panel k-1 is fetched while the rows above panel k are updated.  Read from global memory inside the dot product, every one
    // of its up to 31 steps waited for an L2 round trip (18 us per panel, 0.3 ms of the n = 512 factorisation).
    T (*Db[2])[DNB + 1] = {Ds, reinterpret_cast<T (*)[DNB + 1]>(Li)};
    auto fetch_diag = [&](int kbk, T (*D)[DNB + 1]) {
        const int k0 = kbk * DNB;
        const int kb = (n - k0 < DNB) ? (n - k0) : DNB;
        for (int q = tid; q < DNB * DNB; q += NT) {
            const int rr = q / DNB, c = q - rr * DNB;
            D[rr][c] = (rr < kb && c < kb) ? Ab[(size_t)(k0 + rr) * n + k0 + c] : T(0);
        }
    };
    __syncthreads();
#ifdef PACOH_CHOL_STAMPS
    long long bs_[3] = {0, 0, 0}, bt_ = wall_clock64();
#define BSTAMP(k) do { __syncthreads(); const long long t_ = wall_clock64(); bs_[k] += t_ - bt_; bt_ = t_; } while (0)
#else
#define BSTAMP(k) do {} while (0)
#endif
    fetch_diag(nblk - 1, Db[(nblk - 1) & 1]);
    for (int kbk = nblk - 1; kbk >= 0; --kbk) {
        const int k0 = kbk * DNB;
        const int kb = (n - k0 < DNB) ? (n - k0) : DNB;
        T (*D)[DNB + 1] = Db[kbk & 1];
        __syncthreads();
        // alpha_k[t] = w[t] / L[t][t] + sum_{c > t} Z11[c][t] w[c]: one product per thread (t = tid / 32, c = tid % 32; a
        // 256-thread workgroup takes four passes), summed over the 32 lanes of a row
        T a_new = 0;
        for (int t = tid >> 5; t < DNB; t += NT >> 5) {
            const int c = tid & 31;
            T pr = 0;
            if (t < kb && c < kb) {
                if (c == t) pr = rv[k0 + t] / D[t][t];
                else if (c > t) pr = D[t][c] * rv[k0 + c];
            }
            pr = subwave_sum<T>(pr, 32);
            if (c == 0 && t < kb) red[32 + t] = pr;               // (red[32..96): scratch, unused since the register factor)
        }
        __syncthreads();
        if (tid < kb) rv[k0 + tid] = red[32 + tid];
        __syncthreads();
        BSTAMP(0);
        if (kbk > 0) fetch_diag(kbk - 1, Db[(kbk - 1) & 1]);      // (no dependence on this panel's alpha)
        BSTAMP(1);
        for (int i = tid; i < k0; i += NT) {
            T sacc = rv[i];
            if (kb == DNB) {
#pragma unroll
                for (int c = 0; c < DNB; ++c) sacc = fma(-Ab[(size_t)(k0 + c) * n + i], rv[k0 + c], sacc);   // (32 independent loads in flight)
            } else {
                for (int c = 0; c < kb; ++c) sacc = fma(-Ab[(size_t)(k0 + c) * n + i], rv[k0 + c], sacc);
            }
            rv[i] = sacc;
        }
        BSTAMP(2);
    }
#undef BSTAMP
    __syncthreads();
#ifdef PACOH_CHOL_STAMPS
    if (tid == 0 && blockIdx.x == 0)
        printf("chol backward solve (us): %.1f (dot %.1f | fetch next diagonal block %.1f | update rows above %.1f)\n",
               (wall_clock64() - tq_) * 0.01, bs_[0] * 0.01, bs_[1] * 0.01, bs_[2] * 0.01);
#endif
    for (int q = tid; q < n; q += NT) alpha_out[(size_t)blockIdx.x * n + q] = ok ? rv[q] : T(NAN);
}

// returns 1 when the panel does not fit in LDS (caller falls back to the VALU kernel of dense.hip)
template <typename T>
static int launch_dense_mfma(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale,
                             int B, int n, int attempt, int u_only, hipStream_t s) {
    const int mpad = n > DNB ? (n - DNB + 15) / 16 * 16 : 16;        // rows of the largest panel, in 16-row blocks
    const size_t elems = (size_t)DNB * (DNB + 1) + (size_t)DNB * DLP + (size_t)mpad * DLP + 128 + n;
    const size_t lds = elems * sizeof(T);
    if (lds > 160u * 1024u) return 1;
    // one workgroup per matrix: large matrices get 16 wavefronts (4 per SIMD) so that the trailing update's independent
    // 16x16 blocks hide each other's L2 and MFMA latency (256 threads = 1 wave per SIMD ran the n = 512 factorisation 2x slower)
#define PACOH_CHOL_LAUNCH(nt) do { auto kern = chol_dense_mfma_kernel<T, nt>; \
        if (lds > 64u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return 1; \
        hipLaunchKernelGGL(kern, dim3(B), dim3(nt), lds, s, (T*)A, (const T*)resid, (T*)logp, (T*)alpha_out, info, (T)scale, n, mpad, attempt, u_only); } while (0)
    if (n >= 256) PACOH_CHOL_LAUNCH(1024); else if (n >= 96) PACOH_CHOL_LAUNCH(512); else PACOH_CHOL_LAUNCH(256);
#undef PACOH_CHOL_LAUNCH
    return launch_status();
}

bool dense_mfma_fits(int n, int dtype) {
    const int mpad = n > DNB ? (n - DNB + 15) / 16 * 16 : 16;
    const size_t elems = (size_t)DNB * (DNB + 1) + (size_t)DNB * DLP + (size_t)mpad * DLP + 128 + n;
    return elems * (dtype == PACOH_F32 ? 4 : 8) <= 160u * 1024u;
}

int dense_mfma_try(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n,
                   int dtype, int attempt, int u_only, hipStream_t s) {
    return dtype == PACOH_F32 ? launch_dense_mfma<float>(A, resid, logp, alpha_out, info, scale, B, n, attempt, u_only, s)
                              : launch_dense_mfma<double>(A, resid, logp, alpha_out, info, scale, B, n, attempt, u_only, s);
}

}  // namespace pacoh
