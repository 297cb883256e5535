// SVGD update direction with the IMQ (inverse multi-quadratic) particle kernel and its per-dimension
// median bandwidth: replaces SVGD.phi + IMQSteinKernel.forward/_bandwidth (meta_learn/svgd.py:12-23, 63-97),
// selected by GPRegressionMetaLearnedSVGD(kernel='IMQ') (GPR_meta_svgd.py:176-177).
//
//   h_d      = lower-median_{a<b} (x_bd - x_ad)^2 / log(P+1)              (or the caller's scalar bandwidth)
//   base_ij  = alpha + sum_d (x_jd - x_id)^2 / h_d,   k_ij = base_ij^beta,   kb_ij = beta base_ij^(beta-1)
//   phi_jd   = ( sum_i k_ji s_id - (2/h_d) sum_i kb_ij (x_jd - x_id)
//                + [j == b_d] (sum_il kb_il (x_ld - x_id)^2 / h_d^2) 2 (x_bd - x_ad) / log(P+1) ) / P
// The last term is the derivative THROUGH the median bandwidth: the reference builds h from the differentiable
// squared differences, so autograd sends a gradient to the later particle b of each dimension's median pair.
//
// All three kernels are HBM/latency-trivial (P*D = 50k elements at the headline shape); they are laid out so that
// every global access is coalesced along d and the P^2 matrices are read through scalar loads.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pacoh_gp.h"
#include "common.h"

using namespace pacoh;

namespace {

template <typename T> struct BitsOf;
template <> struct BitsOf<float> { using U = uint32_t; static constexpr int NB = 32; };
template <> struct BitsOf<double> { using U = uint64_t; static constexpr int NB = 64; };
__device__ __forceinline__ uint32_t to_bits(float v) { return __float_as_uint(v); }
__device__ __forceinline__ uint64_t to_bits(double v) { return (uint64_t)__double_as_longlong(v); }
__device__ __forceinline__ float from_bits(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ double from_bits(uint64_t u) { return __longlong_as_double((long long)u); }

// ---- stage 1: per-dimension lower median over the P(P-1)/2 pairs --------------------------------------------
// 16 dimensions per 256-thread workgroup, 16 lanes per dimension.  The k-th smallest of the (non-negative)
// squared differences is found by bisection on the IEEE bit pattern (monotone for v >= 0): NB-1 rounds of
// "count values below the candidate", each lane counting the pairs (a, b > a) of its rows a = ln, ln + 16, ...
// (enumerated on the fly: no pair table, so any P whose 16 coordinate columns fit in LDS -- 1024 particles and beyond),
// 4 shuffles per round.
template <typename T>
__global__ void __launch_bounds__(256) imq_bw_kernel(const T* __restrict__ X, T* __restrict__ h, T* __restrict__ dh,
                                                     int32_t* __restrict__ bidx, T log_p1, int P, int D) {
    using U = typename BitsOf<T>::U;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Pp = P | 1;
    T* xs = reinterpret_cast<T*>(smem_raw);                          // [16][Pp]
    const long npairs = (long)P * (P - 1) / 2;
    const int d0 = blockIdx.x * 16;
    for (int idx = threadIdx.x; idx < P * 16; idx += 256) {
        const int p = idx >> 4, c = idx & 15;
        const int d = min(d0 + c, D - 1);
        xs[c * Pp + p] = X[(long)p * D + d];
    }
    __syncthreads();
    const int dl = threadIdx.x >> 4, ln = threadIdx.x & 15;
    const T* xd = xs + dl * Pp;
    const long k = (npairs - 1) / 2;
    U result = 0;
    for (int bit = BitsOf<T>::NB - 2; bit >= 0; --bit) {
        const U cand = result | (U(1) << bit);
        long cnt = 0;
        for (int a = ln; a < P - 1; a += 16) {
            const T xa = xd[a];
            int c = 0;
            for (int b = a + 1; b < P; ++b) { const T df = xd[b] - xa; c += to_bits(df * df) < cand ? 1 : 0; }
            cnt += c;
        }
        cnt += __shfl_xor(cnt, 1, 64); cnt += __shfl_xor(cnt, 2, 64);
        cnt += __shfl_xor(cnt, 4, 64); cnt += __shfl_xor(cnt, 8, 64);
        if (cnt <= k) result = cand;
    }
    // first pair (row-major over a < b: what torch.median's index refers to) that attains the median
    long qmin = 0x7fffffffffffffffL;
    for (int a = ln; a < P - 1 && qmin == 0x7fffffffffffffffL; a += 16) {       // (a lane's rows ascend: its first hit is its smallest)
        const T xa = xd[a];
        for (int b = a + 1; b < P; ++b) {
            const T df = xd[b] - xa;
            if (to_bits(df * df) == result) { qmin = (long)a * P + b; break; }   // (a P + b orders the pairs like the row-major index)
        }
    }
    qmin = min(qmin, __shfl_xor(qmin, 1, 64)); qmin = min(qmin, __shfl_xor(qmin, 2, 64));
    qmin = min(qmin, __shfl_xor(qmin, 4, 64)); qmin = min(qmin, __shfl_xor(qmin, 8, 64));
    const int d = d0 + dl;
    if (ln == 0 && d < D) {
        const int a = (int)(qmin / P), b = (int)(qmin - (long)a * P);
        h[d] = from_bits(result) / log_p1;
        dh[d] = T(2) * (xd[b] - xd[a]) / log_p1;
        bidx[d] = b;
    }
}

// sum / min over the 16 lanes of a DPP row, result in every lane: quad xor 1, quad xor 2, mirror within 8, mirror within 16 (vector-rate
// moves; the __shfl_xor form goes through the LDS crossbar: 4 x ~100 cycles per bisection round, 3/4 of the small kernel's time)
__device__ __forceinline__ int row16_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);
    return v;
}
__device__ __forceinline__ int row16_min(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false));
    return v;
}

// The same for P <= 32 (the reference's sweeps use 10 / 50 particles, BASELINE config #3 20): a lane keeps its <= 31 pair values
// in registers -- pair q = ln + 16 m in row-major order over a < b --, so that a bisection round is 31 compares and 4 shuffles
// instead of a loop over LDS (41 -> 5 us at P = 20, D = 2534: the kernel is one round of workgroups, i.e. pure latency).  Same
// values, same bit patterns, same first-pair rule as the general kernel above.
template <typename T>
__global__ void __launch_bounds__(256) imq_bw_small_kernel(const T* __restrict__ X, T* __restrict__ h, T* __restrict__ dh,
                                                           int32_t* __restrict__ bidx, T log_p1, int P, int D) {
    using U = typename BitsOf<T>::U;
    constexpr int M = 31;                                            // ceil(32 * 31 / 2 / 16)
    __shared__ T xs[16][33];
    const int npairs = P * (P - 1) / 2;
    const int d0 = blockIdx.x * 16;
    for (int idx = threadIdx.x; idx < P * 16; idx += 256) {
        const int p = idx >> 4, c = idx & 15;
        xs[c][p] = X[(long)p * D + min(d0 + c, D - 1)];
    }
    __syncthreads();
    const int dl = threadIdx.x >> 4, ln = threadIdx.x & 15;
    const T* xd = xs[dl];
    U vals[M];
    {
        // pair q -> (a, b): rows a hold P - 1 - a pairs each; walk the rows once (q ascends by 16 per step)
        int a = 0, row0 = 0;                                         // row0 = index of pair (a, a + 1)
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const int q = ln + 16 * m;
            while (a < P - 2 && q >= row0 + (P - 1 - a)) { row0 += P - 1 - a; ++a; }
            const int b = a + 1 + (q - row0);
            const bool ok = q < npairs;
            const T df = xd[ok ? b : 0] - xd[ok ? a : 0];
            vals[m] = ok ? to_bits(df * df) : ~U(0);                 // (never below a candidate: the sign bit of a candidate is clear)
        }
    }
    const int k = (npairs - 1) / 2;
    U result = 0;
    for (int bit = BitsOf<T>::NB - 2; bit >= 0; --bit) {
        const U cand = result | (U(1) << bit);
        int cnt = 0;
#pragma unroll
        for (int m = 0; m < M; ++m) cnt += vals[m] < cand ? 1 : 0;
        cnt = row16_sum(cnt);
        if (cnt <= k) result = cand;
    }
    int qmin = 0x7fffffff;                                           // first pair in row-major order that attains the median
#pragma unroll
    for (int m = M - 1; m >= 0; --m) if (vals[m] == result) qmin = ln + 16 * m;
    qmin = row16_min(qmin);
    const int d = d0 + dl;
    if (ln == 0 && d < D) {
        int a = 0, row0 = 0;
        while (a < P - 2 && qmin >= row0 + (P - 1 - a)) { row0 += P - 1 - a; ++a; }
        const int b = a + 1 + (qmin - row0);
        h[d] = from_bits(result) / log_p1;
        dh[d] = T(2) * (xd[b] - xd[a]) / log_p1;
        bidx[d] = b;
    }
}

// ---- stage 2: base_ij, k_ij, kb_ij; one workgroup per pair j <= i ---------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) imq_kmat_kernel(const T* __restrict__ X, const T* __restrict__ h, T h_fixed,
                                                      T alpha, T beta, T* __restrict__ Kmat, T* __restrict__ Kb,
                                                      int P, int D) {
    const int i = blockIdx.x / P, j = blockIdx.x - i * P;
    if (j > i) return;
    const T* xi = X + (long)i * D;
    const T* xj = X + (long)j * D;
    __shared__ T part[4];
    T acc = 0;
    for (int d = threadIdx.x; d < D; d += 256) {
        const T df = xi[d] - xj[d];
        acc += df * df / (h ? h[d] : h_fixed);
    }
    acc = subwave_sum<T>(acc, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        acc = (part[0] + part[1]) + (part[2] + part[3]);             // fixed order
        const T base = alpha + acc;
        const T kv = t_exp<T>(beta * t_log<T>(base));
        const T kb = beta * kv / base;
        Kmat[i * P + j] = kv; Kmat[j * P + i] = kv;
        Kb[i * P + j] = kb; Kb[j * P + i] = kb;
    }
}

// ---- stage 3: phi; TD dimensions per 64-thread block (TD = 64 while the 2 P TD coordinates fit in LDS, fewer for many
//      particles), particle columns staged in LDS; the 64 / TD thread groups of a block split the output particles j ----------
template <typename T>
__global__ void __launch_bounds__(64) imq_phi_kernel(const T* __restrict__ X, const T* __restrict__ score,
                                                     const T* __restrict__ Kmat, const T* __restrict__ Kb,
                                                     const T* __restrict__ h, const T* __restrict__ dh,
                                                     const int32_t* __restrict__ bidx, T h_fixed, int neg,
                                                     T* __restrict__ phi, int P, int D, int TD) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* xs = reinterpret_cast<T*>(smem_raw);          // [P][TD]
    T* ss = xs + (size_t)P * TD;                     // [P][TD]
    T* Sred = ss + (size_t)P * TD;                   // [64]
    const int t = threadIdx.x;
    const int tl = t & (TD - 1), grp = t / TD, ngrp = 64 / TD;
    const int d = blockIdx.x * TD + tl;
    const int dc = min(d, D - 1);
    for (int p = grp; p < P; p += ngrp) {
        xs[(size_t)p * TD + tl] = X[(long)p * D + dc];
        ss[(size_t)p * TD + tl] = score[(long)p * D + dc];
    }
    __syncthreads();
    const T hd = h ? h[dc] : h_fixed;
    const T two_over_h = T(2) / hd;
    const T sgn = (neg ? T(-1) : T(1)) / T(P);
    T S = 0;
    for (int j = grp; j < P; j += ngrp) {
        const T xj = xs[(size_t)j * TD + tl];
        const T* Kj = Kmat + (size_t)j * P;
        const T* Kbj = Kb + (size_t)j * P;
        T acc = 0, g = 0;
        for (int i = 0; i < P; ++i) {
            const T df = xj - xs[(size_t)i * TD + tl];
            const T kb = Kbj[i];
            acc = fma(Kj[i], ss[(size_t)i * TD + tl], acc);
            g = fma(kb, df, g);
            S = fma(kb * df, df, S);
        }
        if (d < D) phi[(long)j * D + d] = sgn * (acc - two_over_h * g);
    }
    Sred[t] = S;
    __syncthreads();                                 // (also: every phi row of this block's dimensions is written)
    if (h && grp == 0 && d < D) {
        T Sd = 0;
        for (int q = 0; q < ngrp; ++q) Sd += Sred[q * TD + tl];     // fixed order
        const long q = (long)bidx[d] * D + d;
        __threadfence_block();
        phi[q] += sgn * (Sd / (hd * hd)) * dh[d];
    }
}

template <typename T>
int imq_launch(const void* X, const void* score, double alpha, double beta, double bandwidth, int neg, void* phi,
               void* h_out, void* workspace, int P, int D, hipStream_t s) {
    T* Kmat = (T*)workspace;
    T* Kb = Kmat + (size_t)P * P;
    T* dh = Kb + P * P;
    int32_t* bidx = (int32_t*)(dh + D);
    T* harr = (T*)h_out;
    const bool median = !(bandwidth > 0.0);
    if (median) {
        const int Pp = P | 1;
        const size_t lds = (size_t)16 * Pp * sizeof(T);
        if (lds > 64u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(imq_bw_kernel<T>),
                                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PACOH_ELIMIT;
        if (P <= 32)
            hipLaunchKernelGGL(imq_bw_small_kernel<T>, dim3((D + 15) / 16), dim3(256), 0, s, (const T*)X, harr, dh, bidx,
                               (T)log((double)P + 1.0), P, D);
        else
            hipLaunchKernelGGL(imq_bw_kernel<T>, dim3((D + 15) / 16), dim3(256), lds, s, (const T*)X, harr, dh, bidx,
                               (T)log((double)P + 1.0), P, D);
    }
    hipLaunchKernelGGL(imq_kmat_kernel<T>, dim3(P * P), dim3(256), 0, s, (const T*)X, median ? (const T*)harr : (const T*)nullptr,
                       (T)bandwidth, (T)alpha, (T)beta, Kmat, Kb, P, D);
    int TD = 64;                                     // dimensions per block: the two [P][TD] column images within 96 KB of LDS
    while (TD > 1 && (size_t)2 * P * TD * sizeof(T) > 96u * 1024u) TD >>= 1;
    while (TD > 16 && (D + TD - 1) / TD < 256) TD >>= 1;            // (40 one-wave blocks for D = 2534 were pure latency: 26 -> 8 us)
    const size_t lds3 = (size_t)2 * P * TD * sizeof(T) + 64 * sizeof(T);
    if (lds3 > 64u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(imq_phi_kernel<T>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3) != hipSuccess) return PACOH_ELIMIT;
    hipLaunchKernelGGL(imq_phi_kernel<T>, dim3((D + TD - 1) / TD), dim3(64), lds3, s, (const T*)X,
                       (const T*)score, (const T*)Kmat, (const T*)Kb, median ? (const T*)harr : (const T*)nullptr,
                       (const T*)dh, (const int32_t*)bidx, (T)bandwidth, neg, (T*)phi, P, D, TD);
    return launch_status();
}

}  // namespace

extern "C" size_t pacoh_svgd_imq_workspace_bytes(int P, int D, int dtype) {
    if (P <= 0 || D <= 0) return 0;
    const size_t e = dtype == PACOH_F64 ? 8 : 4;
    return ((size_t)2 * P * P + D) * e + (size_t)D * sizeof(int32_t) + 16;
}

extern "C" int pacoh_svgd_phi_imq(const void* X, const void* score, double alpha, double beta, double bandwidth,
                                  int neg, void* phi, void* h_out, void* workspace, int P, int D, int dtype,
                                  void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!X || !score || !phi || !workspace || P <= 0 || D <= 0) return PACOH_EINVAL;
    if (!(alpha > 0.0) || !(beta < 0.0)) return PACOH_EINVAL;        // svgd.py:72-73
    const bool median = !(bandwidth > 0.0);
    if (median && (!h_out || P < 2)) return PACOH_EINVAL;
    if (P > PACOH_SVGD_MAX_PARTICLES) return PACOH_ELIMIT;
    if (dtype == PACOH_F32)
        return imq_launch<float>(X, score, alpha, beta, bandwidth, neg, phi, h_out, workspace, P, D, (hipStream_t)stream);
    return imq_launch<double>(X, score, alpha, beta, bandwidth, neg, phi, h_out, workspace, P, D, (hipStream_t)stream);
}
