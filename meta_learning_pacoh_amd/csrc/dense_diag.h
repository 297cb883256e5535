// Diagonal-block primitives of the dense (large-n) Cholesky kernels (dense_mfma.hip, dense_ll.hip): MFMA wrappers for both
// element types, the fp64 pivot (v_rsq_f64 + Newton), v_readlane broadcasts fused with their fma, and the register-resident
// right-looking elimination / in-place triangular inverse of a 32 x 32 block, row per lane.  The *Range forms run a
// compile-time sub-range of the 32 steps: dense_ll.hip's factorisation wave interleaves them with the workgroup's barriers.
#pragma once
#include "common.h"

namespace pacoh {

using f32x4_t = __attribute__((ext_vector_type(4))) float;
using f64x4_t = __attribute__((ext_vector_type(4))) double;

template <typename T> struct Mf;
template <> struct Mf<float> {
    using acc = f32x4_t;
    static __device__ __forceinline__ acc mma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int g, int q) { return 4 * g + q; }          // C/D row of register q
    static __host__ __device__ constexpr int g_of(int cc) { return cc >> 2; }              // row(g_of(cc), q_of(cc)) == cc
    static __host__ __device__ constexpr int q_of(int cc) { return cc & 3; }
};
template <> struct Mf<double> {
    using acc = f64x4_t;
    static __device__ __forceinline__ acc mma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int g, int q) { return g + 4 * q; }          // f64 uses a different C/D map
    static __host__ __device__ constexpr int g_of(int cc) { return cc & 3; }
    static __host__ __device__ constexpr int q_of(int cc) { return cc >> 2; }
};

constexpr int DNB = 32;            // panel width
constexpr int DLP = 36;            // leading dimension of the LDS panel / inverse images

// d = sqrt(piv), inv = 1 / d.  fp64: the libm sqrt followed by a division is ~60 dependent double-precision instructions on the
// critical path of every elimination step; v_rsq_f64 (2^-26) with two Newton steps and one correction each for d and inv is 14.
template <typename T> __device__ __forceinline__ void pivot_sqrt_inv(T piv, T& d, T& inv);
template <> __device__ __forceinline__ void pivot_sqrt_inv<float>(float piv, float& d, float& inv) { d = sqrtf(piv); inv = 1.0f / d; }
template <> __device__ __forceinline__ void pivot_sqrt_inv<double>(double x, double& d, double& inv) {
    double y = __builtin_amdgcn_rsq(x);
    double e = fma(-x * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    e = fma(-x * y, y, 1.0);
    y = fma(0.5 * y, e, y);
    d = x * y;
    d = fma(0.5 * y, fma(-d, d, x), d);
    inv = fma(y, fma(-d, y, 1.0), y);
}

// value of v in lane `src` (compile-time constant), as a wave-uniform value
template <typename T> __device__ __forceinline__ T bcast_lane(T v, int src);
template <> __device__ __forceinline__ float bcast_lane<float>(float v, int src) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
template <> __device__ __forceinline__ double bcast_lane<double>(double v, int src) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// a -= lj * (lj of lane SRC): the broadcast (v_readlane into a scalar register pair) and the fma that consumes it as ONE unit.
// (s_nop 1: gfx940+ needs two wait states between a VALU write of a scalar register and a VALU read of it; with one the fma
// now and then sees the previous broadcast.)  Written as separate operations the compiler issues the thirty-odd broadcasts of an elimination step first and the fmas
// after them, runs out of scalar registers and spills each value through v_writelane / v_readlane (550 spills per block).
template <int SRC> __device__ __forceinline__ void bcast_fnma(float& a, float lj) {
    asm volatile("v_readlane_b32 s90, %1, %2\n\ts_nop 1\n\tv_fma_f32 %0, -%1, s90, %0" : "+v"(a) : "v"(lj), "n"(SRC) : "s90");
}
template <int SRC> __device__ __forceinline__ void bcast_fnma(double& a, double lj) {
    const long long b = __double_as_longlong(lj);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    asm volatile("v_readlane_b32 s90, %1, %3\n\tv_readlane_b32 s91, %2, %3\n\ts_nop 1\n\tv_fma_f64 %0, -%4, s[90:91], %0"
                 : "+v"(a) : "v"(lo), "v"(hi), "n"(SRC), "v"(lj) : "s90", "s91");
}
// acc += x * (v of lane SRC)
template <int SRC> __device__ __forceinline__ void bcast_fma(float& acc, float x, float v) {
    asm volatile("v_readlane_b32 s90, %2, %3\n\ts_nop 1\n\tv_fma_f32 %0, %1, s90, %0" : "+v"(acc) : "v"(x), "v"(v), "n"(SRC) : "s90");
}
template <int SRC> __device__ __forceinline__ void bcast_fma(double& acc, double x, double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    asm volatile("v_readlane_b32 s90, %2, %4\n\tv_readlane_b32 s91, %3, %4\n\ts_nop 1\n\tv_fma_f64 %0, %1, s[90:91], %0"
                 : "+v"(acc) : "v"(x), "v"(lo), "v"(hi), "n"(SRC) : "s90", "s91");
}
// two of them per block: the broadcasts of the second cover the wait states of the first
template <int S0, int S1> __device__ __forceinline__ void bcast_fnma2(float& a0, float& a1, float lj) {
    asm volatile("v_readlane_b32 s90, %2, %3\n\tv_readlane_b32 s91, %2, %4\n\ts_nop 0\n\tv_fma_f32 %0, -%2, s90, %0\n\tv_fma_f32 %1, -%2, s91, %1"
                 : "+v"(a0), "+v"(a1) : "v"(lj), "n"(S0), "n"(S1) : "s90", "s91");
}
template <int S0, int S1> __device__ __forceinline__ void bcast_fnma2(double& a0, double& a1, double lj) {
    const long long b = __double_as_longlong(lj);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    asm volatile("v_readlane_b32 s90, %2, %4\n\tv_readlane_b32 s91, %3, %4\n\tv_readlane_b32 s92, %2, %5\n\tv_readlane_b32 s93, %3, %5\n\t"
                 "s_nop 0\n\tv_fma_f64 %0, -%6, s[90:91], %0\n\tv_fma_f64 %1, -%6, s[92:93], %1"
                 : "+v"(a0), "+v"(a1) : "v"(lo), "v"(hi), "n"(S0), "n"(S1), "v"(lj) : "s90", "s91", "s92", "s93");
}
template <int S0, int S1> __device__ __forceinline__ void bcast_fma2(float& acc0, float& acc1, float x0, float x1, float v) {
    asm volatile("v_readlane_b32 s90, %4, %5\n\tv_readlane_b32 s91, %4, %6\n\ts_nop 0\n\tv_fma_f32 %0, %2, s90, %0\n\tv_fma_f32 %1, %3, s91, %1"
                 : "+v"(acc0), "+v"(acc1) : "v"(x0), "v"(x1), "v"(v), "n"(S0), "n"(S1) : "s90", "s91");
}
template <int S0, int S1> __device__ __forceinline__ void bcast_fma2(double& acc0, double& acc1, double x0, double x1, double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    asm volatile("v_readlane_b32 s90, %4, %6\n\tv_readlane_b32 s91, %5, %6\n\tv_readlane_b32 s92, %4, %7\n\tv_readlane_b32 s93, %5, %7\n\t"
                 "s_nop 0\n\tv_fma_f64 %0, %2, s[90:91], %0\n\tv_fma_f64 %1, %3, s[92:93], %1"
                 : "+v"(acc0), "+v"(acc1) : "v"(x0), "v"(x1), "v"(lo), "v"(hi), "n"(S0), "n"(S1) : "s90", "s91", "s92", "s93");
}
// Row rr of X = L^-1, in place over row rr of L (X L = I, columns from the right): X[rr][C] L[C][C] = delta - sum_{k>C} X[rr][k] L[k][C].
// Column C of L is read (broadcast from the lanes k > C) for the last time in step C, which is also where X[rr][C] is born:
// the entry takes its register.
template <typename T, int C, int K> struct InvRow {
    static __device__ __forceinline__ void run(const T (&a)[DNB], T& acc0, T& acc1) {
        if constexpr (K + 1 < DNB) { bcast_fma2<K, K + 1>(acc0, acc1, a[K], a[K + 1], a[C]); InvRow<T, C, K + 2>::run(a, acc0, acc1); }
        else { bcast_fma<K>(acc0, a[K], a[C]); }
    }
};
template <typename T, int C> struct InvRow<T, C, DNB> { static __device__ __forceinline__ void run(const T (&)[DNB], T&, T&) {} };
template <typename T, int C> struct InvSteps {
    static __device__ __forceinline__ void run(T (&a)[DNB], const T* __restrict__ invd, int rr) {
        T acc0 = 0, acc1 = 0;
        InvRow<T, C, C + 1>::run(a, acc0, acc1);
        T xc = (((rr == C) ? T(1) : T(0)) - (acc0 + acc1)) * invd[C];
        asm volatile("s_nop 1" : "+v"(xc));          // (the next step's first broadcast reads registers written just now, see above)
        a[C] = xc;
        InvSteps<T, C - 1>::run(a, invd, rr);
    }
};
template <typename T> struct InvSteps<T, -1> { static __device__ __forceinline__ void run(T (&)[DNB], const T*, int) {} };

template <typename T, int J, int C> struct ElimRow {
    static __device__ __forceinline__ void run(T (&a)[DNB], T lj) {
        if constexpr (C + 1 < DNB) { bcast_fnma2<C, C + 1>(a[C], a[C + 1], lj); ElimRow<T, J, C + 2>::run(a, lj); }
        else { bcast_fnma<C>(a[C], lj); }
    }
};
template <typename T, int J> struct ElimRow<T, J, DNB> { static __device__ __forceinline__ void run(T (&)[DNB], T) {} };
template <typename T, int J> struct ElimSteps {
    static __device__ __forceinline__ void run(T (&a)[DNB], T* __restrict__ invd, bool& bad, int lane) {
        T piv = bcast_lane<T>(a[J], J);
        if (!(piv > T(0))) { bad = true; piv = 1; }
        T d, inv;
        pivot_sqrt_inv<T>(piv, d, inv);
        if (lane == 0) invd[J] = inv;
        T lj = a[J] * inv;                           // L[rr][J] (meaningful for rr > J; lane J: piv / d = d up to one rounding)
        asm volatile("s_nop 1" : "+v"(lj));          // VALU write -> v_readlane of the same register needs a wait state the compiler
                                                     // cannot place: the first broadcast below sits inside an asm block
        ElimRow<T, J, J + 1>::run(a, lj);            // a[rr][c] -= L[rr][J] L[c][J], c > J
        a[J] = lj;
        ElimSteps<T, J + 1>::run(a, invd, bad, lane);
    }
};
template <typename T> struct ElimSteps<T, DNB> { static __device__ __forceinline__ void run(T (&)[DNB], T*, bool&, int) {} };

// steps [J, JEND) of the elimination / steps C, C-1, ..., CEND of the inverse (same arithmetic as the full recursions above)
template <typename T, int J, int JEND> struct ElimRange {
    static __device__ __forceinline__ void run(T (&a)[DNB], T* __restrict__ invd, bool& bad, int lane) {
        if constexpr (J < JEND) {
            T piv = bcast_lane<T>(a[J], J);
            if (!(piv > T(0))) { bad = true; piv = 1; }
            T d, inv;
            pivot_sqrt_inv<T>(piv, d, inv);
            if (lane == 0) invd[J] = inv;
            T lj = a[J] * inv;
            asm volatile("s_nop 1" : "+v"(lj));
            ElimRow<T, J, J + 1>::run(a, lj);
            a[J] = lj;
            ElimRange<T, J + 1, JEND>::run(a, invd, bad, lane);
        }
    }
};
template <typename T, int C, int CEND> struct InvRange {
    static __device__ __forceinline__ void run(T (&a)[DNB], const T* __restrict__ invd, int rr) {
        if constexpr (C >= CEND) {
            T acc0 = 0, acc1 = 0;
            InvRow<T, C, C + 1>::run(a, acc0, acc1);
            T xc = (((rr == C) ? T(1) : T(0)) - (acc0 + acc1)) * invd[C];
            asm volatile("s_nop 1" : "+v"(xc));
            a[C] = xc;
            InvRange<T, C - 1, CEND>::run(a, invd, rr);
        }
    }
};

}  // namespace pacoh
