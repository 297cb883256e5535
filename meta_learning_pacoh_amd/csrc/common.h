// Shared device/host helpers for the PACOH task-GP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <atomic>
#include "../../include/pacoh_gp.h"
#include "switches.h"

#define PACOH_WAVE 64

namespace pacoh {

template <typename T> struct VecOf;
template <> struct VecOf<float> { using type = float4; static constexpr int W = 4; };
template <> struct VecOf<double> { using type = double2; static constexpr int W = 2; };

template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float t_exp<float>(float x) { return expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T t_log(T x);
template <> __device__ __forceinline__ float t_log<float>(float x) { return logf(x); }
template <> __device__ __forceinline__ double t_log<double>(double x) { return log(x); }
template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T t_tanh(T x);
template <> __device__ __forceinline__ float t_tanh<float>(float x) { return tanhf(x); }
template <> __device__ __forceinline__ double t_tanh<double>(double x) { return tanh(x); }
template <typename T> __device__ __forceinline__ T t_log1p(T x);
template <> __device__ __forceinline__ float t_log1p<float>(float x) { return log1pf(x); }
template <> __device__ __forceinline__ double t_log1p<double>(double x) { return log1p(x); }

// exp(x) for the RBF kernel entries (x <= 0).  fp32: hardware exp2 on x*log2(e) with the rounding
// error of that product folded back in (rel. error ~1e-7, vs ~|x|*6e-8 for the bare __expf); fp64: own polynomial.
template <typename T> __device__ __forceinline__ T rbf_exp(T x);
template <> __device__ __forceinline__ double rbf_exp<double>(double x) {
    // exp(x) = 2^k exp(r), k = rint(x log2 e), r = x - k ln2 (two-term ln2), exp(r) by its degree-13 Taylor polynomial
    // (|r| <= 0.347: truncation 4e-18), scaled with v_ldexp_f64.  ~20 fp64 instructions, branch-free, against ~50 with
    // special-case branches in the libm exp: the fp64 Gram build and gradient contractions are bound by this function
    // (n = 512, d = 8 Gram: 2.0 -> 3.6 TB/s).  Max. error 1 ulp on [-745, 0]; below -745 the result flushes to 0.
    x = fmax(x, -745.0);
    const double kf = rint(x * 1.4426950408889634);
    double r = fma(-kf, 6.93147180369123816490e-01, x);
    r = fma(-kf, 1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;               // 1/13!
    p = fma(p, r, 2.08767569878681e-09);             // 1/12!
    p = fma(p, r, 2.505210838544172e-08);            // 1/11!
    p = fma(p, r, 2.755731922398589e-07);            // 1/10!
    p = fma(p, r, 2.7557319223985893e-06);           // 1/9!
    p = fma(p, r, 2.48015873015873e-05);             // 1/8!
    p = fma(p, r, 1.984126984126984e-04);            // 1/7!
    p = fma(p, r, 1.388888888888889e-03);            // 1/6!
    p = fma(p, r, 8.333333333333333e-03);            // 1/5!
    p = fma(p, r, 4.1666666666666664e-02);           // 1/4!
    p = fma(p, r, 1.6666666666666666e-01);           // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)kf);
}
template <> __device__ __forceinline__ float rbf_exp<float>(float x) {
    const float L2E = 1.4426950408889634f, L2E_LO = 1.9259629911266175e-8f;
    const float hi = x * L2E;
    const float lo = fmaf(x, L2E, -hi) + x * L2E_LO;
    const float e = __builtin_amdgcn_exp2f(hi);
    return fmaf(e, lo * 0.6931471805599453f, e);
}

// ---- kernel families ---------------------------------------------------------------------------------------------------------
// The `f` argument of the GP entry points carries the kernel family in the bits above the feature count (PACOH_KERNEL_* in
// pacoh_gp.h): f_arg = f | (kernel << PACOH_KERNEL_SHIFT).  In scaled coordinates u = z / lengthscale, with s2 = |u_i - u_j|^2:
//   RBF     k / os = exp(-s2 / 2)          -(d k / d u_ic) / ((u_i - u_j)_c os) = the same value
//   COSINE  k / os = cos(pi sqrt(s2))      ... = pi sin(pi s) / s   (-> pi^2 for s -> 0)        [gpytorch.kernels.CosineKernel]
// kern_eval returns both: every gradient of the LML is a contraction with one of the two (d/d outputscale with the first, d/d z
// and d/d lengthscale with the second), so the kernels that evaluate the RBF family evaluate any family given this pair.
__host__ __device__ inline int kernel_of(int f_arg) { return f_arg >> PACOH_KERNEL_SHIFT; }
__host__ __device__ inline int features_of(int f_arg) { return f_arg & ((1 << PACOH_KERNEL_SHIFT) - 1); }
template <typename T> __device__ __forceinline__ void sincospi_t(T x, T* s, T* c);
template <> __device__ __forceinline__ void sincospi_t<float>(float x, float* s, float* c) { sincospif(x, s, c); }
template <> __device__ __forceinline__ void sincospi_t<double>(double x, double* s, double* c) { sincospi(x, s, c); }
template <typename T> __device__ __forceinline__ void kern_eval(int kind, T s2, T& kv, T& kd) {
    if (kind == PACOH_KERNEL_RBF) { kv = kd = rbf_exp<T>(T(-0.5) * s2); return; }
    const T s = t_sqrt<T>(s2);
    T sn, cs;
    sincospi_t<T>(s, &sn, &cs);
    kv = cs;
    kd = s > T(1e-12) ? T(3.141592653589793) * sn / s : T(9.869604401089358);
}
template <typename T> __device__ __forceinline__ T kern_val(int kind, T s2) { T kv, kd; kern_eval<T>(kind, s2, kv, kd); return kv; }

// tanh for the MLP activations.  fp32: 1 - 2 / (1 + exp(2x)) on the hardware exp2 / rcp: five instructions, no branches,
// correct limits for both signs (exp2 overflow -> rcp(inf) = 0 -> +1; exp2 underflow -> 1 - 2 = -1).  Its ABSOLUTE error is
// ~1e-7 everywhere (one rounding of a value near 1); the relative error grows like 1e-7 / |x| for tiny |x|, which is
// irrelevant for activations (they enter the next layer additively) -- the earlier variant that switched to an odd
// polynomial below 0.25 for full relative accuracy cost 17 instructions + two exec-mask branches per activation, and tanh
// is what bounds the MLP kernels (52 % of the backward kernel is the forward recompute).  fp64: libm.
template <typename T> __device__ __forceinline__ T act_tanh(T x);
template <> __device__ __forceinline__ double act_tanh<double>(double x) { return tanh(x); }
template <> __device__ __forceinline__ float act_tanh<float>(float x) {
    const float e = __builtin_amdgcn_exp2f(2.885390081777927f * x);            // exp(2x)
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + e), 1.0f);
}

// Leading dimension (in elements) of an LDS matrix whose rows are read both "own row per lane"
// (ds_read_b128, lane i at row i) and "one row broadcast to all lanes".  A multiple of the vector
// width W whose quotient is odd spreads the 16 lanes of a ds_read_b128 lane-group over all 64 banks
// (MI355X_MICROARCH.md, LDS): stride = 4*odd dwords -> 16 distinct 16-byte slots.
template <typename T> __host__ __device__ inline int lds_ld(int n) {
    const int W = VecOf<T>::W;
    int q = (n + W - 1) / W;
    if ((q & 1) == 0) q += 1;
    return q * W;
}

// dot product of two LDS rows over the element range [lo, hi) widened to vector boundaries; the
// caller guarantees that every widened element is zero in at least one of the two rows.
template <typename T>
__device__ __forceinline__ T dot_rows(const T* __restrict__ a, const T* __restrict__ b, int lo, int hi) {
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    const V* av = reinterpret_cast<const V*>(a);
    const V* bv = reinterpret_cast<const V*>(b);
    int v0 = lo / W, v1 = (hi + W - 1) / W;
    T s0 = 0, s1 = 0;
    if constexpr (W == 4) {
        T s2 = 0, s3 = 0;
#pragma unroll 4
        for (int v = v0; v < v1; ++v) {
            V x = av[v], y = bv[v];
            s0 = fma(x.x, y.x, s0); s1 = fma(x.y, y.y, s1);
            s2 = fma(x.z, y.z, s2); s3 = fma(x.w, y.w, s3);
        }
        return (s0 + s1) + (s2 + s3);
    } else {
#pragma unroll 4
        for (int v = v0; v < v1; ++v) {
            V x = av[v], y = bv[v];
            s0 = fma(x.x, y.x, s0); s1 = fma(x.y, y.y, s1);
        }
        return s0 + s1;
    }
}

template <typename T> __device__ __forceinline__ T shfl_xor_t(T v, int mask);
template <> __device__ __forceinline__ float shfl_xor_t<float>(float v, int mask) { return __shfl_xor(v, mask, 64); }
template <> __device__ __forceinline__ double shfl_xor_t<double>(double v, int mask) { return __shfl_xor(v, mask, 64); }

// sum over the `gs` consecutive lanes (power of two <= 64) that contain this lane
template <typename T> __device__ __forceinline__ T subwave_sum(T v, int gs) {
    for (int m = 1; m < gs; m <<= 1) v += shfl_xor_t<T>(v, m);
    return v;
}

// check_dtype() is the first call of every launcher: it also clears HIP's sticky per-thread "last error", which
// the host application may have set with benign codes (PyTorch polls events: hipErrorNotReady) -- otherwise
// launch_status() would blame our launch for somebody else's status.
inline int check_dtype(int dtype) {
    (void)hipGetLastError();
    return (dtype == PACOH_F32 || dtype == PACOH_F64) ? 0 : PACOH_EDTYPE;
}
inline int launch_status() {
    const hipError_t e = hipGetLastError();
    return (e == hipSuccess || e == hipErrorNotReady) ? PACOH_OK : PACOH_ELAUNCH;
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: opt in once per (kernel, device) -- `done` is the
// calling launcher's own mask of devices (a process-wide bool would leave a second device without the opt-in: ADVICE r5).  Two threads
// racing here both set the attribute, which is harmless.
inline int lds_opt_in(const void* kernel, int bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return PACOH_ELAUNCH; }
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return PACOH_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { (void)hipGetLastError(); return PACOH_ELIMIT; }
    done.fetch_or(bit, std::memory_order_release);
    return PACOH_OK;
}

// softplus with torch's threshold (F.softplus: x for x > 20)
template <typename T> __device__ __forceinline__ T softplus_t(T x) { return x > T(20) ? x : t_log1p<T>(t_exp<T>(x)); }

}  // namespace pacoh
