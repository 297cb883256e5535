// Marginal cdf, quantiles and calibration error of the (mixture) posterior predictive -- SURVEY 8(f) ranks 1 and 2.
//   cdf        EqualWeightedMixtureDist.cdf (meta_learn/models.py:124-131) over AffineTransformedDistribution components (models.py:15-43)
//   quantiles  EqualWeightedMixtureDist.icdf (models.py:136-140) = the interval-halving search of meta_learn/util.py:9-42,
//              here ONE launch: the whole search (about 47 rounds of a P x m cdf) runs inside one workgroup
//   calibration error  _calib_error (meta_learn/abstract.py:260-272)
// Components are held in normalised space (mu[P,m], var[P,m]); y = y_mean + y_std * y_n is applied on the fly.
#include <hip/hip_runtime.h>

#include "common.h"
#include "pacoh_gp.h"

namespace pacoh {

template <typename T> __device__ __forceinline__ T t_erf(T x);
template <> __device__ __forceinline__ float t_erf<float>(float x) { return erff(x); }
template <> __device__ __forceinline__ double t_erf<double>(double x) { return erf(x); }
template <typename T> __device__ __forceinline__ T t_erfinv(T x);
template <> __device__ __forceinline__ float t_erfinv<float>(float x) { return erfinvf(x); }
template <> __device__ __forceinline__ double t_erfinv<double>(double x) { return erfinv(x); }

// sum over the components p = p0, p0 + stride, ... of Phi((v - loc_p) / scale_p) for test point j
template <typename T>
__device__ __forceinline__ T cdf_partial(const T* __restrict__ mu, const T* __restrict__ var, int P, int m, int j, T v, T ym, T ys,
                                         int p0, int stride) {
    T acc = T(0);
    for (int p = p0; p < P; p += stride) {
        const T loc = mu[(long)p * m + j] * ys + ym;
        const T scale = t_sqrt(var[(long)p * m + j]) * ys;
        acc += T(0.5) * (T(1) + t_erf((v - loc) / scale * T(0.70710678118654752440)));
    }
    return acc;
}

template <typename T>
__global__ void mixture_cdf_kernel(const T* __restrict__ mu, const T* __restrict__ var, const T* __restrict__ value, T* __restrict__ out,
                                   T ym, T ys, int P, int m) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const long t = blockIdx.y;                                  // task of the batch: mu[t,P,m], value[t,m]
    out[t * m + j] = cdf_partial(mu + t * P * m, var + t * P * m, P, m, j, value[t * m + j], ym, ys, 0, 1) / T(P);
}

template <typename T>
__global__ void gaussian_icdf_kernel(const T* __restrict__ mu, const T* __restrict__ var, const T* __restrict__ q, T* __restrict__ out,
                                     T ym, T ys, int m) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    out[j] = ym + ys * (mu[j] + t_sqrt(var[j]) * T(1.41421356237309504880) * t_erfinv(T(2) * q[j] - T(1)));
}

constexpr int ICDF_THREADS = 1024;

// One workgroup owns all m quantile searches so that the stopping rule is the reference's: every element starts from [lo, hi], all
// are halved together until the LARGEST half-width is <= eps, the result is the last midpoint; more than max_iter rounds -> NaN for
// every element.  tpe (power of two <= 64) lanes share one element's sum over the P components.  Bounds live in LDS (2 m values).
// A round in which no bound moved can never converge (floating-point fixed point above eps): NaN at once instead of after max_iter.
template <typename T>
__global__ void __launch_bounds__(ICDF_THREADS) mixture_icdf_kernel(const T* __restrict__ mu, const T* __restrict__ var,
                                                                    const T* __restrict__ q, T* __restrict__ out, T ym, T ys, T lo, T hi,
                                                                    T eps, int max_iter, int P, int m, int tpe) {
    extern __shared__ unsigned char lds_raw[];
    T* left = reinterpret_cast<T*>(lds_raw);
    T* right = left + m;
    __shared__ T red[ICDF_THREADS / PACOH_WAVE];
    __shared__ int moved[ICDF_THREADS / PACOH_WAVE];
    const int tid = threadIdx.x, lane = tid & (PACOH_WAVE - 1), wave = tid / PACOH_WAVE;
    const int group = tid / tpe, sub = tid & (tpe - 1), groups = ICDF_THREADS / tpe;
    for (int e = tid; e < m; e += ICDF_THREADS) { left[e] = lo; right[e] = hi; }
    __syncthreads();
    bool converged = false;
    for (int it = 0; it < max_iter && !converged; ++it) {
        T width = T(0);
        int any = 0;
        for (int e0 = 0; e0 < m; e0 += groups) {               // uniform trip count: the shuffles below need whole waves
            const int e = e0 + group;
            const bool live = e < m;
            T l = T(0), r = T(0), mid = T(0), part = T(0);
            if (live) {
                l = left[e]; r = right[e];
                mid = (r + l) / T(2);
                part = cdf_partial(mu, var, P, m, e, mid, ym, ys, sub, tpe);
            }
            for (int s = tpe >> 1; s > 0; s >>= 1) part += shfl_xor_t(part, s);
            if (live) {
                const bool below = part / T(P) - q[e] < T(0);
                const T nl = below ? mid : l, nr = below ? r : mid;
                any |= (nl != l) | (nr != r);
                if (sub == 0) { left[e] = nl; right[e] = nr; out[e] = mid; }
                const T w = nr - nl;
                width = fmax(width, (w < T(0) ? -w : w) / T(2));
            }
        }
        for (int s = PACOH_WAVE >> 1; s > 0; s >>= 1) {
            width = fmax(width, shfl_xor_t(width, s));
            any |= __shfl_xor(any, s, PACOH_WAVE);
        }
        if (lane == 0) { red[wave] = width; moved[wave] = any; }
        __syncthreads();
        T wmax = T(0);
        int many = 0;
        for (int w = 0; w < ICDF_THREADS / PACOH_WAVE; ++w) { wmax = fmax(wmax, red[w]); many |= moved[w]; }
        __syncthreads();
        converged = !(wmax > eps);
        if (!converged && !many) break;
    }
    if (!converged) {
        const T nan = T(NAN);
        for (int e = tid; e < m; e += ICDF_THREADS) out[e] = nan;
    }
}

// per task t of the batch: out[t] = sqrt(mean_k (#{j: cdf[j] <= level_k} / m - level_k)^2), level = linspace(0.05, 0.95, 20) in fp32 as torch builds it
template <typename T>
__global__ void __launch_bounds__(256) calib_error_kernel(const T* __restrict__ cdf, T* __restrict__ out, int m) {
    constexpr int NL = 20;
    __shared__ int count[NL];
    const int tid = threadIdx.x;
    cdf += (long)blockIdx.x * m;                                // one workgroup per task of the batch
    out += blockIdx.x;
    if (tid < NL) count[tid] = 0;
    __syncthreads();
    const float step = (0.95f - 0.05f) / float(NL - 1);
    float level[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) level[k] = k < NL / 2 ? 0.05f + step * float(k) : 0.95f - step * float(NL - 1 - k);
    int mine[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) mine[k] = 0;
    for (int j = tid; j < m; j += 256) {
        const T c = cdf[j];
#pragma unroll
        for (int k = 0; k < NL; ++k) mine[k] += c <= T(level[k]) ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        int v = mine[k];
        for (int s = PACOH_WAVE >> 1; s > 0; s >>= 1) v += __shfl_xor(v, s, PACOH_WAVE);
        if ((tid & (PACOH_WAVE - 1)) == 0) atomicAdd(&count[k], v);
    }
    __syncthreads();
    if (tid == 0) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const float d = float(count[k]) / float(m) - level[k];
            acc += d * d;
        }
        out[0] = T(sqrtf(acc / float(NL)));
    }
}

}  // namespace pacoh

using namespace pacoh;

extern "C" int pacoh_mixture_cdf(const void* mu, const void* var, const void* value, void* cdf, double y_mean, double y_std, int T, int P,
                                 int m, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!mu || !var || !value || !cdf || T <= 0 || T > 65535 || P <= 0 || m <= 0 || !(y_std > 0)) return PACOH_EINVAL;
    const dim3 blocks((unsigned)((m + 63) / 64), (unsigned)T);
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(mixture_cdf_kernel<float>, blocks, dim3(64), 0, (hipStream_t)stream, (const float*)mu, (const float*)var,
                           (const float*)value, (float*)cdf, (float)y_mean, (float)y_std, P, m);
    else
        hipLaunchKernelGGL(mixture_cdf_kernel<double>, blocks, dim3(64), 0, (hipStream_t)stream, (const double*)mu, (const double*)var,
                           (const double*)value, (double*)cdf, y_mean, y_std, P, m);
    return launch_status();
}

extern "C" int pacoh_mixture_icdf(const void* mu, const void* var, const void* quantile, void* out, double y_mean, double y_std, double lo,
                                  double hi, double eps, int max_iter, int closed_form, int P, int m, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!mu || !var || !quantile || !out || P <= 0 || m <= 0 || !(y_std > 0) || !(lo <= hi) || max_iter < 0) return PACOH_EINVAL;
    if (closed_form) {
        if (P != 1) return PACOH_EINVAL;
        const unsigned blocks = (unsigned)((m + 63) / 64);
        if (dtype == PACOH_F32)
            hipLaunchKernelGGL(gaussian_icdf_kernel<float>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, (const float*)mu,
                               (const float*)var, (const float*)quantile, (float*)out, (float)y_mean, (float)y_std, m);
        else
            hipLaunchKernelGGL(gaussian_icdf_kernel<double>, dim3(blocks), dim3(64), 0, (hipStream_t)stream, (const double*)mu,
                               (const double*)var, (const double*)quantile, (double*)out, y_mean, y_std, m);
        return launch_status();
    }
    if (m > PACOH_MAX_QUANTILES) return PACOH_ELIMIT;
    int tpe = 1;
    while (tpe < PACOH_WAVE && (long)m * (tpe * 2) <= ICDF_THREADS && tpe * 2 <= P) tpe *= 2;
    const size_t esz = dtype == PACOH_F32 ? sizeof(float) : sizeof(double);
    const size_t lds = 2 * (size_t)m * esz;
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(mixture_icdf_kernel<float>, dim3(1), dim3(ICDF_THREADS), lds, (hipStream_t)stream, (const float*)mu,
                           (const float*)var, (const float*)quantile, (float*)out, (float)y_mean, (float)y_std, (float)lo, (float)hi,
                           (float)eps, max_iter, P, m, tpe);
    else
        hipLaunchKernelGGL(mixture_icdf_kernel<double>, dim3(1), dim3(ICDF_THREADS), lds, (hipStream_t)stream, (const double*)mu,
                           (const double*)var, (const double*)quantile, (double*)out, y_mean, y_std, lo, hi, eps, max_iter, P, m, tpe);
    return launch_status();
}

extern "C" int pacoh_calib_error(const void* cdf, void* out, int T, int m, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!cdf || !out || T <= 0 || m <= 0) return PACOH_EINVAL;
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(calib_error_kernel<float>, dim3(T), dim3(256), 0, (hipStream_t)stream, (const float*)cdf, (float*)out, m);
    else
        hipLaunchKernelGGL(calib_error_kernel<double>, dim3(T), dim3(256), 0, (hipStream_t)stream, (const double*)cdf, (double*)out, m);
    return launch_status();
}
