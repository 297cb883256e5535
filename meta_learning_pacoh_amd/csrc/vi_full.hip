// PACOH-VI with a full-covariance Gaussian hyper-posterior: replaces
// MultivariateNormal(loc, scale_tril=tril(tril_cov)).rsample / .log_prob and the autograd backward of the ELBO
// (meta_learn/random_gp.py:249-251, GPR_meta_vi.py:220-224).
//   posterior[D+1, D]: row 0 = loc, rows 1..D = tril_cov (row-major; entries above the diagonal are ignored)
//   theta[s,i] = loc[i] + sum_{j<=i} L[i,j] eps[s,j]
//   log_q[s]   = -0.5 |eps_s|^2 - sum_d log L[d,d] - D/2 log(2 pi)
//   grad[0,i]    = -mean_s score[s,i]
//   grad[1+i,j]  = j <= i ? -mean_s score[s,i] eps[s,j] - [i==j] prior_factor / L[i,i] : 0
// Both kernels stream the D x D factor exactly once (25.7 MB at D = 2534, fp32): they are HBM-bound, one wave per row
// with the row read / written as contiguous 256-byte segments; eps[S,D] (100 KB) stays L2-resident.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/pacoh_gp.h"
#include "common.h"

using namespace pacoh;

namespace {

constexpr int SC = 8;        // samples accumulated per pass over a row

template <typename T>
__global__ void __launch_bounds__(256) vi_full_sample_kernel(const T* __restrict__ post, const T* __restrict__ eps,
                                                             T* __restrict__ theta, int S, int D) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= D) return;
    const T* Lrow = post + (long)(1 + i) * D;
    const T loc = post[i];
    for (int s0 = 0; s0 < S; s0 += SC) {
        T acc[SC];
#pragma unroll
        for (int u = 0; u < SC; ++u) acc[u] = 0;
        for (int j = lane; j <= i; j += 64) {
            const T l = Lrow[j];
#pragma unroll
            for (int u = 0; u < SC; ++u)
                if (s0 + u < S) acc[u] = fma(l, eps[(long)(s0 + u) * D + j], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < SC; ++u) {
            const T r = subwave_sum<T>(acc[u], 64);
            if (lane == 0 && s0 + u < S) theta[(long)(s0 + u) * D + i] = loc + r;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) vi_full_logq_kernel(const T* __restrict__ post, const T* __restrict__ eps,
                                                           T* __restrict__ log_q, int D) {
    __shared__ T red[4];
    const int s_ = blockIdx.x;
    const T HALF_LOG2PI = T(0.9189385332046727);
    T acc = 0;
    for (int d = threadIdx.x; d < D; d += 256) {
        const T e = eps[(long)s_ * D + d];
        acc += T(-0.5) * e * e - t_log<T>(post[(long)(1 + d) * D + d]) - HALF_LOG2PI;
    }
    acc = subwave_sum<T>(acc, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) log_q[s_] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one 256-thread workgroup per row of grad[D+1, D]
template <typename T>
__global__ void __launch_bounds__(256) vi_full_grad_kernel(const T* __restrict__ post, const T* __restrict__ eps,
                                                           const T* __restrict__ score, T prior_factor,
                                                           T* __restrict__ grad, int S, int D) {
    const int r = blockIdx.x;
    const T inv_s = T(1) / T(S);
    T* out = grad + (long)r * D;
    if (r == 0) {
        for (int d = threadIdx.x; d < D; d += 256) {
            T g = 0;
            for (int s_ = 0; s_ < S; ++s_) g += score[(long)s_ * D + d];
            out[d] = -g * inv_s;
        }
        return;
    }
    const int i = r - 1;
    for (int j = threadIdx.x; j < D; j += 256) {
        T g = 0;
        if (j <= i) {
            for (int s_ = 0; s_ < S; ++s_) g = fma(score[(long)s_ * D + i], eps[(long)s_ * D + j], g);   // score: scalar loads
            g = -g * inv_s;
            if (j == i) g -= prior_factor / post[(long)r * D + i];
        }
        out[j] = g;
    }
}

}  // namespace

extern "C" int pacoh_vi_sample_full(const void* posterior, const void* eps, void* theta, void* log_q, int S, int D,
                                    int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!posterior || !eps || !theta || S <= 0 || D <= 0) return PACOH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((D + 3) / 4);
    if (dtype == PACOH_F32) {
        hipLaunchKernelGGL(vi_full_sample_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)posterior, (const float*)eps,
                           (float*)theta, S, D);
        if (log_q) hipLaunchKernelGGL(vi_full_logq_kernel<float>, dim3(S), dim3(256), 0, s, (const float*)posterior,
                                      (const float*)eps, (float*)log_q, D);
    } else {
        hipLaunchKernelGGL(vi_full_sample_kernel<double>, dim3(blocks), dim3(256), 0, s, (const double*)posterior, (const double*)eps,
                           (double*)theta, S, D);
        if (log_q) hipLaunchKernelGGL(vi_full_logq_kernel<double>, dim3(S), dim3(256), 0, s, (const double*)posterior,
                                      (const double*)eps, (double*)log_q, D);
    }
    return launch_status();
}

extern "C" int pacoh_vi_grad_full(const void* posterior, const void* eps, const void* score, double prior_factor, void* grad,
                                  int S, int D, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!posterior || !eps || !score || !grad || S <= 0 || D <= 0) return PACOH_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(vi_full_grad_kernel<float>, dim3(D + 1), dim3(256), 0, s, (const float*)posterior, (const float*)eps,
                           (const float*)score, (float)prior_factor, (float*)grad, S, D);
    else
        hipLaunchKernelGGL(vi_full_grad_kernel<double>, dim3(D + 1), dim3(256), 0, s, (const double*)posterior, (const double*)eps,
                           (const double*)score, prior_factor, (double*)grad, S, D);
    return launch_status();
}
