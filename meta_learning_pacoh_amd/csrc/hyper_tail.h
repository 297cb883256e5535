// The hyper-parameter reduction of a step (pacoh_hyper_bwd: sums over the tasks of the per-problem lengthscale / outputscale /
// noise / constant-mean gradients with the softplus chain rule, the likelihood sum, the Cholesky-failure flag) as a device
// function over "virtual blocks" w = 0 .. P (f + 4) - 1 of 256 threads: run by its own kernel (misc.hip) or, since round 3, by extra
// workgroups of the MLP backward's slab reduction (mlp_fused.hip: one launch and one launch boundary less per step).
#pragma once
#include "common.h"
#include "step_tail.h"

namespace pacoh {

template <typename T>
struct HyperBwdArgs {
    const T* theta; long stride; int P, Tt, off_ls, f, off_os, off_noise, off_const;
    const T* d_ls; const T* d_os; const T* d_noise; const T* d_const;
    T* grad; long gstride;
    const T* lml; T* lik; T lik_scale;
    const int32_t* info; int32_t* fail_flag;
    int tie;                  // kernel families with ONE raw scale for all f dimensions (PACOH_KERNEL_COSINE): grad[off_ls] takes the
                              // sum over the f per-dimension gradients, the entries behind it do not exist
    const T* sv_d2; int sv_P; T* sv_bw;      // SVGD: ONE more virtual block computes the step's median-heuristic bandwidth from the
                                             // particles' distance matrix (step_tail.h, svgd_bandwidth_block) | sv_bw = nullptr
};

// virtual blocks of the reduction itself, and with the optional bandwidth block behind them
template <typename T> __host__ __device__ inline int hyper_bwd_blocks(const HyperBwdArgs<T>& a) { return a.P * (a.f + 4); }
template <typename T> __host__ __device__ inline int hyper_tail_blocks(const HyperBwdArgs<T>& a) { return hyper_bwd_blocks(a) + (a.sv_bw ? 1 : 0); }

template <typename T> __device__ __forceinline__ T hyper_sigmoid(T x) { return x > T(20) ? T(1) : T(1) / (T(1) + t_exp<T>(-x)); }

// red: 4 values of LDS; blockDim.x == 256
template <typename T>
__device__ __forceinline__ void hyper_bwd_block(const HyperBwdArgs<T>& a, int w, T* red) {
    const int per = a.f + 4;
    // the step's numerical status rides along: any problem whose jittered Cholesky failed (info < 0) raises the caller's flag
    // (gpytorch's psd_safe_cholesky raises NotPSDError at that point; the host checks the flag at its next synchronisation)
    if (a.info && a.fail_flag && w % per == a.f + 1) {
        const int pp = w / per;
        bool bad = false;
        for (int t = threadIdx.x; t < a.Tt; t += 256) bad |= a.info[(long)t * a.P + pp] < 0;
        if (bad) atomicOr(a.fail_flag, 1);
    }
    const int p = w / per, e = w - p * per;
    const T* src; int width, col, off;
    int ncol = 1;
    if (e < a.f) {
        src = a.d_ls; width = a.f; col = e; off = a.off_ls + e;
        if (a.tie) { if (e > 0) return; ncol = a.f; }
    }
    else if (e == a.f) { src = a.d_os; width = 1; col = 0; off = a.off_os; }
    else if (e == a.f + 1) { src = a.d_noise; width = 1; col = 0; off = a.off_noise; }
    else if (e == a.f + 2) { src = a.d_const; width = 1; col = 0; off = a.off_const; }
    else { src = a.lml; width = 1; col = 0; off = 0; }
    if (!src || off < 0) return;
    T s = 0;
    for (int t = threadIdx.x; t < a.Tt; t += 256)
        for (int c = 0; c < ncol; ++c) s += src[((long)t * a.P + p) * width + col + c];
    s = subwave_sum<T>(s, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0] + red[1]) + (red[2] + red[3]);
        if (e == a.f + 3) { if (a.lik) a.lik[p] = a.lik_scale * s; return; }
        const T chain = (e == a.f + 2) ? T(1) : hyper_sigmoid<T>(a.theta[(long)p * a.stride + off]);
        a.grad[(long)p * a.gstride + off] = s * chain;
    }
}

// block w of hyper_tail_blocks(a): the reduction's blocks, then the bandwidth block
template <typename T>
__device__ __forceinline__ void hyper_tail_block(const HyperBwdArgs<T>& a, int w, T* red) {
    if (w < hyper_bwd_blocks(a)) hyper_bwd_block<T>(a, w, red);
    else if (a.sv_bw) svgd_bandwidth_block<T>(a.sv_d2, a.sv_P, a.sv_bw);
}

}  // namespace pacoh
