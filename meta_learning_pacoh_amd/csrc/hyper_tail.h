// The hyper-parameter reduction of a step (pacoh_hyper_bwd: sums over the tasks of the per-problem lengthscale / outputscale /
// noise / constant-mean gradients with the softplus chain rule, the likelihood sum, the Cholesky-failure flag) as a device
// function over "virtual blocks" w = 0 .. P (f + 4) - 1 of 256 threads: run by its own kernel (misc.hip) or, since round 3, by extra
// workgroups of the MLP backward's slab reduction (mlp_fused.hip: one launch and one launch boundary less per step).
#pragma once
#include "common.h"
#include "step_tail.h"

namespace pacoh {

// The optimizer step of a PACOH-MAP iteration applied by the threads that FINISH a gradient entry (round 3): Adam is elementwise, the
// slab reduction's threads hold the networks' gradient entries and the hyper-parameter reduction's blocks the rest, so at world size
// 1 (no exchange between gradient and update) the AdamW launch disappears.  One parameter row (P = 1); op order of adam_dev_kernel.
template <typename T>
struct AdamInline {
    T* param; T* m; T* v;                     // param == nullptr: off
    const T* sc;                              // {1 - lr * weight_decay, lr / (1 - beta1^t), sqrt(1 - beta2^t), eps} in device memory
    T one_minus_b1, b2, one_minus_b2;
    int nseg; int lo[4], hi[4];               // trained column ranges (learning_mode)
    long* step_counter; T* cum;               // the feed's counter (+1 per step) and the running sum of the logged loss, by the likelihood block
    const long* counter; const T* sc2; int n_sc;   // pipelined feed (step_tail.h): the scalars are those of row sc2[*counter & 1] instead of sc
};
// One AdamW update, every operation rounded on its own in the order of torch.optim._single_tensor_adam (mul_(1 - lr wd); lerp_;
// mul_(beta2).addcmul_; sqrt / bias correction + eps; addcdiv_).  No fused multiply-adds: which of `a * b + c * d`'s products the
// compiler folds into the addition depends on the code around it, and the separate AdamW launch (adam_dev_kernel) and the update
// folded into the gradient epilogue (adam_inline) must give the same bits.
template <typename T>
__device__ __forceinline__ void adam_update(T& p, T g, T& m, T& v, T decay_mul, T one_minus_b1, T b2, T one_minus_b2, T step_size,
                                            T bc2_sqrt, T eps) {
#pragma clang fp contract(off)
    T pp = p * decay_mul;
    T d1 = g - m;
    T d2 = d1 * one_minus_b1;
    T mq = m + d2;
    T v1 = v * b2;
    T g1 = one_minus_b2 * g;
    T g2 = g1 * g;
    T vq = v1 + g2;
    T denom = t_sqrt<T>(vq) / bc2_sqrt + eps;
    T r = mq / denom;
    T u = step_size * r;
    p = pp - u; m = mq; v = vq;
}
// -> the entry's value after the call (unchanged where it is not trained)
template <typename T>
__device__ __forceinline__ T adam_inline(const AdamInline<T>& o, long q, T g) {
    if (!o.param) return T(0);
    bool in = false;
    for (int s = 0; s < o.nseg; ++s) in |= q >= o.lo[s] && q < o.hi[s];
    T p = o.param[q];
    if (!in) return p;
    const T* sc = o.counter ? o.sc2 + (*o.counter & 1) * o.n_sc + PACOH_SC_ADAM : o.sc;
    T mq = o.m[q], vq = o.v[q];
    adam_update<T>(p, g, mq, vq, sc[0], o.one_minus_b1, o.b2, o.one_minus_b2, sc[1], sc[2], sc[3]);
    o.param[q] = p; o.m[q] = mq; o.v[q] = vq;
    return p;
}

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
    AdamInline<T> opt;                       // PACOH-MAP at world size 1: the AdamW step on every entry this reduction finishes
    StepNextArgs<T> nx;                      // ... and the pipelined feed: the updated hyper-parameters' transforms are published by the
                                             // blocks that update them, tb + 1 more virtual blocks fetch the next step's operands
};

// virtual blocks of the reduction itself, and with the optional bandwidth block behind them
template <typename T> __host__ __device__ inline int hyper_bwd_blocks(const HyperBwdArgs<T>& a) { return a.P * (a.f + 4); }
template <typename T> __host__ __device__ inline int hyper_tail_blocks(const HyperBwdArgs<T>& a) {
    return hyper_bwd_blocks(a) + (a.sv_bw ? 1 : 0) + (a.nx.counter ? a.nx.tb + 1 : 0);
}

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
        for (int t0 = threadIdx.x; t0 < a.Tt; t0 += 4 * 256) {
            int v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int t = t0 + 256 * u; v[u] = a.info[(long)(t < a.Tt ? t : a.Tt - 1) * a.P + pp]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) bad |= v[u] < 0;      // (a clamped copy of the last task's value: the same verdict)
        }
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
    if (ncol == 1) {
        // (four tasks per trip, their loads requested together, added in the same order: one task per trip was one memory round trip
        //  per 256 tasks -- four in a row at cfg #3's 1 024 tasks per step)
        for (int t0 = threadIdx.x; t0 < a.Tt; t0 += 4 * 256) {
            T v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int t = t0 + 256 * u; v[u] = src[((long)(t < a.Tt ? t : a.Tt - 1) * a.P + p) * width + col]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) if (t0 + 256 * u < a.Tt) s += v[u];
        }
    } else {
        for (int t = threadIdx.x; t < a.Tt; t += 256)
            for (int c = 0; c < ncol; ++c) s += src[((long)t * a.P + p) * width + col + c];
    }
    s = subwave_sum<T>(s, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        s = (red[0] + red[1]) + (red[2] + red[3]);
        if (e == a.f + 3) {
            if (a.lik) a.lik[p] = a.lik_scale * s;
            if (a.opt.param) {                             // (as adam_dev_kernel's first thread: feed counter, running loss sum)
                if (a.opt.step_counter) *a.opt.step_counter += 1;
                if (a.opt.cum) *a.opt.cum += a.lik_scale * s;
            }
            return;
        }
        const T chain = (e == a.f + 2) ? T(1) : hyper_sigmoid<T>(a.theta[(long)p * a.stride + off]);      // (read BEFORE the update below)
        const T gv = s * chain;
        a.grad[(long)p * a.gstride + off] = gv;
        const T now = adam_inline<T>(a.opt, off, gv);
        if (a.opt.param && a.nx.counter && a.nx.ls) {       // the next step's transformed hyper-parameters, from the value just stored
            if (e < a.f) { const T v1 = softplus_t<T>(now); for (int c = 0; c < ncol; ++c) a.nx.ls[p * a.f + e + c] = v1; }
            else if (e == a.f) { if (a.nx.os) a.nx.os[p] = softplus_t<T>(now); }
            else if (e == a.f + 1) a.nx.noise[p] = softplus_t<T>(now) + a.nx.noise_floor;
        }
    }
}

// block w of hyper_tail_blocks(a): the reduction's blocks, then the bandwidth block
template <typename T>
__device__ __forceinline__ void hyper_tail_block(const HyperBwdArgs<T>& a, int w, T* red) {
    const int hb = hyper_bwd_blocks(a);
    if (w < hb) hyper_bwd_block<T>(a, w, red);
    else if (a.sv_bw && w == hb) svgd_bandwidth_block<T>(a.sv_d2, a.sv_P, a.sv_bw);
    else if (a.nx.counter) step_next_tail<T>(a.nx, w - hb - (a.sv_bw ? 1 : 0), a.nx.tb + 1);
}

}  // namespace pacoh
