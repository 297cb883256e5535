// Small kernels around the GP core: parameter transforms (A3), hyper-prior (A7), SVGD update
// direction (A9), fused Adam/AdamW step (A8-A10).  All are launch-latency sized (tens of particles,
// D ~ 10^3 parameters); they exist so that a whole meta-training step stays on the device and can be
// captured into one hipGraph.
#include "common.h"
#include "hyper_tail.h"
#include "step_tail.h"

#include <stdlib.h>
#include <string.h>

namespace pacoh {

// ---- the environment switches (switches.h): read when the library is loaded, again only through pacoh_reload_env --------------
void read_switches(Switches& s) {
    auto off0 = [](const char* name) { const char* e = getenv(name); return !(e && e[0] == '0'); };     // on unless NAME=0
    auto num = [](const char* name, int dflt) { const char* e = getenv(name); return (e && e[0]) ? atoi(e) : dflt; };
    s.chol_ll = off0("PACOH_CHOL_LL"); s.trtri_ll = off0("PACOH_TRTRI_LL"); s.retry_fused = off0("PACOH_RETRY_FUSED");
    s.gemm_tile = off0("PACOH_GEMM_TILE"); s.trtri_blocked = off0("PACOH_TRTRI_BLOCKED"); s.chol_blocked = off0("PACOH_CHOL_BLOCKED");
    s.grad_mfma = off0("PACOH_GRAD_MFMA"); s.grad_mfma_f32 = off0("PACOH_GRAD_MFMA_F32"); s.gram_mfma = off0("PACOH_GRAM_MFMA");
    s.dense_pad = off0("PACOH_DENSE_PAD");
    s.mfma = num("PACOH_DISABLE_MFMA", 0) != 1;
    s.gp8 = off0("PACOH_GP8");
    s.gp_reg = off0("PACOH_GP_REG"); s.gp_reg_predict = off0("PACOH_GP_REG_PREDICT"); s.gp_reg_max_n = num("PACOH_GP_REG_MAX_N", 128);
    s.fused_mlp = num("PACOH_DISABLE_FUSED_MLP", 0) == 0;
    const char* e = getenv("PACOH_MLP_PATH");
    s.mlp_path = !e ? 0 : (!strcmp(e, "mfma") ? 1 : (!strcmp(e, "layers") ? 3 : 0));
    s.mlp_stash = num("PACOH_MLP_STASH", -1);
    s.lds_pad_gp = num("PACOH_LDS_PAD_GP", -1); s.lds_pad_mlp = num("PACOH_LDS_PAD_MLP", -1);
    s.mt_nt = num("PACOH_MT_NT", 0);
    s.fused_bwd_pb = num("PACOH_FUSED_BWD_PB", 0); s.fused_fwd_pb = num("PACOH_FUSED_FWD_PB", 0); s.fused_fwd_tpw = num("PACOH_FUSED_FWD_TPW", 0);
}
Switches g_sw = []() { Switches s; read_switches(s); return s; }();

// ---- softplus (torch.nn.functional.softplus: beta=1, threshold=20) ------------------------------
template <typename T>
__global__ void softplus_fwd_kernel(const T* __restrict__ raw, T* __restrict__ out, T floor_, long count) {
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= count) return;
    T x = raw[q];
    out[q] = (x > T(20) ? x : t_log1p<T>(t_exp<T>(x))) + floor_;
}

template <typename T>
__global__ void softplus_bwd_kernel(const T* __restrict__ raw, const T* __restrict__ g, T* __restrict__ d_raw,
                                    int accumulate, long count) {
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= count) return;
    T x = raw[q];
    T sg = x > T(20) ? T(1) : T(1) / (T(1) + t_exp<T>(-x));
    T v = g[q] * sg;
    d_raw[q] = accumulate ? d_raw[q] + v : v;
}

// ---- all GP hyper-parameter transforms of one step in one launch (A3) -------------------------------
// (softplus_t: common.h)
template <typename T> __device__ __forceinline__ T sigmoid_t(T x) { return x > T(20) ? T(1) : T(1) / (T(1) + t_exp<T>(-x)); }

template <typename T>
__global__ void hyper_fwd_kernel(const T* __restrict__ theta, long stride, int P, int off_ls, int f, int off_os, int off_noise,
                                 T noise_floor, T* __restrict__ ls, T* __restrict__ os, T* __restrict__ noise, int tie) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    const int per = f + 2;
    if (q >= P * per) return;
    const int p = q / per, e = q - p * per;
    const T* th = theta + (long)p * stride;
    if (e < f) ls[p * f + e] = softplus_t<T>(th[off_ls + (tie ? 0 : e)]);     // (tie: one raw scale shared by all f dimensions)
    else if (e == f) { if (os && off_os >= 0) os[p] = softplus_t<T>(th[off_os]); }
    else noise[p] = softplus_t<T>(th[off_noise]) + noise_floor;
}

// grad[p, off_*] = sigmoid(raw) * sum_t d_*[t, p, .]   (softplus chain rule), constant mean: plain sum; optionally also
// lik[p] = lik_scale * sum_t lml[t, p] (the likelihood term of the meta log-probability rides along: same loop over tasks).
// One 256-thread workgroup per entry, fixed summation order (deterministic).
template <typename T>
__global__ void __launch_bounds__(256) hyper_bwd_kernel(HyperBwdArgs<T> a) {
    __shared__ T red[4];
    hyper_tail_block<T>(a, blockIdx.x, red);              // (hyper_tail.h: shared with the slab reduction of the fused MLP backward)
}

// ---- hyper-prior: independent Normals over all D entries ----------------------------------------
template <typename T>
__global__ void __launch_bounds__(1024) prior_kernel(const T* __restrict__ theta, const T* __restrict__ mu,
                                                     const T* __restrict__ sd, T* __restrict__ logp,
                                                     T* __restrict__ grad, T grad_scale, int D, const T* __restrict__ in_scale = nullptr) {
    // one 1024-thread workgroup per parameter row: only P (= particles) rows exist, so the kernel is pure latency; 16 waves
    // per row keep the dependent load -> log -> store chain to 2-3 trips (8.2 -> ~4 us at P = 20, D = 2534)
    __shared__ T red[16];
    const int p = blockIdx.x;
    const T* th = theta + (long)p * D;
    const T HALF_LOG2PI = T(0.9189385332046727);
    T acc = 0;
    const T gin = in_scale ? in_scale[0] : T(1);                    // (pacoh_prior_score_dev: the incoming score's factor, device-side)
    for (int d = threadIdx.x; d < D; d += 1024) {
        T s = sd[d];
        T zv = (th[d] - mu[d]) / s;
        acc += T(-0.5) * zv * zv - t_log<T>(s) - HALF_LOG2PI;
        if (grad) grad[(long)p * D + d] = (in_scale ? gin * grad[(long)p * D + d] : grad[(long)p * D + d]) + grad_scale * (-zv / s);
    }
    acc = subwave_sum<T>(acc, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && logp) {
        T tot = 0;
        for (int w = 0; w < 16; ++w) tot += red[w];                 // fixed order: deterministic
        logp[p] = tot;
    }
}

// ---- SVGD ---------------------------------------------------------------------------------------
// stage 1: squared distances, one 256-thread workgroup per (i,j) pair, direct differences (svgd_dist_block: step_tail.h)
template <typename T>
__global__ void __launch_bounds__(256) svgd_dist_kernel(const T* __restrict__ X, T* __restrict__ d2, int P, int D, T* __restrict__ snap = nullptr) {
    svgd_dist_block<T>(X, d2, P, D, snap, (int)blockIdx.x);
}
// the same behind a forward pass that has no tail of its own (step_tail.h): distances + snapshot + the step counter's increment
template <typename T>
__global__ void __launch_bounds__(256) svgd_dist_advance_kernel(SvgdDistTail<T> t) { svgd_dist_tail<T>(t, (int)blockIdx.x, (int)gridDim.x); }

// (wave_median_full_matrix -- the median of the full PxP matrix by one wavefront -- lives in step_tail.h: the step's bandwidth is
// computed ahead of the update by a workgroup riding in the hyper-parameter reduction, svgd_bandwidth_block)
// More than 64 particles (the register sort above holds 2048 pair values): the two middle order statistics of the full matrix by
// bisection on the IEEE bit pattern (monotone for values >= 0), one 1024-thread workgroup per statistic: per round every thread
// counts its share of the pairs below the candidate straight from d2 (L2-resident), 31 / 63 rounds.  mids[h] receives entry
// (N-1)/2 resp. N/2 of the sorted P x P matrix; the consumers average them.  No limit on P from this kernel.
template <typename T> struct MedBits;
template <> struct MedBits<float> { using U = uint32_t; static constexpr int NB = 32;
    static __device__ __forceinline__ U to(float v) { return __float_as_uint(v); } static __device__ __forceinline__ float from(U u) { return __uint_as_float(u); } };
template <> struct MedBits<double> { using U = uint64_t; static constexpr int NB = 64;
    static __device__ __forceinline__ U to(double v) { return (U)__double_as_longlong(v); } static __device__ __forceinline__ double from(U u) { return __longlong_as_double((long long)u); } };
template <typename T>
__global__ void __launch_bounds__(1024) svgd_median_large_kernel(const T* __restrict__ d2, int P, T* __restrict__ mids) {
    using U = typename MedBits<T>::U;
    __shared__ int part[16];
    __shared__ int total_s;
    const long N = (long)P * P;
    const long m = blockIdx.x == 0 ? (N - 1) / 2 : N / 2;
    if (m < P) { if (threadIdx.x == 0) mids[blockIdx.x] = T(0); return; }        // (P zeros lead the sorted matrix)
    const int k = (int)((m - P) >> 1);               // index into the sorted list of the P(P-1)/2 pair values
    U result = 0;
    for (int bit = MedBits<T>::NB - 2; bit >= 0; --bit) {
        const U cand = result | (U(1) << bit);
        int cnt = 0;
        for (long q = threadIdx.x; q < N; q += 1024) {
            const int i = (int)(q / P), j = (int)(q - (long)i * P);
            if (i < j) cnt += MedBits<T>::to(d2[q]) < cand ? 1 : 0;
        }
#pragma unroll
        for (int w = 1; w < 64; w <<= 1) cnt += __shfl_xor(cnt, w, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += part[w]; total_s = t; }
        __syncthreads();
        if (total_s <= k) result = cand;             // (uniform: every thread reads the same total)
        __syncthreads();
    }
    if (threadIdx.x == 0) mids[blockIdx.x] = MedBits<T>::from(result);
}

// stage 2: bandwidth (median heuristic), kernel matrix and its row sums.  One workgroup; wave 0 finds the median.
template <typename T>
__global__ void __launch_bounds__(256) svgd_kmat_kernel(const T* __restrict__ d2, T bandwidth, T* __restrict__ Kmat,
                                                        T* __restrict__ rowsum, T* __restrict__ gamma_out,
                                                        T* __restrict__ bw_out, int P, const T* __restrict__ mids = nullptr) {
    __shared__ T gam_s;
    const int N = P * P;
    T bw = bandwidth;
    if (!(bandwidth > T(0))) {
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const int npairs = P * (P - 1) / 2;
            T med;
            if (mids) med = (mids[0] + mids[1]) * T(0.5);                    // P > 64: svgd_median_large_kernel ran in front
            else if (npairs <= 256) med = wave_median_full_matrix<T, 4>(d2, P, lane);
            else if (npairs <= 512) med = wave_median_full_matrix<T, 8>(d2, P, lane);
            else if (npairs <= 1024) med = wave_median_full_matrix<T, 16>(d2, P, lane);
            else med = wave_median_full_matrix<T, 32>(d2, P, lane);
            if (lane == 0) {
                T h = med / (T(2) * t_log<T>(T(P + 1)));
                T b = t_sqrt<T>(h);
                gam_s = T(1) / (T(1e-8) + T(2) * b * b);
                if (bw_out) *bw_out = b;
            }
        }
    } else if (threadIdx.x == 0) {
        gam_s = T(1) / (T(1e-8) + T(2) * bw * bw);
        if (bw_out) *bw_out = bw;
    }
    __syncthreads();
    const T gam = gam_s;
    if (threadIdx.x == 0) *gamma_out = gam;
    for (int q = threadIdx.x; q < N; q += 256) Kmat[q] = t_exp<T>(-gam * d2[q]);
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += 256) {
        T s = 0;
        for (int j = 0; j < P; ++j) s += t_exp<T>(-gam * d2[i * P + j]);
        rowsum[i] = s;
    }
}

// stage 3: phi[i,d] = (sum_j K_ij (s_jd - 2 gamma x_jd) + 2 gamma x_id rowsum_i) / P; one thread per (i, d)
template <typename T>
__global__ void __launch_bounds__(256) svgd_phi_kernel(const T* __restrict__ X, const T* __restrict__ score,
                                                       const T* __restrict__ Kmat, const T* __restrict__ rowsum,
                                                       const T* __restrict__ gamma_p, int neg, T* __restrict__ phi,
                                                       int P, int D) {
    const int i = blockIdx.y;
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    const T gam2 = T(2) * gamma_p[0];
    const T* Ki = Kmat + (long)i * P;                // wave-uniform -> scalar loads
    T acc = 0;
    for (int j = 0; j < P; ++j) acc = fma(Ki[j], score[(long)j * D + d] - gam2 * X[(long)j * D + d], acc);
    const T r = (acc + gam2 * X[(long)i * D + d] * rowsum[i]) / T(P);
    phi[(long)i * D + d] = neg ? -r : r;
}

// stages 2 + 3, fused variant used by the SVGD learner's step.  Every workgroup (particle i, 256 dimensions) recomputes the
// bandwidth from the P x P distance matrix (one wavefront, ~1.5 us, all workgroups in parallel) and its own kernel row
// k_i. = exp(-gamma d2[i,.]) -- cheaper than a separate single-workgroup launch in front.  Then the hyper-prior's score is
// added on the fly (score_j += prior_factor * d log N(x_j; mu, sd) / dx), phi is formed as above and the optimizer step on
// particle i is applied in the same thread (Adam with the op order of adam_kernel, or plain SGD), written to X_out (other
// threads still read X).  Replaces prior.log_prob's backward + SVGD.phi + optimizer.step (random_gp.py:128-157,
// svgd.py:12-28) for one step.
template <typename T>
__global__ void __launch_bounds__(256) svgd_update_kernel(const T* __restrict__ X, const T* __restrict__ score,
                                                          const T* __restrict__ mu, const T* __restrict__ sd, T prior_factor,
                                                          const T* __restrict__ d2, T bandwidth, T* __restrict__ bw_out,
                                                          int use_adam, T lr, T one_minus_b1, T b2,
                                                          T one_minus_b2, T step_size, T bc2_sqrt, T eps,
                                                          T* __restrict__ m, T* __restrict__ v, T* __restrict__ X_out, int P, int D,
                                                          const T* __restrict__ sc = nullptr, const T* __restrict__ mids = nullptr,
                                                          long* __restrict__ step_counter = nullptr,
                                                          StepNextArgs<T> nx = StepNextArgs<T>{}) {
    if (nx.counter) {
        // pipelined step (step_tail.h): rows blockIdx.y >= P of the grid fetch the next step's operands; the others find this step's
        // scalars in the row of sc2 the step counter selects
        if ((int)blockIdx.y >= P) {
            step_next_tail<T>(nx, ((int)blockIdx.y - P) * (int)gridDim.x + (int)blockIdx.x, ((int)gridDim.y - P) * (int)gridDim.x);
            return;
        }
    }
    if (step_counter && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *step_counter += 1;   // (as in adam_dev_kernel)
    __shared__ T Ki[PACOH_SVGD_MAX_PARTICLES];
    __shared__ T gam_s, rowsum_s;
    __shared__ T sc_s[5];                                  // score scale, lr, Adam step size, sqrt of bias correction 2, eps
    const int i = blockIdx.y;
    // The kernel is a chain of L2 round trips, not arithmetic (P x 2 loads per thread behind the bandwidth / kernel-row phase of
    // wave 0: 11 us at P = 20 with four particles' loads in flight behind the barrier).  Everything that does not depend on that
    // phase is requested BEFORE it: the first twenty particles' coordinates and scores, the own coordinate, the optimizer state.
    constexpr int PF = 20;
    const int d = blockIdx.x * 256 + threadIdx.x;
    const int dc = d < D ? d : D - 1;
    const int npre = P < PF ? P : PF;
    T xpre[PF], spre[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        xpre[u] = u < npre ? X[(long)u * D + dc] : T(0);
        spre[u] = u < npre ? score[(long)u * D + dc] : T(0);
    }
    const long q = (long)i * D + dc;
    const T xi = X[q];
    const T md = mu ? mu[dc] : T(0), sdv = mu ? sd[dc] : T(1);
    T mq = use_adam ? m[q] : T(0), vq0 = use_adam ? v[q] : T(0);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        T bw = bandwidth;
        if (!(bandwidth > T(0))) {
            if (nx.bw_pre) bw = *nx.bw_pre;                                  // computed ahead, beside the hyper-parameter reduction
            else if (mids) bw = t_sqrt<T>(((mids[0] + mids[1]) * T(0.5)) / (T(2) * t_log<T>(T(P + 1))));   // P > 64: svgd_median_large_kernel ran in front
            else bw = svgd_median_bandwidth<T>(d2, P, lane);
        }
        const T gam = T(1) / (T(1e-8) + T(2) * bw * bw);
        T ksum = 0;
        for (int j = lane; j < P; j += 64) { const T kv = t_exp<T>(-gam * d2[i * P + j]); Ki[j] = kv; ksum += kv; }
        const T rs = subwave_sum<T>(ksum, 64);
        if (lane == 0) { gam_s = gam; rowsum_s = rs; if (bw_out && blockIdx.x == 0 && i == 0) *bw_out = bw; }
    } else if (threadIdx.x == 64) {
        // the step scalars, fetched by a wave that only waits for the bandwidth phase anyway: in the pipelined step their address
        // hangs on the step counter (two dependent round trips, which would otherwise sit in front of the prefetch above)
        const T* scp = nx.counter ? nx.sc2 + (*nx.counter & 1) * nx.n_sc : sc;
        if (scp) { sc_s[0] = scp[0]; sc_s[1] = scp[1]; sc_s[2] = scp[5]; sc_s[3] = scp[6]; sc_s[4] = scp[7]; }     // PACOH_SC_* (pacoh_gp.h)
        else { sc_s[0] = T(1); sc_s[1] = lr; sc_s[2] = step_size; sc_s[3] = bc2_sqrt; sc_s[4] = eps; }
    }
    __syncthreads();
    if (d >= D) return;
    const T score_scale = sc_s[0];
    lr = sc_s[1]; step_size = sc_s[2]; bc2_sqrt = sc_s[3]; eps = sc_s[4];
    const T gam2 = T(2) * gam_s;
    const T pscale = mu ? prior_factor / (sdv * sdv) : T(0);
    T acc = 0;
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        if (u < npre) {
            const T sj = score_scale * spre[u] - pscale * (xpre[u] - md);
            acc = fma(Ki[u], sj - gam2 * xpre[u], acc);
        }
    }
    int j = npre;
    for (; j + 10 <= P; j += 10) {                  // ten particles' loads in flight per round
        T xj[10], scv[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) { xj[u] = X[(long)(j + u) * D + d]; scv[u] = score[(long)(j + u) * D + d]; }
#pragma unroll
        for (int u = 0; u < 10; ++u) {
            const T sj = score_scale * scv[u] - pscale * (xj[u] - md);
            acc = fma(Ki[j + u], sj - gam2 * xj[u], acc);
        }
    }
    for (; j + 4 <= P; j += 4) {
        T xj[4], sc4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { xj[u] = X[(long)(j + u) * D + d]; sc4[u] = score[(long)(j + u) * D + d]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const T sj = score_scale * sc4[u] - pscale * (xj[u] - md);
            acc = fma(Ki[j + u], sj - gam2 * xj[u], acc);
        }
    }
    for (; j < P; ++j) {
        const T xj = X[(long)j * D + d];
        const T sj = score_scale * score[(long)j * D + d] - pscale * (xj - md);
        acc = fma(Ki[j], sj - gam2 * xj, acc);
    }
    const T r = (acc + gam2 * xi * rowsum_s) / T(P);          // phi[i,d]
    T xn;
    if (use_adam) {
        const T g = -r;                                       // particles.grad = -phi (svgd.py:27)
        mq = mq + (g - mq) * one_minus_b1;
        const T vq = vq0 * b2 + one_minus_b2 * g * g;
        const T denom = t_sqrt<T>(vq) / bc2_sqrt + eps;
        xn = xi - step_size * (mq / denom);
        m[q] = mq; v[q] = vq;
    } else {
        xn = fma(lr, r, xi);
    }
    X_out[q] = xn;
    if (nx.counter && nx.ls) {
        // the transformed hyper-parameters of the updated particle (what pacoh_step_begin computes from theta at the next step),
        // by the threads that hold their raw values
        if (d == nx.off_noise) nx.noise[i] = softplus_t<T>(xn) + nx.noise_floor;
        else if (nx.os && d == nx.off_os) nx.os[i] = softplus_t<T>(xn);
        else if (nx.tie) { if (d == nx.off_ls) { const T v1 = softplus_t<T>(xn); for (int e = 0; e < nx.f; ++e) nx.ls[i * nx.f + e] = v1; } }
        else if (d >= nx.off_ls && d < nx.off_ls + nx.f) nx.ls[i * nx.f + (d - nx.off_ls)] = softplus_t<T>(xn);
    }
}

// ---- Adam / AdamW, op order of torch.optim._single_tensor_adam -----------------------------------
template <typename T>
__global__ void adam_kernel(T* __restrict__ param, const T* __restrict__ grad, T* __restrict__ m, T* __restrict__ v,
                            T decay_mul, T one_minus_b1, T b2, T one_minus_b2, T step_size, T bc2_sqrt, T eps, long count) {
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= count) return;
    T g = grad[q];
    T p = param[q] * decay_mul;
    T mq = m[q];
    mq = mq + (g - mq) * one_minus_b1;                 // exp_avg.lerp_(grad, 1 - beta1)
    T vq = v[q] * b2 + one_minus_b2 * g * g;           // mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    T denom = t_sqrt<T>(vq) / bc2_sqrt + eps;
    p = p - step_size * (mq / denom);
    param[q] = p; m[q] = mq; v[q] = vq;
}

// same update with the step-dependent scalars read from device memory, so that the launch can be captured once
// into a hipGraph and replayed every iteration: sc = {decay_mul, step_size, bc2_sqrt, eps}
template <typename T>
__global__ void adam_dev_kernel(T* __restrict__ param, const T* __restrict__ grad, T* __restrict__ m, T* __restrict__ v,
                                const T* __restrict__ sc, T one_minus_b1, T b2, T one_minus_b2, long count,
                                long* __restrict__ step_counter = nullptr, T* __restrict__ cum = nullptr,
                                const T* __restrict__ loss = nullptr) {
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q == 0 && step_counter) *step_counter += 1;  // the step's last launch also advances the feed (pacoh_step_begin, advance = 0)
    if (q == 0 && cum) *cum += *loss;                // ... and keeps the running sum of the logged loss (GPR_meta_mll.py:119-125)
    if (q >= count) return;
    T p = param[q], mq = m[q], vq = v[q];
    adam_update<T>(p, grad[q], mq, vq, sc[0], one_minus_b1, b2, one_minus_b2, sc[1], sc[2], sc[3]);          // (hyper_tail.h)
    param[q] = p; m[q] = mq; v[q] = vq;
}

// ---- PACOH-VI: reparameterised sample of the diagonal Gaussian posterior and the ELBO gradient (A10) ----
// theta[s,d] = loc[d] + exp(scale[d]) * eps[s,d];  log_q[s] = sum_d (-eps^2/2 - scale[d] - log(2 pi)/2)
// sum over a block of NT threads, in thread 0: wave sums, then the waves' sums in a fixed tree (NT = 256: the four-term expression
// every kernel of this file has always used -- same bits)
template <typename T, int NT>
__device__ __forceinline__ T block_sum_(T acc, T* red /*[NT / 64]*/) {
    acc = subwave_sum<T>(acc, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    T tot = 0;
    if (threadIdx.x == 0) {
        if constexpr (NT == 256) tot = (red[0] + red[1]) + (red[2] + red[3]);
        else {
            T q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = (red[4 * k] + red[4 * k + 1]) + (red[4 * k + 2] + red[4 * k + 3]);
            tot = (q[0] + q[1]) + (q[2] + q[3]);
        }
    }
    return tot;
}
// threads per sample block: 1 024 where a row is long (D > 2 048: PACOH-VI at the launchers' 4 x 32 networks has D = 6 566 -- 26
// entries with an exp each per thread of a 256-thread block were 5.5 us of the step's first launch), else 256
__host__ __device__ constexpr int vi_sample_nt(int D) { return D > 2048 ? 1024 : 256; }

template <typename T, int NT>
__global__ void __launch_bounds__(NT) vi_sample_kernel(const T* __restrict__ post /*[2,D]*/, const T* __restrict__ eps,
                                                       T* __restrict__ theta, T* __restrict__ log_q, int D) {
    __shared__ T red[NT / 64];
    const int s_ = blockIdx.x;
    const T HALF_LOG2PI = T(0.9189385332046727);
    T acc = 0;
    for (int d = threadIdx.x; d < D; d += NT) {
        const T e = eps[(long)s_ * D + d], sc = post[D + d];
        theta[(long)s_ * D + d] = post[d] + t_exp<T>(sc) * e;
        acc += T(-0.5) * e * e - sc - HALF_LOG2PI;
    }
    const T tot = block_sum_<T, NT>(acc, red);
    if (threadIdx.x == 0) log_q[s_] = tot;
}

// grad[0,d] = -mean_s score[s,d];  grad[1,d] = -mean_s (score[s,d] * exp(scale[d]) * eps[s,d] + prior_factor)
template <typename T>
__global__ void vi_grad_kernel(const T* __restrict__ post, const T* __restrict__ eps, const T* __restrict__ score,
                               T prior_factor, T* __restrict__ grad, int S, int D) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const T sig = t_exp<T>(post[D + d]);
    T gl = 0, gs = 0;
    for (int s_ = 0; s_ < S; ++s_) {
        const T sc = score[(long)s_ * D + d];
        gl += sc;
        gs += sc * sig * eps[(long)s_ * D + d] + prior_factor;
    }
    grad[d] = -gl / T(S);
    grad[D + d] = -gs / T(S);
}

// The whole update half of a PACOH-VI step (diagonal posterior, Adam) in ONE launch -- pre-factor on the likelihood score, hyper-prior
// score and log-density, ELBO value, reparameterisation gradient (vi_grad_kernel) and the Adam step with its scalars from device
// memory: the seven launches it replaces cost ~5 us each between dependent kernels, more than their work.  sc = a step-scalar row
// (PACOH_SC_*).  One thread per dimension d; the scalar loss from per-block partial sums, combined in block order by the last block to finish.
//   s~[s,d]   = sc[0] score[s,d] - prior_factor (theta[s,d] - mu[d]) / sd[d]^2
//   grad_loc  = -mean_s s~[s,d];   grad_scale = -mean_s (s~[s,d] exp(scale[d]) eps[s,d] + prior_factor)          (GPR_meta_vi.py:216-224)
//   loss      = -mean_s (sc[0] lik[s] + prior_factor log N(theta_s; mu, sd)) + prior_factor mean_s log_q[s]
template <typename T>
__global__ void __launch_bounds__(256) vi_update_kernel(T* __restrict__ post, const T* __restrict__ eps, const T* __restrict__ theta,
                                                        const T* __restrict__ score, const T* __restrict__ lik, const T* __restrict__ log_q,
                                                        const T* __restrict__ mu, const T* __restrict__ sd, T prior_factor,
                                                        const T* __restrict__ sc, T one_minus_b1, T b2, T one_minus_b2,
                                                        T* __restrict__ m, T* __restrict__ v, T* __restrict__ loss,
                                                        long* __restrict__ step_counter, T* __restrict__ partial /*[gridDim.x]*/,
                                                        unsigned* __restrict__ ticket, int S, int D) {
    __shared__ T red[4];
    __shared__ bool last_s;
    const T pref = sc[0];
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x == 0 && step_counter) *step_counter += 1;
    T lp = 0;                                          // this dimension's share of sum_s log N(theta_s; mu, sd)
    if (d < D) {
        const T sig = t_exp<T>(post[D + d]);
        const T md = mu[d], sdv = sd[d];
        const T pscale = prior_factor / (sdv * sdv);
        const T lconst = t_log<T>(sdv) + T(0.9189385332046727);
        T gl = 0, gs = 0;
        // (twelve samples per trip, their loads requested together: one sample per trip was one memory round trip per sample)
        constexpr int U = 12;
        for (int s0 = 0; s0 < S; s0 += U) {
            T tq[U], sq[U], eq[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long q = (long)(s0 + u < S ? s0 + u : S - 1) * D + d;
                tq[u] = theta[q]; sq[u] = score[q]; eq[u] = eps[q];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (s0 + u >= S) break;
                const T dv = tq[u] - md, zv = dv / sdv;
                lp += T(-0.5) * zv * zv - lconst;
                const T st = pref * sq[u] - pscale * dv;
                gl += st;
                gs += st * sig * eq[u] + prior_factor;
            }
        }
        const T decay_mul = sc[4], step_size = sc[5], bc2_sqrt = sc[6], epsv = sc[7];
#pragma unroll
        for (int h = 0; h < 2; ++h) {                  // loc, then log-scale: the op order of adam_dev_kernel
            const long q = (long)h * D + d;
            const T g = -(h == 0 ? gl : gs) / T(S);
            T p = post[q] * decay_mul;
            T mq = m[q];
            mq = mq + (g - mq) * one_minus_b1;
            const T vq = v[q] * b2 + one_minus_b2 * g * g;
            const T denom = t_sqrt<T>(vq) / bc2_sqrt + epsv;
            p = p - step_size * (mq / denom);
            post[q] = p; m[q] = mq; v[q] = vq;
        }
    }
    // ---- the loss (a logged value): block partials of the log-prior, combined in block order by whichever block finishes last
    lp = subwave_sum<T>(lp, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lp;
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
        __threadfence();
        last_s = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (last_s) {
        // the same sums in the same order as one thread walking the three arrays -- but the values come in through 256 parallel loads
        // per trip and are added up out of LDS: as dependent global loads by one thread (26 partials + 2 x 10 samples at the launchers'
        // shape) the walk was ~36 memory round trips, most of this launch's 11 us
        __shared__ T stage[256];
        __threadfence();
        T tot = 0, lq = 0;
        for (unsigned k0 = 0; k0 < gridDim.x; k0 += 256) {
            const unsigned cnt = gridDim.x - k0 < 256u ? gridDim.x - k0 : 256u;
            if (threadIdx.x < cnt) stage[threadIdx.x] = __hip_atomic_load(partial + k0 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (threadIdx.x == 0) for (unsigned k = 0; k < cnt; ++k) tot += stage[k];
            __syncthreads();
        }
        tot *= prior_factor;
        for (int s0 = 0; s0 < S; s0 += 128) {
            const int cnt = S - s0 < 128 ? S - s0 : 128;
            if ((int)threadIdx.x < cnt) { stage[threadIdx.x] = lik[s0 + threadIdx.x]; stage[128 + threadIdx.x] = log_q[s0 + threadIdx.x]; }
            __syncthreads();
            if (threadIdx.x == 0) for (int k = 0; k < cnt; ++k) { tot += pref * stage[k]; lq += stage[128 + k]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            *loss = (prior_factor * lq - tot) / T(S);
            *ticket = 0;                               // (ready for the next launch / replay)
        }
    }
}

// y += alpha * x  (plain SGD step of the optimizer='SGD' option)
template <typename T>
__global__ void axpy_kernel(T* __restrict__ y, const T* __restrict__ x, T alpha, long count) {
    long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < count) y[q] = fma(alpha, x[q], y[q]);
}

// gather of a step's task batch: out_x[b] = x[idx[b]], out_y[b] = y[idx[b]], out_nv[b] = n_valid[idx[b]]; one workgroup per
// selected task, 16-byte copies when the row sizes allow
template <typename T>
__global__ void __launch_bounds__(256) gather_tasks_kernel(const T* __restrict__ x, const T* __restrict__ y,
                                                           const int32_t* __restrict__ n_valid, const long* __restrict__ idx,
                                                           T* __restrict__ ox, T* __restrict__ oy, int32_t* __restrict__ onv,
                                                           int nx, int ny) {
    const long b = blockIdx.x, t = idx[b];
    const T* sx = x + t * nx;
    const T* sy = y + t * ny;
    T* dx = ox + b * nx;
    T* dy = oy + b * ny;
    for (int q = threadIdx.x; q < nx; q += 256) dx[q] = sx[q];
    for (int q = threadIdx.x; q < ny; q += 256) dy[q] = sy[q];
    if (threadIdx.x == 0 && n_valid) onv[b] = n_valid[t];
}

// this step's row of the pre-uploaded task draws and step scalars, selected by a device-side counter (then advanced): what
// lets a whole meta-training step be captured once in a hipGraph and replayed with nothing but the replay call on the host
template <typename T>
__global__ void __launch_bounds__(256) step_select_kernel(const long* __restrict__ idx_all, int tb, const T* __restrict__ sc_all, int n_sc,
                                                          const T* __restrict__ aux_all, long n_aux, long* __restrict__ counter,
                                                          long* __restrict__ idx_out, T* __restrict__ sc_out, T* __restrict__ aux_out) {
    const long row = *counter;
    if (blockIdx.x == 0) {
        for (int q = threadIdx.x; q < tb; q += 256) idx_out[q] = idx_all[row * tb + q];
        for (int q = threadIdx.x; q < n_sc; q += 256) sc_out[q] = sc_all[row * n_sc + q];
    } else {                                    // blocks 1.. copy the auxiliary payload (PACOH-VI: the step's reparameterisation noise)
        for (long q = (long)(blockIdx.x - 1) * 256 + threadIdx.x; q < n_aux; q += (long)(gridDim.x - 1) * 256) aux_out[q] = aux_all[row * n_aux + q];
    }
}
// (the counter is advanced by its own one-thread launch after every block has read it)
__global__ void step_advance_kernel(long* __restrict__ counter) { *counter += 1; }

// The first launch of a captured step: row = *counter selects this step's operands, and in ONE launch
//   blocks [0, tb)      gather the step's task batch  out_x[b] = x[idx_all[row, b]] ... (A12)
//   block  tb           copies the step scalars (and, blocks tb+2.., the auxiliary payload) into their fixed buffers
//   block  tb+1         runs the hyper-parameter transforms of the step (A3) when theta is given
// The counter is advanced by a one-thread launch behind it (a last-block ticket inside this kernel -- device-scope fence plus an
// atomic per block -- cost 75 us at 1026 blocks, against 4.5 us for the extra launch).
template <typename T>
struct StepBeginArgs {
    const long* idx_all; int tb; const T* sc_all; int n_sc; const T* aux_all; long n_aux; long* counter; int* ticket;
    T* sc_out; T* aux_out;
    const T* x; const T* y; const int32_t* n_valid; T* ox; T* oy; int32_t* onv; int nx, ny;
    const T* theta; long stride; int P, off_ls, f, off_os, off_noise; T noise_floor; T* ls; T* os; T* noise;
    int aux_blocks;                                  // blocks [tb+2, tb+2+aux_blocks) copy the auxiliary payload
    const T* sv_X; T* sv_d2; T* sv_snap; int sv_P, sv_D;         // blocks behind them: SVGD pairwise distances + particle snapshot
    int tie;                                         // one raw scale for all f dimensions (kernel families other than ARD-RBF)
    // PACOH-VI (diagonal posterior): the blocks behind all others draw the step's samples theta[s] = loc + exp(scale) * eps[s] from
    // the step's noise row (aux_all[row], NOT the copy another block of this launch is making), their log q, and the transformed
    // hyper-parameters of the samples -- pacoh_vi_sample + pacoh_hyper_fwd without their launches
    const T* vi_post; T* vi_theta; T* vi_logq; int vi_S, vi_D;
};

template <typename T, int NT>
__global__ void __launch_bounds__(NT) step_begin_kernel(StepBeginArgs<T> a) {
    const long row = *a.counter;
    const int blk = blockIdx.x;
    if (blk < a.tb) {
        const long t = a.idx_all[row * a.tb + blk];
        const T* sx = a.x + t * a.nx;
        const T* sy = a.y + t * a.ny;
        T* dx = a.ox + (long)blk * a.nx;
        T* dy = a.oy + (long)blk * a.ny;
        for (int q = threadIdx.x; q < a.nx; q += NT) dx[q] = sx[q];
        for (int q = threadIdx.x; q < a.ny; q += NT) dy[q] = sy[q];
        if (threadIdx.x == 0 && a.n_valid) a.onv[blk] = a.n_valid[t];
    } else if (blk == a.tb) {
        for (int q = threadIdx.x; q < a.n_sc; q += NT) a.sc_out[q] = a.sc_all[row * a.n_sc + q];
    } else if (blk == a.tb + 1) {
        if (a.theta) {
            const int per = a.f + 2;
            for (int q = threadIdx.x; q < a.P * per; q += NT) {
                const int p = q / per, e = q - p * per;
                const T* th = a.theta + (long)p * a.stride;
                if (e < a.f) a.ls[p * a.f + e] = softplus_t<T>(th[a.off_ls + (a.tie ? 0 : e)]);
                else if (e == a.f) { if (a.os && a.off_os >= 0) a.os[p] = softplus_t<T>(th[a.off_os]); }
                else a.noise[p] = softplus_t<T>(th[a.off_noise]) + a.noise_floor;
            }
        }
    } else if (blk < a.tb + 2 + a.aux_blocks) {
        // (2 048 entries per block and trip, eight per thread, loaded before any is stored: the members are not __restrict__, and as a
        //  load-store loop every entry of a thread was a memory round trip of its own -- most of this launch's 15 us at PACOH-VI's
        //  launcher shape, 10 x 6 566 noise entries in 33 blocks)
        const int nb = a.aux_blocks;
        const T* __restrict__ src = a.aux_all + row * a.n_aux;
        T* __restrict__ dst = a.aux_out;
        for (long q0 = (long)(blk - a.tb - 2) * (8 * NT); q0 < a.n_aux; q0 += (long)nb * (8 * NT)) {
            T v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const long q = q0 + threadIdx.x + NT * u; v[u] = src[q < a.n_aux ? q : a.n_aux - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const long q = q0 + threadIdx.x + NT * u; if (q < a.n_aux) dst[q] = v[u]; }
        }
    } else if (a.vi_post == nullptr || blk < a.tb + 2 + a.aux_blocks + (a.sv_X ? a.sv_P * a.sv_P : 0)) {
        // the particles do not change before the step's update: their distance matrix (and the snapshot the in-place update reads)
        // can be had here, a launch earlier and off the path behind the all-reduce
        svgd_dist_block<T>(a.sv_X, a.sv_d2, a.sv_P, a.sv_D, a.sv_snap, blk - (a.tb + 2 + a.aux_blocks));
    } else {
        __shared__ T red[NT / 64];                               // (arithmetic and summation order of vi_sample_kernel<T, NT> / hyper_fwd_kernel)
        const int s_ = blk - (a.tb + 2 + a.aux_blocks + (a.sv_X ? a.sv_P * a.sv_P : 0)), D = a.vi_D;
        const T* eps = a.aux_all + row * a.n_aux + (long)s_ * D;
        const T HALF_LOG2PI = T(0.9189385332046727);
        T acc = 0;
        // (sixteen entries per thread and trip, all their loads requested before the first is used: as one entry per trip the 26 trips
        //  of the launchers' D = 6 566 were 26 dependent memory round trips -- 15 us for a launch of ten such blocks.  With that gone,
        //  5.5 of the launch's 10.5 us were still these blocks' own work at 256 threads, 26 entries with an exp each per thread:
        //  long rows get 1 024 threads -- vi_sample_nt)
        constexpr int U = NT >= 1024 ? 8 : 16;
        for (int d0 = threadIdx.x; d0 < D; d0 += NT * U) {
            T e[U], sc[U], lc[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int d = d0 + NT * u, dc = d < D ? d : D - 1;
                e[u] = eps[dc]; sc[u] = a.vi_post[D + dc]; lc[u] = a.vi_post[dc];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int d = d0 + NT * u;
                if (d >= D) break;
                const T th = lc[u] + t_exp<T>(sc[u]) * e[u];
                a.vi_theta[(long)s_ * D + d] = th;
                acc += T(-0.5) * e[u] * e[u] - sc[u] - HALF_LOG2PI;
                if (a.ls) {
                    if (d == a.off_noise) a.noise[s_] = softplus_t<T>(th) + a.noise_floor;
                    else if (a.os && d == a.off_os) a.os[s_] = softplus_t<T>(th);
                    else if (a.tie) { if (d == a.off_ls) { const T v1 = softplus_t<T>(th); for (int e2 = 0; e2 < a.f; ++e2) a.ls[s_ * a.f + e2] = v1; } }
                    else if (d >= a.off_ls && d < a.off_ls + a.f) a.ls[s_ * a.f + (d - a.off_ls)] = softplus_t<T>(th);
                }
            }
        }
        const T tot = block_sum_<T, NT>(acc, red);
        if (threadIdx.x == 0) a.vi_logq[s_] = tot;
    }
}

template <typename T>
__global__ void scale_dev_kernel(T* __restrict__ buf, const T* __restrict__ sc, long count) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < count) buf[q] *= sc[0];
}

}  // namespace pacoh

using namespace pacoh;

extern "C" int pacoh_step_select(const int64_t* idx_all, int tb, const void* sc_all, int n_sc, const void* aux_all, long n_aux,
                                 int64_t* counter, int64_t* idx_out, void* sc_out, void* aux_out, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!counter || tb < 0 || n_sc < 0 || n_aux < 0 || (tb > 0 && (!idx_all || !idx_out)) || (n_sc > 0 && (!sc_all || !sc_out)) ||
        (n_aux > 0 && (!aux_all || !aux_out))) return PACOH_EINVAL;
    long ab = n_aux > 0 ? (n_aux + 2047) / 2048 : 0;
    if (ab > 256) ab = 256;
    const dim3 grid((unsigned)(1 + ab));
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(step_select_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const long*)idx_all, tb, (const float*)sc_all,
                           n_sc, (const float*)aux_all, n_aux, (long*)counter, (long*)idx_out, (float*)sc_out, (float*)aux_out);
    else
        hipLaunchKernelGGL(step_select_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (const long*)idx_all, tb, (const double*)sc_all,
                           n_sc, (const double*)aux_all, n_aux, (long*)counter, (long*)idx_out, (double*)sc_out, (double*)aux_out);
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long*)counter);
    return launch_status();
}

template <typename T>
static int step_begin_launch(const int64_t* idx_all, int tb, const void* sc_all, int n_sc, const void* aux_all, long n_aux, int64_t* counter,
                             int32_t* ticket, void* sc_out, void* aux_out, const void* x, const void* y, const int32_t* n_valid, void* out_x,
                             void* out_y, int32_t* out_n_valid, int n, int d, const void* theta, long theta_stride, int P, int off_ls, int f,
                             int off_os, int off_noise, double noise_floor, void* ls, void* os, void* noise, int advance,
                             const void* svgd_X, void* svgd_workspace, int svgd_P, int svgd_D, hipStream_t s,
                             const void* vi_post = nullptr, void* vi_theta = nullptr, void* vi_logq = nullptr, int vi_S = 0, int vi_D = 0) {
    long ab = n_aux > 0 ? (n_aux + 2047) / 2048 : 0;
    if (ab > 256) ab = 256;
    T* d2 = (T*)svgd_workspace;                       // (layout of pacoh_svgd_update_dev_workspace_bytes: distances | snapshot | median pair)
    StepBeginArgs<T> a = {(const long*)idx_all, tb, (const T*)sc_all, n_sc, (const T*)aux_all, n_aux, (long*)counter, ticket, (T*)sc_out,
                          (T*)aux_out, (const T*)x, (const T*)y, n_valid, (T*)out_x, (T*)out_y, out_n_valid, n * d, n, (const T*)theta,
                          theta_stride, P, off_ls, features_of(f), off_os, off_noise, (T)noise_floor, (T*)ls, (T*)os, (T*)noise,
                          (int)ab, (const T*)svgd_X, d2, svgd_X ? d2 + svgd_P * svgd_P : nullptr, svgd_P, svgd_D,
                          kernel_of(f) != PACOH_KERNEL_RBF, (const T*)vi_post, (T*)vi_theta, (T*)vi_logq, vi_S, vi_D};
    const long sb = svgd_X ? (long)svgd_P * svgd_P : 0;
    const dim3 grid((unsigned)(tb + 2 + ab + sb + (vi_post ? vi_S : 0)));
    if (vi_post && vi_sample_nt(vi_D) == 1024) hipLaunchKernelGGL((step_begin_kernel<T, 1024>), grid, dim3(1024), 0, s, a);
    else hipLaunchKernelGGL((step_begin_kernel<T, 256>), grid, dim3(256), 0, s, a);
    if (advance) hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(1), 0, s, (long*)counter);
    return launch_status();
}

extern "C" int pacoh_step_begin(const int64_t* idx_all, int tb, const void* sc_all, int n_sc, const void* aux_all, long n_aux,
                                int64_t* counter, int32_t* ticket, void* sc_out, void* aux_out,
                                const void* x, const void* y, const int32_t* n_valid, void* out_x, void* out_y, int32_t* out_n_valid, int n, int d,
                                const void* theta, long theta_stride, int P, int off_ls, int f, int off_os, int off_noise, double noise_floor,
                                void* ls, void* os, void* noise, int advance, const void* svgd_X, void* svgd_workspace, int svgd_P, int svgd_D,
                                int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!counter || !ticket || tb < 0 || n_sc < 0 || n_aux < 0 || (n_sc > 0 && (!sc_all || !sc_out)) || (n_aux > 0 && (!aux_all || !aux_out)))
        return PACOH_EINVAL;
    if (tb > 0 && (!idx_all || !x || !y || !out_x || !out_y || n <= 0 || d <= 0 || (n_valid == nullptr) != (out_n_valid == nullptr))) return PACOH_EINVAL;
    if (theta && (!ls || !noise || P <= 0 || features_of(f) <= 0)) return PACOH_EINVAL;
    if (svgd_X && (!svgd_workspace || svgd_P <= 0 || svgd_D <= 0)) return PACOH_EINVAL;
    if (svgd_X && svgd_P > PACOH_SVGD_MAX_PARTICLES) return PACOH_ELIMIT;
    if (dtype == PACOH_F32)
        return step_begin_launch<float>(idx_all, tb, sc_all, n_sc, aux_all, n_aux, counter, ticket, sc_out, aux_out, x, y, n_valid, out_x, out_y,
                                        out_n_valid, n, d, theta, theta_stride, P, off_ls, f, off_os, off_noise, noise_floor, ls, os, noise,
                                        advance, svgd_X, svgd_workspace, svgd_P, svgd_D, (hipStream_t)stream);
    return step_begin_launch<double>(idx_all, tb, sc_all, n_sc, aux_all, n_aux, counter, ticket, sc_out, aux_out, x, y, n_valid, out_x, out_y,
                                     out_n_valid, n, d, theta, theta_stride, P, off_ls, f, off_os, off_noise, noise_floor, ls, os, noise,
                                     advance, svgd_X, svgd_workspace, svgd_P, svgd_D, (hipStream_t)stream);
}

// pacoh_step_begin for a PACOH-VI step with a diagonal posterior: additionally draws the step's S samples theta[S, D] = loc +
// exp(scale) * eps from posterior[2, D] and the step's noise row (n_aux = S * D), their log q[S], and the samples' transformed
// hyper-parameters ls[S, f] / os[S] / noise[S] -- pacoh_vi_sample and pacoh_hyper_fwd in extra workgroups of the same launch
extern "C" int pacoh_step_begin_vi(const int64_t* idx_all, int tb, const void* sc_all, int n_sc, const void* aux_all, long n_aux,
                                   int64_t* counter, int32_t* ticket, void* sc_out, void* aux_out,
                                   const void* x, const void* y, const int32_t* n_valid, void* out_x, void* out_y, int32_t* out_n_valid, int n, int d,
                                   const void* posterior, int S, int D, void* theta_out, void* log_q_out,
                                   int off_ls, int f, int off_os, int off_noise, double noise_floor, void* ls, void* os, void* noise,
                                   int advance, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!counter || !ticket || tb < 0 || n_sc <= 0 || !sc_all || !sc_out || !aux_all || !aux_out) return PACOH_EINVAL;
    if (!posterior || !theta_out || !log_q_out || S <= 0 || D <= 0 || n_aux != (long)S * D) return PACOH_EINVAL;
    if (tb > 0 && (!idx_all || !x || !y || !out_x || !out_y || n <= 0 || d <= 0 || (n_valid == nullptr) != (out_n_valid == nullptr))) return PACOH_EINVAL;
    if (ls && (!noise || features_of(f) <= 0 || off_ls < 0 || off_noise < 0 || off_noise >= D || off_os >= D)) return PACOH_EINVAL;
    if (dtype == PACOH_F32)
        return step_begin_launch<float>(idx_all, tb, sc_all, n_sc, aux_all, n_aux, counter, ticket, sc_out, aux_out, x, y, n_valid, out_x, out_y,
                                        out_n_valid, n, d, nullptr, 0, S, off_ls, f, off_os, off_noise, noise_floor, ls, os, noise,
                                        advance, nullptr, nullptr, 0, 0, (hipStream_t)stream, posterior, theta_out, log_q_out, S, D);
    return step_begin_launch<double>(idx_all, tb, sc_all, n_sc, aux_all, n_aux, counter, ticket, sc_out, aux_out, x, y, n_valid, out_x, out_y,
                                     out_n_valid, n, d, nullptr, 0, S, off_ls, f, off_os, off_noise, noise_floor, ls, os, noise,
                                     advance, nullptr, nullptr, 0, 0, (hipStream_t)stream, posterior, theta_out, log_q_out, S, D);
}

extern "C" int pacoh_scale_dev(void* buf, const void* scalar, long count, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!buf || !scalar || count <= 0) return PACOH_EINVAL;
    const unsigned blocks = (unsigned)((count + 255) / 256);
    if (dtype == PACOH_F32) hipLaunchKernelGGL(scale_dev_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)buf, (const float*)scalar, count);
    else hipLaunchKernelGGL(scale_dev_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (double*)buf, (const double*)scalar, count);
    return launch_status();
}

extern "C" int pacoh_gather_tasks(const void* x, const void* y, const int32_t* n_valid, const int64_t* idx, void* out_x,
                                  void* out_y, int32_t* out_n_valid, int Tb, int n, int d, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!x || !y || !idx || !out_x || !out_y || Tb <= 0 || n <= 0 || d <= 0) return PACOH_EINVAL;
    if ((n_valid == nullptr) != (out_n_valid == nullptr)) return PACOH_EINVAL;
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(gather_tasks_kernel<float>, dim3(Tb), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)y,
                           n_valid, (const long*)idx, (float*)out_x, (float*)out_y, out_n_valid, n * d, n);
    else
        hipLaunchKernelGGL(gather_tasks_kernel<double>, dim3(Tb), dim3(256), 0, (hipStream_t)stream, (const double*)x, (const double*)y,
                           n_valid, (const long*)idx, (double*)out_x, (double*)out_y, out_n_valid, n * d, n);
    return launch_status();
}

extern "C" int pacoh_abi_version(void) { return 14; }
extern "C" void pacoh_reload_env(void) { read_switches(g_sw); }

extern "C" int pacoh_hyper_fwd(const void* theta, long theta_stride, int P, int off_ls, int f, int off_os, int off_noise,
                               double noise_floor, void* ls, void* os, void* noise, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    const int tie = kernel_of(f) != PACOH_KERNEL_RBF;
    f = features_of(f);
    if (!theta || !ls || !noise || P <= 0 || f <= 0 || off_ls < 0 || off_noise < 0) return PACOH_EINVAL;
    unsigned blocks = (unsigned)((P * (f + 2) + 255) / 256);
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(hyper_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)theta, theta_stride,
                           P, off_ls, f, off_os, off_noise, (float)noise_floor, (float*)ls, (float*)os, (float*)noise, tie);
    else
        hipLaunchKernelGGL(hyper_fwd_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double*)theta, theta_stride,
                           P, off_ls, f, off_os, off_noise, noise_floor, (double*)ls, (double*)os, (double*)noise, tie);
    return launch_status();
}

// svgd_workspace (optional; pacoh_svgd_update_dev_workspace_bytes(svgd_P, svgd_D)): one more workgroup of the launch computes the
// median-heuristic bandwidth from the distance matrix at the head of that workspace into its bandwidth slot (P <= 64)
extern "C" int pacoh_hyper_bwd(const void* theta, long theta_stride, int P, int T_, int off_ls, int f, int off_os, int off_noise,
                               int off_const, const void* d_ls, const void* d_os, const void* d_noise, const void* d_const,
                               void* grad, long grad_stride, const void* lml, void* lik, double lik_scale,
                               const int32_t* info, int32_t* fail_flag, void* svgd_workspace, int svgd_P, int svgd_D,
                               const pacoh_adam_inline* opt, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (opt && (!opt->param || !opt->exp_avg || !opt->exp_avg_sq || !opt->scalars || opt->n_seg < 1 || opt->n_seg > 4 || P != 1 || !lml))
        return PACOH_EINVAL;
    if (opt && opt->next) return PACOH_ELIMIT;         // (the pipelined feed's counter is advanced by the fused MLP backward launch)
    const int tie = kernel_of(f) != PACOH_KERNEL_RBF;
    f = features_of(f);
    if (!theta || !grad || !d_ls || !d_noise || P <= 0 || T_ <= 0 || f <= 0) return PACOH_EINVAL;
    if ((lml == nullptr) != (lik == nullptr)) return PACOH_EINVAL;
    if (svgd_workspace && (svgd_P <= 0 || svgd_D <= 0)) return PACOH_EINVAL;
    if (svgd_workspace && svgd_P > 64) return PACOH_ELIMIT;
    if (dtype == PACOH_F32) {
        float* ws = (float*)svgd_workspace;
        HyperBwdArgs<float> a = {(const float*)theta, theta_stride, P, T_, off_ls, f, off_os, off_noise, off_const, (const float*)d_ls,
                                 (const float*)d_os, (const float*)d_noise, (const float*)d_const, (float*)grad, grad_stride,
                                 (const float*)lml, (float*)lik, (float)lik_scale, info, fail_flag, tie,
                                 ws, svgd_P, ws ? ws + svgd_bw_slot(svgd_P, svgd_D) : nullptr, AdamInline<float>{}, StepNextArgs<float>{}};
        if (opt) {
            a.opt = {(float*)opt->param, (float*)opt->exp_avg, (float*)opt->exp_avg_sq, (const float*)opt->scalars, (float)(1.0 - opt->beta1),
                     (float)opt->beta2, (float)(1.0 - opt->beta2), opt->n_seg, {0, 0, 0, 0}, {0, 0, 0, 0}, (long*)opt->step_counter, (float*)opt->loss_cum,
                     nullptr, nullptr, 0};
            for (int k = 0; k < opt->n_seg; ++k) { a.opt.lo[k] = opt->seg_lo[k]; a.opt.hi[k] = opt->seg_hi[k]; }
        }
        hipLaunchKernelGGL(hyper_bwd_kernel<float>, dim3((unsigned)hyper_tail_blocks(a)), dim3(256), 0, (hipStream_t)stream, a);
    } else {
        double* ws = (double*)svgd_workspace;
        HyperBwdArgs<double> a = {(const double*)theta, theta_stride, P, T_, off_ls, f, off_os, off_noise, off_const, (const double*)d_ls,
                                  (const double*)d_os, (const double*)d_noise, (const double*)d_const, (double*)grad, grad_stride,
                                  (const double*)lml, (double*)lik, lik_scale, info, fail_flag, tie,
                                  ws, svgd_P, ws ? ws + svgd_bw_slot(svgd_P, svgd_D) : nullptr, AdamInline<double>{}, StepNextArgs<double>{}};
        if (opt) {
            a.opt = {(double*)opt->param, (double*)opt->exp_avg, (double*)opt->exp_avg_sq, (const double*)opt->scalars, 1.0 - opt->beta1,
                     opt->beta2, 1.0 - opt->beta2, opt->n_seg, {0, 0, 0, 0}, {0, 0, 0, 0}, (long*)opt->step_counter, (double*)opt->loss_cum,
                     nullptr, nullptr, 0};
            for (int k = 0; k < opt->n_seg; ++k) { a.opt.lo[k] = opt->seg_lo[k]; a.opt.hi[k] = opt->seg_hi[k]; }
        }
        hipLaunchKernelGGL(hyper_bwd_kernel<double>, dim3((unsigned)hyper_tail_blocks(a)), dim3(256), 0, (hipStream_t)stream, a);
    }
    return launch_status();
}

extern "C" int pacoh_softplus_fwd(const void* raw, void* out, double floor_, long count, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!raw || !out || count <= 0) return PACOH_EINVAL;
    unsigned blocks = (unsigned)((count + 255) / 256);
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(softplus_fwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)raw, (float*)out, (float)floor_, count);
    else
        hipLaunchKernelGGL(softplus_fwd_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double*)raw, (double*)out, floor_, count);
    return launch_status();
}

extern "C" int pacoh_softplus_bwd(const void* raw, const void* g, void* d_raw, int accumulate, long count, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!raw || !g || !d_raw || count <= 0) return PACOH_EINVAL;
    unsigned blocks = (unsigned)((count + 255) / 256);
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(softplus_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)raw, (const float*)g, (float*)d_raw, accumulate, count);
    else
        hipLaunchKernelGGL(softplus_bwd_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double*)raw, (const double*)g, (double*)d_raw, accumulate, count);
    return launch_status();
}

extern "C" int pacoh_prior_logprob_grad(const void* theta, const void* prior_mean, const void* prior_std,
                                        void* logp, void* grad, double grad_scale, int P, int D, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!theta || !prior_mean || !prior_std || P <= 0 || D <= 0) return PACOH_EINVAL;
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(prior_kernel<float>, dim3(P), dim3(1024), 0, (hipStream_t)stream, (const float*)theta,
                           (const float*)prior_mean, (const float*)prior_std, (float*)logp, (float*)grad, (float)grad_scale, D);
    else
        hipLaunchKernelGGL(prior_kernel<double>, dim3(P), dim3(1024), 0, (hipStream_t)stream, (const double*)theta,
                           (const double*)prior_mean, (const double*)prior_std, (double*)logp, (double*)grad, grad_scale, D);
    return launch_status();
}

extern "C" int pacoh_prior_score_dev(const void* theta, const void* prior_mean, const void* prior_std, void* score,
                                     double prior_factor, const void* score_scale, int P, int D, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!theta || !prior_mean || !prior_std || !score || !score_scale || P <= 0 || D <= 0) return PACOH_EINVAL;
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(prior_kernel<float>, dim3(P), dim3(1024), 0, (hipStream_t)stream, (const float*)theta,
                           (const float*)prior_mean, (const float*)prior_std, (float*)nullptr, (float*)score, (float)prior_factor, D,
                           (const float*)score_scale);
    else
        hipLaunchKernelGGL(prior_kernel<double>, dim3(P), dim3(1024), 0, (hipStream_t)stream, (const double*)theta,
                           (const double*)prior_mean, (const double*)prior_std, (double*)nullptr, (double*)score, prior_factor, D,
                           (const double*)score_scale);
    return launch_status();
}

extern "C" size_t pacoh_svgd_workspace_bytes(int P, int D, int dtype) {
    (void)D;
    if (P <= 0) return 0;
    return (size_t)(2 * P * P + P + 8) * (dtype == PACOH_F64 ? 8 : 4);
}

extern "C" size_t pacoh_svgd_update_dev_workspace_bytes(int P, int D, int dtype) {
    if (P <= 0 || D <= 0) return 0;
    return (size_t)(P * P + (long)P * D + 4) * (dtype == PACOH_F64 ? 8 : 4);      // distances + snapshot of the particles + median pair + bandwidth (svgd_bw_slot)
}

template <typename T>
static int svgd_update_dev_launch(void* X, const void* score, const void* mu, const void* sd, double prior_factor, double bandwidth,
                                  int use_adam, const void* scalars, double beta1, double beta2, void* m, void* v, void* bw_out,
                                  void* workspace, int P, int D, int dist_done, int64_t* step_counter, hipStream_t s) {
    T* d2 = (T*)workspace;
    T* snap = d2 + P * P;
    T* mids = nullptr;
    if (!dist_done) hipLaunchKernelGGL(svgd_dist_kernel<T>, dim3(P * P), dim3(256), 0, s, (const T*)X, d2, P, D, snap);
    if (P > 64 && !(bandwidth > 0.0)) {
        mids = snap + (long)P * D;
        hipLaunchKernelGGL(svgd_median_large_kernel<T>, dim3(2), dim3(1024), 0, s, (const T*)d2, P, mids);
    }
    hipLaunchKernelGGL(svgd_update_kernel<T>, dim3((D + 255) / 256, P), dim3(256), 0, s, (const T*)snap, (const T*)score, (const T*)mu,
                       (const T*)sd, (T)prior_factor, (const T*)d2, (T)bandwidth, (T*)bw_out, use_adam, T(0),
                       (T)(1.0 - beta1), (T)beta2, (T)(1.0 - beta2), T(0), T(1), T(0), (T*)m, (T*)v, (T*)X, P, D, (const T*)scalars,
                       (const T*)mids, (long*)step_counter);
    return launch_status();
}

extern "C" int pacoh_svgd_update_dev(void* X, const void* score, const void* prior_mean, const void* prior_std,
                                     double prior_factor, double bandwidth, int use_adam, const void* scalars, double beta1,
                                     double beta2, void* exp_avg, void* exp_avg_sq, void* bw_out, void* workspace, int P, int D,
                                     int dist_done, int64_t* step_counter, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!X || !score || !scalars || !workspace || P <= 0 || D <= 0) return PACOH_EINVAL;
    if ((prior_mean == nullptr) != (prior_std == nullptr)) return PACOH_EINVAL;
    if (use_adam && (!exp_avg || !exp_avg_sq)) return PACOH_EINVAL;
    if (P > PACOH_SVGD_MAX_PARTICLES) return PACOH_ELIMIT;
    if (dtype == PACOH_F32)
        return svgd_update_dev_launch<float>(X, score, prior_mean, prior_std, prior_factor, bandwidth, use_adam, scalars, beta1, beta2,
                                             exp_avg, exp_avg_sq, bw_out, workspace, P, D, dist_done, step_counter, (hipStream_t)stream);
    return svgd_update_dev_launch<double>(X, score, prior_mean, prior_std, prior_factor, bandwidth, use_adam, scalars, beta1, beta2,
                                          exp_avg, exp_avg_sq, bw_out, workspace, P, D, dist_done, step_counter, (hipStream_t)stream);
}

extern "C" int pacoh_svgd_dist_advance(const void* X, void* workspace, int P, int D, int64_t* counter, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!X || !workspace || P <= 0 || D <= 0) return PACOH_EINVAL;
    if (P > PACOH_SVGD_MAX_PARTICLES) return PACOH_ELIMIT;
    if (dtype == PACOH_F32) {
        float* d2 = (float*)workspace;
        SvgdDistTail<float> t = {(const float*)X, d2, d2 + P * P, P, D, (long*)counter};
        hipLaunchKernelGGL(svgd_dist_advance_kernel<float>, dim3(P * P), dim3(256), 0, (hipStream_t)stream, t);
    } else {
        double* d2 = (double*)workspace;
        SvgdDistTail<double> t = {(const double*)X, d2, d2 + P * P, P, D, (long*)counter};
        hipLaunchKernelGGL(svgd_dist_advance_kernel<double>, dim3(P * P), dim3(256), 0, (hipStream_t)stream, t);
    }
    return launch_status();
}

template <typename T>
static int svgd_update_next_launch(void* X, const void* score, const void* mu, const void* sd, double prior_factor, double bandwidth,
                                   int use_adam, double beta1, double beta2, void* m, void* v, void* bw_out, void* workspace, int P, int D,
                                   const StepNextArgs<T>& nx, hipStream_t s) {
    T* d2 = (T*)workspace;
    T* snap = d2 + P * P;
    T* mids = nullptr;
    if (P > 64 && !(bandwidth > 0.0)) {
        mids = snap + (long)P * D;
        hipLaunchKernelGGL(svgd_median_large_kernel<T>, dim3(2), dim3(1024), 0, s, (const T*)d2, P, mids);
    }
    const int gx = (D + 255) / 256;
    const int tail_rows = (nx.tb + 1 + gx - 1) / gx;                 // tb gathers + the scalars' row
    hipLaunchKernelGGL(svgd_update_kernel<T>, dim3(gx, P + tail_rows), dim3(256), 0, s, (const T*)snap, (const T*)score, (const T*)mu,
                       (const T*)sd, (T)prior_factor, (const T*)d2, (T)bandwidth, (T*)bw_out, use_adam, T(0),
                       (T)(1.0 - beta1), (T)beta2, (T)(1.0 - beta2), T(0), T(1), T(0), (T*)m, (T*)v, (T*)X, P, D, (const T*)nullptr,
                       (const T*)mids, (long*)nullptr, nx);
    return launch_status();
}

// pacoh_svgd_update_dev for the pipelined step (step_tail.h): distances already in the workspace, scalars from the ping-pong rows
// the counter selects, the updated particles' transformed hyper-parameters written by the update itself, and -- in extra workgroups
// of the same launch -- the NEXT step's scalars and task batch fetched
extern "C" int pacoh_svgd_update_next(void* X, const void* score, const void* prior_mean, const void* prior_std, double prior_factor,
                                      double bandwidth, int use_adam, double beta1, double beta2, void* exp_avg, void* exp_avg_sq,
                                      void* bw_out, void* workspace, int P, int D,
                                      const int64_t* counter, void* sc2, int n_sc, const int64_t* idx_all, int tb, const void* sc_all,
                                      const void* x, const void* y, const int32_t* n_valid, void* out_x, void* out_y,
                                      int32_t* out_n_valid, int n, int d,
                                      int off_ls, int f, int off_os, int off_noise, double noise_floor, void* ls, void* os, void* noise,
                                      int bandwidth_ready, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (bandwidth_ready && P > 64) return PACOH_ELIMIT;
    if (!X || !score || !workspace || P <= 0 || D <= 0 || !counter || !sc2 || !sc_all || n_sc < PACOH_SC_COUNT) return PACOH_EINVAL;
    if ((prior_mean == nullptr) != (prior_std == nullptr)) return PACOH_EINVAL;
    if (use_adam && (!exp_avg || !exp_avg_sq)) return PACOH_EINVAL;
    if (tb < 0 || (tb > 0 && (!idx_all || !x || !y || !out_x || !out_y || n <= 0 || d <= 0 || (n_valid == nullptr) != (out_n_valid == nullptr))))
        return PACOH_EINVAL;
    const int fdim = features_of(f);
    if (ls && (!noise || fdim <= 0 || off_ls < 0 || off_noise < 0 || off_noise >= D || off_ls + (kernel_of(f) != PACOH_KERNEL_RBF ? 1 : fdim) > D ||
               off_os >= D)) return PACOH_EINVAL;
    if (P > PACOH_SVGD_MAX_PARTICLES) return PACOH_ELIMIT;
    if (dtype == PACOH_F32) {
        StepNextArgs<float> nx = {(const long*)counter, (float*)sc2, n_sc, (const long*)idx_all, tb, (const float*)sc_all, (const float*)x,
                                  (const float*)y, n_valid, (float*)out_x, (float*)out_y, out_n_valid, n * d, n, off_ls, fdim, off_os, off_noise,
                                  kernel_of(f) != PACOH_KERNEL_RBF, (float)noise_floor, (float*)ls, (float*)os, (float*)noise,
                                  bandwidth_ready ? (const float*)workspace + svgd_bw_slot(P, D) : nullptr};
        return svgd_update_next_launch<float>(X, score, prior_mean, prior_std, prior_factor, bandwidth, use_adam, beta1, beta2, exp_avg,
                                              exp_avg_sq, bw_out, workspace, P, D, nx, (hipStream_t)stream);
    }
    StepNextArgs<double> nx = {(const long*)counter, (double*)sc2, n_sc, (const long*)idx_all, tb, (const double*)sc_all, (const double*)x,
                               (const double*)y, n_valid, (double*)out_x, (double*)out_y, out_n_valid, n * d, n, off_ls, fdim, off_os, off_noise,
                               kernel_of(f) != PACOH_KERNEL_RBF, noise_floor, (double*)ls, (double*)os, (double*)noise,
                               bandwidth_ready ? (const double*)workspace + svgd_bw_slot(P, D) : nullptr};
    return svgd_update_next_launch<double>(X, score, prior_mean, prior_std, prior_factor, bandwidth, use_adam, beta1, beta2, exp_avg,
                                           exp_avg_sq, bw_out, workspace, P, D, nx, (hipStream_t)stream);
}

template <typename T>
static int svgd_launch(const void* X, const void* score, double bandwidth, int neg, void* phi, void* bw_out,
                       void* workspace, int P, int D, hipStream_t s) {
    T* d2 = (T*)workspace;
    T* Kmat = d2 + P * P;
    T* rowsum = Kmat + P * P;
    T* gamma = rowsum + P;
    T* mids = nullptr;                                // (the workspace has 8 spare elements behind rowsum: [0] gamma, [2..3] median pair)
    hipLaunchKernelGGL(svgd_dist_kernel<T>, dim3(P * P), dim3(256), 0, s, (const T*)X, d2, P, D);
    if (P > 64 && !(bandwidth > 0.0)) {
        mids = gamma + 2;
        hipLaunchKernelGGL(svgd_median_large_kernel<T>, dim3(2), dim3(1024), 0, s, (const T*)d2, P, mids);
    }
    hipLaunchKernelGGL(svgd_kmat_kernel<T>, dim3(1), dim3(256), 0, s, (const T*)d2, (T)bandwidth, Kmat,
                       rowsum, gamma, (T*)bw_out, P, (const T*)mids);
    hipLaunchKernelGGL(svgd_phi_kernel<T>, dim3((D + 255) / 256, P), dim3(256), 0, s,
                       (const T*)X, (const T*)score, (const T*)Kmat, (const T*)rowsum, (const T*)gamma, neg, (T*)phi, P, D);
    return launch_status();
}

extern "C" int pacoh_svgd_phi(const void* X, const void* score, double bandwidth, int neg, void* phi,
                              void* bw_out, void* workspace, int P, int D, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!X || !score || !phi || !workspace || P <= 0 || D <= 0) return PACOH_EINVAL;
    if (P > PACOH_SVGD_MAX_PARTICLES) return PACOH_ELIMIT;
    if (dtype == PACOH_F32) return svgd_launch<float>(X, score, bandwidth, neg, phi, bw_out, workspace, P, D, (hipStream_t)stream);
    return svgd_launch<double>(X, score, bandwidth, neg, phi, bw_out, workspace, P, D, (hipStream_t)stream);
}

template <typename T>
static int svgd_update_launch(const void* X, const void* score, const void* mu, const void* sd, double prior_factor, double bandwidth,
                              int use_adam, double lr, double beta1, double beta2, double eps, long step, void* m, void* v,
                              void* X_out, void* bw_out, void* workspace, int P, int D, hipStream_t s) {
    T* d2 = (T*)workspace;
    T* mids = nullptr;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipLaunchKernelGGL(svgd_dist_kernel<T>, dim3(P * P), dim3(256), 0, s, (const T*)X, d2, P, D);
    if (P > 64 && !(bandwidth > 0.0)) {
        mids = d2 + 2 * P * P + P + 2;                // (pacoh_svgd_workspace_bytes: same slots as in pacoh_svgd_phi)
        hipLaunchKernelGGL(svgd_median_large_kernel<T>, dim3(2), dim3(1024), 0, s, (const T*)d2, P, mids);
    }
    hipLaunchKernelGGL(svgd_update_kernel<T>, dim3((D + 255) / 256, P), dim3(256), 0, s, (const T*)X, (const T*)score, (const T*)mu,
                       (const T*)sd, (T)prior_factor, (const T*)d2, (T)bandwidth, (T*)bw_out, use_adam, (T)lr,
                       (T)(1.0 - beta1), (T)beta2, (T)(1.0 - beta2), (T)(lr / bc1), (T)sqrt(bc2), (T)eps, (T*)m, (T*)v, (T*)X_out, P, D,
                       (const T*)nullptr, (const T*)mids);
    return launch_status();
}

extern "C" int pacoh_svgd_update(const void* X, const void* score, const void* prior_mean, const void* prior_std,
                                 double prior_factor, double bandwidth, int use_adam, double lr, double beta1, double beta2,
                                 double eps, long step, void* exp_avg, void* exp_avg_sq, void* X_out, void* bw_out,
                                 void* workspace, int P, int D, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!X || !score || !X_out || X_out == X || !workspace || P <= 0 || D <= 0) return PACOH_EINVAL;
    if ((prior_mean == nullptr) != (prior_std == nullptr)) return PACOH_EINVAL;
    if (use_adam && (!exp_avg || !exp_avg_sq || step <= 0)) return PACOH_EINVAL;
    if (P > PACOH_SVGD_MAX_PARTICLES) return PACOH_ELIMIT;
    if (dtype == PACOH_F32)
        return svgd_update_launch<float>(X, score, prior_mean, prior_std, prior_factor, bandwidth, use_adam, lr, beta1, beta2, eps, step,
                                         exp_avg, exp_avg_sq, X_out, bw_out, workspace, P, D, (hipStream_t)stream);
    return svgd_update_launch<double>(X, score, prior_mean, prior_std, prior_factor, bandwidth, use_adam, lr, beta1, beta2, eps, step,
                                      exp_avg, exp_avg_sq, X_out, bw_out, workspace, P, D, (hipStream_t)stream);
}

extern "C" int pacoh_adam_step(void* param, const void* grad, void* exp_avg, void* exp_avg_sq,
                               double lr, double beta1, double beta2, double eps, double weight_decay,
                               long step, long count, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!param || !grad || !exp_avg || !exp_avg_sq || count <= 0 || step <= 0) return PACOH_EINVAL;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const double step_size = lr / bc1;
    const double bc2_sqrt = sqrt(bc2);
    const double decay_mul = 1.0 - lr * weight_decay;
    unsigned blocks = (unsigned)((count + 255) / 256);
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(adam_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)param, (const float*)grad,
                           (float*)exp_avg, (float*)exp_avg_sq, (float)decay_mul, (float)(1.0 - beta1), (float)beta2,
                           (float)(1.0 - beta2), (float)step_size, (float)bc2_sqrt, (float)eps, count);
    else
        hipLaunchKernelGGL(adam_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (double*)param, (const double*)grad,
                           (double*)exp_avg, (double*)exp_avg_sq, decay_mul, 1.0 - beta1, beta2, 1.0 - beta2, step_size, bc2_sqrt, eps, count);
    return launch_status();
}

extern "C" int pacoh_adam_step_dev(void* param, const void* grad, void* exp_avg, void* exp_avg_sq, const void* scalars,
                                   double beta1, double beta2, long count, int64_t* step_counter, void* loss_cum, const void* loss,
                                   int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !scalars || count <= 0 || (loss_cum && !loss)) return PACOH_EINVAL;
    unsigned blocks = (unsigned)((count + 255) / 256);
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(adam_dev_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)param, (const float*)grad,
                           (float*)exp_avg, (float*)exp_avg_sq, (const float*)scalars, (float)(1.0 - beta1), (float)beta2,
                           (float)(1.0 - beta2), count, (long*)step_counter, (float*)loss_cum, (const float*)loss);
    else
        hipLaunchKernelGGL(adam_dev_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (double*)param, (const double*)grad,
                           (double*)exp_avg, (double*)exp_avg_sq, (const double*)scalars, 1.0 - beta1, beta2, 1.0 - beta2, count,
                           (long*)step_counter, (double*)loss_cum, (const double*)loss);
    return launch_status();
}

extern "C" int pacoh_vi_sample(const void* posterior, const void* eps, void* theta, void* log_q, int S, int D, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!posterior || !eps || !theta || !log_q || S <= 0 || D <= 0) return PACOH_EINVAL;
#define PACOH_VI_SAMPLE(T_, NT_) hipLaunchKernelGGL((vi_sample_kernel<T_, NT_>), dim3(S), dim3(NT_), 0, (hipStream_t)stream, (const T_*)posterior, \
                                                  (const T_*)eps, (T_*)theta, (T_*)log_q, D)
    if (vi_sample_nt(D) == 1024) { if (dtype == PACOH_F32) PACOH_VI_SAMPLE(float, 1024); else PACOH_VI_SAMPLE(double, 1024); }
    else { if (dtype == PACOH_F32) PACOH_VI_SAMPLE(float, 256); else PACOH_VI_SAMPLE(double, 256); }
#undef PACOH_VI_SAMPLE
    return launch_status();
}

extern "C" int pacoh_vi_grad(const void* posterior, const void* eps, const void* score, double prior_factor, void* grad,
                             int S, int D, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!posterior || !eps || !score || !grad || S <= 0 || D <= 0) return PACOH_EINVAL;
    unsigned blocks = (unsigned)((D + 255) / 256);
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(vi_grad_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)posterior, (const float*)eps,
                           (const float*)score, (float)prior_factor, (float*)grad, S, D);
    else
        hipLaunchKernelGGL(vi_grad_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const double*)posterior, (const double*)eps,
                           (const double*)score, prior_factor, (double*)grad, S, D);
    return launch_status();
}

extern "C" size_t pacoh_vi_update_dev_workspace_bytes(int D, int dtype) {
    return D > 0 ? 16 + (size_t)((D + 255) / 256) * (dtype == PACOH_F64 ? 8 : 4) : 0;
}

extern "C" int pacoh_vi_update_dev(void* posterior, const void* eps, const void* theta, const void* score, const void* lik,
                                   const void* log_q, const void* prior_mean, const void* prior_std, double prior_factor,
                                   const void* scalars, double beta1, double beta2, void* exp_avg, void* exp_avg_sq, void* loss_out,
                                   int64_t* step_counter, void* workspace, int S, int D, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!posterior || !eps || !theta || !score || !lik || !log_q || !prior_mean || !prior_std || !scalars || !exp_avg || !exp_avg_sq ||
        !loss_out || !workspace || S <= 0 || D <= 0) return PACOH_EINVAL;
    const unsigned blocks = (unsigned)((D + 255) / 256);
    unsigned* ticket = (unsigned*)workspace;           // [0]: ticket (zero before the first launch; every launch leaves it zero)
    void* partial = (char*)workspace + 16;             // [16 ..): one partial sum per block
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(vi_update_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)posterior, (const float*)eps,
                           (const float*)theta, (const float*)score, (const float*)lik, (const float*)log_q, (const float*)prior_mean,
                           (const float*)prior_std, (float)prior_factor, (const float*)scalars, (float)(1.0 - beta1), (float)beta2,
                           (float)(1.0 - beta2), (float*)exp_avg, (float*)exp_avg_sq, (float*)loss_out, (long*)step_counter,
                           (float*)partial, ticket, S, D);
    else
        hipLaunchKernelGGL(vi_update_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (double*)posterior, (const double*)eps,
                           (const double*)theta, (const double*)score, (const double*)lik, (const double*)log_q, (const double*)prior_mean,
                           (const double*)prior_std, prior_factor, (const double*)scalars, 1.0 - beta1, beta2, 1.0 - beta2,
                           (double*)exp_avg, (double*)exp_avg_sq, (double*)loss_out, (long*)step_counter, (double*)partial, ticket, S, D);
    return launch_status();
}

extern "C" int pacoh_axpy(void* y, const void* x, double alpha, long count, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!y || !x || count <= 0) return PACOH_EINVAL;
    unsigned blocks = (unsigned)((count + 255) / 256);
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(axpy_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float*)y, (const float*)x, (float)alpha, count);
    else
        hipLaunchKernelGGL(axpy_kernel<double>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (double*)y, (const double*)x, alpha, count);
    return launch_status();
}
