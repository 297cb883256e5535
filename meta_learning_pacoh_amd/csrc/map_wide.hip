// PACOH-MAP iteration with WIDE networks at a tiny batch (round 6): forward of the networks, GP LML + gradient and the networks'
// backward of ALL tasks of an iteration in ONE launch of one workgroup per network, for hidden widths up to 128 -- the reference's own
// PACOH-MAP launcher runs 2 tasks x 5 points per iteration through two 4 x 128 networks (experiments/meta_GPR_mll_base_exp.py:29-47).  The
// general path takes ~62 launches for that iteration (layer by layer: repack, GEMM, delta, weight gradient per layer and network:
// 0.21 ms, all of it launch latency); the task-fused kernel of round 5 (map_task.hip) keeps a network as an LDS image, which two
// 128-wide networks (400 KB) cannot be.  Here the weights stay where they are -- theta -- and stream through the matrix cores:
//   * a hidden layer is at most eight 16-unit output tiles, one per wave: the lane's row of the weight matrix as 16-byte loads along k
//     (requested a layer AHEAD, into a register set of its own), the activations of the <= 16 points of the batch from LDS as the B
//     operand under the same quad permutation of k (map_net.h), bias + tanh in the epilogue, the result back to LDS for the next layer
//     (one workgroup barrier per layer);
//   * the two networks' workgroups meet once, in front of the GP, which needs the mean AND the features: an exchange of <= 64 floats
//     through write-through atomic stores and a pair of arrival counts (see the kernel); both then run the GP (gp8_body / gp_reg_body of
//     map_net.h, one wave per task) on the same operands, and each takes its own network's gradient back down;
//   * backwards, layer by layer: the delta recursion as the transposed product (four strided loads per MFMA group, requested under the
//     weight-gradient tiles), the weight gradient tiles as products over the points (as many MFMA steps as the points fill), written
//     straight into ONE gradient slab per network in theta's own layout -- which the slab reduction of map_task.hip / mlp_fused.hip
//     (fused_reduce_launch) turns into the AdamW step, the hyper-parameter tail and the next iteration's batch: an iteration is two
//     launches.
// How it got to 31 us per launch (in-kernel stamps, tools/svgd_task_stamps.py ref_map; 4 x 128 networks, 10 points): both networks in ONE
// workgroup 112 000 cycles = 46 us -- one compute unit's matrix cores were the bound: a 128 x 128 layer of the pair is 512
// v_mfma_f32_16x16x4_f32, 128 per SIMD at 32 cycles = 4 100 cycles per layer forward, per delta product and per weight-gradient layer
// (warming the XCD's L2 from helper workgroups changed nothing: tried, measured, removed) --; one workgroup per network with
// __threadfence() on either side of the wait 104 000 (the hand-off alone 18 000); payload as agent-scope atomic stores, one
// thread's release: 90 000; a weight register set per layer instead of a rotating pair (whose copy waited for the loads just
// issued): 72 000 = 30 us.  What is left per workgroup: prologue 4 400, first layer 6 400, hidden layers 5 900 / 5 700 / 3 100, output
// layer + hand-off 5 400, GP 5 400, top of the backward pass 3 800, three backward steps of 5 700 (weight tiles) + 2 500 (bias sums)
// + 2 500 (delta), first layer's gradient 2 000.
// Limits: fp32, RBF, tb x n <= 32 points per iteration (tasks of n <= 16), d <= 4, f <= 4, 1 .. 4 hidden layers of equal or different widths that are
// multiples of 16 and <= 128 (narrower networks take map_task.hip / map_persist.hip).
// Reference lines replaced: GPR_meta_mll.py:104-117, models.py:206-217, 505-519.
#include "map_net.h"

namespace pacoh {

int fused_reduce_launch(const float* slab0, int wd0, long off0, const float* slab1, int wd1, long off1, int nets, float* d_theta,
                        long d_theta_stride, int slabs, const HyperBwdArgs<float>* tail, float* img_th, const int* img_map, hipStream_t s, int P = 1);

namespace {

constexpr int MW_NT = 1024;
constexpr int MW_MAXL = 5;           // up to 4 hidden layers + the output layer
constexpr int MW_MAXW = 128;
constexpr int MW_PT = 32;            // points per iteration: one or two 16-point MFMA tiles

struct MwLayer { int in, out, w_flat, b_flat; };
struct MwNet {
    int nl; MwLayer L[MW_MAXL];
    int flat0, dnet; float* slab;
    int o_act;                       // LDS: activations of the hidden layers [nl - 1][MW_PT][S]
    int o_out, s_out;                // LDS: the network's outputs [pts][s_out] (mean: 1, features: f)
    int o_gout;                      // LDS: upstream gradients [pts][s_out]
    int o_del;                       // LDS: delta ping-pong [2][MW_PT][S]
};
struct MwArgs {
    const float* theta; const float* bx; const float* by; const int32_t* bnv;
    const float* hyp_ls; const float* hyp_os; const float* hyp_noise;
    MwNet net[2]; int nets;
    int n, d, f, tb, pts, mean_mode, kernel_nn, off_const, gp8, S;
    float* lml_g; int32_t* info_g; float* dls_g; float* dos_g; float* dnz_g; float* dc_g;
    long* adv_counter;
    float* xchg; int* sync;          // two networks = two workgroups: their outputs [2][128] and arrival counts [2] (see the kernel)
    int o_hp, o_x, o_xs, o_y, o_nv, o_gl, o_gp, gpw, o_dummy, total;
};

typedef float __attribute__((ext_vector_type(4), aligned(4))) f4u;      // 16 bytes at 4-byte alignment (rows of theta)

// out[p][u] = tanh(b[u] + sum_c W[u][c] x[p][c]) of the first layer (in = d <= 4): one (point, unit) per thread and round
// (the first two entries of a thread -- all of them up to 2 048 point x unit pairs -- get their bias and weight row REQUESTED at the top
//  of the kernel, beside the task's points: behind the prologue's barrier they were two more cold round trips, 6 600 cycles of the
//  launcher's 75 000)
struct MwFirst { float b[2]; float w[2][4]; };
__device__ __forceinline__ void mw_first_load(const MwArgs& a, const MwNet& N, int tl, int nthr, MwFirst& F) {
    const MwLayer& L = N.L[0];
    const float* W = a.theta + L.w_flat; const float* b = a.theta + L.b_flat;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = tl + k * nthr;
        const int u = e < a.pts * L.out ? e % L.out : 0;
        F.b[k] = b[u];
#pragma unroll
        for (int c = 0; c < 4; ++c) F.w[k][c] = W[u * L.in + (c < L.in ? c : 0)];
    }
}
__device__ __forceinline__ void mw_first_layer(const MwArgs& a, const MwNet& N, float* lds, int tl, int nthr, const MwFirst& F) {
    const MwLayer& L = N.L[0];
    const float* W = a.theta + L.w_flat; const float* b = a.theta + L.b_flat;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = tl + k * nthr;
        if (e < a.pts * L.out) {
            const int p = e / L.out, u = e - p * L.out;
            float v = F.b[k];
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < L.in) v = fmaf(F.w[k][c], lds[a.o_x + p * 4 + c], v);
            lds[N.o_act + p * a.S + u] = act_tanh<float>(v);
        }
    }
    for (int e = tl + 2 * nthr; e < a.pts * L.out; e += nthr) {
        const int p = e / L.out, u = e - p * L.out;
        float v = b[u];
        for (int c = 0; c < L.in; ++c) v = fmaf(W[u * L.in + c], lds[a.o_x + p * 4 + c], v);
        lds[N.o_act + p * a.S + u] = act_tanh<float>(v);
    }
}

// hidden layer l >= 1 (in, out multiples of 16, <= 128: at most eight 16-unit output tiles, one per wave of the network's nw >= 8 waves).
// The lane's row of the weight matrix is REQUESTED a layer ahead (mw_load_w: the weights were written by the previous iteration's
// AdamW on other XCDs -- an L2 miss of ~2 000 cycles per layer if it were waited for where it is used) and applied behind the barrier.
struct MwW { f4u wq[MW_MAXW / 16]; f4u bq; };
__device__ __forceinline__ void mw_load_w(const MwArgs& a, const MwNet& N, int l, int wl, int r, int g, MwW& w) {
    const MwLayer& L = N.L[l];
    const int nc = L.in >> 4;
    const int U = wl & 7;                                // (waves 8 .. 15: the same tiles for the second point tile)
    const bool mine = l >= 1 && l + 1 < N.nl && U < (L.out >> 4) && 16 * (wl >> 3) < a.pts;
    const float* wrow = a.theta + L.w_flat + (long)(16 * U + r) * L.in + 4 * g;
#pragma unroll
    for (int c = 0; c < MW_MAXW / 16; ++c) w.wq[c] = (mine && c < nc) ? *reinterpret_cast<const f4u*>(wrow + 16 * c) : f4u{0.f, 0.f, 0.f, 0.f};
    w.bq = mine ? *reinterpret_cast<const f4u*>(a.theta + L.b_flat + 16 * U + 4 * g) : f4u{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void mw_hidden_layer(const MwArgs& a, const MwNet& N, int l, float* lds, int wl, int r, int g, const MwW& w) {
    const MwLayer& L = N.L[l];
    const int U = wl & 7, Pt = wl >> 3;                  // output tile, point tile
    if (U >= (L.out >> 4) || 16 * Pt >= a.pts) return;
    const int S = a.S, nc = L.in >> 4, p = 16 * Pt + r;
    const float* ain = lds + N.o_act + (l - 1) * MW_PT * S + p * S + 4 * g;
    float* aout = lds + N.o_act + l * MW_PT * S + p * S + 4 * g;
    gpreg::f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < MW_MAXW / 16; ++c) {
        if (c < nc) {
            const float4 av = *reinterpret_cast<const float4*>(ain + 16 * c);
            acc = gpreg::mfma_(w.wq[c][0], av.x, acc); acc = gpreg::mfma_(w.wq[c][1], av.y, acc);
            acc = gpreg::mfma_(w.wq[c][2], av.z, acc); acc = gpreg::mfma_(w.wq[c][3], av.w, acc);
        }
    }
    // lane (r, g) holds units 16 U + 4 g + s of point p
    float4 v;
    v.x = act_tanh<float>(acc[0] + w.bq[0]); v.y = act_tanh<float>(acc[1] + w.bq[1]);
    v.z = act_tanh<float>(acc[2] + w.bq[2]); v.w = act_tanh<float>(acc[3] + w.bq[3]);
    if (p < a.pts) *reinterpret_cast<float4*>(aout + 16 * U) = v;
}

// the output layer (out <= 4): out[p][o] = b[o] + W[o][:] . h[p][:], 16 lanes per (p, o) entry
// (the output layer's weight pieces and bias of a thread's first entry -- the only one up to 64 point x output pairs -- are requested a
//  layer ahead, like the hidden layers' rows)
struct MwOut { f4u w[2]; float b; };
__device__ __forceinline__ void mw_output_load(const MwArgs& a, const MwNet& N, int tl, int nthr, MwOut& O) {
    const MwLayer& L = N.L[N.nl - 1];
    const int sub = tl & 15, e = tl >> 4;
    const int o = e < a.pts * L.out ? e % L.out : 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int k4 = sub + 16 * k;
        O.w[k] = k4 < (L.in >> 2) ? *reinterpret_cast<const f4u*>(a.theta + L.w_flat + o * L.in + 4 * k4) : f4u{0.f, 0.f, 0.f, 0.f};
    }
    O.b = a.theta[L.b_flat + o];
}
__device__ __forceinline__ void mw_output_layer(const MwArgs& a, const MwNet& N, float* lds, int tl, int nthr, const MwOut& O) {
    const MwLayer& L = N.L[N.nl - 1];
    const int sub = tl & 15, S = a.S;
    const float* h = lds + N.o_act + (N.nl - 2) * MW_PT * S;
    {
        const int e = tl >> 4;
        if (e < a.pts * L.out) {
            const int p = e / L.out, o = e - p * L.out;
            float v = 0.0f;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int k4 = sub + 16 * k;
                if (k4 < (L.in >> 2)) {
                    const float4 hv = *reinterpret_cast<const float4*>(h + p * S + 4 * k4);
                    v = fmaf(O.w[k][0], hv.x, v); v = fmaf(O.w[k][1], hv.y, v); v = fmaf(O.w[k][2], hv.z, v); v = fmaf(O.w[k][3], hv.w, v);
                }
            }
            for (int k4 = sub + 32; k4 < (L.in >> 2); k4 += 16) {      // (never at widths <= 128)
                const f4u w = *reinterpret_cast<const f4u*>(a.theta + L.w_flat + o * L.in + 4 * k4);
                const float4 hv = *reinterpret_cast<const float4*>(h + p * S + 4 * k4);
                v = fmaf(w[0], hv.x, v); v = fmaf(w[1], hv.y, v); v = fmaf(w[2], hv.z, v); v = fmaf(w[3], hv.w, v);
            }
            v = gpreg::row_sum_(v);
            if (sub == 0) lds[N.o_out + p * N.s_out + o] = v + O.b;
        }
    }
    for (int e = (tl >> 4) + (nthr >> 4); e < a.pts * L.out; e += nthr >> 4) {
        const int p = e / L.out, o = e - p * L.out;
        float v = 0.0f;
        for (int k4 = sub; k4 < (L.in >> 2); k4 += 16) {
            const f4u w = *reinterpret_cast<const f4u*>(a.theta + L.w_flat + o * L.in + 4 * k4);
            const float4 hv = *reinterpret_cast<const float4*>(h + p * S + 4 * k4);
            v = fmaf(w[0], hv.x, v); v = fmaf(w[1], hv.y, v); v = fmaf(w[2], hv.z, v); v = fmaf(w[3], hv.w, v);
        }
        v = gpreg::row_sum_(v);
        if (sub == 0) lds[N.o_out + p * N.s_out + o] = v + a.theta[L.b_flat + o];
    }
}

__global__ void __launch_bounds__(MW_NT) map_wide_kernel(MwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = t & 15, g = (t >> 4) & 3;
    const int n = a.n, d = a.d, f = a.f, tb = a.tb, pts = a.pts, S = a.S;
    float* hp = lds + a.o_hp;
#ifdef PACOH_MP_STAMPS
    if (t == 0) { mp_st_on = 1; mp_st_n[0] = mp_st_n[1] = 0; }
    __syncthreads();
    MP_STAMP();
#endif

    // ---- prologue: everything requested first, LDS zeroed (padding points and columns must read 0), then landed -----------------------
    const int epl = n * (d + 1);
    const bool mover = t < tb * epl;
    const int mv_s = mover ? t / epl : 0, mv_r = mover ? t - mv_s * epl : 0;
    float mv_val = 0.0f, hp_val = 0.0f;
    int nv_val = 0;
    if (mover) mv_val = mv_r < n * d ? a.bx[(long)mv_s * (n * d) + mv_r] : a.by[(long)mv_s * n + (mv_r - n * d)];
    if (t < tb && a.bnv) nv_val = a.bnv[t];
    {
        const float* hsrc = t < f ? a.hyp_ls + t
                          : (t == 4 ? a.hyp_os : (t == 5 ? a.hyp_noise : ((t == 6 && a.off_const >= 0) ? a.theta + a.off_const : nullptr)));
        if (t < 7 && hsrc) hp_val = *hsrc;
    }
    // (the first layer's operands and layer 1's weight rows too: requested here, beside the task's points, they travel under the LDS
    //  zeroing and its barrier)
    const int k_net = (int)blockIdx.x;
    const int wl = wave, tl = t, nthr = MW_NT;
    const MwNet& N = a.net[k_net];
    const int max_nl = N.nl;                             // (a workgroup follows its own network's depth)
    MwFirst wf;
    MwW w1, w2, w3;
    mw_first_load(a, N, tl, nthr, wf);
    mw_load_w(a, N, 1, wl, r, g, w1);
    asm volatile("" ::: "memory");                       // (the loads are issued HERE; the compiler would sink them to their use)
    {
        float4* l4 = reinterpret_cast<float4*>(lds);
        for (int q = t; q < (a.total + 3) >> 2; q += MW_NT) l4[q] = float4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    if (t < 7 && (t < f || t >= 4)) hp[t] = hp_val;
    if (t < tb) lds[a.o_gl + t] = -1.0f;               // loss = -sum_t mll_t (GPR_meta_mll.py:109-113)
    if (mover) {
        if (mv_r < n * d) {
            const int i = mv_r / d, c = mv_r - i * d;
            lds[a.o_x + (mv_s * n + i) * 4 + c] = mv_val; lds[a.o_xs + (mv_s * n + i) * d + c] = mv_val;
        } else lds[a.o_y + mv_s * n + (mv_r - n * d)] = mv_val;
    }
    if (t < tb && a.bnv) reinterpret_cast<int*>(lds + a.o_nv)[t] = nv_val;
    if (t == 0 && blockIdx.x == 0 && a.adv_counter) *a.adv_counter += 1;
    __syncthreads();
    MP_STAMP();

    // ONE WORKGROUP PER NETWORK (the first version ran both on one compute unit: its matrix cores were the bound, 4 100 cycles per
    // 128 x 128 layer of the pair).  The two meet once, in front of the GP, which needs the mean AND the features: each publishes its
    // network's outputs (<= 64 floats), counts its arrival and waits for the other's count to catch up -- both arrive exactly once per
    // launch, so the counts stay in step without ever being reset (graph replays included) --, then BOTH run the GP on the same operands
    // (5 000 cycles, in parallel) and each takes the gradient of its own network back down.  Two 1024-thread workgroups are always
    // co-resident on this part; the wait spins on an agent-scope atomic load.

    // ---- forward ---------------------------------------------------------------------------------------------------------------------
    // (three register sets, one per hidden layer >= 1, each loaded a layer ahead of its use: a rotating pair `wcur = wnext` made the copy
    //  wait for the loads it had just issued -- the whole memory latency exposed once per layer.  TWO layers ahead was measured too:
    //  0.0379 against 0.0382 ms per iteration with two registers spilled -- not kept)
    MwOut wo;
    mw_output_load(a, N, tl, nthr, wo);                  // (the output layer's pieces: four layers ahead, nine registers)
    asm volatile("" ::: "memory");
    mw_first_layer(a, N, lds, tl, nthr, wf);
    __syncthreads();
    MP_STAMP();
    if (2 < max_nl) {
        mw_load_w(a, N, 2, wl, r, g, w2);
        asm volatile("" ::: "memory");
        mw_hidden_layer(a, N, 1, lds, wl, r, g, w1);
        __syncthreads();
        MP_STAMP();
    }
    if (3 < max_nl) {
        mw_load_w(a, N, 3, wl, r, g, w3);
        asm volatile("" ::: "memory");
        mw_hidden_layer(a, N, 2, lds, wl, r, g, w2);
        __syncthreads();
        MP_STAMP();
    }
    if (4 < max_nl) {
        mw_hidden_layer(a, N, 3, lds, wl, r, g, w3);
        __syncthreads();
        MP_STAMP();
    }
    mw_output_layer(a, N, lds, tl, nthr, wo);
    __syncthreads();
    if (a.nets > 1) {
        const MwNet& No = a.net[1 - k_net];
        // (the payload as agent-scope atomic stores -- written through, no cache line to flush -- and ONE thread's release on the count: a
        //  __threadfence() by all 1 024 threads on either side of the wait made this hand-off 18 000 cycles)
        for (int e = t; e < pts * N.s_out; e += MW_NT) __hip_atomic_store(a.xchg + k_net * 128 + e, lds[N.o_out + e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        __shared__ int arrived;
        if (t == 0) {
            const int mine = __hip_atomic_fetch_add(a.sync + k_net, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) + 1;
            // (bounded: counts that were never zeroed -- a workspace without pacoh_map_task_setup -- must not hang the device; the
            //  partner's outputs then read NaN and the iteration's loss says so)
            int ok = 0;
            for (int spin = 0; spin < (1 << 20); ++spin) {
                if (__hip_atomic_load(a.sync + (1 - k_net), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= mine) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            arrived = ok;
        }
        __syncthreads();
        for (int e = t; e < pts * No.s_out; e += MW_NT)
            lds[No.o_out + e] = arrived ? __hip_atomic_load(a.xchg + (1 - k_net) * 128 + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : NAN;
        __syncthreads();
    }
    MP_STAMP();

    // ---- GP: one wave per task ----------------------------------------------------------------------------------------------------------
    if (wave < tb) {
        GpMfmaArgs gq;
        const int mean_mode = sg(a.mean_mode), kernel_nn = sg(a.kernel_nn), has_os = a.hyp_os != nullptr;
        const MwNet& Nm = a.net[0];
        const MwNet& Nk = a.net[a.nets - 1];
        gq.z = kernel_nn ? lds + Nk.o_out : lds + a.o_xs; gq.z_div = 1;
        gq.mean = mean_mode == PACOH_MEAN_VECTOR ? lds + Nm.o_out : (mean_mode == PACOH_MEAN_CONST ? hp + 6 : nullptr);
        gq.mean_mode = mean_mode;
        gq.y = lds + a.o_y; gq.y_div = 1;
        gq.ls = hp; gq.os = has_os ? hp + 4 : nullptr; gq.noise = hp + 5;
        gq.n_valid = a.bnv ? reinterpret_cast<int*>(lds + a.o_nv) : nullptr;
        gq.g_lml = lds + a.o_gl;
        // (the per-task outputs belong to workgroup 0; the other one's copies go to a scratch corner of its LDS)
        const bool owner = blockIdx.x == 0;
        float* dum = lds + a.o_dummy;
        gq.lml = owner ? a.lml_g : dum; gq.info = owner ? a.info_g : reinterpret_cast<int32_t*>(dum + 16);
        gq.d_z = kernel_nn ? lds + Nk.o_gout : nullptr;
        gq.d_mean = mean_mode == PACOH_MEAN_VECTOR ? lds + Nm.o_gout : (mean_mode == PACOH_MEAN_CONST ? (owner ? a.dc_g : dum + 32) : nullptr);
        gq.d_ls = owner ? a.dls_g : dum + 48; gq.d_os = has_os ? (owner ? a.dos_g : dum + 112) : nullptr; gq.d_noise = owner ? a.dnz_g : dum + 128;
        gq.B = tb; gq.P = 1; gq.n = sg(n); gq.f = sg(f);
        if (sg(a.gp8)) {
            if (f <= 2) gpreg::gp8_body<2>(gq, gpreg::WaveCtx{(unsigned)wave}); else gpreg::gp8_body<4>(gq, gpreg::WaveCtx{(unsigned)wave});
        } else {
            float* ws = lds + sg(a.o_gp) + wave * sg(a.gpw);
            constexpr int NP = 16;
            if (f <= 2) gpreg::gp_reg_body<1, 2, true, true>(gq, gpreg::WaveCtx{(unsigned)wave}, ws, ws + NP * 2, ws + NP * 2 + NP, ws + NP * 2 + 2 * NP,
                                                             ws + NP * 2 + 2 * NP, ws + NP * 2 + 2 * NP + gpreg::GPR_SCR, ws + 2 * NP * 2 + 2 * NP + gpreg::GPR_SCR);
            else gpreg::gp_reg_body<1, 4, true, true>(gq, gpreg::WaveCtx{(unsigned)wave}, ws, ws + NP * 4, ws + NP * 4 + NP, ws + NP * 4 + 2 * NP,
                                                      ws + NP * 4 + 2 * NP, ws + NP * 4 + 2 * NP + gpreg::GPR_SCR, ws + 2 * NP * 4 + 2 * NP + gpreg::GPR_SCR);
        }
    }
    __syncthreads();
    MP_STAMP();

    // ---- backward.  Top: the output layer's gradients (slab) and the delta of the last hidden layer, on the vector units ----------------
    float* slab = N.slab;
    {
        const MwLayer& L = N.L[N.nl - 1];
        const float* h = lds + N.o_act + (N.nl - 2) * MW_PT * S;
        const float* gout = lds + N.o_gout;
        // d W_out[o][k] = sum_p g[p][o] h[p][k], d b_out[o] = sum_p g[p][o]
        for (int e = tl; e < L.out * (L.in + 1); e += nthr) {
            const int o = e / (L.in + 1), k = e - o * (L.in + 1);
            float v = 0.0f;
            for (int p = 0; p < pts; ++p) v = fmaf(gout[p * N.s_out + o], k < L.in ? h[p * S + k] : 1.0f, v);
            slab[(k < L.in ? L.w_flat + o * L.in + k : L.b_flat + o) - N.flat0] = v;
        }
        // delta of the last hidden layer: (sum_o W_out[o][i] g[p][o]) (1 - h[p][i]^2) -> delta buffer 0
        float* del = lds + N.o_del;
        for (int e = tl; e < pts * L.in; e += nthr) {
            const int p = e / L.in, i = e - p * L.in;
            float v = 0.0f;
            for (int o = 0; o < L.out; ++o) v = fmaf(a.theta[L.w_flat + o * L.in + i], gout[p * N.s_out + o], v);
            const float hv = h[p * S + i];
            del[p * S + i] = v * fmaf(-hv, hv, 1.0f);
        }
    }
    __syncthreads();
    MP_STAMP();
    // hidden layers l = nl - 2 .. 1: weight gradient tiles of layer l (its delta x the activations below) and the delta of layer l - 1
    for (int step = 0; step + 2 < max_nl; ++step) {
        const int l = N.nl - 2 - step;                   // this network's layer of the step (networks of different depth: the shallower one idles)
        if (l >= 1) {
            const MwLayer& L = N.L[l];
            const float* del = lds + N.o_del + (step & 1) * MW_PT * S;           // delta of layer l [pts][out]
            float* dnx = lds + N.o_del + ((step + 1) & 1) * MW_PT * S;           // delta of layer l - 1 [pts][in]
            const float* ain = lds + N.o_act + (l - 1) * MW_PT * S;             // activations below [pts][in]
            const int nJ = L.out >> 4, nI = L.in >> 4;
            // the transposed weights of the delta product (wave wl: input tile I = wl) are requested FIRST: W[16 c + 4 g + s][16 wl + r],
            // four strided loads per group of 16 outputs, under the weight-gradient tiles' LDS work
            float wt[MW_MAXW / 16][4];
            const int Iw = wl & 7, Pw = wl >> 3;         // the delta product's tile of this wave: input block, point tile
            const bool dmine = Iw < nI && 16 * Pw < pts;
            {
                const float* wcol = a.theta + L.w_flat + 16 * Iw + r + (long)(4 * g) * L.in;
#pragma unroll
                for (int c = 0; c < MW_MAXW / 16; ++c)
#pragma unroll
                    for (int s = 0; s < 4; ++s) wt[c][s] = (dmine && c < nJ) ? wcol[(long)(16 * c + s) * L.in] : 0.0f;
                asm volatile("" ::: "memory");           // (issued here, used behind the weight tiles)
            }
            // weight tiles: a wave takes row block J (its delta operand is read once) and every other column block I;
            // D[j][i] = sum_p delta[p][16 J + j] a[p][16 I + i]; lane (r, g) holds rows 4 g + s, column r
            const int Jw = wl & 7, Ih = wl >> 3;         // sixteen waves: row block J = wl mod 8, every other column block
            if (Jw < nJ) {
                const int nks = (pts + 3) >> 2;          // MFMA steps the points fill (the rows behind them are zero: nothing to add)
                float dj[MW_PT / 4];
#pragma unroll
                for (int ks = 0; ks < MW_PT / 4; ++ks) dj[ks] = ks < nks ? del[(4 * ks + g) * S + 16 * Jw + r] : 0.0f;
                float* dst0 = slab + (L.w_flat - N.flat0) + (long)(16 * Jw + 4 * g) * L.in + r;
                // the bias gradient of the wave's 16 rows, sum_p delta[p][16 J + r], from the operand it holds anyway: a vector add per
                // step and two cross-row adds (as a phase of its own -- 128 threads walking the points through LDS one by one -- it
                // was 2 100 cycles per layer)
                if (Ih == 0) {
                    float bs = 0.0f;
#pragma unroll
                    for (int ks = 0; ks < MW_PT / 4; ++ks) bs += dj[ks];
                    bs += __shfl_xor(bs, 16, 64); bs += __shfl_xor(bs, 32, 64);
                    if (g == 0) slab[L.b_flat - N.flat0 + 16 * Jw + r] = bs;
                }
#pragma unroll 2
                for (int I = Ih; I < nI; I += 2) {
                    gpreg::f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < MW_PT / 4; ++ks) if (ks < nks) acc = gpreg::mfma_(dj[ks], ain[(4 * ks + g) * S + 16 * I + r], acc);
                    float* dst = dst0 + 16 * I;
#pragma unroll
                    for (int s = 0; s < 4; ++s) dst[(long)s * L.in] = acc[s];
                }
            }
            MP_STAMP();
            // delta of the layer below: D[i][p] = sum_j W[j][16 I + i] delta[p][j], times (1 - a^2)
            if (dmine) {
                gpreg::f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < MW_MAXW / 16; ++c) {
                    if (c < nJ) {
                        const float4 dv = *reinterpret_cast<const float4*>(del + (16 * Pw + r) * S + 16 * c + 4 * g);
                        acc = gpreg::mfma_(wt[c][0], dv.x, acc); acc = gpreg::mfma_(wt[c][1], dv.y, acc);
                        acc = gpreg::mfma_(wt[c][2], dv.z, acc); acc = gpreg::mfma_(wt[c][3], dv.w, acc);
                    }
                }
                const float4 hv = *reinterpret_cast<const float4*>(ain + (16 * Pw + r) * S + 16 * Iw + 4 * g);
                float4 o4;
                o4.x = acc[0] * fmaf(-hv.x, hv.x, 1.0f); o4.y = acc[1] * fmaf(-hv.y, hv.y, 1.0f);
                o4.z = acc[2] * fmaf(-hv.z, hv.z, 1.0f); o4.w = acc[3] * fmaf(-hv.w, hv.w, 1.0f);
                if (16 * Pw + r < pts) *reinterpret_cast<float4*>(dnx + (16 * Pw + r) * S + 16 * Iw + 4 * g) = o4;
            }
        }
        __syncthreads();
        MP_STAMP();
    }
    // the first layer: d W_0[j][c] = sum_p delta_0[p][j] x[p][c], d b_0[j] = sum_p delta_0[p][j]
    {
        const MwLayer& L = N.L[0];
        const float* del = lds + N.o_del + ((N.nl - 2) & 1) * MW_PT * S;
        for (int e = tl; e < L.out * (L.in + 1); e += nthr) {
            const int j = e / (L.in + 1), c = e - j * (L.in + 1);
            float v = 0.0f;
            for (int p = 0; p < pts; ++p) v = fmaf(del[p * S + j], c < L.in ? lds[a.o_x + p * 4 + c] : 1.0f, v);
            slab[(c < L.in ? L.w_flat + j * L.in + c : L.b_flat + j) - N.flat0] = v;
        }
    }
#ifdef PACOH_MP_STAMPS
    MP_STAMP();
    __syncthreads();
    if (t == 0 && a.adv_counter && *a.adv_counter == 3)
        for (int q = 1; q < mp_st_n[0]; ++q) printf("mw stamp %d: +%lld cycles\n", q, mp_st[0][q] - mp_st[0][q - 1]);
#endif
}

// -> PACOH_OK and the filled arguments, or PACOH_ELIMIT when the shape is outside this kernel's plan
int mw_plan(MwArgs& a, int n, int d, int tb, int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden, int kernel_nn,
            int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f) {
    memset(&a, 0, sizeof(a));
    if (n < 1 || n > 16 || d < 1 || d > 4 || f < 1 || f > 4 || tb < 1 || tb > MW_NT / 64 || tb * n > MW_PT || tb * n * (d + 1) > MW_NT) return PACOH_ELIMIT;
    if (!kernel_nn && f != d) return PACOH_EINVAL;
    a.n = n; a.d = d; a.f = f; a.tb = tb; a.pts = tb * n; a.mean_mode = mean_mode; a.kernel_nn = kernel_nn;
    a.gp8 = (n <= 8 && g_sw.gp8) ? 1 : 0;
    int maxw = 16;
    auto net = [&](int k, int off, const int32_t* hidden, int nh, int d_out) -> int {
        if (nh < 1 || nh > MW_MAXL - 1) return PACOH_ELIMIT;
        MwNet& N = a.net[k];
        N.nl = nh + 1;
        int prev = d, q = off;
        for (int l = 0; l <= nh; ++l) {
            const int out = l < nh ? hidden[l] : d_out;
            if (l < nh && (out < 16 || out > MW_MAXW || (out & 15))) return PACOH_ELIMIT;
            N.L[l].in = prev; N.L[l].out = out; N.L[l].b_flat = q; N.L[l].w_flat = q + out; q += out * (prev + 1);
            if (l < nh && out > maxw) maxw = out;
            prev = out;
        }
        N.flat0 = off; N.dnet = q - off;
        return PACOH_OK;
    };
    a.nets = 0;
    if (mean_mode == PACOH_MEAN_VECTOR) { const int rc = net(a.nets, off_mean, mean_hidden, n_mean_hidden, 1); if (rc) return rc; a.net[a.nets].s_out = 1; a.nets++; }
    if (kernel_nn) { const int rc = net(a.nets, off_kernel, kernel_hidden, n_kernel_hidden, f); if (rc) return rc; a.net[a.nets].s_out = f; a.nets++; }
    if (a.nets < 1) return PACOH_ELIMIT;
    bool wide = false;                                   // (networks of <= 32 units per layer belong to map_task.hip / map_persist.hip)
    for (int k = 0; k < a.nets; ++k) for (int l = 0; l + 1 < a.net[k].nl; ++l) wide = wide || a.net[k].L[l].out > 32;
    if (!wide) return PACOH_ELIMIT;
    a.S = maxw + 4;                                      // row stride of the activation / delta images: 16-byte rows, 4 (odd) quads off a bank period
    int top = 0;
    auto take = [&](int count) { const int o = top; top += (count + 3) & ~3; return o; };
    a.o_hp = take(8);
    a.o_x = take(MW_PT * 4); a.o_xs = take(MW_PT * d); a.o_y = take(MW_PT); a.o_nv = take(16); a.o_gl = take(16);
    // (a workgroup works on ONE network: the activation and delta images of the two share their place; the small output / upstream-
    //  gradient arrays exist for both -- the GP reads both networks' outputs)
    int max_hidden = 1;
    for (int k = 0; k < a.nets; ++k) if (a.net[k].nl - 1 > max_hidden) max_hidden = a.net[k].nl - 1;
    const int o_act = take(max_hidden * MW_PT * a.S), o_del = take(2 * MW_PT * a.S);
    for (int k = 0; k < a.nets; ++k) {
        MwNet& N = a.net[k];
        N.o_act = o_act; N.o_del = o_del;
        N.o_out = take(MW_PT * N.s_out); N.o_gout = take(MW_PT * N.s_out);
    }
    const int FPp = f <= 2 ? 2 : 4;
    a.gpw = (2 * 16 * FPp + 2 * 16 + gpreg::GPR_SCR + 4 + 3) & ~3;
    a.o_gp = take(a.gp8 ? 4 : a.gpw * tb);
    a.o_dummy = take(160);
    a.total = top;
    if ((size_t)top * sizeof(float) > (size_t)MP_LDS_BYTES) return PACOH_ELIMIT;
    return PACOH_OK;
}

}  // namespace

// the launch pair of map_task_launch for wide networks: plan_only 1 -> *need_bytes only; 2 -> nothing to set up (no parameter image)
int map_wide_launch(const void* theta, const void* bx, const void* by, const int32_t* bnv, int n, int d, int tb_total,
                    int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                    int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                    const void* hyp_ls, const void* hyp_os, const void* hyp_noise, void* workspace, size_t workspace_bytes,
                    void* d_theta, long d_theta_stride, const HyperBwdArgs<float>* tail_in, int plan_only, size_t* need_bytes, int D, hipStream_t stream) {
    MwArgs a;
    const int rc = mw_plan(a, n, d, tb_total, mean_mode, off_mean, mean_hidden, n_mean_hidden, kernel_nn, off_kernel, kernel_hidden, n_kernel_hidden, f);
    if (rc != PACOH_OK) return rc;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t o_slab[2] = {0, 0};
    for (int k = 0; k < a.nets; ++k) o_slab[k] = carve((size_t)a.net[k].dnet * sizeof(float));
    const size_t B_ = (size_t)tb_total;
    const size_t o_lml = carve(B_ * 4), o_dls = carve(B_ * f * 4), o_dos = carve(B_ * 4), o_dnz = carve(B_ * 4), o_dc = carve(B_ * 4), o_info = carve(B_ * 4);
    const size_t o_xchg = carve(2 * 128 * 4), o_sync = carve(2 * 4);
    if (need_bytes) *need_bytes = off;
    if (plan_only == 1) return PACOH_OK;
    if (plan_only == 2) {                               // (setup: the two workgroups' arrival counts start at zero, once per workspace)
        if (!workspace || workspace_bytes < off) return PACOH_EINVAL;
        return hipMemsetAsync((char*)workspace + o_sync, 0, 8, stream) == hipSuccess ? PACOH_OK : PACOH_ELAUNCH;
    }
    for (int k = 0; k < a.nets; ++k)
        if (a.net[k].flat0 < 0 || a.net[k].flat0 + a.net[k].dnet > D || a.net[k].flat0 + a.net[k].dnet > d_theta_stride) return PACOH_EINVAL;
    if (!workspace || workspace_bytes < off) return PACOH_EINVAL;
    char* ws = (char*)workspace;
    a.theta = (const float*)theta;
    a.off_const = mean_mode == PACOH_MEAN_CONST ? off_mean : -1;
    a.bx = (const float*)bx; a.by = (const float*)by; a.bnv = bnv;
    a.hyp_ls = (const float*)hyp_ls; a.hyp_os = (const float*)hyp_os; a.hyp_noise = (const float*)hyp_noise;
    for (int k = 0; k < a.nets; ++k) a.net[k].slab = (float*)(ws + o_slab[k]);
    a.lml_g = (float*)(ws + o_lml); a.dls_g = (float*)(ws + o_dls); a.dos_g = (float*)(ws + o_dos); a.dnz_g = (float*)(ws + o_dnz);
    a.dc_g = (float*)(ws + o_dc); a.info_g = (int32_t*)(ws + o_info);
    a.xchg = (float*)(ws + o_xchg); a.sync = (int*)(ws + o_sync);
    HyperBwdArgs<float> tail = *tail_in;
    tail.d_ls = a.dls_g; tail.d_os = hyp_os ? a.dos_g : nullptr; tail.d_noise = a.dnz_g; tail.d_const = mean_mode == PACOH_MEAN_CONST ? a.dc_g : nullptr;
    tail.lml = tail.lik ? a.lml_g : nullptr; tail.info = tail.fail_flag ? a.info_g : nullptr;
    a.adv_counter = const_cast<long*>(tail.nx.counter);
    static std::atomic<uint64_t> attr_done{0};
    { const int rc_a = lds_opt_in((const void*)map_wide_kernel, MP_LDS_BYTES, attr_done); if (rc_a != PACOH_OK) return rc_a; }
    hipLaunchKernelGGL(map_wide_kernel, dim3((unsigned)a.nets), dim3(MW_NT), (size_t)a.total * sizeof(float), stream, a);
    if (launch_status() != PACOH_OK) return PACOH_ELAUNCH;
    return fused_reduce_launch(a.net[0].slab, a.net[0].dnet, a.net[0].flat0, a.nets > 1 ? a.net[1].slab : nullptr, a.nets > 1 ? a.net[1].dnet : 0,
                               a.nets > 1 ? a.net[1].flat0 : 0, a.nets, (float*)d_theta, d_theta_stride, 1, &tail, nullptr, nullptr, stream, 1);
}

}  // namespace pacoh
