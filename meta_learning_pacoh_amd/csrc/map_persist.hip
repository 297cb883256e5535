// Persistent PACOH-MAP iteration kernel (round 5): K whole meta-training iterations of GPRegressionMetaLearned per LAUNCH, for the
// reference's own regime -- a handful of small tasks per iteration (demo.py: 5 tasks x 5 points, two 32-wide networks).  There an
// iteration is a few microseconds of arithmetic, and as four dependent launches (forward, GP, backward, slab reduction + AdamW) it
// costs four kernel latencies: 0.028 ms.  Here ONE 1024-thread workgroup keeps the parameters, both Adam moments and every
// activation of the batch in LDS and runs
//     task gather -> both networks forward -> Gram / Cholesky / LML and its gradient (one wave per task: gp_reg_body.h, the
//     register-resident matrix-core kernel of the big configurations) -> both networks backward -> AdamW on every parameter ->
//     softplus transforms of the hyper-parameters
// K times from the pre-uploaded task draws and step scalars of engine.StepFeed; the launch latency is paid once per K iterations
// and nothing but the next iteration's task rows (prefetched one iteration ahead) is read from memory inside the loop.
// Reference lines replaced: GPR_meta_mll.py:104-117 (the loop body of meta_fit: task loop, ExactMarginalLogLikelihood, backward,
// AdamW step), models.py:505-519 (LearnedGPRegressionModel.forward), models.py:206-217 (NeuralNetwork.forward).
//
// Data layout in LDS (floats; offsets computed by the host, MpPlan):
//   parameter image th / m / v : every linear layer as an [out][S] matrix, S = (in + 1) rounded up to 4: columns 0..in-1 the
//                        weights, column `in` the bias, zero padding -- its input activations carry a constant 1 in column `in`,
//                        so a layer is one product and a weight-and-bias gradient one outer-product sum.  S = 4 * odd for the
//                        widths in use (36, 20, 12, 4): "own row per lane" ds_read_b128 without bank conflicts
//   flat[]             : flat index in theta of every image entry that is TRAINED (learning_mode), -1 for padding / frozen entries
//   A0 / Xs / Y / nv   : the iteration's task batch (double-buffered: the next batch lands while this one is in use)
//   H[net][l]          : hidden activations [pts][S of the next layer];  Mn / Zk: the networks' outputs = the GP's mean and features
//   Dl[net][l]         : gradient w.r.t. the pre-activation outputs of hidden layer l;  dMn / dZk: the GP's gradients
// Work mapping: lane j of a 32-lane group = output unit j, the groups stride over the points (forward and delta steps); a
// weight-gradient "unit" = four consecutive columns of one row summed over the points by one thread, which then applies AdamW to
// them in place (op order of adam_update, hyper_tail.h) -- a layer's weights are updated one phase after their last reader.
// Phases per iteration for L hidden layers: L + 1 forward, GP, L + 1 backward: 2 L + 3 workgroup barriers.

#include "map_net.h"
#include <type_traits>

namespace pacoh {
namespace {

template <int NB, int FP>
__global__ void __launch_bounds__(MP_NT) map_persist_kernel(MpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ __attribute__((aligned(16))) int ltab[2 * MP_MAXL * 16];      // the layer table (mp_layer_lds)
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int pts = a.pts, n = a.n, d = a.d, f = a.f, tb = a.tb;
    float* th = lds + a.o_th;
    float* mm = lds + a.o_m;
    float* vv = lds + a.o_v;
    int* flat = reinterpret_cast<int*>(lds + a.o_flat);
    float* hp = lds + a.o_hp;
    int* info_l = reinterpret_cast<int*>(lds + a.o_info);
    int* nv_l = reinterpret_cast<int*>(lds + a.o_nv);

    // ---- prologue: zero LDS, parameter image, constant columns, hyper-parameters, step scalars, first batch -------------------
    if (t < 2 * MP_MAXL * 16) {
        typedef const int __attribute__((address_space(4))) * kint_t;
        kint_t kt = (kint_t)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(MpArgs, L));
        ltab[t] = kt[t];
    }
    for (int q = t; q < a.total; q += MP_NT) lds[q] = 0.0f;
    __syncthreads();
    for (int q = t; q < a.DP; q += MP_NT) flat[q] = -1;
    __syncthreads();
    for (int k = 0; k < a.nets; ++k)
        for (int l = 0; l < a.nl[k]; ++l) {
            const MpLayer L = mp_layer_karg(k, l);
            for (int e = t; e < L.out * (L.in + 1); e += MP_NT) {
                const int j = e / (L.in + 1), i = e - j * (L.in + 1);
                const int q = i < L.in ? L.w_flat + j * L.in + i : L.b_flat + j;
                const int li = L.w_lds + j * L.S + i;
                th[li] = a.theta[q]; mm[li] = a.m[q]; vv[li] = a.v[q];
                flat[li] = mp_trained(a, q) ? q : -1;
            }
            // the constant-1 column of this layer's input activations
            if (l == 0) { for (int p = t; p < 2 * pts; p += MP_NT) lds[a.o_a0 + (p / pts) * a.a0_sz + (p % pts) * L.S + L.in] = 1.0f; }
            else for (int p = t; p < pts; p += MP_NT) lds[L.a_in + p * L.S + L.in] = 1.0f;
        }
    if (t < 7) {
        const int e = t;
        int q = -1;
        if (e < f) q = a.off_ls + e; else if (e == 4) q = a.off_os; else if (e == 5) q = a.off_noise; else if (e == 6) q = a.off_const;
        if ((e < f || e >= 4) && q >= 0) {
            const float raw = a.theta[q];
            hp[HP_RAW + e] = raw; hp[HP_M + e] = a.m[q]; hp[HP_V + e] = a.v[q];
            hp[e] = e == 6 ? raw : (mp_softplus(raw) + (e == 5 ? a.noise_floor : 0.0f));
        }
    }
    if (t < tb) lds[a.o_gl + t] = -1.0f;               // loss = -sum_t mll_t (GPR_meta_mll.py:109-113): upstream gradient of every LML

    // the batch elements this thread moves: element r of slot s -- r < n d: x[i][c], else y[i]
    const int epl = n * (d + 1);
    const bool mover = t < tb * epl;
    const int mv_s = mover ? t / epl : 0, mv_r = mover ? t - mv_s * epl : 0;
    const bool mv_x = mv_r < n * d;
    const int mv_i = mv_x ? mv_r / d : mv_r - n * d, mv_c = mv_x ? mv_r - mv_i * d : 0;
    // the task ids of a row are fetched TWO iterations ahead and the values one ahead: an id load whose result the value load needs
    // at once would stall every iteration for a memory round trip
    long pf_id = 0, pf_idn = 0;
    auto fetch_ids = [&](int row) {
        if (mover) pf_id = a.idx_all[(long)row * tb + mv_s];
        if (t < tb && a.n_valid) pf_idn = a.idx_all[(long)row * tb + t];
    };
    auto fetch_vals = [&](float& val, int& nvv) {
        if (mover) val = mv_x ? a.x[pf_id * (long)(n * d) + mv_r] : a.y[pf_id * (long)n + mv_i];
        if (t < tb && a.n_valid) nvv = a.n_valid[pf_idn];
    };
    auto land = [&](int buf, float val, int nvv) {
        if (mover) {
            const int p = mv_s * n + mv_i;
            if (mv_x) { lds[a.o_a0 + buf * a.a0_sz + p * a.S0 + mv_c] = val; lds[a.o_xs + buf * a.xs_sz + p * d + mv_c] = val; }
            else lds[a.o_y + buf * a.y_sz + p] = val;
        }
        if (t < tb && a.n_valid) nv_l[buf * 16 + t] = nvv;
    };
    float pf_val = 0.0f; int pf_nv = 0;
    fetch_ids(0);
    fetch_vals(pf_val, pf_nv);
    land(0, pf_val, pf_nv);
    if (a.K > 1) fetch_ids(1);
    __syncthreads();

    const int lane = t & 63, r16 = t & 15, g4 = (t >> 4) & 3;
    int max_nl = a.nets > 0 ? a.nl[0] : 0;
    if (a.nets == 2 && a.nl[1] > max_nl) max_nl = a.nl[1];
    constexpr int NW = MP_NT / 64;
    __shared__ int ntask[4];
    MpTask* tasks = reinterpret_cast<MpTask*>(lds + a.o_tasks);
    if (t == 0 && max_nl > 0) mp_plan(a, ltab, tasks, ntask);
    __syncthreads();
    const int slots = a.slots;
    // task q of phase ph -> the wave's tile.  The table does not change during a launch and wave w's first task of a phase is always
    // task w: its descriptor is read and made scalar ONCE, in front of the iteration loop (as four LDS reads and their
    // v_readfirstlane per phase and iteration the decode was ~700 cycles at the head of every phase, right behind a barrier)
    auto run_task = [&](const int4 d0, const int4 d1, const int4 d2, const int4 d3, int a0_off, const MpAdam& ad) {
        const int kind = sgi(d0.x);
        const bool w32 = sgi(d3.x) & 4;
        if (kind == MP_FWD) { if (w32) mp_fwd_chain32(ltab, sgi(d0.z), sgi(d2.z), sgi(d0.w), pts, th, lds, a0_off, r16, g4); else mp_fwd_chain(ltab, sgi(d0.z), sgi(d2.z), sgi(d0.w), pts, th, lds, a0_off, r16, g4); }
        else if (kind == MP_DELTA) { if (w32) mp_delta_chain32(ltab, sgi(d0.z), sgi(d2.z), sgi(d0.w), pts, th, lds, r16, g4); else mp_delta_chain(ltab, sgi(d0.z), sgi(d2.z), sgi(d0.w), pts, th, lds, r16, g4); }
        else if (kind == MP_WGRAD) mp_wgrad_tile(d0, d1, d2, d3, th, mm, vv, flat, lds, a0_off, pts, r16, g4, ad);
    };
    int4 pre[3][4];
    bool has[3];
    int ntk[3];
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
        ntk[ph] = max_nl > 0 ? sgi(ntask[ph]) : 0;
        has[ph] = wave < ntk[ph];
        const int4* e = reinterpret_cast<const int4*>(tasks + ph * slots + (has[ph] ? wave : 0));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int4 v = max_nl > 0 ? e[k] : int4{0, 0, 0, 0};
            pre[ph][k] = int4{sgi(v.x), sgi(v.y), sgi(v.z), sgi(v.w)};
        }
    }
    auto run_phase = [&](auto ph_c, int a0_off, const MpAdam& ad) {
        constexpr int ph = decltype(ph_c)::value;
        const int nt = ntk[ph];
        if (has[ph]) {
            if constexpr (ph < 2) run_task(pre[ph][0], pre[ph][1], pre[ph][2], pre[ph][3], a0_off, ad);
            else {                                        // (the weight tiles' sixteen words stay in LDS: as scalars held across the loop they spilled)
                const int4* e = reinterpret_cast<const int4*>(tasks + ph * slots + wave);
                run_task(e[0], e[1], e[2], e[3], a0_off, ad);
            }
        }
        // later rounds (more tasks than waves) from the last wave down: the tasks are sorted by weight (mp_plan), so a second task
        // goes to a wave whose first one was light
        for (int rd = 1; rd * NW < nt; ++rd) {
            const int q = rd * NW + ((rd & 1) ? NW - 1 - wave : wave);
            if (q >= nt) continue;
            const int4* e = reinterpret_cast<const int4*>(tasks + ph * slots + q);
            run_task(e[0], e[1], e[2], e[3], a0_off, ad);
        }
    };

    float loss_cum = 0.0f, loss_last = 0.0f;           // (thread 0 of the hyper step's wave)
    int any_fail = 0;
#ifdef PACOH_MP_STAMPS
    if (t == 0) { mp_st_on = 0; mp_st_n[0] = mp_st_n[1] = 0; }
    __syncthreads();
#endif

    for (int it = 0; it < a.K; ++it) {
        const int buf = it & 1;
#ifdef PACOH_MP_STAMPS
        if (it == a.K - 1) { __syncthreads(); if (t == 0) mp_st_on = 1; __syncthreads(); }
#endif
        // the iteration's step scalars: a scalar load issued here, used in the backward phases (its latency is long gone by then)
        const float* scr = a.sc_all + (long)it * a.n_sc + PACOH_SC_ADAM;
        const float dm = scr[0], ss = scr[1], bc2 = scr[2], eps = scr[3];
        if (it + 1 < a.K) {                                               // lands at the end of this iteration
            fetch_vals(pf_val, pf_nv);
            if (it + 2 < a.K) fetch_ids(it + 2);
        }
        MP_STAMP();

        // ---- forward: one wave per (network, 16-point tile) runs the network's layers back to back ----------------------------------
        const int a0_off = a.o_a0 + buf * a.a0_sz;
        if (max_nl > 0) {
            run_phase(std::integral_constant<int, 0>{}, a0_off, MpAdam{});
            MP_STAMP();
            __syncthreads();
            MP_STAMP();
        }

        // ---- GP: one wave per task of the batch.  (Tried: several whole tasks per wave as ONE block-diagonal 16 x 16 problem -- three of
        //      demo.py's 5-point tasks per wave, two GP waves instead of five.  The body is ~1 850 instructions per wave whatever its
        //      neighbours do, the per-task sums and masks added 25 %: 10 700 cycles against 8 400 + 1 300 of barrier wait.  Dropped.) ----
        if (wave < tb) {
            const int mean_mode = sg(a.mean_mode), kernel_nn = sg(a.kernel_nn), has_os = sg(a.off_os) >= 0;
            // (wrap: the block body wants its 17 pointers in scalar registers -- sp(); the small-context body addresses per lane anyway,
            //  and the scalar copies cost it ~100 v_readlane of spilled scalars)
            auto gp_args = [&](auto wrap) {
                GpMfmaArgs g;
                g.z = wrap(kernel_nn ? lds + a.o_zk : lds + a.o_xs + buf * a.xs_sz); g.z_div = 1;
                g.mean = wrap(mean_mode == PACOH_MEAN_VECTOR ? lds + a.o_mn : (mean_mode == PACOH_MEAN_CONST ? hp + 6 : (float*)nullptr));
                g.mean_mode = mean_mode;
                g.y = wrap(lds + a.o_y + buf * a.y_sz); g.y_div = 1;
                g.ls = wrap(hp); g.os = wrap(has_os ? hp + 4 : (float*)nullptr); g.noise = wrap(hp + 5);
                g.n_valid = wrap(a.n_valid ? nv_l + buf * 16 : (int*)nullptr);
                g.g_lml = wrap(lds + a.o_gl);
                g.lml = wrap(lds + a.o_lml); g.info = wrap(info_l);
                g.d_z = wrap(kernel_nn ? lds + a.o_dzk : (float*)nullptr);
                g.d_mean = wrap(mean_mode == PACOH_MEAN_VECTOR ? lds + a.o_dmn : (mean_mode == PACOH_MEAN_CONST ? lds + a.o_dc : (float*)nullptr));
                g.d_ls = wrap(lds + a.o_dls); g.d_os = wrap(has_os ? lds + a.o_dos : (float*)nullptr); g.d_noise = wrap(lds + a.o_dnz);
                g.B = sg(tb); g.P = 1; g.n = sg(n); g.f = sg(f);
                return g;
            };
            // contexts of <= 8 points (the reference's demo: 5): one matrix entry per lane, no 16 x 16 blocks (gp8_body.h)
            bool small8 = false;
            if constexpr (NB == 1) small8 = sg(a.gp8) != 0;
            if (small8) gpreg::gp8_body<FP>(gp_args([](auto* q) { return q; }), gpreg::WaveCtx{(unsigned)wave});
            else {
                const GpMfmaArgs g = gp_args([](auto* q) { return sp(q); });
                constexpr int NP = 16 * NB;
                float* ws = lds + sg(a.o_gp) + wave * sg(a.gpw);
                gpreg::gp_reg_body<NB, FP, true, true>(g, gpreg::WaveCtx{(unsigned)wave}, ws, ws + NP * FP, ws + NP * FP + NP,
                                                       ws + NP * FP + 2 * NP, ws + NP * FP + 2 * NP, ws + NP * FP + 2 * NP + gpreg::GPR_SCR,
                                                       ws + 2 * NP * FP + 2 * NP + gpreg::GPR_SCR);
            }
        }
        MP_STAMP();
        __syncthreads();
        MP_STAMP();

        // hyper-parameters: sums over the tasks, softplus chain rule, AdamW, next iteration's transforms; the loss.  One WAVE per entry
        // (the last eight waves: entries 0 .. 6 and the loss) while waves 0 .. 3 run the delta chains: as one wave's work (~1 300
        // instructions with the IEEE division / log1p sequences) it outlasted the chains by 3 000 cycles.  Lane k reads task k's value,
        // the sum runs in task order over v_readlane; nothing below reads what these waves write.
        if (wave >= NW - 8) {
            const int e = wave - (NW - 8), lk = t & 63;
            if (e < 7) {
                if (e < f || e >= 4) {
                    int q = -1, src = 0, width = 1;
                    if (e < f) { q = a.off_ls + e; src = a.o_dls + e; width = f; }
                    else if (e == 4) { q = a.off_os; src = a.o_dos; }
                    else if (e == 5) { q = a.off_noise; src = a.o_dnz; }
                    else { q = a.mean_mode == PACOH_MEAN_CONST ? a.off_const : -1; src = a.o_dc; }
                    if (q >= 0 && mp_trained(a, q)) {
                        const float val = lds[src + (lk < tb ? lk : 0) * width];
                        float sum = 0.0f;
                        for (int k = 0; k < tb; ++k) sum += gpreg::readlane_(val, k);
                        // (sigmoid, AdamW and softplus on the hardware's exp2 / log2 / rcp / sqrt: ~40 instructions; the library
                        //  sequences -- IEEE division, log1p -- were ~200 per entry, and what a phase costs is instructions per SIMD)
                        const float raw = hp[HP_RAW + e], m1 = hp[HP_M + e], v1 = hp[HP_V + e];
                        const float gv = e == 6 ? sum : sum * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * raw));
                        const float mq = fmaf(gv - m1, a.one_minus_b1, m1);
                        const float vq = fmaf(v1, a.b2, a.one_minus_b2 * gv * gv);
                        const float denom = fmaf(__builtin_amdgcn_sqrtf(vq), 1.0f / bc2, eps);
                        const float now = fmaf(-ss * mq, __builtin_amdgcn_rcpf(denom), raw * dm);
                        if (lk == 0) {
                            hp[HP_RAW + e] = now; hp[HP_M + e] = mq; hp[HP_V + e] = vq;
                            hp[e] = e == 6 ? now : (mp_softplus(now) + (e == 5 ? a.noise_floor : 0.0f));
                        }
                    }
                }
            } else {
                const float val = lds[a.o_lml + (lk < tb ? lk : 0)];
                const int bad = info_l[lk < tb ? lk : 0] < 0 && lk < tb;
                float sum = 0.0f;
                for (int k = 0; k < tb; ++k) sum += gpreg::readlane_(val, k);
                any_fail |= __builtin_amdgcn_ballot_w64(bad) != 0ull;
                loss_last = -sum;
                loss_cum += loss_last;
            }
        }
        MP_STAMP();
        // ---- backward: the delta chains (one wave per network and point tile), then every layer's weight step + AdamW ----------------
        const MpAdam ad = {dm, a.one_minus_b1, a.b2, a.one_minus_b2, ss, 1.0f / bc2, eps};
        if (max_nl > 0) {
            run_phase(std::integral_constant<int, 1>{}, a0_off, ad);
            MP_STAMP();
            __syncthreads();
            MP_STAMP();
            run_phase(std::integral_constant<int, 2>{}, a0_off, ad);
            if (it + 1 < a.K) land(buf ^ 1, pf_val, pf_nv);
            __syncthreads();
            MP_STAMP();
        }
        if (max_nl == 0) {                                  // no network at all: the GP and the hyper-parameters only
            if (it + 1 < a.K) land(buf ^ 1, pf_val, pf_nv);
            __syncthreads();
        }
    }

#ifdef PACOH_MP_STAMPS
    __syncthreads();
    if (t == 0) for (int q = 1; q < 48; ++q) if (q < mp_st_n[0] || q < mp_st_n[1])
        printf("mp stamp %d: wave 0 +%lld  wave 15 +%lld cycles\n", q, q < mp_st_n[0] ? mp_st[0][q] - mp_st[0][q - 1] : -1, q < mp_st_n[1] ? mp_st[1][q] - mp_st[1][q - 1] : -1);
#endif
    // ---- epilogue: trained entries back to memory ----------------------------------------------------------------------------------
    for (int li = t; li < a.DP; li += MP_NT) {
        const int q = flat[li];
        if (q >= 0) { a.theta[q] = th[li]; a.m[q] = mm[li]; a.v[q] = vv[li]; }
    }
    if (t < 7) {
        const int e = t;
        int q = -1;
        if (e < f) q = a.off_ls + e; else if (e == 4) q = a.off_os; else if (e == 5) q = a.off_noise;
        else if (e == 6) q = a.mean_mode == PACOH_MEAN_CONST ? a.off_const : -1;
        if ((e < f || e >= 4) && q >= 0 && mp_trained(a, q)) { a.theta[q] = hp[HP_RAW + e]; a.m[q] = hp[HP_M + e]; a.v[q] = hp[HP_V + e]; }
    }
    if (t == MP_NT - 1) {                                  // (wave 15 keeps the loss)
        if (a.loss_last) *a.loss_last = loss_last;
        if (a.loss_cum) *a.loss_cum += loss_cum;
        if (any_fail && a.fail_flag) atomicOr(a.fail_flag, 1);
    }
}

}  // namespace
}  // namespace pacoh

using namespace pacoh;

extern "C" int pacoh_map_persist_supported(int n, int d, int tb, int mean_mode, const int32_t* mean_hidden, int n_mean_hidden, int kernel_nn,
                                           const int32_t* kernel_hidden, int n_kernel_hidden, int f, int dtype) {
    if (dtype != PACOH_F32) return 0;
    MpArgs a; int nb, fp;
    return map_persist_plan(a, n, d, tb, 1024, mean_mode, 0, mean_hidden, n_mean_hidden, kernel_nn, 0, kernel_hidden, n_kernel_hidden, f, &nb, &fp) == PACOH_OK;
}

extern "C" int pacoh_map_persist(void* theta, void* exp_avg, void* exp_avg_sq, int D, const void* x, const void* y, const int32_t* n_valid,
                                 int n, int d, const int64_t* idx_rows, int tb, const void* sc_rows, int n_sc, int K,
                                 int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                                 int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                                 int off_ls, int off_os, int off_noise, double noise_floor,
                                 const int32_t* seg_lo, const int32_t* seg_hi, int n_seg, double beta1, double beta2,
                                 void* loss_last, void* loss_cum, int32_t* fail_flag, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (dtype != PACOH_F32) return PACOH_ELIMIT;
    if (f != features_of(f)) return PACOH_ELIMIT;          // (f carries no kernel-family bits here: the persistent kernel is RBF-only)
    if (!theta || !exp_avg || !exp_avg_sq || !x || !y || !idx_rows || !sc_rows || D <= 0 || n_seg < 0 || n_seg > 4 || n_sc < PACOH_SC_COUNT ||
        off_ls < 0 || off_noise < 0 || (n_seg > 0 && (!seg_lo || !seg_hi))) return PACOH_EINVAL;
    MpArgs a; int NB, FP;
    const int rc = map_persist_plan(a, n, d, tb, K, mean_mode, off_mean, mean_hidden, n_mean_hidden, kernel_nn, off_kernel, kernel_hidden,
                                    n_kernel_hidden, f, &NB, &FP);
    if (rc != PACOH_OK) return rc;
    // every range of the parameter row the kernel reads AND writes (theta and both Adam moments, prologue and epilogue) must lie
    // inside [0, D): a wrong layout from a direct C caller would otherwise be a silent out-of-bounds device write
    {
        auto inside = [D](long lo, long len) { return lo >= 0 && len >= 0 && lo + len <= (long)D; };
        for (int k = 0; k < a.nets; ++k)
            for (int l = 0; l < a.nl[k]; ++l) {
                const MpLayer& Ly = a.L[k][l];
                if (!inside(Ly.b_flat, Ly.out) || !inside(Ly.w_flat, (long)Ly.out * Ly.in)) return PACOH_EINVAL;
            }
        if (!inside(off_ls, f) || !inside(off_noise, 1) || (off_os >= 0 && !inside(off_os, 1)) ||
            (mean_mode == PACOH_MEAN_CONST && !inside(off_mean, 1))) return PACOH_EINVAL;
        for (int s = 0; s < n_seg; ++s) if (seg_lo[s] < 0 || seg_hi[s] < seg_lo[s] || seg_hi[s] > D) return PACOH_EINVAL;
    }
    a.theta = (float*)theta; a.m = (float*)exp_avg; a.v = (float*)exp_avg_sq; a.D = D;
    a.x = (const float*)x; a.y = (const float*)y; a.n_valid = n_valid;
    a.idx_all = idx_rows; a.sc_all = (const float*)sc_rows; a.n_sc = n_sc;
    a.loss_last = (float*)loss_last; a.loss_cum = (float*)loss_cum; a.fail_flag = fail_flag;
    a.off_ls = off_ls; a.off_os = off_os; a.off_noise = off_noise; a.off_const = mean_mode == PACOH_MEAN_CONST ? off_mean : -1;
    a.noise_floor = (float)noise_floor; a.one_minus_b1 = (float)(1.0 - beta1); a.b2 = (float)beta2; a.one_minus_b2 = (float)(1.0 - beta2);
    a.nseg = n_seg;
    for (int s = 0; s < n_seg; ++s) { a.lo[s] = seg_lo[s]; a.hi[s] = seg_hi[s]; }
    const size_t bytes = (size_t)a.total * sizeof(float);
#define PACOH_MP_LAUNCH(nb, fp) do { \
        static std::atomic<uint64_t> attr_done{0}; \
        { const int rc_a = lds_opt_in((const void*)map_persist_kernel<nb, fp>, MP_LDS_BYTES, attr_done); if (rc_a != PACOH_OK) return rc_a; } \
        hipLaunchKernelGGL((map_persist_kernel<nb, fp>), dim3(1), dim3(MP_NT), bytes, (hipStream_t)stream, a); } while (0)
    if (NB == 1 && FP == 2) PACOH_MP_LAUNCH(1, 2);
    else if (NB == 1) PACOH_MP_LAUNCH(1, 4);
    else if (FP == 2) PACOH_MP_LAUNCH(2, 2);
    else PACOH_MP_LAUNCH(2, 4);
#undef PACOH_MP_LAUNCH
    return launch_status();
}
