// Task-fused PACOH-MAP iteration (round 5): forward of the networks, Gram / Cholesky / LML and its gradient, and the networks'
// backward pass of ONE task per workgroup in ONE launch, for task batches too large for the persistent one-workgroup kernel
// (map_persist.hip) -- BASELINE config #2: 256 tasks x 32 points per iteration.  As separate launches that iteration was
// forward (6.7 us) -> GP (8.0) -> backward (10.5) -> slab reduction + AdamW (5.0): four kernel latencies, each kernel a few hundred
// instructions per wave.  Here a workgroup keeps its task's activations in LDS between the three stages (map_net.h: the parameter
// image, the in-register MFMA chains, gp_reg_body.h for the GP) and writes ONE gradient slab per network in the layout the existing
// slab reduction reads (mlp_fused.hip, fused_reduce_slab_kernel: sum over tasks + AdamW + next batch + hyper-parameter transforms in its
// tail) -- an iteration is two launches.
// Reference lines replaced: GPR_meta_mll.py:104-117 (the task loop and backward), models.py:505-519, models.py:206-217.
#include "map_net.h"

namespace pacoh {

// mlp_fused.hip: the slab reduction as a launch of its own (what mlp_fused_bwd issues behind its backward kernel)
int fused_reduce_launch(const float* slab0, int wd0, long off0, const float* slab1, int wd1, long off1, int nets, float* d_theta,
                        long d_theta_stride, int slabs, const HyperBwdArgs<float>* tail, float* img_th, const int* img_map, hipStream_t s, int P = 1);

namespace {

constexpr int MT_NT = 512;           // 8 waves: 256 registers per lane (the two-block GP needs 90-112), cheaper barriers ...
constexpr int MT_NT_DEEP = 1024;     // ... 16 waves where the weight tiles would take 8 waves more than two rounds (two 4 x 32 networks: 32 tiles;
                                     // the kernel needs 102 registers there): PACOH-SVGD at the launchers' shape 0.0297 -> 0.0283 ms per step
constexpr int MT_MAXTASKS = 36;      // task descriptors that fit the kernel-argument segment beside the layer table

struct MtArgs {
    MpArgs p;                        // (first: mp_layer_karg reads the layer table at its offset in the kernel-argument segment)
    const float* bx; const float* by; const int32_t* bnv;            // the gathered batch [tb_total, n, d] / [tb_total, n] / [tb_total] | null
    const float* hyp_ls; const float* hyp_os; const float* hyp_noise;   // transformed hyper-parameters of the one parameter row
    float* slab[2]; int dnet[2]; int flat0[2];                       // per network: slabs [workgroups][dnet], first column of its block in theta
    float* lml_g; int32_t* info_g; float* dls_g; float* dos_g; float* dnz_g; float* dc_g;   // per-task GP outputs [tb_total (, f)]
    long* adv_counter;               // the pipelined feed's step counter: advanced by workgroup 0 (as mlp_fused_bwd_kernel does)
    const float* thimg;              // the parameter image in memory (map_task_setup_kernel; kept current by the slab reduction's AdamW)
    int tb_total;
    // MULTI (round 6, PACOH-SVGD / PACOH-VI: P parameter rows -- particles / posterior samples --, problem b = task * P + row, one
    // workgroup per (task group, row)): the rows' stride, the image's gather map (entry q of the image = element src_map[q] of a row,
    // -1: zero; a row changes every step, so every workgroup decodes its own image), and the SVGD step's distance matrix + counter
    // in extra workgroups behind the groups * P task workgroups (step_tail.h)
    long theta_stride; int P, groups;
    const int* src_map;
    SvgdDistTail<float> sv;
    int ntask[3];                    // tasks per phase of a FULL workgroup, planned by the host (mt_plan) ...
    MpTask plan[MT_MAXTASKS];        // ... phase 0 = plan[0 .. ntask[0]), phase 2 = plan[ntask[0] ..): the delta chains are phase 0's tasks again
};
static_assert(sizeof(MtArgs) <= 4096, "the kernel-argument segment holds 4 KB");

// The task descriptors of a workgroup, planned ONCE on the host and handed over in the kernel arguments (planned inside the kernel
// -- by one thread, then by 64 in parallel -- the plan cost 2-4 us of a 7 us kernel in every launch).  Chains (network, point tile)
// first: they are phase 0 and, as delta chains, phase 1; then per layer its weight tiles (flags bit 3: the tile also sums the bias).  The slab fields of a
// weight tile: dst = its first entry relative to the network's block, s_dst = in (row stride in theta), kmax = the bias entries of its
// rows, pad1 = the tile column that is the bias (none if >= 16), pad0 = the network.  A workgroup with fewer tasks than a full one
// skips the chains of point tiles it does not have; the MFMA steps of a weight tile follow from its own point count.
int mt_plan(MtArgs& ka) {
    const MpArgs& a = ka.p;
    const int nPt = (a.pts + 15) >> 4;
    int q = 0;
    for (int k = 0; k < a.nets; ++k)
        for (int Pt = 0; Pt < nPt; ++Pt) {
            if (q >= MT_MAXTASKS) return PACOH_ELIMIT;
            MpTask tk = {};
            tk.kind = MP_FWD; tk.w = k; tk.src = Pt; tk.n1 = a.nl[k];
            bool w32 = a.L[k][0].S <= 8 && a.L[k][a.nl[k] - 1].out <= 4;
            for (int l = 0; l + 1 < a.nl[k]; ++l) w32 = w32 && a.L[k][l].out == 32;
            tk.flags = w32 ? 4 : 0;
            ka.plan[q++] = tk;
        }
    ka.ntask[0] = ka.ntask[1] = q;
    for (int k = 0; k < a.nets; ++k)
        for (int l = 0; l < a.nl[k]; ++l) {
            const MpLayer& L = a.L[k][l];
            for (int J = 0; J < (L.out + 15) >> 4; ++J)
                for (int I = 0; I < (L.in + 15) >> 4; ++I) {
                    if (q >= MT_MAXTASKS) return PACOH_ELIMIT;
                    MpTask tk = {};
                    tk.kind = MP_WGRAD; tk.S = L.S; tk.pad0 = k; tk.w = L.w_lds + 16 * J * L.S + 16 * I;
                    tk.src = L.d_out + 16 * J; tk.s_src = L.s_d; tk.aux = (l == 0 ? 0 : L.a_in) + 16 * I; tk.flags = l == 0 ? 2 : 0;
                    tk.lim_a = L.out - 16 * J; tk.lim_b = L.S - 16 * I; tk.n2 = L.s_d - 16 * J;
                    tk.dst = (L.w_flat - ka.flat0[k]) + 16 * J * L.in + 16 * I; tk.s_dst = L.in;
                    tk.kmax = (L.b_flat - ka.flat0[k]) + 16 * J; tk.pad1 = L.in - 16 * I;
                    if (I == 0 && (L.in & 15) == 0) tk.flags |= 8;      // (no tile covers the bias column: this one sums it alongside)
                    ka.plan[q++] = tk;
                }
        }
    ka.ntask[2] = q - ka.ntask[0];
    return PACOH_OK;
}

// The parameter image in MEMORY (once per training call, and whenever theta was changed from outside): thimg[DP] as the kernels keep it
// in LDS, and for every network entry of theta its place in the image (-1 elsewhere) -- with it the slab reduction's AdamW step writes
// each updated entry into the image too, so that the task kernel's prologue is a copy.
// (MULTI: src_map[DP], the inverse map, instead: thimg == nullptr)
__global__ void __launch_bounds__(MT_NT) map_task_setup_kernel(MtArgs ka, float* thimg, int* img_map, int Dmax, int* src_map) {
    const MpArgs& a = ka.p;
    const int t = threadIdx.x;
    if (src_map) {
        for (int q = t; q < a.DP; q += MT_NT) src_map[q] = -1;
        __syncthreads();
        for (int k = 0; k < a.nets; ++k)
            for (int l = 0; l < a.nl[k]; ++l) {
                const MpLayer L = mp_layer_karg(k, l);
                for (int e = t; e < L.out * (L.in + 1); e += MT_NT) {
                    const int j = e / (L.in + 1), i = e - j * (L.in + 1);
                    src_map[L.w_lds + j * L.S + i] = i < L.in ? L.w_flat + j * L.in + i : L.b_flat + j;
                }
            }
        return;
    }
    for (int q = t; q < a.DP; q += MT_NT) thimg[q] = 0.0f;
    for (int q = t; q < Dmax; q += MT_NT) img_map[q] = -1;
    __syncthreads();
    for (int k = 0; k < a.nets; ++k)
        for (int l = 0; l < a.nl[k]; ++l) {
            const MpLayer L = mp_layer_karg(k, l);
            for (int e = t; e < L.out * (L.in + 1); e += MT_NT) {
                const int j = e / (L.in + 1), i = e - j * (L.in + 1);
                const int q = i < L.in ? L.w_flat + j * L.in + i : L.b_flat + j;
                const int li = L.w_lds + j * L.S + i;
                thimg[li] = a.theta[q];
                img_map[q] = li;
            }
        }
}

template <int NB, int FP, bool MULTI, int NT>
__global__ void __launch_bounds__(NT) map_task_kernel(MtArgs ka) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ __attribute__((aligned(16))) int ltab[2 * MP_MAXL * 16];
    const MpArgs& a = ka.p;
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int lane = t & 63, r16 = t & 15, g4 = (t >> 4) & 3;
    constexpr int NW = NT / 64;
    const int n = a.n, d = a.d, f = a.f;
    const int P = MULTI ? ka.P : 1;
    if (MULTI && (int)blockIdx.x >= ka.groups * P) {       // the SVGD step's distance matrix, snapshot and counter increment
        svgd_dist_tail<float>(ka.sv, (int)blockIdx.x - ka.groups * P, (int)gridDim.x - ka.groups * P);
        return;
    }
    const int grp = MULTI ? (int)blockIdx.x / P : (int)blockIdx.x;       // task group of this workgroup ...
    const int pp = MULTI ? (int)blockIdx.x - grp * P : 0;               // ... and its parameter row
    const float* trow = a.theta + (MULTI ? (long)pp * ka.theta_stride : 0L);
    const int task0 = grp * a.tb;
    const int tb = ka.tb_total - task0 < a.tb ? ka.tb_total - task0 : a.tb;      // tasks of this workgroup
    const int pts = tb * n;
    float* th = lds + a.o_th;
    float* hp = lds + a.o_hp;
    MpTask* tasks = reinterpret_cast<MpTask*>(lds + a.o_tasks);

#ifdef PACOH_MP_STAMPS
    if (t == 0) { mp_st_on = 1; mp_st_n[0] = mp_st_n[1] = 0; }
    __syncthreads();
    MP_STAMP();
#endif
    // ---- prologue (every launch pays it: 18 000 cycles in the first version, a third of the kernel): the layer table from the
    //      kernel-argument segment, LDS zeroed 16 bytes per store; then ONE memory round trip for everything else -- the parameter
    //      image as a flat loop over image entries (layer found by comparison, row by a float division: no per-layer loops with their
    //      dependent scalar loads), the task's points, the hyper-parameters -- while the last wave plans the tasks -------------------
    typedef const int __attribute__((address_space(4))) * kint_t;
    if (t < 2 * MP_MAXL * 16) {
        kint_t kt = (kint_t)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(MpArgs, L));
        ltab[t] = kt[t];
    }
    const int nchain = ka.ntask[0], nweight = ka.ntask[2];
    // MULTI: this row's image is gathered through a map (two dependent round trips).  The map words are requested HERE, in front of
    // the LDS zeroing and its barrier, and the row entries right behind that barrier in one burst: as a loop of load / gather / store
    // per 16 bytes the compiler made six dependent round trips of it -- 10 000-11 000 of the kernel's 52 000 cycles
    constexpr int GI = NT >= 1024 ? 4 : 7;                 // 7 x 448 (4 x 960) lanes x 4 = 12 544 (15 360) image entries >= two networks of five 32 x 36 layers
    constexpr int IT = NT - 64;                         // (the last wave copies the plan instead)
    const int ng = a.DP >> 2;
    int4 m4[GI];
    if (MULTI && wave != NW - 1) {
        const int4* sm = reinterpret_cast<const int4*>(ka.src_map);
#pragma unroll
        for (int u = 0; u < GI; ++u) { const int q = t + u * IT; m4[u] = sm[q < ng ? q : 0]; }
    }
    {
        float4* l4 = reinterpret_cast<float4*>(lds);
        for (int q = t; q < (a.total + 3) >> 2; q += NT) l4[q] = float4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    MP_STAMP();
    // (every global load of the prologue is issued before the first one is waited for: theta was written by the previous
    //  iteration's reduction on other XCDs -- a load of it is a fabric round trip of ~2 000 cycles, and three of them in a row per
    //  thread were most of the second version's prologue)
    const int epl = n * (d + 1);
    const bool mover = t < tb * epl;
    const int mv_s = mover ? t / epl : 0, mv_r = mover ? t - mv_s * epl : 0;
    float mv_val = 0.0f, hp_val = 0.0f;
    int nv_val = 0;
    if (mover) mv_val = mv_r < n * d ? ka.bx[(long)(task0 + mv_s) * (n * d) + mv_r] : ka.by[(long)(task0 + mv_s) * n + (mv_r - n * d)];
    if (t < tb && ka.bnv) nv_val = ka.bnv[task0 + t];
    {   // (ONE load through a selected pointer: as a chain of branches each arm waited for the loads in flight before it)
        const float* hsrc = t < f ? ka.hyp_ls + pp * f + t
                          : (t == 4 ? (ka.hyp_os ? ka.hyp_os + pp : nullptr)
                          : (t == 5 ? ka.hyp_noise + pp : ((t == 6 && a.off_const >= 0) ? trow + a.off_const : nullptr)));
        if (t < 7 && hsrc) hp_val = *hsrc;
    }
    if (wave == NW - 1) {                               // the last wave copies the plan out of the kernel arguments
        kint_t kt = (kint_t)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(MtArgs, plan));
        int* dst = reinterpret_cast<int*>(tasks);
        constexpr int PW = MT_MAXTASKS * 16 / 64;        // words per lane: all loads first, then the stores (ONE memory round trip)
        int w[PW];
        const int nw = (nchain + nweight) * 16;
#pragma unroll
        for (int u = 0; u < PW; ++u) { const int q = (t & 63) + 64 * u; w[u] = q < nw ? kt[q] : 0; }
#pragma unroll
        for (int u = 0; u < PW; ++u) { const int q = (t & 63) + 64 * u; if (q < nw) dst[q] = w[u]; }
    } else {
        const MpLayer* lt = reinterpret_cast<const MpLayer*>(ltab);
        const int nl0 = a.nets > 0 ? a.nl[0] : 0, nl1 = a.nets > 1 ? a.nl[1] : 0;
        // the constant-1 column of every layer's input activations, one (layer, point) per lane (pts <= 32): as a loop over the layers
        // with the points on the lanes it was ten dependent reads of the layer table, 4 000 cycles of every workgroup's prologue
        auto ones_columns = [&]() {
            for (int e = t; e < (nl0 + nl1) * 32; e += IT) {
                const int q = e >> 5, p = e & 31;
                if (p < pts) {
                    const MpLayer& L = lt[q < nl0 ? q : MP_MAXL + (q - nl0)];
                    const bool first = q == 0 || q == nl0;
                    lds[(first ? a.o_a0 : L.a_in) + p * L.S + L.in] = 1.0f;
                }
            }
        };
        // the parameter image, 16 bytes per lane from its copy in memory (decoding it from theta's layout cost every workgroup ~300
        // instructions per launch: 5 000 cycles)
        if (MULTI) {                                      // this row's image through the gather map: two round trips (the map is shared by all)
            float4* dst = reinterpret_cast<float4*>(th);
            float4 v[GI];
#pragma unroll
            for (int u = 0; u < GI; ++u) {
                v[u].x = trow[m4[u].x >= 0 ? m4[u].x : 0]; v[u].y = trow[m4[u].y >= 0 ? m4[u].y : 0];
                v[u].z = trow[m4[u].z >= 0 ? m4[u].z : 0]; v[u].w = trow[m4[u].w >= 0 ? m4[u].w : 0];
            }
            MP_STAMP();
            ones_columns();                               // (LDS-only work under the gather's round trip)
            MP_STAMP();
#pragma unroll
            for (int u = 0; u < GI; ++u) {
                const int q = t + u * IT;
                if (q < ng) {
                    float4 w;
                    w.x = m4[u].x >= 0 ? v[u].x : 0.0f; w.y = m4[u].y >= 0 ? v[u].y : 0.0f;
                    w.z = m4[u].z >= 0 ? v[u].z : 0.0f; w.w = m4[u].w >= 0 ? v[u].w : 0.0f;
                    dst[q] = w;
                }
            }
            MP_STAMP();
            const int4* sm = reinterpret_cast<const int4*>(ka.src_map);
            for (int q = t + GI * IT; q < ng; q += IT) {   // (never at the plan's sizes)
                const int4 mm = sm[q];
                float4 w;
                w.x = mm.x >= 0 ? trow[mm.x] : 0.0f; w.y = mm.y >= 0 ? trow[mm.y] : 0.0f;
                w.z = mm.z >= 0 ? trow[mm.z] : 0.0f; w.w = mm.w >= 0 ? trow[mm.w] : 0.0f;
                dst[q] = w;
            }
        } else {
            const float4* src = reinterpret_cast<const float4*>(ka.thimg);
            float4* dst = reinterpret_cast<float4*>(th);
            for (int q = t; q < a.DP >> 2; q += IT) dst[q] = src[q];
        }
        if (!MULTI) ones_columns();
    }
    MP_STAMP();
    if (t < 7 && (t < f || t >= 4)) hp[t] = hp_val;
    if (t < tb) lds[a.o_gl + t] = MULTI ? 1.0f : -1.0f;       // MAP: loss = -sum_t mll_t (GPR_meta_mll.py:109-113); SVGD / VI: the score of +sum_t mll_t
    if (mover) {
        if (mv_r < n * d) {
            const int i = mv_r / d, c = mv_r - i * d;
            lds[a.o_a0 + (mv_s * n + i) * a.S0 + c] = mv_val; lds[a.o_xs + (mv_s * n + i) * d + c] = mv_val;
        } else lds[a.o_y + mv_s * n + (mv_r - n * d)] = mv_val;
    }
    if (t < tb && ka.bnv) reinterpret_cast<int*>(lds + a.o_nv)[t] = nv_val;
    if (blockIdx.x == 0 && t == 0 && ka.adv_counter) *ka.adv_counter += 1;
    MP_STAMP();
    __syncthreads();

    const int a0_off = a.o_a0;
    auto run_phase = [&](int ph) {
        const int nt = ph == 2 ? nweight : nchain, first = ph == 2 ? nchain : 0;
        for (int q = wave; q < nt; q += NW) {
            const int4* e = reinterpret_cast<const int4*>(tasks + first + q);
            const int4 d0 = e[0], d1 = e[1], d2 = e[2], d3 = e[3];
            const int kind = ph == 1 ? MP_DELTA : sgi(d0.x);          // (phase 1 = the chains of phase 0, backwards)
            const bool w32 = sgi(d3.x) & 4;
            if (kind != MP_WGRAD && 16 * sgi(d0.w) >= pts) continue;      // (a point tile this workgroup's tasks do not reach)
            if (kind == MP_FWD) { if (w32) mp_fwd_chain32(ltab, sgi(d0.z), sgi(d2.z), sgi(d0.w), pts, th, lds, a0_off, r16, g4); else mp_fwd_chain(ltab, sgi(d0.z), sgi(d2.z), sgi(d0.w), pts, th, lds, a0_off, r16, g4); }
            else if (kind == MP_DELTA) { if (w32) mp_delta_chain32(ltab, sgi(d0.z), sgi(d2.z), sgi(d0.w), pts, th, lds, r16, g4); else mp_delta_chain(ltab, sgi(d0.z), sgi(d2.z), sgi(d0.w), pts, th, lds, r16, g4); }
            else if (kind == MP_WGRAD) {
                // the tile's 16 x 16 block of the weight-and-bias gradient into this workgroup's slab, in theta's own layout: weight
                // [j][i] row-major, then the bias column where the tile covers it (lane (r, g): row r, columns 4 g .. 4 g + 3)
                float bias;
                const f32x4 acc = mp_wgrad_acc(d0, d1, d2, d3, lds, a0_off, pts, r16, g4, bias);
                const int k = sgi(d3.z);
                float* sl = ka.slab[k] + (long)blockIdx.x * ka.dnet[k];
                if ((d3.x & 8) && g4 == 0 && r16 < d2.x) sl[d3.y + r16] = bias;
                if (r16 < d2.x) {
                    float* row = sl + d1.z + r16 * d1.w + 4 * g4;
                    if (4 * g4 + 3 < d3.w) {                // four consecutive entries of a weight row: one 16-byte store (4-byte aligned)
                        typedef float __attribute__((ext_vector_type(4), aligned(4))) f4u;
                        *reinterpret_cast<f4u*>(row) = f4u{acc[0], acc[1], acc[2], acc[3]};
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            const int c = 4 * g4 + s;
                            if (c < d3.w) row[s] = acc[s];
                            else if (c == d3.w) sl[d3.y + r16] = acc[s];
                        }
                    }
                }
            }
        }
    };

    MP_STAMP();
    if (a.nets > 0) { run_phase(0); __syncthreads(); }
    MP_STAMP();

    // ---- GP: one wave per task of the workgroup; its per-task outputs go straight to the arrays the hyper-parameter reduction reads --
    if (wave < tb) {
        GpMfmaArgs g;
        const int mean_mode = sg(a.mean_mode), kernel_nn = sg(a.kernel_nn), has_os = ka.hyp_os != nullptr;
        // the body indexes everything by b = the wave (the workgroup's LDS arrays); the global per-task arrays are handed over
        // already offset to this workgroup's first task
        g.z = sp(kernel_nn ? lds + a.o_zk : lds + a.o_xs); g.z_div = 1;
        g.mean = sp(mean_mode == PACOH_MEAN_VECTOR ? lds + a.o_mn : (mean_mode == PACOH_MEAN_CONST ? hp + 6 : nullptr));
        g.mean_mode = mean_mode;
        g.y = sp(lds + a.o_y); g.y_div = 1;
        g.ls = sp(hp); g.os = sp(has_os ? hp + 4 : nullptr); g.noise = sp(hp + 5);
        g.n_valid = sp(ka.bnv ? reinterpret_cast<int*>(lds + a.o_nv) : nullptr);
        g.g_lml = sp(lds + a.o_gl);
        g.d_z = sp(kernel_nn ? lds + a.o_dzk : nullptr);
        if (MULTI) {
            // (consecutive tasks of a row are P problems apart in the per-problem arrays: the body writes the workgroup's LDS slots, which
            //  the first threads scatter behind the next barrier)
            g.lml = sp(lds + a.o_lml); g.info = sp(reinterpret_cast<int32_t*>(lds + a.o_info));
            g.d_mean = sp(mean_mode == PACOH_MEAN_VECTOR ? lds + a.o_dmn : (mean_mode == PACOH_MEAN_CONST ? lds + a.o_dc : nullptr));
            g.d_ls = sp(lds + a.o_dls); g.d_os = sp(has_os ? lds + a.o_dos : nullptr); g.d_noise = sp(lds + a.o_dnz);
        } else {
            g.lml = ka.lml_g + task0; g.info = ka.info_g + task0;
            g.d_mean = mean_mode == PACOH_MEAN_VECTOR ? sp(lds + a.o_dmn) : (mean_mode == PACOH_MEAN_CONST ? ka.dc_g + task0 : nullptr);
            g.d_ls = ka.dls_g + (long)task0 * f; g.d_os = has_os ? ka.dos_g + task0 : nullptr; g.d_noise = ka.dnz_g + task0;
        }
        g.B = ka.tb_total; g.P = 1; g.n = sg(n); g.f = sg(f);
        // contexts of <= 8 points (the reference's demo: 5): one matrix entry per lane, no 16 x 16 blocks (gp8_body.h)
        bool small8 = false;
        if constexpr (NB == 1) small8 = sg(a.gp8) != 0;
        if (small8) gpreg::gp8_body<FP>(g, gpreg::WaveCtx{(unsigned)wave});
        else {
            constexpr int NP = 16 * NB;
            float* ws = lds + sg(a.o_gp) + wave * sg(a.gpw);
            gpreg::gp_reg_body<NB, FP, true, true>(g, gpreg::WaveCtx{(unsigned)wave}, ws, ws + NP * FP, ws + NP * FP + NP,
                                                   ws + NP * FP + 2 * NP, ws + NP * FP + 2 * NP, ws + NP * FP + 2 * NP + gpreg::GPR_SCR,
                                                   ws + 2 * NP * FP + 2 * NP + gpreg::GPR_SCR);
        }
    }
    MP_STAMP();
    if (a.nets > 0) {
        __syncthreads();
        if (MULTI && t < tb) {
            const long b = (long)(task0 + t) * P + pp;
            ka.lml_g[b] = lds[a.o_lml + t]; ka.info_g[b] = reinterpret_cast<const int32_t*>(lds + a.o_info)[t]; ka.dnz_g[b] = lds[a.o_dnz + t];
            if (ka.hyp_os) ka.dos_g[b] = lds[a.o_dos + t];
            if (a.mean_mode == PACOH_MEAN_CONST) ka.dc_g[b] = lds[a.o_dc + t];
            for (int c = 0; c < f; ++c) ka.dls_g[b * f + c] = lds[a.o_dls + t * f + c];
        }
        MP_STAMP();
        run_phase(1);
        __syncthreads();
        MP_STAMP();
        run_phase(2);
    }
    MP_STAMP();
#ifdef PACOH_MP_STAMPS
    if (t == 0 && blockIdx.x == 7 && ((ka.adv_counter && *ka.adv_counter == 3) || (MULTI && ka.sv.counter && *ka.sv.counter == 3)))
        for (int q = 1; q < mp_st_n[0]; ++q) printf("mt stamp %d: +%lld cycles\n", q, mp_st[0][q] - mp_st[0][q - 1]);
#endif
}

}  // namespace

// The launch pair of one iteration's likelihood + gradient + update: the task kernel, then the slab reduction with the step's tail
// (hyper-parameter reduction, AdamW, next batch).  -> PACOH_ELIMIT when the shape is outside the plan.
int map_task_launch(const void* theta, const void* bx, const void* by, const int32_t* bnv, int n, int d, int tb_total,
                    int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                    int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                    const void* hyp_ls, const void* hyp_os, const void* hyp_noise, void* workspace, size_t workspace_bytes,
                    void* d_theta, long d_theta_stride, const HyperBwdArgs<float>* tail_in, int plan_only, size_t* need_bytes, int D, hipStream_t stream,
                    int multi, int P, long theta_stride, const SvgdDistTail<float>* sv, int one_round_only) {
    // multi: P parameter rows of stride theta_stride (PACOH-SVGD's particles / PACOH-VI's posterior samples), hyp_* = [P, f] / [P],
    // problem b = task * P + row; sv: the SVGD step's distance tail | nullptr
    if (!multi) P = 1;
    if (P < 1) return PACOH_EINVAL;
    MtArgs ka;
    memset(&ka, 0, sizeof(ka));
    MpArgs& a = ka.p;
    int NB, FP;
    // tasks per workgroup: as many whole tasks as fit ONE 16-point tile (a chain per network and tile), at least one
    int tpw = n > 0 ? 16 / n : 1;
    if (tpw < 1) tpw = 1;
    if (tpw > MT_NT / 64) tpw = MT_NT / 64;
    if (tpw > tb_total) tpw = tb_total;
    const int rc = map_persist_plan(a, n, d, tpw, 1, mean_mode, off_mean, mean_hidden, n_mean_hidden, kernel_nn, off_kernel, kernel_hidden,
                                    n_kernel_hidden, f, &NB, &FP, false);
    if (rc != PACOH_OK) return rc;
    if (a.nets < 1 || tpw * n * (d + 1) > MT_NT) return PACOH_ELIMIT;
    const int groups = (tb_total + tpw - 1) / tpw;
    const int wgs = groups * P;
    // workspace: slabs [wgs][dnet] per network, then lml / d_ls (f) / d_os / d_noise / d_const [tb_total * P] and info
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    size_t o_slab[2] = {0, 0};
    for (int k = 0; k < a.nets; ++k) {
        const MpLayer& L0 = a.L[k][0]; const MpLayer& Ll = a.L[k][a.nl[k] - 1];
        ka.flat0[k] = L0.b_flat;
        ka.dnet[k] = Ll.w_flat + Ll.out * Ll.in - L0.b_flat;
        o_slab[k] = carve((size_t)wgs * ka.dnet[k] * sizeof(float));
    }
    if (mt_plan(ka) != PACOH_OK) return PACOH_ELIMIT;        // (needs flat0[]: the slab-relative entries)
    // threads per workgroup: 16 waves where 8 would walk the weight tiles in more than two rounds (PACOH_MT_NT=512 | 1024 forces one: A/B)
    const int nt = g_sw.mt_nt == MT_NT || g_sw.mt_nt == MT_NT_DEEP ? g_sw.mt_nt : (ka.ntask[2] > 2 * (MT_NT / 64) ? MT_NT_DEEP : MT_NT);
    const int Dmax = D;                                 // (the index map covers the whole parameter row)
    for (int k = 0; k < a.nets; ++k)
        if (plan_only != 1 && (ka.flat0[k] < 0 || ka.flat0[k] + ka.dnet[k] > D || (plan_only == 0 && ka.flat0[k] + ka.dnet[k] > d_theta_stride))) return PACOH_EINVAL;
    const size_t o_img = carve((size_t)a.DP * 4), o_map = carve(multi ? 256 : (size_t)Dmax * 4);      // (multi: o_img holds the gather map)
    const size_t B_ = (size_t)tb_total * P;
    const size_t o_lml = carve(B_ * 4), o_dls = carve(B_ * f * 4), o_dos = carve(B_ * 4), o_dnz = carve(B_ * 4), o_dc = carve(B_ * 4),
                 o_info = carve(B_ * 4);
    if (need_bytes) *need_bytes = off;
    if (plan_only == 1 && one_round_only) {
        // The task-fused launch wins while every workgroup is resident at once -- ONE round of workgroups, each a latency chain of ~20 us --
        // and loses to the throughput kernels as soon as a second round starts (profiles/r06_task_fused_crossover.txt: 4 x 32 networks at
        // n = 20, 139 KB of LDS = one workgroup per CU: 0.034 vs 0.069 ms at 160 problems, 0.061 vs 0.054 at 320).  Resident workgroups =
        // CUs x what the occupancy calculator says for this instantiation and LDS plan; no device (build host): no verdict, the plan decides.
        int dev = 0, cus = 0, per_cu = 0;
        const size_t lds_bytes = (size_t)a.total * sizeof(float);
        hipError_t e = hipGetDevice(&dev);
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
#define PACOH_MT_OCC(nb, fp, mu, nt_) do { \
            static std::atomic<uint64_t> attr_done{0}; \
            if (lds_opt_in((const void*)map_task_kernel<nb, fp, mu, nt_>, MP_LDS_BYTES, attr_done) != PACOH_OK) e = hipErrorUnknown; \
            else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, map_task_kernel<nb, fp, mu, nt_>, nt_, lds_bytes); } while (0)
#define PACOH_MT_OCC_NT(nb, fp, mu) do { if (nt == MT_NT_DEEP) PACOH_MT_OCC(nb, fp, mu, MT_NT_DEEP); else PACOH_MT_OCC(nb, fp, mu, MT_NT); } while (0)
        if (e == hipSuccess) {
            if (multi) { if (NB == 1 && FP == 2) PACOH_MT_OCC_NT(1, 2, true); else if (NB == 1) PACOH_MT_OCC_NT(1, 4, true); else if (FP == 2) PACOH_MT_OCC_NT(2, 2, true); else PACOH_MT_OCC_NT(2, 4, true); }
            else { if (NB == 1 && FP == 2) PACOH_MT_OCC_NT(1, 2, false); else if (NB == 1) PACOH_MT_OCC_NT(1, 4, false); else if (FP == 2) PACOH_MT_OCC_NT(2, 2, false); else PACOH_MT_OCC_NT(2, 4, false); }
        }
#undef PACOH_MT_OCC_NT
#undef PACOH_MT_OCC
        (void)hipGetLastError();
        if (e == hipSuccess && cus > 0 && per_cu > 0 && (long)wgs > (long)cus * per_cu) return PACOH_ELIMIT;
    }
    if (plan_only == 1) return PACOH_OK;
    if (!workspace || workspace_bytes < off) return PACOH_EINVAL;
    char* ws = (char*)workspace;
    a.theta = (float*)const_cast<void*>(theta);
    a.off_const = mean_mode == PACOH_MEAN_CONST ? off_mean : -1;
    ka.bx = (const float*)bx; ka.by = (const float*)by; ka.bnv = bnv;
    ka.hyp_ls = (const float*)hyp_ls; ka.hyp_os = (const float*)hyp_os; ka.hyp_noise = (const float*)hyp_noise;
    for (int k = 0; k < a.nets; ++k) ka.slab[k] = (float*)(ws + o_slab[k]);
    ka.lml_g = (float*)(ws + o_lml); ka.dls_g = (float*)(ws + o_dls); ka.dos_g = (float*)(ws + o_dos); ka.dnz_g = (float*)(ws + o_dnz);
    ka.dc_g = (float*)(ws + o_dc); ka.info_g = (int32_t*)(ws + o_info);
    ka.tb_total = tb_total;
    ka.thimg = (const float*)(ws + o_img);
    ka.theta_stride = theta_stride; ka.P = P; ka.groups = groups; ka.src_map = (const int*)(ws + o_img);
    if (plan_only == 2) {                               // the image and its index map (pacoh_map_task_setup) / the gather map (pacoh_svgd_task_setup)
        hipLaunchKernelGGL(map_task_setup_kernel, dim3(1), dim3(MT_NT), 0, stream, ka, multi ? nullptr : (float*)(ws + o_img), (int*)(ws + o_map), Dmax,
                           multi ? (int*)(ws + o_img) : nullptr);
        return launch_status();
    }
    int tail_wgs = 0;
    if (multi && sv && sv->X) {                         // one extra workgroup per particle pair of the lower triangle's rows (<= 256)
        ka.sv = *sv;
        tail_wgs = sv->P * sv->P < 256 ? sv->P * sv->P : 256;
    }
    HyperBwdArgs<float> tail = *tail_in;
    tail.d_ls = ka.dls_g; tail.d_os = hyp_os ? ka.dos_g : nullptr; tail.d_noise = ka.dnz_g; tail.d_const = mean_mode == PACOH_MEAN_CONST ? ka.dc_g : nullptr;
    tail.lml = tail.lik ? ka.lml_g : nullptr; tail.info = tail.fail_flag ? ka.info_g : nullptr;
    ka.adv_counter = multi ? nullptr : const_cast<long*>(tail.nx.counter);
    const size_t bytes = (size_t)a.total * sizeof(float);
#define PACOH_MT_LAUNCH_NT(nb, fp, mu, nt_) do { \
        static std::atomic<uint64_t> attr_done{0}; \
        { const int rc_a = lds_opt_in((const void*)map_task_kernel<nb, fp, mu, nt_>, MP_LDS_BYTES, attr_done); if (rc_a != PACOH_OK) return rc_a; } \
        hipLaunchKernelGGL((map_task_kernel<nb, fp, mu, nt_>), dim3((unsigned)(wgs + tail_wgs)), dim3(nt_), bytes, stream, ka); } while (0)
#define PACOH_MT_LAUNCH(nb, fp, mu) do { if (nt == MT_NT_DEEP) PACOH_MT_LAUNCH_NT(nb, fp, mu, MT_NT_DEEP); else PACOH_MT_LAUNCH_NT(nb, fp, mu, MT_NT); } while (0)
#define PACOH_MT_PICK(mu) do { \
        if (NB == 1 && FP == 2) PACOH_MT_LAUNCH(1, 2, mu); \
        else if (NB == 1) PACOH_MT_LAUNCH(1, 4, mu); \
        else if (FP == 2) PACOH_MT_LAUNCH(2, 2, mu); \
        else PACOH_MT_LAUNCH(2, 4, mu); } while (0)
    if (multi) PACOH_MT_PICK(true); else PACOH_MT_PICK(false);
#undef PACOH_MT_PICK
#undef PACOH_MT_LAUNCH
#undef PACOH_MT_LAUNCH_NT
    if (launch_status() != PACOH_OK) return PACOH_ELAUNCH;
    return fused_reduce_launch(ka.slab[0], ka.dnet[0], ka.flat0[0], a.nets > 1 ? ka.slab[1] : nullptr, a.nets > 1 ? ka.dnet[1] : 0,
                               a.nets > 1 ? ka.flat0[1] : 0, a.nets, (float*)d_theta, d_theta_stride, groups, &tail,
                               multi ? nullptr : (float*)(ws + o_img), multi ? nullptr : (const int*)(ws + o_map), stream, P);
}

}  // namespace pacoh
