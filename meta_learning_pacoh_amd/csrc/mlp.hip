// C-ABI entry points of the per-particle MLP.  Three implementations, picked per call by shape and dtype (round 5 removed the fourth, a
// thread-per-point VALU kernel: on the 18 shapes of tests/test_gpu_kernels.py it was never the fastest where anything else applied but
// a network without hidden layers, and the dispatcher's order cost fp64 networks 2-6x against `layers`: profiles/r05_mlp_paths.txt):
//   fused   (mlp_fused.hip)  fp32, 1-4 hidden layers of width <= 32, d_in <= 4, d_out <= 2: register-resident MFMA kernels,
//                            one launch for the mean AND the kernel-feature network (pacoh_mlp2_*)
//   mfma    (mlp_mfma.hip)   fp32, 1-2 hidden layers of width <= 32, d_in <= 16, d_out <= 8: the same idea with padded io layers
//   layers  (mlp_layers.hip) everything else (any depth, any width, fp32/fp64): layer-by-layer MFMA GEMMs through a workspace
// PACOH_MLP_PATH=fused|mfma|layers restricts the choice to that path and the ones after it (tests, A/B timing).
#include "common.h"
#include "hyper_tail.h"
#include "step_tail.h"
#include <stdlib.h>
#include <string.h>

namespace pacoh {
// mlp_mfma.hip; return 1 if not applicable
int mlp_mfma_fwd(const void*, int, const void*, long, int, int, const int32_t*, int, int, void*, int, int, hipStream_t);
size_t mlp_mfma_bwd_workspace(int, int, int, int, const int32_t*, int, int);
int mlp_mfma_bwd(const void*, int, const void*, long, int, int, const int32_t*, int, int, const void*, void*, long, int, void*, int, int, hipStream_t);
bool mlp_mfma_applicable(int d_in, const int32_t* hidden, int n_hidden, int d_out);
// mlp_fused.hip
bool mlp_fused_applicable(int d_in, const int32_t* hidden, int n_hidden, int d_out);
int mlp_fused_fwd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in, const int32_t* hidden,
                  int n_hidden, int nets, const long* off, const int* d_out, void* const* out, void* stash, int B, int n,
                  hipStream_t s, const SvgdDistTail<float>* tail = nullptr, bool* tail_taken = nullptr);
size_t mlp_fused_stash_bytes(int B, int P, int n, int n_hidden, int nets);
size_t mlp_fused_bwd_workspace(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out, int nets);
int mlp_fused_bwd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in, const int32_t* hidden,
                  int n_hidden, int nets, const long* off, const int* d_out, const void* const* g_out, void* d_theta,
                  long d_theta_stride, int accumulate, void* workspace, const void* stash, int B, int n, hipStream_t s,
                  const HyperBwdArgs<float>* tail, long col_base = 0);
// mlp_layers.hip
size_t mlp_layers_workspace(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out, int dtype, int bwd);
size_t mlp_layers_stash_bytes(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out, int dtype);
int mlp_layers_fwd(const void*, int, const void*, long, int, int, const int32_t*, int, int, void*, void*, int, int, int, hipStream_t, void* stash = nullptr);
int mlp_layers_bwd(const void*, int, const void*, long, int, int, const int32_t*, int, int, const void*, void*, long, int, void*, int, int, int, hipStream_t,
                   const void* stash = nullptr);

// out[o] (+)= scale * sum_c in[c, o]; one wavefront per output element, lanes stride over c, fixed order
template <typename T>
__global__ void __launch_bounds__(256) reduce_tasks_kernel(const T* __restrict__ in, T* __restrict__ out, T scale, int accumulate,
                                                           int C, int P, int Wd) {
    const long tot = (long)P * Wd;
    const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= tot) return;
    const int lane = threadIdx.x & 63;
    T s = 0;
    for (int c = lane; c < C; c += 64) s += in[(long)c * tot + idx];
    s = subwave_sum<T>(s, 64);
    if (lane == 0) out[idx] = accumulate ? out[idx] + scale * s : scale * s;
}

enum MlpPath { PATH_FUSED = 0, PATH_MFMA = 1, PATH_LAYERS = 3 };

static int first_allowed_path() { return g_sw.mlp_path; }      // (PACOH_MLP_PATH: switches.h; tests switch it through pacoh_reload_env)

static int args_ok(int d_in, const int32_t* hidden, int n_hidden, int d_out) {
    if (d_in <= 0 || d_out <= 0 || n_hidden < 0 || (n_hidden > 0 && !hidden)) return PACOH_EINVAL;
    if (n_hidden > PACOH_MLP_MAX_HIDDEN_LAYERS) return PACOH_ELIMIT;
    for (int l = 0; l < n_hidden; ++l) {
        if (hidden[l] <= 0) return PACOH_EINVAL;
        if (hidden[l] > PACOH_MLP_MAX_WIDTH) return PACOH_ELIMIT;
    }
    return PACOH_OK;
}

static MlpPath pick_path(int dtype, int d_in, const int32_t* hidden, int n_hidden, int d_out, long rows_total) {
    const int first = first_allowed_path();
    const bool f32 = dtype == PACOH_F32 && rows_total <= 0x3fffffffL;
    if (first <= PATH_FUSED && f32 && mlp_fused_applicable(d_in, hidden, n_hidden, d_out)) return PATH_FUSED;
    if (first <= PATH_MFMA && f32 && mlp_mfma_applicable(d_in, hidden, n_hidden, d_out)) return PATH_MFMA;
    return PATH_LAYERS;
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
}  // namespace pacoh

using namespace pacoh;

extern "C" size_t pacoh_mlp_fwd_workspace_bytes(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out,
                                                int dtype) {
    if (P <= 0 || B <= 0 || n <= 0 || args_ok(d_in, hidden, n_hidden, d_out)) return 0;
    if (pick_path(dtype, d_in, hidden, n_hidden, d_out, (long)B * n) != PATH_LAYERS) return 0;
    return mlp_layers_workspace(B, P, n, d_in, hidden, n_hidden, d_out, dtype, 0);
}

static int mlp_fwd_impl(const void* x, int x_div, const void* theta, long theta_stride, int P,
                        int d_in, const int32_t* hidden, int n_hidden, int d_out, void* out, void* workspace, void* stash,
                        int B, int n, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!out || !x || !theta || x_div <= 0 || P <= 0 || B <= 0 || n <= 0 || B % P != 0) return PACOH_EINVAL;
    int rc = args_ok(d_in, hidden, n_hidden, d_out);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (pick_path(dtype, d_in, hidden, n_hidden, d_out, (long)B * n)) {
    case PATH_FUSED: {
        const long off = 0;
        void* const outs[1] = {out};
        return mlp_fused_fwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, 1, &off, &d_out, outs, stash, B, n, s);
    }
    case PATH_MFMA:
        return mlp_mfma_fwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, out, B, n, s);
    default:
        return mlp_layers_fwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, out, workspace, B, n, dtype, s, stash);
    }
}


extern "C" int pacoh_mlp_fwd(const void* x, int x_div, const void* theta, long theta_stride, int P,
                             int d_in, const int32_t* hidden, int n_hidden, int d_out, void* out, void* workspace,
                             int B, int n, int dtype, void* stream) {
    return mlp_fwd_impl(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, out, workspace, nullptr, B, n, dtype, stream);
}

// the activation stash of pacoh_mlp2_fwd / pacoh_mlp2_bwd for ONE network: bytes (0: this shape keeps none), and the forward that fills it
extern "C" size_t pacoh_mlp_stash_bytes(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out, int dtype) {
    if (P <= 0 || B <= 0 || n <= 0 || B % P != 0 || args_ok(d_in, hidden, n_hidden, d_out)) return 0;
    const MlpPath path = pick_path(dtype, d_in, hidden, n_hidden, d_out, (long)B * n);
    // (round 6) the layer-by-layer path keeps its packed weights and hidden activations for the backward too: 9 launches less per
    // network and step (5 repacks -> 0, 4 recomputed layers -> 0 at four hidden layers)
    if (path == PATH_LAYERS) return align256(mlp_layers_stash_bytes(B, P, n, d_in, hidden, n_hidden, d_out, dtype));
    if (path != PATH_FUSED) return 0;
    return mlp_fused_stash_bytes(B, P, n, n_hidden, 1);
}
extern "C" int pacoh_mlp_fwd_stash(const void* x, int x_div, const void* theta, long theta_stride, int P,
                                   int d_in, const int32_t* hidden, int n_hidden, int d_out, void* out, void* workspace, void* stash,
                                   int B, int n, int dtype, void* stream) {
    return mlp_fwd_impl(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, out, workspace, stash, B, n, dtype, stream);
}

extern "C" size_t pacoh_mlp_bwd_workspace_bytes(int B, int P, int n, int d_in, const int32_t* hidden,
                                                int n_hidden, int d_out, int dtype) {
    if (P <= 0 || B <= 0 || n <= 0 || args_ok(d_in, hidden, n_hidden, d_out)) return 0;
    switch (pick_path(dtype, d_in, hidden, n_hidden, d_out, (long)B * n)) {
    case PATH_FUSED: return mlp_fused_bwd_workspace(B, P, n, d_in, hidden, n_hidden, d_out, 1);
    case PATH_MFMA: return mlp_mfma_bwd_workspace(B, P, n, d_in, hidden, n_hidden, d_out);
    default: return mlp_layers_workspace(B, P, n, d_in, hidden, n_hidden, d_out, dtype, 1);
    }
}

extern "C" int pacoh_mlp_bwd(const void* x, int x_div, const void* theta, long theta_stride, int P,
                             int d_in, const int32_t* hidden, int n_hidden, int d_out, const void* g_out,
                             void* d_theta, long d_theta_stride, int accumulate, void* workspace,
                             int B, int n, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!g_out || !d_theta || !workspace || !x || !theta || x_div <= 0 || P <= 0 || B <= 0 || n <= 0 || B % P != 0) return PACOH_EINVAL;
    int rc = args_ok(d_in, hidden, n_hidden, d_out);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (pick_path(dtype, d_in, hidden, n_hidden, d_out, (long)B * n)) {
    case PATH_FUSED: {
        const long off = 0;
        const void* const gs[1] = {g_out};
        return mlp_fused_bwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, 1, &off, &d_out, gs, d_theta, d_theta_stride,
                             accumulate, workspace, nullptr, B, n, s, nullptr);
    }
    case PATH_MFMA:
        return mlp_mfma_bwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, g_out, d_theta, d_theta_stride,
                            accumulate, workspace, B, n, s);
    default:
        return mlp_layers_bwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, g_out, d_theta, d_theta_stride,
                              accumulate, workspace, B, n, dtype, s);
    }
}

// ---- two networks of the same hidden shape on the same inputs (the mean and the kernel-feature network of a step) --------
extern "C" size_t pacoh_mlp2_fwd_workspace_bytes(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out_a,
                                                 int d_out_b, int dtype) {
    const size_t a = pacoh_mlp_fwd_workspace_bytes(B, P, n, d_in, hidden, n_hidden, d_out_a, dtype);
    const size_t b = pacoh_mlp_fwd_workspace_bytes(B, P, n, d_in, hidden, n_hidden, d_out_b, dtype);
    return a > b ? a : b;                       // the two networks run one after the other on the general path
}

extern "C" size_t pacoh_mlp2_stash_bytes(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out_a,
                                         int d_out_b, int dtype) {
    if (P <= 0 || B <= 0 || n <= 0 || B % P != 0 || args_ok(d_in, hidden, n_hidden, d_out_a) || args_ok(d_in, hidden, n_hidden, d_out_b)) return 0;
    const MlpPath pa = pick_path(dtype, d_in, hidden, n_hidden, d_out_a, (long)B * n), pb = pick_path(dtype, d_in, hidden, n_hidden, d_out_b, (long)B * n);
    if (pa == PATH_LAYERS && pb == PATH_LAYERS)          // (one stash per network, back to back: mlp2_layers_stash)
        return align256(mlp_layers_stash_bytes(B, P, n, d_in, hidden, n_hidden, d_out_a, dtype)) +
               align256(mlp_layers_stash_bytes(B, P, n, d_in, hidden, n_hidden, d_out_b, dtype));
    if (pa != PATH_FUSED || pb != PATH_FUSED) return 0;
    return mlp_fused_stash_bytes(B, P, n, n_hidden, 2);
}

static int mlp2_fwd_impl(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                         const int32_t* hidden, int n_hidden, long off_a, int d_out_a, void* out_a, long off_b, int d_out_b,
                         void* out_b, void* workspace, void* stash, int B, int n, int dtype, void* stream,
                         const SvgdDistTail<float>* tail, bool* tail_done) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!out_a || !out_b || !x || !theta || x_div <= 0 || P <= 0 || B <= 0 || n <= 0 || B % P != 0 || off_a < 0 || off_b < 0) return PACOH_EINVAL;
    int rc = args_ok(d_in, hidden, n_hidden, d_out_a);
    if (rc || (rc = args_ok(d_in, hidden, n_hidden, d_out_b))) return rc;
    if (pick_path(dtype, d_in, hidden, n_hidden, d_out_a, (long)B * n) == PATH_FUSED &&
        pick_path(dtype, d_in, hidden, n_hidden, d_out_b, (long)B * n) == PATH_FUSED) {
        const long off[2] = {off_a, off_b};
        const int dout[2] = {d_out_a, d_out_b};
        void* const outs[2] = {out_a, out_b};
        return mlp_fused_fwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, 2, off, dout, outs, stash, B, n, (hipStream_t)stream, tail,
                             tail_done);
    }
    const size_t es = dtype == PACOH_F64 ? 8 : 4;
    void* st_a = nullptr; void* st_b = nullptr;          // the layer-by-layer path's per-network stashes (pacoh_mlp2_stash_bytes)
    if (stash && pick_path(dtype, d_in, hidden, n_hidden, d_out_a, (long)B * n) == PATH_LAYERS &&
        pick_path(dtype, d_in, hidden, n_hidden, d_out_b, (long)B * n) == PATH_LAYERS) {
        st_a = stash;
        st_b = (char*)stash + align256(mlp_layers_stash_bytes(B, P, n, d_in, hidden, n_hidden, d_out_a, dtype));
    }
    rc = mlp_fwd_impl(x, x_div, (const char*)theta + off_a * es, theta_stride, P, d_in, hidden, n_hidden, d_out_a, out_a, workspace, st_a, B, n, dtype, stream);
    if (rc) return rc;
    return mlp_fwd_impl(x, x_div, (const char*)theta + off_b * es, theta_stride, P, d_in, hidden, n_hidden, d_out_b, out_b, workspace, st_b, B, n, dtype, stream);
}

extern "C" int pacoh_mlp2_fwd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                              const int32_t* hidden, int n_hidden, long off_a, int d_out_a, void* out_a, long off_b, int d_out_b,
                              void* out_b, void* workspace, void* stash, int B, int n, int dtype, void* stream) {
    return mlp2_fwd_impl(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, off_a, d_out_a, out_a, off_b, d_out_b, out_b, workspace,
                         stash, B, n, dtype, stream, nullptr, nullptr);
}

extern "C" int pacoh_svgd_dist_advance(const void* X, void* workspace, int P, int D, int64_t* counter, int dtype, void* stream);

// pacoh_mlp2_fwd + pacoh_svgd_dist_advance: the first launch of a pipelined SVGD step (step_tail.h).  On the fused fp32 path the
// particles' distance matrix, their snapshot and the step counter's increment run in extra workgroups of the forward launch;
// elsewhere the two calls in sequence.
extern "C" int pacoh_mlp2_fwd_svgd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                                   const int32_t* hidden, int n_hidden, long off_a, int d_out_a, void* out_a, long off_b, int d_out_b,
                                   void* out_b, void* workspace, void* stash, int B, int n,
                                   const void* svgd_X, void* svgd_workspace, int svgd_P, int svgd_D, int64_t* counter,
                                   int dtype, void* stream) {
    if (!svgd_X || !svgd_workspace || svgd_P <= 0 || svgd_D <= 0) return PACOH_EINVAL;
    if (svgd_P > PACOH_SVGD_MAX_PARTICLES) return PACOH_ELIMIT;
    bool tail_done = false;
    int rc;
    if (dtype == PACOH_F32) {
        float* d2 = (float*)svgd_workspace;
        SvgdDistTail<float> tail = {(const float*)svgd_X, d2, d2 + svgd_P * svgd_P, svgd_P, svgd_D, (long*)counter};
        rc = mlp2_fwd_impl(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, off_a, d_out_a, out_a, off_b, d_out_b, out_b,
                           workspace, stash, B, n, dtype, stream, &tail, &tail_done);
    } else {
        rc = mlp2_fwd_impl(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, off_a, d_out_a, out_a, off_b, d_out_b, out_b,
                           workspace, stash, B, n, dtype, stream, nullptr, nullptr);
    }
    if (rc || tail_done) return rc;
    return pacoh_svgd_dist_advance(svgd_X, svgd_workspace, svgd_P, svgd_D, counter, dtype, stream);
}

extern "C" size_t pacoh_mlp2_bwd_workspace_bytes(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out_a,
                                                 int d_out_b, int dtype) {
    if (P <= 0 || B <= 0 || n <= 0 || args_ok(d_in, hidden, n_hidden, d_out_a) || args_ok(d_in, hidden, n_hidden, d_out_b)) return 0;
    if (pick_path(dtype, d_in, hidden, n_hidden, d_out_a, (long)B * n) == PATH_FUSED &&
        pick_path(dtype, d_in, hidden, n_hidden, d_out_b, (long)B * n) == PATH_FUSED)
        return mlp_fused_bwd_workspace(B, P, n, d_in, hidden, n_hidden, d_out_a, 2) + mlp_fused_bwd_workspace(B, P, n, d_in, hidden, n_hidden, d_out_b, 2);
    const size_t a = pacoh_mlp_bwd_workspace_bytes(B, P, n, d_in, hidden, n_hidden, d_out_a, dtype);
    const size_t b = pacoh_mlp_bwd_workspace_bytes(B, P, n, d_in, hidden, n_hidden, d_out_b, dtype);
    return align256(a > b ? a : b);
}

// pacoh_adam_inline (include/pacoh_gp.h) -> the device-side struct of the fused path
static bool adam_inline_ok(const pacoh_adam_inline* o, int P, const void* lml) {
    return o->param && o->exp_avg && o->exp_avg_sq && o->scalars && o->n_seg >= 0 && o->n_seg <= 4 && P == 1 && lml != nullptr;
}
static AdamInline<float> adam_inline_f32(const pacoh_adam_inline* o) {
    AdamInline<float> a = {(float*)o->param, (float*)o->exp_avg, (float*)o->exp_avg_sq, (const float*)o->scalars, (float)(1.0 - o->beta1),
                           (float)o->beta2, (float)(1.0 - o->beta2), o->n_seg, {0, 0, 0, 0}, {0, 0, 0, 0}, (long*)o->step_counter, (float*)o->loss_cum,
                           nullptr, nullptr, 0};
    for (int k = 0; k < o->n_seg; ++k) { a.lo[k] = o->seg_lo[k]; a.hi[k] = o->seg_hi[k]; }
    if (o->next) {                                   // pipelined feed: scalars by the counter, which the backward launch advances
        a.counter = (const long*)o->next->counter; a.sc2 = (const float*)o->next->sc2; a.n_sc = o->next->n_sc;
        a.step_counter = nullptr;
    }
    return a;
}
static bool step_next_ok(const pacoh_step_next* x) {
    return x->counter && x->sc2 && x->sc_all && x->n_sc >= PACOH_SC_COUNT && x->tb > 0 && x->idx_all && x->x && x->y && x->out_x && x->out_y &&
           x->n > 0 && x->d > 0 && (x->n_valid == nullptr) == (x->out_n_valid == nullptr) && (x->ls == nullptr || x->noise != nullptr);
}
// off_ls .. off_noise: the hyper-parameter layout of the call (the transforms the next step needs are of the entries this call updates)
static StepNextArgs<float> step_next_f32(const pacoh_step_next* x, int off_ls, int f, int off_os, int off_noise) {
    return StepNextArgs<float>{(const long*)x->counter, (float*)x->sc2, x->n_sc, (const long*)x->idx_all, x->tb, (const float*)x->sc_all,
                               (const float*)x->x, (const float*)x->y, x->n_valid, (float*)x->out_x, (float*)x->out_y, x->out_n_valid,
                               x->n * x->d, x->n, off_ls, features_of(f), off_os, off_noise, kernel_of(f) != PACOH_KERNEL_RBF,
                               (float)x->noise_floor, (float*)x->ls, (float*)x->os, (float*)x->noise, nullptr};
}
// the same step as separate launches (paths without the fused slab reduction): one pacoh_adam_step_dev per trained segment, the last
// one advancing the feed's counter and adding the loss to its running sum
static int adam_inline_fallback(const pacoh_adam_inline* o, const void* grad_rows, const void* loss, int dtype, void* stream) {
    const size_t es = dtype == PACOH_F64 ? 8 : 4;
    for (int k = 0; k < o->n_seg; ++k) {
        const bool last = k == o->n_seg - 1;
        const long lo = o->seg_lo[k], cnt = o->seg_hi[k] - lo;
        if (cnt <= 0) continue;
        const int rc = pacoh_adam_step_dev((char*)o->param + lo * es, (const char*)grad_rows + lo * es, (char*)o->exp_avg + lo * es,
                                           (char*)o->exp_avg_sq + lo * es, o->scalars, o->beta1, o->beta2, cnt,
                                           last ? o->step_counter : nullptr, last ? o->loss_cum : nullptr, last ? loss : nullptr, dtype, stream);
        if (rc) return rc;
    }
    return PACOH_OK;
}

static int mlp2_bwd_impl(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                         const int32_t* hidden, int n_hidden, long off_a, int d_out_a, const void* g_a, long off_b,
                         int d_out_b, const void* g_b, void* d_theta, long d_theta_stride, int accumulate,
                         void* workspace, const void* stash, int B, int n, int dtype, void* stream,
                         const HyperBwdArgs<float>* tail, bool* tail_done) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!g_a || !g_b || !d_theta || !workspace || !x || !theta || x_div <= 0 || P <= 0 || B <= 0 || n <= 0 || B % P != 0 || off_a < 0 || off_b < 0)
        return PACOH_EINVAL;
    int rc = args_ok(d_in, hidden, n_hidden, d_out_a);
    if (rc || (rc = args_ok(d_in, hidden, n_hidden, d_out_b))) return rc;
    if (pick_path(dtype, d_in, hidden, n_hidden, d_out_a, (long)B * n) == PATH_FUSED &&
        pick_path(dtype, d_in, hidden, n_hidden, d_out_b, (long)B * n) == PATH_FUSED) {
        const long off[2] = {off_a, off_b};
        const int dout[2] = {d_out_a, d_out_b};
        const void* const gs[2] = {g_a, g_b};
        if (tail_done) *tail_done = tail != nullptr;
        return mlp_fused_bwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, 2, off, dout, gs, d_theta, d_theta_stride,
                             accumulate, workspace, stash, B, n, (hipStream_t)stream, tail);
    }
    const size_t es = dtype == PACOH_F64 ? 8 : 4;
    if (stash && pick_path(dtype, d_in, hidden, n_hidden, d_out_a, (long)B * n) == PATH_LAYERS &&
        pick_path(dtype, d_in, hidden, n_hidden, d_out_b, (long)B * n) == PATH_LAYERS) {
        // the forward of the same step left each network's packed weights and hidden activations in its half of the stash
        const void* st_b = (const char*)stash + align256(mlp_layers_stash_bytes(B, P, n, d_in, hidden, n_hidden, d_out_a, dtype));
        rc = mlp_layers_bwd(x, x_div, (const char*)theta + off_a * es, theta_stride, P, d_in, hidden, n_hidden, d_out_a, g_a,
                            (char*)d_theta + off_a * es, d_theta_stride, accumulate, workspace, B, n, dtype, (hipStream_t)stream, stash);
        if (rc) return rc;
        return mlp_layers_bwd(x, x_div, (const char*)theta + off_b * es, theta_stride, P, d_in, hidden, n_hidden, d_out_b, g_b,
                              (char*)d_theta + off_b * es, d_theta_stride, accumulate, workspace, B, n, dtype, (hipStream_t)stream, st_b);
    }
    rc = pacoh_mlp_bwd(x, x_div, (const char*)theta + off_a * es, theta_stride, P, d_in, hidden, n_hidden, d_out_a, g_a,
                       (char*)d_theta + off_a * es, d_theta_stride, accumulate, workspace, B, n, dtype, stream);
    if (rc) return rc;
    return pacoh_mlp_bwd(x, x_div, (const char*)theta + off_b * es, theta_stride, P, d_in, hidden, n_hidden, d_out_b, g_b,
                         (char*)d_theta + off_b * es, d_theta_stride, accumulate, workspace, B, n, dtype, stream);
}

extern "C" int pacoh_mlp2_bwd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                              const int32_t* hidden, int n_hidden, long off_a, int d_out_a, const void* g_a, long off_b,
                              int d_out_b, const void* g_b, void* d_theta, long d_theta_stride, int accumulate,
                              void* workspace, const void* stash, int B, int n, int dtype, void* stream) {
    return mlp2_bwd_impl(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, off_a, d_out_a, g_a, off_b, d_out_b, g_b, d_theta,
                         d_theta_stride, accumulate, workspace, stash, B, n, dtype, stream, nullptr, nullptr);
}

// pacoh_mlp2_bwd + pacoh_hyper_bwd: the whole gradient epilogue of a step.  On the fused fp32 path the hyper-parameter
// reduction runs in extra workgroups of the slab-reduction launch (one launch less); elsewhere the two calls in sequence.
extern "C" int pacoh_mlp2_bwd_hyper(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                                    const int32_t* hidden, int n_hidden, long off_a, int d_out_a, const void* g_a, long off_b,
                                    int d_out_b, const void* g_b, void* d_theta, long d_theta_stride, int accumulate,
                                    void* workspace, const void* stash, int B, int n,
                                    int T_, int off_ls, int f, int off_os, int off_noise, int off_const, const void* d_ls,
                                    const void* d_os, const void* d_noise, const void* d_const, const void* lml, void* lik,
                                    double lik_scale, const int32_t* info, int32_t* fail_flag,
                                    void* svgd_workspace, int svgd_P, int svgd_D, const pacoh_adam_inline* opt, int dtype, void* stream) {
    if (svgd_workspace && (svgd_P <= 0 || svgd_D <= 0)) return PACOH_EINVAL;
    if (svgd_workspace && svgd_P > 64) return PACOH_ELIMIT;
    if (opt && (!adam_inline_ok(opt, P, lml) || opt->n_seg < 1)) return PACOH_EINVAL;
    if (opt && opt->next) {
        if (!step_next_ok(opt->next)) return PACOH_EINVAL;
        if (dtype != PACOH_F32 || args_ok(d_in, hidden, n_hidden, d_out_a) || args_ok(d_in, hidden, n_hidden, d_out_b) ||
            pick_path(dtype, d_in, hidden, n_hidden, d_out_a, (long)B * n) != PATH_FUSED ||
            pick_path(dtype, d_in, hidden, n_hidden, d_out_b, (long)B * n) != PATH_FUSED) return PACOH_ELIMIT;     // (pacoh_mlp_fused_path)
    }
    if (!d_ls || !d_noise || T_ <= 0 || features_of(f) <= 0 || (lml == nullptr) != (lik == nullptr)) return PACOH_EINVAL;
    if (accumulate) return PACOH_EINVAL;              // (the tail writes its columns of d_theta; the blocks of the two networks are overwritten)
    bool tail_done = false;
    int rc;
    if (dtype == PACOH_F32) {
        HyperBwdArgs<float> tail = {(const float*)theta, theta_stride, P, T_, off_ls, features_of(f), off_os, off_noise, off_const,
                                    (const float*)d_ls, (const float*)d_os, (const float*)d_noise, (const float*)d_const, (float*)d_theta,
                                    d_theta_stride, (const float*)lml, (float*)lik, (float)lik_scale, info, fail_flag,
                                    kernel_of(f) != PACOH_KERNEL_RBF, (const float*)svgd_workspace, svgd_P,
                                    svgd_workspace ? (float*)svgd_workspace + svgd_bw_slot(svgd_P, svgd_D) : nullptr,
                                    opt ? adam_inline_f32(opt) : AdamInline<float>{},
                                    (opt && opt->next) ? step_next_f32(opt->next, off_ls, f, off_os, off_noise) : StepNextArgs<float>{}};
        rc = mlp2_bwd_impl(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, off_a, d_out_a, g_a, off_b, d_out_b, g_b, d_theta,
                           d_theta_stride, accumulate, workspace, stash, B, n, dtype, stream, &tail, &tail_done);
    } else {
        rc = mlp2_bwd_impl(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, off_a, d_out_a, g_a, off_b, d_out_b, g_b, d_theta,
                           d_theta_stride, accumulate, workspace, stash, B, n, dtype, stream, nullptr, nullptr);
    }
    if (rc || tail_done) return rc;
    rc = pacoh_hyper_bwd(theta, theta_stride, P, T_, off_ls, f, off_os, off_noise, off_const, d_ls, d_os, d_noise, d_const, d_theta,
                         d_theta_stride, lml, lik, lik_scale, info, fail_flag, svgd_workspace, svgd_P, svgd_D, nullptr, dtype, stream);
    if (rc || !opt) return rc;
    return adam_inline_fallback(opt, d_theta, lik, dtype, stream);
}

// pacoh_mlp_bwd + pacoh_hyper_bwd for configurations with ONE network (mean or kernel features): as pacoh_mlp2_bwd_hyper, the
// hyper-parameter reduction rides in the slab reduction's launch on the fused fp32 path.  theta / d_theta: the NETWORK's block inside
// the parameter rows; theta_rows / grad_rows: the rows themselves (what pacoh_hyper_bwd reads and writes).
extern "C" int pacoh_mlp_bwd_hyper(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                                   const int32_t* hidden, int n_hidden, int d_out, const void* g_out, void* d_theta, long d_theta_stride,
                                   void* workspace, const void* stash, int B, int n,
                                   const void* theta_rows, void* grad_rows, int T_, int off_ls, int f, int off_os, int off_noise,
                                   int off_const, const void* d_ls, const void* d_os, const void* d_noise, const void* d_const,
                                   const void* lml, void* lik, double lik_scale, const int32_t* info, int32_t* fail_flag,
                                   void* svgd_workspace, int svgd_P, int svgd_D, const pacoh_adam_inline* opt, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (opt && (!adam_inline_ok(opt, P, lml) || opt->n_seg < 1)) return PACOH_EINVAL;
    if (opt && opt->next) {
        if (!step_next_ok(opt->next)) return PACOH_EINVAL;
        if (dtype != PACOH_F32 || args_ok(d_in, hidden, n_hidden, d_out) ||
            pick_path(dtype, d_in, hidden, n_hidden, d_out, (long)B * n) != PATH_FUSED) return PACOH_ELIMIT;       // (pacoh_mlp_fused_path)
    }
    if (!g_out || !d_theta || !workspace || !x || !theta || !theta_rows || !grad_rows || x_div <= 0 || P <= 0 || B <= 0 || n <= 0 || B % P != 0)
        return PACOH_EINVAL;
    if (!d_ls || !d_noise || T_ <= 0 || features_of(f) <= 0 || (lml == nullptr) != (lik == nullptr)) return PACOH_EINVAL;
    if (svgd_workspace && (svgd_P <= 0 || svgd_D <= 0)) return PACOH_EINVAL;
    if (svgd_workspace && svgd_P > 64) return PACOH_ELIMIT;
    int rc = args_ok(d_in, hidden, n_hidden, d_out);
    if (rc) return rc;
    if (dtype == PACOH_F32 && pick_path(dtype, d_in, hidden, n_hidden, d_out, (long)B * n) == PATH_FUSED) {
        HyperBwdArgs<float> tail = {(const float*)theta_rows, theta_stride, P, T_, off_ls, features_of(f), off_os, off_noise, off_const,
                                    (const float*)d_ls, (const float*)d_os, (const float*)d_noise, (const float*)d_const, (float*)grad_rows,
                                    d_theta_stride, (const float*)lml, (float*)lik, (float)lik_scale, info, fail_flag,
                                    kernel_of(f) != PACOH_KERNEL_RBF, (const float*)svgd_workspace, svgd_P,
                                    svgd_workspace ? (float*)svgd_workspace + svgd_bw_slot(svgd_P, svgd_D) : nullptr,
                                    opt ? adam_inline_f32(opt) : AdamInline<float>{},
                                    (opt && opt->next) ? step_next_f32(opt->next, off_ls, f, off_os, off_noise) : StepNextArgs<float>{}};
        const long off = 0;
        const void* const gs[1] = {g_out};
        return mlp_fused_bwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, 1, &off, &d_out, gs, d_theta, d_theta_stride,
                             0, workspace, stash, B, n, (hipStream_t)stream, &tail, (long)((const float*)d_theta - (const float*)grad_rows));
    }
    if (stash && !args_ok(d_in, hidden, n_hidden, d_out) && pick_path(dtype, d_in, hidden, n_hidden, d_out, (long)B * n) == PATH_LAYERS)
        rc = mlp_layers_bwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, g_out, d_theta, d_theta_stride, 0, workspace, B, n,
                            dtype, (hipStream_t)stream, stash);      // (packed weights + hidden activations from pacoh_mlp_fwd_stash)
    else
        rc = pacoh_mlp_bwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, g_out, d_theta, d_theta_stride, 0, workspace, B, n,
                           dtype, stream);
    if (rc) return rc;
    rc = pacoh_hyper_bwd(theta_rows, theta_stride, P, T_, off_ls, f, off_os, off_noise, off_const, d_ls, d_os, d_noise, d_const, grad_rows,
                         d_theta_stride, lml, lik, lik_scale, info, fail_flag, svgd_workspace, svgd_P, svgd_D, nullptr, dtype, stream);
    if (rc || !opt) return rc;
    return adam_inline_fallback(opt, grad_rows, lik, dtype, stream);
}

// ---- the task-fused PACOH-MAP iteration (map_task.hip): forward + GP + backward of every task in one launch, slab reduction + tail --
namespace pacoh {
int map_task_launch(const void* theta, const void* bx, const void* by, const int32_t* bnv, int n, int d, int tb_total,
                    int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                    int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                    const void* hyp_ls, const void* hyp_os, const void* hyp_noise, void* workspace, size_t workspace_bytes,
                    void* d_theta, long d_theta_stride, const HyperBwdArgs<float>* tail_in, int plan_only, size_t* need_bytes, int D, hipStream_t stream,
                    int multi = 0, int P = 1, long theta_stride = 0, const SvgdDistTail<float>* sv = nullptr, int one_round_only = 0);
// map_wide.hip: the same launch pair for hidden widths up to 128 at <= 16 points per iteration (weights streamed from theta)
int map_wide_launch(const void* theta, const void* bx, const void* by, const int32_t* bnv, int n, int d, int tb_total,
                    int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                    int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                    const void* hyp_ls, const void* hyp_os, const void* hyp_noise, void* workspace, size_t workspace_bytes,
                    void* d_theta, long d_theta_stride, const HyperBwdArgs<float>* tail_in, int plan_only, size_t* need_bytes, int D, hipStream_t stream);
}
extern "C" size_t pacoh_map_task_workspace_bytes(int D, int n, int d, int tb, int mean_mode, const int32_t* mean_hidden, int n_mean_hidden, int kernel_nn,
                                                 const int32_t* kernel_hidden, int n_kernel_hidden, int f, int any_size, int dtype) {
    if (dtype != PACOH_F32 || tb <= 0 || D <= 0 || kernel_of(f) != PACOH_KERNEL_RBF) return 0;      // (the fused kernel is RBF-only)
    size_t need = 0;
    HyperBwdArgs<float> none = {};
    const int rc = map_task_launch(nullptr, nullptr, nullptr, nullptr, n, d, tb, mean_mode, 0, mean_hidden, n_mean_hidden, kernel_nn, 0, kernel_hidden,
                                   n_kernel_hidden, features_of(f), nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, &none, 1, &need, D, nullptr,
                                   0, 1, 0, nullptr, any_size ? 0 : 1);
    if (rc == PACOH_OK) return need;
    // networks wider than the LDS image takes: the weights-from-theta kernel (map_wide.hip), <= 16 points per iteration
    const int rcw = map_wide_launch(nullptr, nullptr, nullptr, nullptr, n, d, tb, mean_mode, 0, mean_hidden, n_mean_hidden, kernel_nn, 0, kernel_hidden,
                                    n_kernel_hidden, features_of(f), nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, &none, 1, &need, D, nullptr);
    return rcw == PACOH_OK ? need : 0;
}
extern "C" int pacoh_map_task_setup(const void* theta, int D, int n, int d, int tb, int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                                    int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                                    void* workspace, size_t workspace_bytes, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (dtype != PACOH_F32) return PACOH_ELIMIT;
    if (kernel_of(f) != PACOH_KERNEL_RBF) return PACOH_ELIMIT;       // (the fused kernel implements the RBF Gram and gradient only)
    if (!theta || !workspace || tb <= 0 || D <= 0) return PACOH_EINVAL;
    HyperBwdArgs<float> none = {};
    const int rc = map_task_launch(theta, nullptr, nullptr, nullptr, n, d, tb, mean_mode, off_mean, mean_hidden, n_mean_hidden, kernel_nn, off_kernel,
                                   kernel_hidden, n_kernel_hidden, features_of(f), nullptr, nullptr, nullptr, workspace, workspace_bytes, nullptr, 0, &none, 2,
                                   nullptr, D, (hipStream_t)stream);
    if (rc != PACOH_ELIMIT) return rc;
    return map_wide_launch(theta, nullptr, nullptr, nullptr, n, d, tb, mean_mode, off_mean, mean_hidden, n_mean_hidden, kernel_nn, off_kernel,
                           kernel_hidden, n_kernel_hidden, features_of(f), nullptr, nullptr, nullptr, workspace, workspace_bytes, nullptr, 0, &none, 2,
                           nullptr, D, (hipStream_t)stream);      // (no parameter image to build: the wide kernel reads theta itself)
}
// every entry of a parameter row the task kernels and their reduction tail read or WRITE (d_theta, and theta itself with the inline
// optimizer step) must lie inside the row: a wrong layout from a direct C caller must be an error, not a device write out of bounds
static bool task_offsets_ok(long row, int off_ls, int f, int off_os, int off_noise, int mean_mode, int off_mean, int kernel_nn, int off_kernel) {
    auto in_row = [row](long lo, long len) { return lo >= 0 && len >= 0 && lo + len <= row; };
    if (!in_row(off_ls, features_of(f)) || !in_row(off_noise, 1) || (off_os >= 0 && !in_row(off_os, 1))) return false;
    if (mean_mode == PACOH_MEAN_CONST && !in_row(off_mean, 1)) return false;
    if (mean_mode == PACOH_MEAN_VECTOR && !in_row(off_mean, 1)) return false;      // (the networks' extents: map_task_launch / map_wide_launch)
    if (kernel_nn && !in_row(off_kernel, 1)) return false;
    return true;
}

extern "C" int pacoh_map_task_step(const void* theta, long theta_stride, const void* batch_x, const void* batch_y, const int32_t* batch_n_valid,
                                   int n, int d, int tb, int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                                   int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                                   const void* ls, const void* os, const void* noise, int off_ls, int off_os, int off_noise,
                                   void* d_theta, long d_theta_stride, void* lik, double lik_scale, int32_t* fail_flag,
                                   void* workspace, size_t workspace_bytes, const pacoh_adam_inline* opt, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (dtype != PACOH_F32) return PACOH_ELIMIT;
    if (kernel_of(f) != PACOH_KERNEL_RBF) return PACOH_ELIMIT;       // (a non-RBF family must not silently train with the RBF kernel)
    if (!theta || !batch_x || !batch_y || !ls || !noise || !d_theta || !workspace || tb <= 0 || off_ls < 0 || off_noise < 0 || (os == nullptr) != (off_os < 0))
        return PACOH_EINVAL;
    if (opt && (!adam_inline_ok(opt, 1, lik) || opt->n_seg < 1)) return PACOH_EINVAL;
    if (opt && opt->next && !step_next_ok(opt->next)) return PACOH_EINVAL;
    if (!task_offsets_ok(theta_stride < d_theta_stride ? theta_stride : d_theta_stride, off_ls, f, off_os, off_noise, mean_mode, off_mean, kernel_nn,
                         off_kernel)) return PACOH_EINVAL;
    HyperBwdArgs<float> tail = {(const float*)theta, theta_stride, 1, tb, off_ls, features_of(f), off_os, off_noise,
                                mean_mode == PACOH_MEAN_CONST ? off_mean : -1, nullptr, nullptr, nullptr, nullptr, (float*)d_theta, d_theta_stride,
                                nullptr, (float*)lik, (float)lik_scale, nullptr, fail_flag, 0, nullptr, 0, nullptr,
                                opt ? adam_inline_f32(opt) : AdamInline<float>{},
                                (opt && opt->next) ? step_next_f32(opt->next, off_ls, f, off_os, off_noise) : StepNextArgs<float>{}};
    const int rc = map_task_launch(theta, batch_x, batch_y, batch_n_valid, n, d, tb, mean_mode, off_mean, mean_hidden, n_mean_hidden, kernel_nn, off_kernel,
                                   kernel_hidden, n_kernel_hidden, features_of(f), ls, os, noise, workspace, workspace_bytes, d_theta, d_theta_stride, &tail, 0,
                                   nullptr, (int)theta_stride, (hipStream_t)stream);
    if (rc != PACOH_ELIMIT) return rc;
    return map_wide_launch(theta, batch_x, batch_y, batch_n_valid, n, d, tb, mean_mode, off_mean, mean_hidden, n_mean_hidden, kernel_nn, off_kernel,
                           kernel_hidden, n_kernel_hidden, features_of(f), ls, os, noise, workspace, workspace_bytes, d_theta, d_theta_stride, &tail, 0,
                           nullptr, (int)theta_stride, (hipStream_t)stream);
}

// ---- the same kernel with P parameter rows (round 6): the likelihood half of a PACOH-SVGD / PACOH-VI step -- forward of both networks,
// GP LML + gradient, both networks' backward of every (task, row) problem in ONE launch, then the slab reduction with the step's
// hyper-parameter tail: two launches where the general path needs four (random_gp.py:204-222, svgd.py:12-28, GPR_meta_vi.py:216-224) --
extern "C" size_t pacoh_svgd_task_workspace_bytes(int D, int P, int n, int d, int tb, int mean_mode, const int32_t* mean_hidden, int n_mean_hidden,
                                                  int kernel_nn, const int32_t* kernel_hidden, int n_kernel_hidden, int f, int any_size, int dtype) {
    if (dtype != PACOH_F32 || tb <= 0 || D <= 0 || P <= 0 || kernel_of(f) != PACOH_KERNEL_RBF) return 0;
    size_t need = 0;
    HyperBwdArgs<float> none = {};
    const int rc = map_task_launch(nullptr, nullptr, nullptr, nullptr, n, d, tb, mean_mode, 0, mean_hidden, n_mean_hidden, kernel_nn, 0, kernel_hidden,
                                   n_kernel_hidden, features_of(f), nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, &none, 1, &need, D, nullptr, 1, P, D,
                                   nullptr, any_size ? 0 : 1);
    return rc == PACOH_OK ? need : 0;
}
extern "C" int pacoh_svgd_task_setup(int D, int P, int n, int d, int tb, int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                                     int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                                     void* workspace, size_t workspace_bytes, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (dtype != PACOH_F32 || kernel_of(f) != PACOH_KERNEL_RBF) return PACOH_ELIMIT;
    if (!workspace || tb <= 0 || D <= 0 || P <= 0) return PACOH_EINVAL;
    HyperBwdArgs<float> none = {};
    // (the gather map depends on the layout only: no parameter row is read here)
    return map_task_launch(workspace, nullptr, nullptr, nullptr, n, d, tb, mean_mode, off_mean, mean_hidden, n_mean_hidden, kernel_nn, off_kernel,
                           kernel_hidden, n_kernel_hidden, features_of(f), nullptr, nullptr, nullptr, workspace, workspace_bytes, nullptr, 0, &none, 2,
                           nullptr, D, (hipStream_t)stream, 1, P, D, nullptr);
}
extern "C" int pacoh_svgd_task_step(const void* theta, long theta_stride, int P, const void* batch_x, const void* batch_y,
                                    const int32_t* batch_n_valid, int n, int d, int tb, int mean_mode, int off_mean, const int32_t* mean_hidden,
                                    int n_mean_hidden, int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f,
                                    const void* ls, const void* os, const void* noise, int off_ls, int off_os, int off_noise,
                                    void* d_theta, long d_theta_stride, void* lik, double lik_scale, int32_t* fail_flag,
                                    void* workspace, size_t workspace_bytes, const void* svgd_X, void* svgd_workspace, int svgd_D, int64_t* counter,
                                    int want_bandwidth, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (dtype != PACOH_F32 || kernel_of(f) != PACOH_KERNEL_RBF) return PACOH_ELIMIT;
    if (!theta || !batch_x || !batch_y || !ls || !noise || !d_theta || !workspace || tb <= 0 || P <= 0 || off_ls < 0 || off_noise < 0 ||
        (os == nullptr) != (off_os < 0) || theta_stride <= 0) return PACOH_EINVAL;
    if ((svgd_X == nullptr) != (svgd_workspace == nullptr) || (svgd_X && svgd_D <= 0) || (want_bandwidth && !svgd_workspace)) return PACOH_EINVAL;
    if (want_bandwidth && P > 64) return PACOH_ELIMIT;
    if (!task_offsets_ok(theta_stride < d_theta_stride ? theta_stride : d_theta_stride, off_ls, f, off_os, off_noise, mean_mode, off_mean, kernel_nn,
                         off_kernel)) return PACOH_EINVAL;
    float* d2 = (float*)svgd_workspace;
    SvgdDistTail<float> sv = {(const float*)svgd_X, d2, d2 ? d2 + P * P : nullptr, P, svgd_D, (long*)counter};
    HyperBwdArgs<float> tail = {(const float*)theta, theta_stride, P, tb, off_ls, features_of(f), off_os, off_noise,
                                mean_mode == PACOH_MEAN_CONST ? off_mean : -1, nullptr, nullptr, nullptr, nullptr, (float*)d_theta, d_theta_stride,
                                nullptr, (float*)lik, (float)lik_scale, nullptr, fail_flag, 0, want_bandwidth ? d2 : nullptr, P,
                                want_bandwidth ? d2 + svgd_bw_slot(P, svgd_D) : nullptr, AdamInline<float>{}, StepNextArgs<float>{}};
    return map_task_launch(theta, batch_x, batch_y, batch_n_valid, n, d, tb, mean_mode, off_mean, mean_hidden, n_mean_hidden, kernel_nn, off_kernel,
                           kernel_hidden, n_kernel_hidden, features_of(f), ls, os, noise, workspace, workspace_bytes, d_theta, d_theta_stride, &tail, 0,
                           nullptr, (int)theta_stride, (hipStream_t)stream, 1, P, theta_stride, svgd_X ? &sv : nullptr);
}

// 1 if these network shapes run on the fused fp32 kernels (mlp_fused.hip) -- where the gradient epilogue can carry the optimizer
// step's pipelined feed (pacoh_adam_inline.next) --, else 0
extern "C" int pacoh_mlp_fused_path(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out, int dtype) {
    if (check_dtype(dtype) || P <= 0 || B <= 0 || n <= 0 || B % P != 0 || args_ok(d_in, hidden, n_hidden, d_out)) return 0;
    return dtype == PACOH_F32 && pick_path(dtype, d_in, hidden, n_hidden, d_out, (long)B * n) == PATH_FUSED;
}

extern "C" int pacoh_reduce_tasks(const void* in, void* out, double scale, int accumulate, int T_, int P, int Wd,
                                  int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!in || !out || T_ <= 0 || P <= 0 || Wd <= 0) return PACOH_EINVAL;
    long tot = (long)P * Wd;
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(reduce_tasks_kernel<float>, dim3((unsigned)((tot + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)in, (float*)out, (float)scale, accumulate, T_, P, Wd);
    else
        hipLaunchKernelGGL(reduce_tasks_kernel<double>, dim3((unsigned)((tot + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const double*)in, (double*)out, scale, accumulate, T_, P, Wd);
    return launch_status();
}
