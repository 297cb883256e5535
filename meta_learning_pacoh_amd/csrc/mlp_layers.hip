// General per-particle MLP: ANY number of tanh hidden layers of ANY width (the reference's NeuralNetworkVectorized takes
// arbitrary layer_sizes, meta_learn/models.py:328-349; experiments/meta_GPR_mll_base_exp.py:29-30 runs 4 x 128), fp32 and
// fp64, layer by layer: every layer is a batched (per particle) GEMM on the matrix cores (v_mfma_f32_16x16x4_f32 /
// v_mfma_f64_16x16x4_f64) with bias + tanh fused into its epilogue; activations live in a caller-provided workspace
// (L2-resident at the sizes PACOH-MAP runs), weights are first repacked into zero-padded 16-aligned images so that the
// GEMM loops carry no bounds checks in the feature dimensions.  Backward: per layer one GEMM contracting over the data
// points (weight + bias gradient, partial slabs over row chunks summed in fixed order -> deterministic) and one GEMM for
// the delta recursion with the (1 - h^2) factor fused.
//
// The register-resident fused kernels (mlp_fused.hip, mlp_mfma.hip) and the thread-per-point kernels (mlp_impl.h) serve the
// shapes they were written for; this path takes everything else.
// Replaces LinearVectorized / NeuralNetworkVectorized forward (models.py:295-317,343-349) and the autograd backward.
#include "common.h"

namespace pacoh {

using lf32x4 = __attribute__((ext_vector_type(4))) float;
using lf64x4 = __attribute__((ext_vector_type(4))) double;

template <typename T> struct LMf;
template <> struct LMf<float> {
    using acc = lf32x4;
    static __device__ __forceinline__ acc mma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int g, int q) { return 4 * g + q; }          // C/D row held in register q
};
template <> struct LMf<double> {
    using acc = lf64x4;
    static __device__ __forceinline__ acc mma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int row(int g, int q) { return g + 4 * q; }
};

constexpr int LMAXL = 64;      // layers (hidden + output) of one network

struct LayerPlan {
    int n_layers;              // hidden layers + 1
    int in_real[LMAXL], out_real[LMAXL], inp[LMAXL], outp[LMAXL];
    long w_off[LMAXL], b_off[LMAXL];       // element offsets inside one particle's packed weight image
    long th_off[LMAXL];                    // element offset of the layer (bias first) inside the theta block
    long w_elems;                          // packed elements per particle
    long act_off[LMAXL];                   // element offset of H_l (output of layer l) inside one particle's activation image, per row
    long act_width;                        // sum of outp over hidden layers
    int max_w;                             // max padded width over all layers
    int D_net;
};

static int pad16(int v) { return (v + 15) & ~15; }

static int make_plan(LayerPlan& pl, int d_in, const int32_t* hidden, int n_hidden, int d_out) {
    if (d_in <= 0 || d_out <= 0 || n_hidden < 0 || n_hidden + 1 > LMAXL || (n_hidden > 0 && !hidden)) return PACOH_EINVAL;
    pl.n_layers = n_hidden + 1;
    int prev = d_in;
    long w = 0, th = 0, act = 0;
    pl.max_w = pad16(d_in);
    for (int l = 0; l <= n_hidden; ++l) {
        const int out = l < n_hidden ? hidden[l] : d_out;
        if (out <= 0 || out > 65536) return out <= 0 ? PACOH_EINVAL : PACOH_ELIMIT;
        pl.in_real[l] = prev; pl.out_real[l] = out; pl.inp[l] = pad16(prev); pl.outp[l] = pad16(out);
        pl.b_off[l] = w; w += pl.outp[l];
        pl.w_off[l] = w; w += (long)pl.outp[l] * pl.inp[l];
        pl.th_off[l] = th; th += (long)out * (prev + 1);
        pl.act_off[l] = act;
        if (l < n_hidden) act += pl.outp[l];
        if (pl.outp[l] > pl.max_w) pl.max_w = pl.outp[l];
        prev = out;
    }
    pl.w_elems = w; pl.act_width = act; pl.D_net = (int)th;
    if (th > 0x7fffffffL) return PACOH_ELIMIT;
    return PACOH_OK;
}

// ---- weight repack: theta block (bias[out] | W[out][in] per layer) -> zero-padded images -------------------------------
template <typename T>
__global__ void __launch_bounds__(256) pack_weights_kernel(const T* __restrict__ theta, long theta_stride, T* __restrict__ wp, long w_elems,
                                                           int in_real, int out_real, int inp, int outp, long w_off, long b_off, long th_off) {
    const int p = blockIdx.y;
    const T* th = theta + (long)p * theta_stride + th_off;
    T* dst = wp + (long)p * w_elems;
    const long tot = (long)outp * (inp + 1);
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < tot; q += (long)gridDim.x * 256) {
        if (q < outp) { dst[b_off + q] = q < out_real ? th[q] : T(0); continue; }
        const long e = q - outp;
        const int o = (int)(e / inp), k = (int)(e - (long)o * inp);
        dst[w_off + e] = (o < out_real && k < in_real) ? th[out_real + (long)o * in_real + k] : T(0);
    }
}

// the same for ALL layers of a network in one launch (blockIdx.z = layer; round 6: a launch per layer was 20 of the ~62 launches of a
// PACOH-MAP iteration with two 4 x 128 networks, each ~4.6 us of pure latency).  Networks of more than PACK_MAXL layers keep the loop.
constexpr int PACK_MAXL = 16;
struct PackPlan { int n_layers; int in_real[PACK_MAXL], out_real[PACK_MAXL], inp[PACK_MAXL], outp[PACK_MAXL]; long w_off[PACK_MAXL], b_off[PACK_MAXL], th_off[PACK_MAXL]; };
template <typename T>
__global__ void __launch_bounds__(256) pack_all_kernel(const T* __restrict__ theta, long theta_stride, T* __restrict__ wp, long w_elems, PackPlan pp) {
    const int p = blockIdx.y, l = blockIdx.z;
    const int in_real = pp.in_real[l], out_real = pp.out_real[l], inp = pp.inp[l], outp = pp.outp[l];
    const long w_off = pp.w_off[l], b_off = pp.b_off[l];
    const T* th = theta + (long)p * theta_stride + pp.th_off[l];
    T* dst = wp + (long)p * w_elems;
    const long tot = (long)outp * (inp + 1);
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < tot; q += (long)gridDim.x * 256) {
        if (q < outp) { dst[b_off + q] = q < out_real ? th[q] : T(0); continue; }
        const long e = q - outp;
        const int o = (int)(e / inp), k = (int)(e - (long)o * inp);
        dst[w_off + e] = (o < out_real && k < in_real) ? th[out_real + (long)o * in_real + k] : T(0);
    }
}

// rows of a particle: row rr = t*n + i -> problem b = t*P + p; x row = (b / x_div)*n + i, output row = b*n + i
struct RowMap { int P, n, R, x_div; };
__device__ __forceinline__ void map_row(const RowMap& m, int p, int rr, long& xrow, long& orow) {
    const int t = (int)((unsigned)rr / (unsigned)m.n), i = rr - t * m.n;
    const int bi = t * m.P + p;
    const int xb = m.x_div == 1 ? bi : (int)((unsigned)bi / (unsigned)m.x_div);
    xrow = (long)xb * m.n + i;
    orow = (long)bi * m.n + i;
}

// ---- forward layer: OUT[rows, outp] = act(IN[rows, inp] W^T + b) -------------------------------------------------------
// one wave per 16 rows x 16 output features; IN is either the packed activation image of the previous layer (IN_X = false)
// or the raw inputs x (gathered rows, d_in real columns); OUT is either the next activation image or the final output
// tensor out[B, n, d_out] (OUT_FINAL).
template <typename T, bool IN_X, bool OUT_FINAL>
__global__ void __launch_bounds__(256) layer_fwd_kernel(const T* __restrict__ in, long in_particle_stride, int in_ld, int in_real,
                                                        const T* __restrict__ wp, long w_elems, long w_off, long b_off, int inp,
                                                        T* __restrict__ out, long out_particle_stride, int out_ld, int out_real,
                                                        RowMap m) {
    using M = LMf<T>;
    const int p = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int row0 = (blockIdx.x * 4 + wave) * 16;
    if (row0 >= m.R) return;
    const int col0 = blockIdx.y * 16;
    const T* W = wp + (long)p * w_elems + w_off + (long)(col0 + r) * inp;
    const int ra = min(row0 + r, m.R - 1);                        // the row this lane feeds as the A operand (clamped)
    const T* arow;
    if (IN_X) { long xr, orw; map_row(m, p, ra, xr, orw); arow = in + xr * in_ld; }
    else arow = in + (long)p * in_particle_stride + (long)ra * in_ld;
    typename M::acc acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < inp; k0 += 16) {
        T av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + 4 * g + s;
            if (IN_X) av[s] = k < in_real ? arow[k] : T(0);
            else av[s] = arow[k];
            bv[s] = W[k];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = M::mma(av[s], bv[s], acc);
    }
    const T bias = wp[(long)p * w_elems + b_off + col0 + r];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = row0 + M::row(g, q);
        if (row >= m.R) continue;
        const T v = acc[q] + bias;
        if (OUT_FINAL) {
            if (col0 + r < out_real) { long xr, orw; map_row(m, p, row, xr, orw); out[orw * out_ld + col0 + r] = v; }
        } else {
            out[(long)p * out_particle_stride + (long)row * out_ld + col0 + r] = act_tanh<T>(v);
        }
    }
}

// ---- delta recursion: DIN[rows, inp] = (DOUT[rows, outp] W) .* (1 - H^2) -----------------------------------------------
// DOUT is the packed delta image of the layer above (D_G = false) or the upstream gradient g_out[B, n, d_out] (D_G = true)
template <typename T, bool D_G>
__global__ void __launch_bounds__(256) layer_delta_kernel(const T* __restrict__ dout, long d_particle_stride, int d_ld, int d_real,
                                                          const T* __restrict__ wp, long w_elems, long w_off, int inp, int outp,
                                                          const T* __restrict__ h, long h_particle_stride, int h_ld,
                                                          T* __restrict__ din, long din_particle_stride, int din_ld, RowMap m) {
    using M = LMf<T>;
    const int p = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int row0 = (blockIdx.x * 4 + wave) * 16;
    if (row0 >= m.R) return;
    const int col0 = blockIdx.y * 16;                              // input-feature block of the layer
    const T* W = wp + (long)p * w_elems + w_off + col0 + r;          // W[k][col0 + r] at k * inp
    const int ra = min(row0 + r, m.R - 1);
    const T* arow;
    if (D_G) { long xr, orw; map_row(m, p, ra, xr, orw); arow = dout + orw * d_ld; }
    else arow = dout + (long)p * d_particle_stride + (long)ra * d_ld;
    typename M::acc acc = {0, 0, 0, 0};
    for (int k0 = 0; k0 < outp; k0 += 16) {
        T av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = k0 + 4 * g + s;
            if (D_G) av[s] = k < d_real ? arow[k] : T(0);
            else av[s] = arow[k];
            bv[s] = W[(long)k * inp];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = M::mma(av[s], bv[s], acc);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int row = row0 + M::row(g, q);
        if (row >= m.R) continue;
        const T hv = h[(long)p * h_particle_stride + (long)row * h_ld + col0 + r];
        din[(long)p * din_particle_stride + (long)row * din_ld + col0 + r] = acc[q] * (T(1) - hv * hv);
    }
}

// ---- weight / bias gradient: dW[o][j] = sum_rows D[row][o] IN[row][j], db[o] = sum_rows D[row][o] ------------------------
// grid (row chunks, (outp/16)*(inp/16), P); the four waves of a workgroup split the chunk's rows and add up through LDS; each
// workgroup writes its block of the chunk's slab in the reference's flattened layout (bias[out] | W[out][in]).
template <typename T, bool D_G, bool IN_X>
__global__ void __launch_bounds__(256) layer_wgrad_kernel(const T* __restrict__ d, long d_particle_stride, int d_ld, int out_real,
                                                          const T* __restrict__ in, long in_particle_stride, int in_ld, int in_real,
                                                          int inp, T* __restrict__ slab, int D_net, long th_off, int rows_per_chunk,
                                                          RowMap m) {
    using M = LMf<T>;
    __shared__ T red[3][5][64];
    const int p = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int jbs = inp >> 4;
    const int ob = blockIdx.y / jbs, jb = blockIdx.y - ob * jbs;
    const int rbeg = blockIdx.x * rows_per_chunk, rend = min(m.R, rbeg + rows_per_chunk);
    typename M::acc acc = {0, 0, 0, 0};
    T bsum = 0;
    for (int row0 = rbeg + wave * 16; row0 < rend; row0 += 64) {
        T av[4], bv[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int row = row0 + 4 * g + s;
            const bool ok = row < rend;
            const int rc = ok ? row : rend - 1;
            long xr = 0, orw = 0;
            if (D_G || IN_X) map_row(m, p, rc, xr, orw);
            T a_, b_;
            if (D_G) a_ = (ob * 16 + r < out_real) ? d[orw * d_ld + ob * 16 + r] : T(0);
            else a_ = d[(long)p * d_particle_stride + (long)rc * d_ld + ob * 16 + r];
            if (IN_X) b_ = (jb * 16 + r < in_real) ? in[xr * in_ld + jb * 16 + r] : T(0);
            else b_ = in[(long)p * in_particle_stride + (long)rc * in_ld + jb * 16 + r];
            av[s] = ok ? a_ : T(0);
            bv[s] = b_;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) { acc = M::mma(av[s], bv[s], acc); bsum += av[s]; }
    }
    // cross-wave sum in fixed order
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave - 1][q][lane] = acc[q];
        red[wave - 1][4][lane] = bsum;
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = ((acc[q] + red[0][q][lane]) + red[1][q][lane]) + red[2][q][lane];
    bsum = ((bsum + red[0][4][lane]) + red[1][4][lane]) + red[2][4][lane];
    bsum += shfl_xor_t<T>(bsum, 16); bsum += shfl_xor_t<T>(bsum, 32);             // over the four row groups g
    T* dst = slab + ((long)blockIdx.x * m.P + p) * D_net + th_off;
    if (jb == 0 && g == 0 && ob * 16 + r < out_real) dst[ob * 16 + r] = bsum;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int o = ob * 16 + M::row(g, q), j = jb * 16 + r;
        if (o < out_real && j < in_real) dst[out_real + (long)o * in_real + j] = acc[q];
    }
}

template <typename T>
__global__ void __launch_bounds__(256) layers_reduce_slab_kernel(const T* __restrict__ in, T* __restrict__ out, long out_stride, int accumulate,
                                                                 int C, int P, int Wd) {
    const long tot = (long)P * Wd;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= tot) return;
    T s = 0;
    for (int c = 0; c < C; ++c) s += in[(long)c * tot + idx];
    const int p = (int)(idx / Wd), w = (int)(idx - (long)p * Wd);
    T* o = out + (long)p * out_stride + w;
    *o = accumulate ? *o + s : s;
}

// ---- host side ----------------------------------------------------------------------------------------------------
static size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct LayersWs { size_t wp, act, delta, slab, total; int chunks, rows_per_chunk; };

static LayersWs layers_ws(const LayerPlan& pl, int P, long R, size_t es, bool bwd) {
    LayersWs w = {};
    w.wp = align_up((size_t)P * pl.w_elems * es);
    if (!bwd) {
        // forward only: two ping-pong activation buffers of the widest layer
        w.act = pl.n_layers > 1 ? align_up((size_t)2 * P * R * pl.max_w * es) : 0;
    } else {
        w.act = align_up((size_t)P * R * (pl.act_width > 0 ? pl.act_width : 1) * es);
        w.delta = pl.n_layers > 1 ? align_up((size_t)2 * P * R * pl.max_w * es) : 0;
        long chunks = (R + 255) / 256;                          // >= 256 rows per chunk, at most 64 chunks
        if (chunks > 64) chunks = 64;
        if (chunks < 1) chunks = 1;
        long rpc = ((R + chunks - 1) / chunks + 63) / 64 * 64;
        w.chunks = (int)((R + rpc - 1) / rpc);
        w.rows_per_chunk = (int)rpc;
        w.slab = align_up((size_t)w.chunks * P * pl.D_net * es);
    }
    w.total = w.wp + w.act + w.delta + w.slab;
    return w;
}

// the forward's state a backward of the SAME inputs and parameters can start from instead of repacking the weights and recomputing the
// hidden layers (round 6; mlp.hip hands it around as the "stash" the fused path has had since round 3): packed weights | every hidden
// layer's activations, in the backward's own layout
size_t mlp_layers_stash_bytes(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out, int dtype) {
    LayerPlan pl;
    if (P <= 0 || B <= 0 || n <= 0 || make_plan(pl, d_in, hidden, n_hidden, d_out)) return 0;
    const LayersWs w = layers_ws(pl, P, (long)(B / P) * n, dtype == PACOH_F64 ? 8 : 4, true);
    return w.wp + w.act;
}

size_t mlp_layers_workspace(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out, int dtype, int bwd) {
    LayerPlan pl;
    if (P <= 0 || B <= 0 || n <= 0 || make_plan(pl, d_in, hidden, n_hidden, d_out)) return 0;
    return layers_ws(pl, P, (long)(B / P) * n, dtype == PACOH_F64 ? 8 : 4, bwd != 0).total;
}

template <typename T>
static void launch_pack(const LayerPlan& pl, const T* theta, long theta_stride, T* wp, int P, hipStream_t s) {
    if (pl.n_layers <= PACK_MAXL) {
        PackPlan pp;
        pp.n_layers = pl.n_layers;
        long maxtot = 0;
        for (int l = 0; l < pl.n_layers; ++l) {
            pp.in_real[l] = pl.in_real[l]; pp.out_real[l] = pl.out_real[l]; pp.inp[l] = pl.inp[l]; pp.outp[l] = pl.outp[l];
            pp.w_off[l] = pl.w_off[l]; pp.b_off[l] = pl.b_off[l]; pp.th_off[l] = pl.th_off[l];
            const long tot = (long)pl.outp[l] * (pl.inp[l] + 1);
            if (tot > maxtot) maxtot = tot;
        }
        long blocks = (maxtot + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(pack_all_kernel<T>, dim3((unsigned)blocks, P, pl.n_layers), dim3(256), 0, s, theta, theta_stride, wp, pl.w_elems, pp);
        return;
    }
    for (int l = 0; l < pl.n_layers; ++l) {
        const long tot = (long)pl.outp[l] * (pl.inp[l] + 1);
        long blocks = (tot + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(pack_weights_kernel<T>, dim3((unsigned)blocks, P), dim3(256), 0, s, theta, theta_stride, wp, pl.w_elems,
                           pl.in_real[l], pl.out_real[l], pl.inp[l], pl.outp[l], pl.w_off[l], pl.b_off[l], pl.th_off[l]);
    }
}

// runs the forward; H_l images: `act` holds per layer l < n_hidden an image [P][R][outp_l] at act + act_base[l]
template <typename T>
static void launch_forward(const LayerPlan& pl, const T* x, int d_in, const T* wp, T* const* H, T* out, int d_out, int P, RowMap m, hipStream_t s) {
    const unsigned rb = (unsigned)((m.R + 63) / 64);
    for (int l = 0; l < pl.n_layers; ++l) {
        const bool first = l == 0, last = l == pl.n_layers - 1;
        const dim3 grid(rb, pl.outp[l] / 16, P);
        const T* in = first ? x : H[l - 1];
        const long in_ps = first ? 0 : (long)m.R * pl.outp[l - 1];
        const int in_ld = first ? d_in : pl.outp[l - 1];
        T* o = last ? out : H[l];
        const long o_ps = last ? 0 : (long)m.R * pl.outp[l];
        const int o_ld = last ? d_out : pl.outp[l];
#define PACOH_LF(INX, FIN) hipLaunchKernelGGL((layer_fwd_kernel<T, INX, FIN>), grid, dim3(256), 0, s, in, in_ps, in_ld, pl.in_real[l], wp, \
                                              pl.w_elems, pl.w_off[l], pl.b_off[l], pl.inp[l], o, o_ps, o_ld, pl.out_real[l], m)
        if (first && last) PACOH_LF(true, true); else if (first) PACOH_LF(true, false); else if (last) PACOH_LF(false, true); else PACOH_LF(false, false);
#undef PACOH_LF
    }
}

template <typename T>
int mlp_layers_fwd_t(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in, const int32_t* hidden,
                     int n_hidden, int d_out, void* out, void* workspace, int B, int n, hipStream_t s, void* stash) {
    LayerPlan pl;
    int rc = make_plan(pl, d_in, hidden, n_hidden, d_out);
    if (rc) return rc;
    if (!workspace && !stash) return PACOH_EINVAL;
    const long R = (long)(B / P) * n;
    if (R > 0x3fffffffL || (long)B * n > 0x7fffffffL) return PACOH_ELIMIT;
    // stash: the packed weights and EVERY hidden layer's activations stay behind for the backward (mlp_layers_stash_bytes)
    const LayersWs w = layers_ws(pl, P, R, sizeof(T), stash != nullptr);
    void* base = stash ? stash : workspace;
    T* wp = (T*)base;
    T* act = (T*)((char*)base + w.wp);
    launch_pack<T>(pl, (const T*)theta, theta_stride, wp, P, s);
    T* H[LMAXL];
    for (int l = 0; l + 1 < pl.n_layers; ++l)
        H[l] = stash ? act + (size_t)P * R * pl.act_off[l] : act + (size_t)(l & 1) * P * R * pl.max_w;     // kept / ping-pong
    RowMap m = {P, n, (int)R, x_div};
    launch_forward<T>(pl, (const T*)x, d_in, wp, H, (T*)out, d_out, P, m, s);
    return launch_status();
}

template <typename T>
int mlp_layers_bwd_t(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in, const int32_t* hidden,
                     int n_hidden, int d_out, const void* g_out, void* d_theta, long d_theta_stride, int accumulate,
                     void* workspace, int B, int n, hipStream_t s, const void* stash) {
    LayerPlan pl;
    int rc = make_plan(pl, d_in, hidden, n_hidden, d_out);
    if (rc) return rc;
    if (!workspace) return PACOH_EINVAL;
    const long R = (long)(B / P) * n;
    if (R > 0x3fffffffL || (long)B * n > 0x7fffffffL) return PACOH_ELIMIT;
    const LayersWs w = layers_ws(pl, P, R, sizeof(T), true);
    // stash: the forward of the same inputs and parameters left the packed weights and the hidden activations there
    T* wp = stash ? (T*)const_cast<void*>(stash) : (T*)workspace;
    T* act = stash ? (T*)((char*)const_cast<void*>(stash) + w.wp) : (T*)((char*)workspace + w.wp);
    T* delta = (T*)((char*)workspace + w.wp + w.act);
    T* slab = (T*)((char*)workspace + w.wp + w.act + w.delta);
    if (!stash) launch_pack<T>(pl, (const T*)theta, theta_stride, wp, P, s);
    RowMap m = {P, n, (int)R, x_div};
    const int L = pl.n_layers;
    T* H[LMAXL];
    for (int l = 0; l + 1 < L; ++l) H[l] = act + (size_t)P * R * pl.act_off[l];
    // forward recompute up to the last hidden layer (unless the forward stashed its activations)
    if (L > 1 && !stash) {
        const unsigned rb = (unsigned)((R + 63) / 64);
        for (int l = 0; l < L - 1; ++l) {
            const dim3 grid(rb, pl.outp[l] / 16, P);
            if (l == 0)
                hipLaunchKernelGGL((layer_fwd_kernel<T, true, false>), grid, dim3(256), 0, s, (const T*)x, 0L, d_in, pl.in_real[0], (const T*)wp,
                                   pl.w_elems, pl.w_off[0], pl.b_off[0], pl.inp[0], H[0], (long)R * pl.outp[0], pl.outp[0], pl.out_real[0], m);
            else
                hipLaunchKernelGGL((layer_fwd_kernel<T, false, false>), grid, dim3(256), 0, s, (const T*)H[l - 1], (long)R * pl.outp[l - 1], pl.outp[l - 1],
                                   pl.in_real[l], (const T*)wp, pl.w_elems, pl.w_off[l], pl.b_off[l], pl.inp[l], H[l], (long)R * pl.outp[l],
                                   pl.outp[l], pl.out_real[l], m);
        }
    }
    // backward, top layer first: delta of layer l in dbuf[l & 1] (the top layer's delta is g_out itself)
    T* dbuf[2] = {delta, delta + (size_t)P * R * pl.max_w};
    const unsigned rb = (unsigned)((R + 63) / 64);
    for (int l = L - 1; l >= 0; --l) {
        const bool top = l == L - 1, first = l == 0;
        const T* dl = top ? (const T*)g_out : dbuf[l & 1];
        const long d_ps = top ? 0 : (long)R * pl.outp[l];
        const int d_ld = top ? d_out : pl.outp[l];
        const T* in = first ? (const T*)x : H[l - 1];
        const long in_ps = first ? 0 : (long)R * pl.outp[l - 1];
        const int in_ld = first ? d_in : pl.outp[l - 1];
        const dim3 gw(w.chunks, (pl.outp[l] / 16) * (pl.inp[l] / 16), P);
#define PACOH_LW(DG, INX) hipLaunchKernelGGL((layer_wgrad_kernel<T, DG, INX>), gw, dim3(256), 0, s, dl, d_ps, d_ld, pl.out_real[l], in, in_ps, \
                                             in_ld, pl.in_real[l], pl.inp[l], slab, pl.D_net, pl.th_off[l], w.rows_per_chunk, m)
        if (top && first) PACOH_LW(true, true); else if (top) PACOH_LW(true, false); else if (first) PACOH_LW(false, true); else PACOH_LW(false, false);
#undef PACOH_LW
        if (!first) {
            const dim3 gd(rb, pl.inp[l] / 16, P);
            T* dn = dbuf[(l - 1) & 1];
            if (top)
                hipLaunchKernelGGL((layer_delta_kernel<T, true>), gd, dim3(256), 0, s, dl, d_ps, d_ld, pl.out_real[l], (const T*)wp, pl.w_elems, pl.w_off[l],
                                   pl.inp[l], pl.outp[l], (const T*)H[l - 1], (long)R * pl.outp[l - 1], pl.outp[l - 1], dn, (long)R * pl.outp[l - 1], pl.outp[l - 1], m);
            else
                hipLaunchKernelGGL((layer_delta_kernel<T, false>), gd, dim3(256), 0, s, dl, d_ps, d_ld, pl.out_real[l], (const T*)wp, pl.w_elems, pl.w_off[l],
                                   pl.inp[l], pl.outp[l], (const T*)H[l - 1], (long)R * pl.outp[l - 1], pl.outp[l - 1], dn, (long)R * pl.outp[l - 1], pl.outp[l - 1], m);
        }
    }
    const long tot = (long)P * pl.D_net;
    hipLaunchKernelGGL(layers_reduce_slab_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, (const T*)slab, (T*)d_theta,
                       d_theta_stride, accumulate, w.chunks, P, pl.D_net);
    return launch_status();
}

int mlp_layers_fwd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in, const int32_t* hidden, int n_hidden,
                   int d_out, void* out, void* workspace, int B, int n, int dtype, hipStream_t s, void* stash) {
    return dtype == PACOH_F32
        ? mlp_layers_fwd_t<float>(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, out, workspace, B, n, s, stash)
        : mlp_layers_fwd_t<double>(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, out, workspace, B, n, s, stash);
}

int mlp_layers_bwd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in, const int32_t* hidden, int n_hidden,
                   int d_out, const void* g_out, void* d_theta, long d_theta_stride, int accumulate, void* workspace, int B, int n,
                   int dtype, hipStream_t s, const void* stash) {
    return dtype == PACOH_F32
        ? mlp_layers_bwd_t<float>(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, g_out, d_theta, d_theta_stride, accumulate, workspace, B, n, s, stash)
        : mlp_layers_bwd_t<double>(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, g_out, d_theta, d_theta_stride, accumulate, workspace, B, n, s, stash);
}

}  // namespace pacoh
