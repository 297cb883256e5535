// Per-particle ("vectorised") MLP forward / backward.
// Replaces NeuralNetworkVectorized / LinearVectorized (meta_learn/models.py:279-384; the bmm at :313)
// and, with P = 1, the shared-weight NeuralNetwork of PACOH-MAP (models.py:190-227).
//
// Mapping: blockIdx.y = particle p, every thread owns one data point (row) of that particle, the
// particle's weights sit zero-padded in LDS ([out][in] rows, width HP) and are read as broadcast
// ds_read_b128; activations live in registers (static arrays, fully unrolled).  The backward
// recomputes the activations (nothing is saved by the forward), forms the per-point deltas, and
// reduces the weight gradients over the points of a tile through LDS; per-workgroup partial
// gradients go to a slab that a second kernel sums in a fixed order (deterministic).
#pragma once
#include "common.h"

namespace pacoh {

constexpr int VALU_MAX_HIDDEN = 3;    // limits of THIS path (the general layer-wise path, mlp_layers.hip, has none)
constexpr int VALU_MAX_WIDTH = 64;
constexpr int DP = 16;   // padded input width
constexpr int OP = 8;    // padded output width

struct MlpDims {
    int d_in, d_out, n_hidden;
    int hidden[VALU_MAX_HIDDEN];
};

template <typename T>
struct MlpArgs {
    const T* x; int x_div;
    const T* theta; long theta_stride;
    T* out;
    const T* g_out;
    T* slab;            // [n_chunks][P][D_net]
    int P, B, n;
    long rows_per_particle;   // T_tasks * n
    int n_chunks;
    MlpDims dims;
    int D_net;
};

__host__ __device__ inline int mlp_param_count(const MlpDims& d) {
    int prev = d.d_in, c = 0;
    for (int l = 0; l < d.n_hidden; ++l) { c += d.hidden[l] * (prev + 1); prev = d.hidden[l]; }
    return c + d.d_out * (prev + 1);
}

// LDS weight image: layer l at offset woff(l): bias[OUTP] then W[OUTP][INP] (zero padded)
template <int HP, int NH> __host__ __device__ constexpr int lds_weight_elems() {
    if (NH == 0) return OP + OP * DP;
    return (HP + HP * DP) + (NH - 1) * (HP + HP * HP) + (OP + OP * HP);
}

template <typename T, int HP, int NH>
__device__ void load_weights(T* wl, const T* __restrict__ th, const MlpDims& d) {
    // zero fill then scatter the real entries
    const int total = lds_weight_elems<HP, NH>();
    for (int q = threadIdx.x; q < total; q += blockDim.x) wl[q] = 0;
    __syncthreads();
    int prev = d.d_in, src = 0, dst = 0;
    for (int l = 0; l <= NH; ++l) {
        const int out = (l < NH) ? d.hidden[l] : d.d_out;
        const int OUTP = (l < NH) ? HP : OP;
        const int INP = (l == 0) ? DP : HP;
        for (int q = threadIdx.x; q < out; q += blockDim.x) wl[dst + q] = th[src + q];
        for (int q = threadIdx.x; q < out * prev; q += blockDim.x) {
            int o = q / prev, k = q - o * prev;
            wl[dst + OUTP + o * INP + k] = th[src + out + q];
        }
        src += out * (prev + 1);
        dst += OUTP + OUTP * INP;
        prev = out;
    }
    __syncthreads();
}

template <typename T, int IN, int OUT, bool ACT>
__device__ __forceinline__ void dense(const T* wl, const T (&in)[IN], T (&out)[OUT]) {
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    const T* bias = wl;
    const T* Wm = wl + OUT;
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        T acc = bias[o];
        const V* row = reinterpret_cast<const V*>(Wm + o * IN);
#pragma unroll
        for (int v = 0; v < IN / W; ++v) {
            V w = row[v];
            if constexpr (W == 4) {
                acc = fma(w.x, in[4 * v], acc); acc = fma(w.y, in[4 * v + 1], acc);
                acc = fma(w.z, in[4 * v + 2], acc); acc = fma(w.w, in[4 * v + 3], acc);
            } else {
                acc = fma(w.x, in[2 * v], acc); acc = fma(w.y, in[2 * v + 1], acc);
            }
        }
        out[o] = ACT ? act_tanh<T>(acc) : acc;
    }
}

// Same layer with the output loop kept as a real (not unrolled) loop: each result goes to a
// thread-private LDS column col[o*TPB] and is read back with static indices.  Used by the forward
// kernel, where the fully unrolled form makes hipcc hoist the whole layer's weight reads in front
// of the FMAs and spill them (2340 B/lane of scratch in the first build of this kernel).
template <typename T, int IN, int OUT, bool ACT, int TPB>
__device__ __forceinline__ void dense_rolled(const T* wl, T* col, const T (&in)[IN], T (&out)[OUT]) {
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    const T* bias = wl;
    const T* Wm = wl + OUT;
#pragma unroll 1
    for (int o = 0; o < OUT; ++o) {
        T acc = bias[o];
        const V* row = reinterpret_cast<const V*>(Wm + o * IN);
#pragma unroll
        for (int v = 0; v < IN / W; ++v) {
            V w = row[v];
            if constexpr (W == 4) {
                acc = fma(w.x, in[4 * v], acc); acc = fma(w.y, in[4 * v + 1], acc);
                acc = fma(w.z, in[4 * v + 2], acc); acc = fma(w.w, in[4 * v + 3], acc);
            } else {
                acc = fma(w.x, in[2 * v], acc); acc = fma(w.y, in[2 * v + 1], acc);
            }
        }
        col[o * TPB] = ACT ? act_tanh<T>(acc) : acc;
    }
#pragma unroll
    for (int o = 0; o < OUT; ++o) out[o] = col[o * TPB];
}

// d_in[k] = sum_o W[o][k] * delta[o]
template <typename T, int IN, int OUT>
__device__ __forceinline__ void dense_back(const T* wl, const T (&delta)[OUT], T (&din)[IN]) {
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    const T* Wm = wl + OUT;
#pragma unroll
    for (int k = 0; k < IN; ++k) din[k] = 0;
#pragma unroll
    for (int o = 0; o < OUT; ++o) {
        const V* row = reinterpret_cast<const V*>(Wm + o * IN);
        const T dl = delta[o];
#pragma unroll
        for (int v = 0; v < IN / W; ++v) {
            V w = row[v];
            if constexpr (W == 4) {
                din[4 * v] = fma(w.x, dl, din[4 * v]); din[4 * v + 1] = fma(w.y, dl, din[4 * v + 1]);
                din[4 * v + 2] = fma(w.z, dl, din[4 * v + 2]); din[4 * v + 3] = fma(w.w, dl, din[4 * v + 3]);
            } else {
                din[2 * v] = fma(w.x, dl, din[2 * v]); din[2 * v + 1] = fma(w.y, dl, din[2 * v + 1]);
            }
        }
    }
}

// row r of particle p -> problem b = t*P + p, point i.  All index math is 32-bit (64-bit integer
// division costs ~150 instructions and a dozen SGPR pairs on gfx950); only the final offsets are 64-bit.
template <typename T>
__device__ __forceinline__ const T* row_x(const MlpArgs<T>& a, int p, int r, long& b, int& i) {
    const int t = (int)((unsigned)r / (unsigned)a.n);
    i = r - t * a.n;
    const int bi = t * a.P + p;
    b = bi;
    const int xb = (a.x_div == 1) ? bi : (int)((unsigned)bi / (unsigned)a.x_div);
    return a.x + ((long)xb * a.n + i) * (long)a.dims.d_in;
}

template <typename T, int HP, int NH, int TPB>
__global__ void __launch_bounds__(TPB) mlp_fwd_kernel(MlpArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* wl = reinterpret_cast<T*>(smem_raw);
    constexpr int WELEMS = (lds_weight_elems<HP, NH>() + 3) & ~3;
    T* col = wl + WELEMS + threadIdx.x;            // thread-private column, stride TPB
    const int p = blockIdx.y;
    load_weights<T, HP, NH>(wl, a.theta + (long)p * a.theta_stride, a.dims);
    const int R = (int)a.rows_per_particle;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
        // compiler barrier: without it the (loop-invariant) LDS weight reads are hoisted out of the row
        // loop into ~1000 registers -> 256 VGPRs + scratch, 1 wave/SIMD
        asm volatile("" ::: "memory");
        long b; int i;
        const T* xp = row_x(a, p, r, b, i);
        T x[DP];
        const int d_in = a.dims.d_in;
#pragma unroll
        for (int c = 0; c < DP; ++c) {           // branch-free: clamped (always valid) load + select
            const T v = xp[c < d_in ? c : d_in - 1];
            x[c] = (c < d_in) ? v : T(0);
        }
        T o[OP];
        if constexpr (NH == 0) {
            dense_rolled<T, DP, OP, false, TPB>(wl, col, x, o);
        } else {
            T h[HP], h2[HP];
            dense_rolled<T, DP, HP, true, TPB>(wl, col, x, h);
            int off = HP + HP * DP;
#pragma unroll
            for (int l = 1; l < NH; ++l) {
                dense_rolled<T, HP, HP, true, TPB>(wl + off, col, h, h2);
#pragma unroll
                for (int k = 0; k < HP; ++k) h[k] = h2[k];
                off += HP + HP * HP;
            }
            dense_rolled<T, HP, OP, false, TPB>(wl + off, col, h, o);
        }
        T* op = a.out + (b * a.n + i) * (long)a.dims.d_out;
#pragma unroll
        for (int c = 0; c < OP; ++c) if (c < a.dims.d_out) op[c] = o[c];
    }
}

// accumulate dW[o][k..] += sum_pts delta[pt][o] * in[pt][k..] for this thread's output quads
template <typename T, int IN, int OUT, int TILE, int NQ>
__device__ __forceinline__ void reduce_tile(const T* __restrict__ dl /*[TILE][OUT+1]*/, const T* __restrict__ inp /*[TILE][IN+4]*/,
                                            int npts, T (&accw)[NQ][4], T& accb) {
    constexpr int QROW = IN / 4;
    constexpr int QTOT = OUT * QROW;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        const int q = threadIdx.x + qi * TILE;
        if (q < QTOT) {
            const int o = q / QROW, k4 = (q - o * QROW) * 4;
            T a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int pt = 0; pt < npts; ++pt) {
                const T d = dl[pt * (OUT + 1) + o];
                const T* ip = inp + pt * (IN + 4) + k4;
                a0 = fma(d, ip[0], a0); a1 = fma(d, ip[1], a1); a2 = fma(d, ip[2], a2); a3 = fma(d, ip[3], a3);
            }
            accw[qi][0] += a0; accw[qi][1] += a1; accw[qi][2] += a2; accw[qi][3] += a3;
        }
    }
    if (threadIdx.x < OUT) {
        T s = 0;
        for (int pt = 0; pt < npts; ++pt) s += dl[pt * (OUT + 1) + threadIdx.x];
        accb += s;
    }
}

template <typename T, int IN, int OUT, int TILE, int NQ>
__device__ __forceinline__ void store_layer(T* __restrict__ dst, int out_real, int in_real, const T (&accw)[NQ][4], T accb) {
    constexpr int QROW = IN / 4;
    constexpr int QTOT = OUT * QROW;
    if (threadIdx.x < out_real) dst[threadIdx.x] = accb;
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        const int q = threadIdx.x + qi * TILE;
        if (q < QTOT) {
            const int o = q / QROW, k4 = (q - o * QROW) * 4;
            if (o < out_real) {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (k4 + c < in_real) dst[out_real + o * in_real + k4 + c] = accw[qi][c];
            }
        }
    }
}

template <typename T, int HP, int NH, int TILE>
__global__ void __launch_bounds__(TILE) mlp_bwd_kernel(MlpArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* wl = reinterpret_cast<T*>(smem_raw);
    constexpr int WELEMS = (lds_weight_elems<HP, NH>() + 3) & ~3;
    constexpr int MAXW = (NH == 0) ? DP : HP;
    T* dl_t = wl + WELEMS;                              // [TILE][MAXO+1]
    constexpr int MAXO = (NH == 0) ? OP : HP;
    T* in_t = dl_t + ((TILE * (MAXO + 1) + 3) & ~3);     // [TILE][MAXW+4]
    const int p = blockIdx.y;
    load_weights<T, HP, NH>(wl, a.theta + (long)p * a.theta_stride, a.dims);

    // per-thread accumulators of this workgroup's partial weight gradient
    constexpr int NQ0 = ((NH == 0 ? OP : HP) * (DP / 4) + TILE - 1) / TILE;       // first layer (IN = DP)
    constexpr int NQH = (NH > 1) ? (HP * (HP / 4) + TILE - 1) / TILE : 1;          // hidden -> hidden
    constexpr int NQO = (NH > 0) ? (OP * (HP / 4) + TILE - 1) / TILE : 1;          // hidden -> out
    T acc0[NQ0][4]; T accb0 = 0;
    T accH[(NH > 1 ? NH - 1 : 1)][NQH][4]; T accbH[(NH > 1 ? NH - 1 : 1)];
    T accO[NQO][4]; T accbO = 0;
#pragma unroll
    for (int q = 0; q < NQ0; ++q) for (int c = 0; c < 4; ++c) acc0[q][c] = 0;
#pragma unroll
    for (int l = 0; l < (NH > 1 ? NH - 1 : 1); ++l) { accbH[l] = 0; for (int q = 0; q < NQH; ++q) for (int c = 0; c < 4; ++c) accH[l][q][c] = 0; }
#pragma unroll
    for (int q = 0; q < NQO; ++q) for (int c = 0; c < 4; ++c) accO[q][c] = 0;

    const int R = (int)a.rows_per_particle;
    const int rows_per_chunk = ((R + a.n_chunks - 1) / a.n_chunks + TILE - 1) / TILE * TILE;
    const long r_begin_l = (long)blockIdx.x * rows_per_chunk;
    const int r_begin = r_begin_l < R ? (int)r_begin_l : R;
    const int r_end = (r_begin_l + rows_per_chunk < R) ? (int)(r_begin_l + rows_per_chunk) : R;

    for (int r0 = r_begin; r0 < r_end; r0 += TILE) {
        const int r = r0 + threadIdx.x;
        const bool has = r < r_end;
        const int npts = (r_end - r0 < TILE) ? (r_end - r0) : TILE;
        T x[DP];
        T go[OP];
#pragma unroll
        for (int c = 0; c < DP; ++c) x[c] = 0;
#pragma unroll
        for (int c = 0; c < OP; ++c) go[c] = 0;
        if (has) {
            long b; int i;
            const T* xp = row_x(a, p, r, b, i);
            const int d_in = a.dims.d_in, d_out = a.dims.d_out;
#pragma unroll
            for (int c = 0; c < DP; ++c) { const T v = xp[c < d_in ? c : d_in - 1]; x[c] = (c < d_in) ? v : T(0); }
            const T* gp = a.g_out + (b * a.n + i) * (long)d_out;
#pragma unroll
            for (int c = 0; c < OP; ++c) { const T v = gp[c < d_out ? c : d_out - 1]; go[c] = (c < d_out) ? v : T(0); }
        }
        if constexpr (NH == 0) {
#pragma unroll
            for (int c = 0; c < OP; ++c) dl_t[threadIdx.x * (OP + 1) + c] = go[c];
#pragma unroll
            for (int c = 0; c < DP; ++c) in_t[threadIdx.x * (DP + 4) + c] = x[c];
            __syncthreads();
            reduce_tile<T, DP, OP, TILE, NQ0>(dl_t, in_t, npts, acc0, accb0);
            __syncthreads();
        } else {
            // forward recompute, keep all activations
            T act[NH][HP];
            dense<T, DP, HP, true>(wl, x, act[0]);
            int off = HP + HP * DP;
#pragma unroll
            for (int l = 1; l < NH; ++l) { dense<T, HP, HP, true>(wl + off, act[l - 1], act[l]); off += HP + HP * HP; }
            // output layer: delta = g_out, input = act[NH-1]
#pragma unroll
            for (int c = 0; c < OP; ++c) dl_t[threadIdx.x * (OP + 1) + c] = go[c];
#pragma unroll
            for (int c = 0; c < HP; ++c) in_t[threadIdx.x * (HP + 4) + c] = act[NH - 1][c];
            __syncthreads();
            reduce_tile<T, HP, OP, TILE, NQO>(dl_t, in_t, npts, accO, accbO);
            __syncthreads();
            T delta[HP], dprev[HP];
            dense_back<T, HP, OP>(wl + off, go, delta);
#pragma unroll
            for (int k = 0; k < HP; ++k) delta[k] *= (T(1) - act[NH - 1][k] * act[NH - 1][k]);
#pragma unroll
            for (int l = NH - 1; l >= 1; --l) {
                off -= HP + HP * HP;
                // layer l: delta (HP) x act[l-1] (HP)
#pragma unroll
                for (int c = 0; c < HP; ++c) dl_t[threadIdx.x * (HP + 1) + c] = delta[c];
#pragma unroll
                for (int c = 0; c < HP; ++c) in_t[threadIdx.x * (HP + 4) + c] = act[l - 1][c];
                __syncthreads();
                reduce_tile<T, HP, HP, TILE, NQH>(dl_t, in_t, npts, accH[l - 1], accbH[l - 1]);
                __syncthreads();
                dense_back<T, HP, HP>(wl + off, delta, dprev);
#pragma unroll
                for (int k = 0; k < HP; ++k) delta[k] = dprev[k] * (T(1) - act[l - 1][k] * act[l - 1][k]);
            }
            // first layer: delta (HP) x x (DP)
#pragma unroll
            for (int c = 0; c < HP; ++c) dl_t[threadIdx.x * (HP + 1) + c] = delta[c];
#pragma unroll
            for (int c = 0; c < DP; ++c) in_t[threadIdx.x * (DP + 4) + c] = x[c];
            __syncthreads();
            reduce_tile<T, DP, HP, TILE, NQ0>(dl_t, in_t, npts, acc0, accb0);
            __syncthreads();
        }
    }
    // ---- write this workgroup's partial gradient in the reference's flattened layout -----------
    T* dst = a.slab + ((long)blockIdx.x * a.P + p) * a.D_net;
    const MlpDims& d = a.dims;
    if constexpr (NH == 0) {
        store_layer<T, DP, OP, TILE, NQ0>(dst, d.d_out, d.d_in, acc0, accb0);
    } else {
        int prev = d.d_in, src = 0;
        store_layer<T, DP, HP, TILE, NQ0>(dst, d.hidden[0], prev, acc0, accb0);
        src += d.hidden[0] * (prev + 1); prev = d.hidden[0];
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            store_layer<T, HP, HP, TILE, NQH>(dst + src, d.hidden[l], prev, accH[l - 1], accbH[l - 1]);
            src += d.hidden[l] * (prev + 1); prev = d.hidden[l];
        }
        store_layer<T, HP, OP, TILE, NQO>(dst + src, d.d_out, prev, accO, accbO);
    }
}

// out[p, w] (+)= scale * sum_c in[c, p, w]
template <typename T>
__global__ void reduce_chunks_kernel(const T* __restrict__ in, T* __restrict__ out, long out_stride, T scale,
                                     int accumulate, int C, int P, int Wd) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)P * Wd) return;
    int p = (int)(idx / Wd), w = (int)(idx - (long)p * Wd);
    T s = 0;
    for (int c = 0; c < C; ++c) s += in[((long)c * P + p) * Wd + w];
    T* o = out + (long)p * out_stride + w;
    *o = accumulate ? *o + scale * s : scale * s;
}

static int fill_dims(MlpDims& d, int d_in, const int32_t* hidden, int n_hidden, int d_out, int& HPsel) {
    if (d_in <= 0 || d_out <= 0 || n_hidden < 0 || (n_hidden > 0 && !hidden)) return PACOH_EINVAL;
    if (d_in > DP || d_out > OP || n_hidden > VALU_MAX_HIDDEN) return PACOH_ELIMIT;
    d.d_in = d_in; d.d_out = d_out; d.n_hidden = n_hidden;
    int mx = 0;
    for (int l = 0; l < VALU_MAX_HIDDEN; ++l) d.hidden[l] = 0;
    for (int l = 0; l < n_hidden; ++l) {
        if (hidden[l] <= 0) return PACOH_EINVAL;
        if (hidden[l] > VALU_MAX_WIDTH) return PACOH_ELIMIT;
        d.hidden[l] = hidden[l];
        mx = hidden[l] > mx ? hidden[l] : mx;
    }
    HPsel = mx <= 32 ? 32 : 64;
    return PACOH_OK;
}

template <typename T> constexpr int bwd_tile(int HP) { return (HP == 32 ? 256 : 128) / (sizeof(T) == 8 ? 2 : 1); }

static int bwd_chunks(long rows_per_particle, int P, int tile) {
    // enough workgroups to fill 256 CUs several times over, but at least one tile of rows each
    long tiles = (rows_per_particle + tile - 1) / tile;
    long want = (2048 + P - 1) / P;
    long c = tiles < want ? tiles : want;
    return (int)(c < 1 ? 1 : c);
}

template <typename T, int HP, int NH>
static int launch_fwd(MlpArgs<T>& a, hipStream_t s) {
    constexpr int TPB = sizeof(T) == 8 ? 128 : 256;
    size_t lds = (size_t)(((lds_weight_elems<HP, NH>() + 3) & ~3) + TPB * (NH == 0 ? OP : HP)) * sizeof(T);
    if (lds > 160u * 1024u) return PACOH_ELIMIT;
    long blocks = (a.rows_per_particle + TPB - 1) / TPB;
    if (blocks > 8192) blocks = 8192;
    auto kern = mlp_fwd_kernel<T, HP, NH, TPB>;
    if (lds > 64u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PACOH_ELIMIT;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks, a.P), dim3(TPB), lds, s, a);
    return launch_status();
}

template <typename T, int HP, int NH>
static int launch_bwd(MlpArgs<T>& a, hipStream_t s) {
    constexpr int TILE = bwd_tile<T>(HP);
    constexpr int MAXW = (NH == 0) ? DP : HP;
    constexpr int MAXO = (NH == 0) ? OP : HP;
    size_t elems = ((lds_weight_elems<HP, NH>() + 3) & ~3) + ((TILE * (MAXO + 1) + 3) & ~3) + TILE * (MAXW + 4);
    size_t lds = elems * sizeof(T);
    if (lds > 160u * 1024u) return PACOH_ELIMIT;
    auto kern = mlp_bwd_kernel<T, HP, NH, TILE>;
    if (lds > 64u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return PACOH_ELIMIT;
    hipLaunchKernelGGL(kern, dim3(a.n_chunks, a.P), dim3(TILE), lds, s, a);
    return launch_status();
}

template <typename T, bool BWD>
static int dispatch(MlpArgs<T>& a, int HP, hipStream_t s) {
#define PACOH_MLP_CASE(hp, nh) if (HP == hp && a.dims.n_hidden == nh) return BWD ? launch_bwd<T, hp, nh>(a, s) : launch_fwd<T, hp, nh>(a, s);
    PACOH_MLP_CASE(32, 0) PACOH_MLP_CASE(32, 1) PACOH_MLP_CASE(32, 2) PACOH_MLP_CASE(32, 3)
    PACOH_MLP_CASE(64, 1) PACOH_MLP_CASE(64, 2) PACOH_MLP_CASE(64, 3)
#undef PACOH_MLP_CASE
    return PACOH_ELIMIT;
}

template <typename T>
static int mlp_common(MlpArgs<T>& a, const void* x, int x_div, const void* theta, long theta_stride, int P,
                      int d_in, const int32_t* hidden, int n_hidden, int d_out, int B, int n, int& HP) {
    if (!x || !theta || x_div <= 0 || P <= 0 || B <= 0 || n <= 0 || B % P != 0) return PACOH_EINVAL;
    int rc = fill_dims(a.dims, d_in, hidden, n_hidden, d_out, HP);
    if (rc) return rc;
    a.x = (const T*)x; a.x_div = x_div; a.theta = (const T*)theta; a.theta_stride = theta_stride;
    a.P = P; a.B = B; a.n = n; a.rows_per_particle = (long)(B / P) * n;
    if (a.rows_per_particle > 0x3fffffffL || (long)B * n > 0x7fffffffL) return PACOH_ELIMIT;
    a.D_net = mlp_param_count(a.dims);
    return PACOH_OK;
}

}  // namespace pacoh


namespace pacoh {

template <typename T>
int mlp_fwd_entry(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                  const int32_t* hidden, int n_hidden, int d_out, void* out, int B, int n, hipStream_t stream) {
    int HP = 32, rc;
    MlpArgs<T> a = {};
    if ((rc = mlp_common(a, x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, B, n, HP))) return rc;
    a.out = (T*)out;
    return dispatch<T, false>(a, HP, stream);
}

template <typename T>
int mlp_bwd_entry(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                  const int32_t* hidden, int n_hidden, int d_out, const void* g_out, void* d_theta,
                  long d_theta_stride, int accumulate, void* workspace, int B, int n, hipStream_t stream) {
    int HP = 32, rc;
    MlpArgs<T> a = {};
    if ((rc = mlp_common(a, x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, B, n, HP))) return rc;
    a.g_out = (const T*)g_out; a.slab = (T*)workspace;
    a.n_chunks = bwd_chunks(a.rows_per_particle, P, bwd_tile<T>(HP));
    if ((rc = dispatch<T, true>(a, HP, stream))) return rc;
    long tot = (long)P * a.D_net;
    hipLaunchKernelGGL(reduce_chunks_kernel<T>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, stream,
                       (const T*)workspace, (T*)d_theta, d_theta_stride, T(1), accumulate, a.n_chunks, P, a.D_net);
    return launch_status();
}

}  // namespace pacoh
