// Fused per-particle MLP, fp32, 1-4 tanh hidden layers of width <= 32, d_in <= 4, d_out <= 2 -- the mean and the
// kernel-feature network of the reference's launchers (experiments/meta_GPR_{SVGD,vi}_base_exp.py: 4 x 32; class default
// 2 x 32) -- forward and backward, ONE launch for both networks of a step (blockIdx.z selects the network).
//
// A wave owns a tile of 16*PB data points of one particle.  Activations are carried transposed, H^T[feature, point], as
// 16x16 blocks in the v_mfma_f32_16x16x4_f32 accumulator layout (lane (r = l&15, g = l>>4), register s  <->
// H^T[feature 4g+s][point r]); with the k index of a block product permuted as k = 4g+s that layout IS the B operand of
// the next layer's product and of the delta recursion, weights are the A operand (one ds_read_b128 per 4 MFMAs).
// Weight gradients contract over the point index and need the other orientation, X[feature r][point 4g+q] ("plain"):
// 16x16 blocks are turned through a per-wave LDS scratch: row stride 17 words
// and a skew of (0, 12, 32, 44) words for the four groups of four rows, which
// makes BOTH directions conflict-free under gfx950's dword-access banking (bank = word mod 32, the two 32-lane halves of a
// wave are serviced separately; see f_turn).
// Activation stash (round 3): the forward can park the top hidden layers' activations in HBM ([particle][16-point block]
// [layer][feature block] -> one 1-KiB wave store each, in the accumulator layout as it stands in the registers) and the
// backward reads them back instead of recomputing them: for the 2 x 32 networks of cfg #3 that removes 64 of the backward's
// 200 MFMAs and 32 of its 64 tanh per tile for 16 KiB of traffic per tile each way.
// The narrow first (d_in <= 4) and output (d_out <= 2) layers run on the VALU; their gradient partial sums are taken in
// the PLAIN orientation, where a lane owns a feature and sums over points: 10 + 6 accumulator registers per lane and
// two cross-lane adds at the very end (the earlier formulation summed in the transposed orientation, where a lane owns
// a point: 66 lane-private accumulators per lane, kept in 67 KB of LDS).  Hidden-layer bias gradients fall out of the
// same plain-orientation deltas the weight gradient needs.  Every workgroup (1-2 hidden layers; every wave for deeper networks) writes its own partial slab; slabs are summed in
// fixed order (deterministic).
//
// Replaces LinearVectorized / NeuralNetworkVectorized forward (meta_learn/models.py:295-317,343-349; the torch.bmm at
// :313) and its autograd backward; P = 1 is NeuralNetwork.forward (models.py:211-217).
#include "common.h"
#include "hyper_tail.h"
#include "step_tail.h"
#include <stdlib.h>

namespace pacoh {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int FMAXNH = 4;
constexpr int LWH = 36;                       // leading dimension of a hidden weight matrix in LDS
constexpr int HBLK = 32 * LWH + 32;           // one hidden layer: W[32][36] | b[32]
constexpr int F_OFF_W1 = 0;                   // W1 [32][4]
constexpr int F_OFF_B1 = 128;                 // b1 [32]
constexpr int F_OFF_H = 160;                  // hidden layers 2..NH
__host__ __device__ constexpr int f_off_out(int nh) { return F_OFF_H + (nh - 1) * HBLK; }      // W_out [2][32] | b_out [2] (+2 pad)
__host__ __device__ constexpr int f_welems(int nh) { return f_off_out(nh) + 68; }
__host__ __device__ constexpr int f_dnet_max(int nh) { return 32 * 5 + (nh - 1) * 32 * 33 + 2 * 33; }   // parameters of a d_in 4 / width 32 / d_out 2 network
__host__ __device__ constexpr bool bwd_wg_slab(int nh) { return nh <= 2; }     // backward: one gradient slab per workgroup (else per wave)
constexpr int TSTRIDE = 4;                    // staged tile: x[point][4] (16-byte rows: conflict-free 16-byte writes and broadcast reads) ...
constexpr int TSTAGE = 6;                     // ... followed by the upstream gradients feature-major, g[o][point]: TSTAGE floats per point
constexpr int TLD = 17;                       // scratch row stride of a block transpose; row i sits tskew(i >> 2) words further on
__host__ __device__ constexpr int tskew(int k) { return 12 * (k & 1) + 32 * (k >> 1); }
constexpr int TRS = 320;                      // one transpose scratch block (16 * 17 + tskew(3) + 4, rounded)

struct FusedNet {
    long theta_off;            // element offset of the network's block inside a theta row
    float* out;                // fwd: [B, n, d_out]
    const float* g_out;        // bwd: [B, n, d_out]
    float* slab;               // bwd: [slabs][P][D_net]
    float* stash;              // activation stash of this network (fwd: written, bwd: read), NULL = none
    int d_out, D_net;
};

struct FusedArgs {
    const float* x; int x_div;
    const float* theta; long theta_stride;
    FusedNet net[2];
    int P, n, R;               // R = rows (points) per particle
    int d_in, nh;
    int h[FMAXNH];
    int tiles_per_wg;
    int n_stash;               // the top n_stash hidden layers travel through FusedNet::stash (0 = recompute everything)
    int nblk;                  // 16-point blocks per particle in the stash (a multiple of 4)
    int tail_z;                // workgroups with blockIdx.z == tail_z (> 0) run a tail instead of a network: forward sv (step_tail.h),
    SvgdDistTail<float> sv;    // backward the SVGD bandwidth block (svgd_bandwidth_block on bw_d2[bw_P, bw_P] -> bw_out)
    const float* bw_d2; float* bw_out; int bw_P;
    long* adv_counter;         // backward: the same workgroup advances the pipelined feed's step counter (PACOH-MAP; the slab reduction
                               // behind this launch reads it)
};

// stash element (particle p, 16-point block blk, slot, feature block fb): 256 floats, lane-major f32x4
__device__ __forceinline__ long stash_off(const FusedArgs& a, int p, int blk, int slot, int fb, int lane) {
    return (((((long)p * a.nblk + blk) * a.n_stash + slot) * 2 + fb) << 8) + (lane << 2);
}

__device__ __forceinline__ f32x4 fmfma(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// tanh(y) = 1 - 2 / (1 + exp2(TSC y)), TSC = 2 log2(e).  The weights and biases of every tanh layer are multiplied by TSC when
// they are copied to LDS, so the matrix cores deliver the exp2 argument itself and an activation costs v_exp + v_add + v_rcp +
// v_fma (one instruction less, on each of the 64 activations per lane, tile and pass).  The backward pass multiplies by the
// SCALED hidden weights in its delta recursion, so delta_l comes out as TSC^(NH - l) times the true one (l = 1..NH): the
// accumulated weight gradients are scaled back once, where the slab is written.
constexpr float TSC = 2.8853900817779268f, INV_TSC = 0.34657359027997264f;
__device__ __forceinline__ float tanh_pre(float a) {
    return fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(a)), 1.0f);
}

// zero-padded weights of one particle's network -> LDS (tanh layers pre-multiplied by TSC)
template <int NH>
__device__ void fused_load_weights(float* wl, const float* __restrict__ th, const FusedArgs& a, int d_out) {
    for (int q = threadIdx.x; q < f_welems(NH); q += blockDim.x) wl[q] = 0.0f;
    __syncthreads();
    int src = 0;
    const int h0 = a.h[0], d_in = a.d_in;
    for (int q = threadIdx.x; q < h0; q += blockDim.x) wl[F_OFF_B1 + q] = TSC * th[q];
    for (int q = threadIdx.x; q < h0 * d_in; q += blockDim.x) { const int o = q / d_in, k = q - o * d_in; wl[F_OFF_W1 + o * 4 + k] = TSC * th[h0 + q]; }
    src += h0 * (d_in + 1);
    int prev = h0;
#pragma unroll
    for (int l = 1; l < NH; ++l) {
        const int hl = a.h[l];
        float* dst = wl + F_OFF_H + (l - 1) * HBLK;
        for (int q = threadIdx.x; q < hl; q += blockDim.x) dst[32 * LWH + q] = TSC * th[src + q];
        for (int q = threadIdx.x; q < hl * prev; q += blockDim.x) { const int o = q / prev, k = q - o * prev; dst[o * LWH + k] = TSC * th[src + hl + q]; }
        src += hl * (prev + 1);
        prev = hl;
    }
    float* dst = wl + f_off_out(NH);
    for (int q = threadIdx.x; q < d_out; q += blockDim.x) dst[64 + q] = th[src + q];
    for (int q = threadIdx.x; q < d_out * prev; q += blockDim.x) { const int o = q / prev, k = q - o * prev; dst[o * 32 + k] = th[src + d_out + q]; }
    __syncthreads();
}

// Stage the tile's inputs (and upstream gradients) in the wave's LDS block: st[pt * 4 + 0..3] = x (zero padded),
// st[16 PB * 4 + o * 16 PB + pt] = g[pt][o] (zero for rows past the end; dword accesses of consecutive lanes: no bank conflicts --
// as 8-byte fields of 32-byte rows the compiler's ds_read2_b64 pairs were 4-way conflicted).
// Returns the output row (problem*n + point) of this lane's point, or -1.
template <int PB, bool BWD>
__device__ __forceinline__ long stage_tile(const FusedArgs& a, const FusedNet& nt, int p, int row0, float* st, int lane) {
    long orow = -1;
    if (lane < 16 * PB) {
        const int row = row0 + lane;
        const bool valid = row < a.R;
        const int rr = valid ? row : a.R - 1;
        const int t = (int)((unsigned)rr / (unsigned)a.n), i = rr - t * a.n;
        const int bi = t * a.P + p;
        const int xb = a.x_div == 1 ? bi : (int)((unsigned)bi / (unsigned)a.x_div);
        const float* xq = a.x + ((long)xb * a.n + i) * (long)a.d_in;
        float4 xv;
        xv.x = xq[0];
        xv.y = a.d_in > 1 ? xq[1] : 0.0f;
        xv.z = a.d_in > 2 ? xq[2] : 0.0f;
        xv.w = a.d_in > 3 ? xq[3] : 0.0f;
        *reinterpret_cast<float4*>(st + lane * TSTRIDE) = xv;
        const long o = (long)bi * a.n + i;
        if (BWD) {
            const float* gq = nt.g_out + o * nt.d_out;
            st[16 * PB * TSTRIDE + lane] = valid ? gq[0] : 0.0f;
            st[16 * PB * (TSTRIDE + 1) + lane] = (valid && nt.d_out > 1) ? gq[1] : 0.0f;
        }
        orow = valid ? o : -1;
    }
    asm volatile("" ::: "memory");                   // LDS operations of one wave execute in order: only the compiler must keep it
    return orow;
}

// H1^T = tanh(W1 x + b1): one MFMA per (feature block, point block), k = g covers d_in <= 4
template <int PB>
__device__ __forceinline__ void f_layer1(const float* wl, const float* st, int r, int g, f32x4 (&H)[2][PB]) {
    float bx[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) bx[pb] = st[(pb * 16 + r) * TSTRIDE + g];
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        f32x4 bias;
#pragma unroll
        for (int s = 0; s < 4; ++s) bias[s] = wl[F_OFF_B1 + fb * 16 + 4 * g + s];
        const float aw = wl[F_OFF_W1 + (fb * 16 + r) * 4 + g];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            f32x4 acc = fmfma(aw, bx[pb], bias);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[s] = tanh_pre(acc[s]);
            H[fb][pb] = acc;
        }
    }
}

// OUT^T = tanh(W IN^T + b), W: LDS [32][LWH] | b[32]
template <int PB>
__device__ __forceinline__ void f_hidden(const float* W, int r, int g, const f32x4 (&IN)[2][PB], f32x4 (&OUT)[2][PB]) {
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        f32x4 bias;
#pragma unroll
        for (int s = 0; s < 4; ++s) bias[s] = W[32 * LWH + ob * 16 + 4 * g + s];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) OUT[ob][pb] = bias;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float4 aw = *reinterpret_cast<const float4*>(W + (ob * 16 + r) * LWH + kb * 16 + 4 * g);
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                f32x4 acc = OUT[ob][pb];
                acc = fmfma(aw.x, IN[kb][pb][0], acc); acc = fmfma(aw.y, IN[kb][pb][1], acc);
                acc = fmfma(aw.z, IN[kb][pb][2], acc); acc = fmfma(aw.w, IN[kb][pb][3], acc);
                OUT[ob][pb] = acc;
            }
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int s = 0; s < 4; ++s) OUT[ob][pb][s] = tanh_pre(OUT[ob][pb][s]);
    }
}

// D_in^T = (W^T D_out^T) .* (1 - H_in^2), written over H_in
template <int PB>
__device__ __forceinline__ void f_delta(const float* W, int r, int g, const f32x4 (&DOUT)[2][PB], f32x4 (&HIN)[2][PB]) {
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        f32x4 acc[PB];
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) acc[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            const float* wp = W + (ob * 16 + 4 * g) * LWH + fb * 16 + r;
            const float a0 = wp[0], a1 = wp[LWH], a2 = wp[2 * LWH], a3 = wp[3 * LWH];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                acc[pb] = fmfma(a0, DOUT[ob][pb][0], acc[pb]); acc[pb] = fmfma(a1, DOUT[ob][pb][1], acc[pb]);
                acc[pb] = fmfma(a2, DOUT[ob][pb][2], acc[pb]); acc[pb] = fmfma(a3, DOUT[ob][pb][3], acc[pb]);
            }
        }
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int s = 0; s < 4; ++s) { const float h = HIN[fb][pb][s]; HIN[fb][pb][s] = acc[pb][s] * (1.0f - h * h); }
    }
}

// 16x16 block: lane (r,g) holds X[feature 4g+s][point r] in register s, returns X[feature r][point 4g+q] in register q.
// Scratch word of element (i, j): 17 i + tskew(i >> 2) + j, tskew = (0, 12, 32, 44) = (0, 12, 0, 12) mod 32.  Dword LDS accesses
// are banked modulo 32 words and serviced per 32-lane half (g in {0,1} | {2,3}).  Writes: register s of lane (r,g) goes to word
// 17 s + r + 68 g + tskew(g): the two lane rows of a half land 80 = 16 (mod 32) banks apart -> 32 distinct banks.  Reads: lane
// (r,g) reads words 17 r + tskew(r >> 2) + 4 g + q; over r = 0..15 the first two terms take every bank with bit 2 clear exactly once
// ({0,17,2,19 | 16,1,18,3 | 8,25,10,27 | 24,9,26,11}), 4 g fills in the others -> 32 distinct banks.  (The plain stride-17 layout of
// rounds 1-2 was 2-way conflicted in both directions: SQ_LDS_BANK_CONFLICT exceeded the LDS-active cycles of the backward kernel.)
// wr = 68 g + tskew(g) + r and rd = 17 r + tskew(r >> 2) + 4 g are computed once per wave.
// (16-byte writes + strided reads, row stride 20, were 22 % slower at 64-point tiles)
__device__ __forceinline__ f32x4 f_turn(float* scr, const f32x4& v, int wr, int rd) {
    f32x4 o;
#pragma unroll
    for (int s = 0; s < 4; ++s) scr[wr + s * TLD] = v[s];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = scr[rd + q];
    asm volatile("" ::: "memory");
    return o;
}

__device__ __forceinline__ void f_wgrad(f32x4& acc, const f32x4& Ap, const f32x4& Bp) {
    acc = fmfma(Ap[0], Bp[0], acc); acc = fmfma(Ap[1], Bp[1], acc);
    acc = fmfma(Ap[2], Bp[2], acc); acc = fmfma(Ap[3], Bp[3], acc);
}

__device__ __forceinline__ void stash_st(const f32x4& v, f32x4* q) {
    __builtin_nontemporal_store(v, q);
}
__device__ __forceinline__ f32x4 stash_ld(const f32x4* q) {
    return *q;
}

template <int PB>
__device__ __forceinline__ void stash_put(const FusedArgs& a, const FusedNet& nt, int p, int blk0, int slot, int lane,
                                          const f32x4 (&H)[2][PB]) {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
            stash_st(H[fb][pb], reinterpret_cast<f32x4*>(nt.stash + stash_off(a, p, blk0 + pb, slot, fb, lane)));
}

// blocks that start past the end of the particle's rows were never written by a forward with a smaller tile: zeros
template <int PB>
__device__ __forceinline__ void stash_get(const FusedArgs& a, const FusedNet& nt, int p, int blk0, int slot, int lane,
                                          f32x4 (&H)[2][PB]) {
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const bool in = (blk0 + pb) * 16 < a.R;
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
            H[fb][pb] = in ? stash_ld(reinterpret_cast<const f32x4*>(nt.stash + stash_off(a, p, blk0 + pb, slot, fb, lane)))
                           : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// ---- forward ------------------------------------------------------------------------------------------------------
template <int NH, int PB, int MINW>
__global__ void __launch_bounds__(256, MINW) mlp_fused_fwd_kernel(FusedArgs a) {
    __shared__ __attribute__((aligned(16))) float wl[f_welems(NH)];
    __shared__ __attribute__((aligned(16))) float stage[4][16 * PB * TSTAGE];
    if (a.tail_z > 0 && (int)blockIdx.z == a.tail_z) {    // the SVGD step's distance matrix rides in this launch
        svgd_dist_tail<float>(a.sv, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
        return;
    }
    const FusedNet& nt = a.net[blockIdx.z];
    const int p = blockIdx.y;
    fused_load_weights<NH>(wl, a.theta + (long)p * a.theta_stride + nt.theta_off, a, nt.d_out);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // (uniform: the tile loop and every stash address on the scalar unit)
    const int r = lane & 15, g = lane >> 4;
    float* st = stage[wave];
    const float* wo = wl + f_off_out(NH);
    float w3r[2][2][4];                                   // W_out[o][feature fb*16+4g+s] (rows >= d_out are zero)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int s = 0; s < 4; ++s) w3r[o][fb][s] = wo[o * 32 + fb * 16 + 4 * g + s];
    const float b30 = wo[64], b31 = wo[65];
    constexpr int TP = 16 * PB;
    for (int tl = wave; tl < a.tiles_per_wg; tl += 4) {
        const int row0 = (blockIdx.x * a.tiles_per_wg + tl) * TP;
        if (row0 >= a.R) break;
        const long orow_l = stage_tile<PB, false>(a, nt, p, row0, st, lane);
        f32x4 HA[2][PB], HB[2][PB];
        const int blk0 = row0 >> 4;
        const int ns = nt.stash ? a.n_stash : 0;
        f_layer1<PB>(wl, st, r, g, HA);
        if (ns >= NH) stash_put<PB>(a, nt, p, blk0, 0, lane, HA);
#pragma unroll
        for (int l = 1; l < NH; ++l) {
            if (l & 1) f_hidden<PB>(wl + F_OFF_H + (l - 1) * HBLK, r, g, HA, HB);
            else f_hidden<PB>(wl + F_OFF_H + (l - 1) * HBLK, r, g, HB, HA);
            if (l >= NH - ns) stash_put<PB>(a, nt, p, blk0, l - (NH - ns), lane, (l & 1) ? HB : HA);
        }
        const f32x4 (&HL)[2][PB] = ((NH - 1) & 1) ? HB : HA;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            float v0 = 0.f, v1 = 0.f;
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int s = 0; s < 4; ++s) { v0 = fmaf(w3r[0][fb][s], HL[fb][pb][s], v0); v1 = fmaf(w3r[1][fb][s], HL[fb][pb][s], v1); }
            v0 += __shfl_xor(v0, 16, 64); v0 += __shfl_xor(v0, 32, 64);           // sum over the four feature groups g
            v1 += __shfl_xor(v1, 16, 64); v1 += __shfl_xor(v1, 32, 64);
            const long orow = __shfl(orow_l, pb * 16 + r, 64);                     // the row of point pb*16 + r
            if (g == 0 && orow >= 0) {
                nt.out[orow * nt.d_out] = v0 + b30;
                if (nt.d_out > 1) nt.out[orow * nt.d_out + 1] = v1 + b31;
            }
        }
    }
}

// ---- backward -----------------------------------------------------------------------------------------------------
template <int NH, int PB, int MINW>
__global__ void __launch_bounds__(256, MINW) mlp_fused_bwd_kernel(FusedArgs a) {
    // one LDS block: weights | staged tiles | transpose scratch.  NH <= 2: once the tile loop is over the same bytes take the four
    // waves' gradient slabs, which are summed there (fixed order) into ONE slab per workgroup -- a quarter of the partial slabs the
    // reduction kernel has to read, written with coalesced stores (bwd_wg_slab; deeper networks keep a slab per wave: 4 D_net
    // floats do not fit beside their larger weight image)
    constexpr int L_WL = (f_welems(NH) + 3) & ~3, L_ST = 4 * 16 * PB * TSTAGE, L_TS = 4 * 2 * TRS;
    constexpr int L_RED = bwd_wg_slab(NH) ? 4 * f_dnet_max(NH) : 0;
    constexpr int L_ALL = (L_WL + L_ST + L_TS) > L_RED ? (L_WL + L_ST + L_TS) : L_RED;
    __shared__ __attribute__((aligned(16))) float lds[L_ALL];
    if (a.tail_z > 0 && (int)blockIdx.z == a.tail_z) {     // the SVGD step's median bandwidth rides in this launch (one workgroup)
        if (blockIdx.x == 0 && blockIdx.y == 0) {
            if (a.bw_out) svgd_bandwidth_block<float>(a.bw_d2, a.bw_P, a.bw_out);
            if (a.adv_counter && threadIdx.x == 0) *a.adv_counter += 1;
        }
        return;
    }
    float* wl = lds;
    const FusedNet& nt = a.net[blockIdx.z];
    const int p = blockIdx.y;
    fused_load_weights<NH>(wl, a.theta + (long)p * a.theta_stride + nt.theta_off, a, nt.d_out);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    float* st = lds + L_WL + wave * (16 * PB * TSTAGE);
    float* sc0 = lds + L_WL + L_ST + wave * (2 * TRS);
    float* sc1 = sc0 + TRS;
    const int twr = 4 * TLD * g + tskew(g) + r, trd = TLD * r + tskew(r >> 2) + 4 * g;
    const float* wo = wl + f_off_out(NH);

    // accumulators (plain orientation: this lane = feature r of a block, partial sums over the points 4g+q of every block)
    f32x4 aW[NH > 1 ? NH - 1 : 1][2][2] = {};             // hidden weight gradients, accumulator layout [out 4g+s][in r]
    float aBh[NH > 1 ? NH - 1 : 1][2] = {};               // hidden bias gradients [layer][feature block]
    float aW1[2][4] = {}, aB1[2] = {};                    // first layer [feature block][k]
    f32x4 aWoT[2][2] = {};                                // output layer, transposed orientation: [o][feature block][s] of this lane's point(s)
    float aBoT[2] = {};

    constexpr int TP = 16 * PB;
    for (int tl = wave; tl < a.tiles_per_wg; tl += 4) {
        const int row0 = (blockIdx.x * a.tiles_per_wg + tl) * TP;
        if (row0 >= a.R) break;
        stage_tile<PB, true>(a, nt, p, row0, st, lane);
        // ---- activations: the top `ns` hidden layers from the forward's stash, the ones below recomputed -------------
        f32x4 H[NH][2][PB];
        const int blk0 = row0 >> 4;
        const int ns = nt.stash ? a.n_stash : 0;
#pragma unroll
        for (int l = NH - 1; l >= 0; --l)
            if (l >= NH - ns) stash_get<PB>(a, nt, p, blk0, l - (NH - ns), lane, H[l]);
        if (ns < NH) f_layer1<PB>(wl, st, r, g, H[0]);
#pragma unroll
        for (int l = 1; l < NH; ++l)
            if (l < NH - ns) f_hidden<PB>(wl + F_OFF_H + (l - 1) * HBLK, r, g, H[l - 1], H[l]);
        // ---- output layer (VALU).  dW_out, db_out are summed in the TRANSPOSED orientation the activations are in (a lane owns a
        //      point and the features 4g+s: 16 + 2 lane-private accumulators, summed over the 16 lanes of a row once, at the very
        //      end) -- no block transposes for this layer (round 3; rounds 1-2 turned both feature blocks of every 16-point block
        //      through LDS: 8 of a tile's 32 round trips).  W_out comes from LDS per tile instead of living in 16 registers. ----
        {
            f32x4 w3[2][2];
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int fb = 0; fb < 2; ++fb) w3[o][fb] = *reinterpret_cast<const f32x4*>(wo + o * 32 + fb * 16 + 4 * g);
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                float2 gr;                                 // upstream gradient of this lane's point
                gr.x = st[16 * PB * TSTRIDE + pb * 16 + r];
                gr.y = st[16 * PB * (TSTRIDE + 1) + pb * 16 + r];
                aBoT[0] += gr.x; aBoT[1] += gr.y;
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const float h = H[NH - 1][fb][pb][s];
                        aWoT[0][fb][s] = fmaf(gr.x, h, aWoT[0][fb][s]);
                        aWoT[1][fb][s] = fmaf(gr.y, h, aWoT[1][fb][s]);
                        const float d = fmaf(w3[1][fb][s], gr.y, w3[0][fb][s] * gr.x);
                        H[NH - 1][fb][pb][s] = d * (1.0f - h * h);
                    }
            }
        }
        // ---- hidden layers NH .. 2: H[l] holds delta_l^T, H[l-1] the activations below it ---------------------------------
#pragma unroll
        for (int l = NH - 1; l >= 1; --l) {
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                f32x4 Dp[2], Hp[2];
                Dp[0] = f_turn(sc0, H[l][0][pb], twr, trd);     Hp[0] = f_turn(sc1, H[l - 1][0][pb], twr, trd);
                Dp[1] = f_turn(sc0, H[l][1][pb], twr, trd);     Hp[1] = f_turn(sc1, H[l - 1][1][pb], twr, trd);
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) {
                    aBh[l - 1][ob] += (Dp[ob][0] + Dp[ob][1]) + (Dp[ob][2] + Dp[ob][3]);
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) f_wgrad(aW[l - 1][ob][kb], Dp[ob], Hp[kb]);
                }
            }
            f_delta<PB>(wl + F_OFF_H + (l - 1) * HBLK, r, g, H[l], H[l - 1]);
        }
        // ---- first layer (VALU, plain orientation): H[0] holds delta_1^T ----------------------------------------------------
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const f32x4 Dp0 = f_turn(sc0, H[0][0][pb], twr, trd);
            const f32x4 Dp1 = f_turn(sc1, H[0][1][pb], twr, trd);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 xp = *reinterpret_cast<const float4*>(st + (pb * 16 + 4 * g + q) * TSTRIDE);
                aW1[0][0] = fmaf(Dp0[q], xp.x, aW1[0][0]); aW1[0][1] = fmaf(Dp0[q], xp.y, aW1[0][1]);
                aW1[0][2] = fmaf(Dp0[q], xp.z, aW1[0][2]); aW1[0][3] = fmaf(Dp0[q], xp.w, aW1[0][3]);
                aW1[1][0] = fmaf(Dp1[q], xp.x, aW1[1][0]); aW1[1][1] = fmaf(Dp1[q], xp.y, aW1[1][1]);
                aW1[1][2] = fmaf(Dp1[q], xp.z, aW1[1][2]); aW1[1][3] = fmaf(Dp1[q], xp.w, aW1[1][3]);
                aB1[0] += Dp0[q]; aB1[1] += Dp1[q];
            }
        }
    }
    // ---- sums over the four point groups g, then this wave's slab in the reference's flattened layout -------------------
    auto rg = [](float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; };
    if (bwd_wg_slab(NH)) __syncthreads();               // (every wave is done with the weights / tiles / scratch the slabs overwrite)
    float* dst = bwd_wg_slab(NH) ? lds + wave * nt.D_net : nt.slab + ((long)(blockIdx.x * 4 + wave) * a.P + p) * nt.D_net;
    const int d_in = a.d_in, h0 = a.h[0];
    float unscale[NH];                                 // layer j+1's delta carries TSC^(NH-1-j) (see tanh_pre)
    unscale[NH - 1] = 1.0f;
#pragma unroll
    for (int j = NH - 2; j >= 0; --j) unscale[j] = unscale[j + 1] * INV_TSC;
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        const int o = fb * 16 + r;
        const float b = rg(aB1[fb]) * unscale[0];
        if (g == 0 && o < h0) dst[o] = b;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const float v = rg(aW1[fb][k]) * unscale[0]; if (g == 0 && o < h0 && k < d_in) dst[h0 + o * d_in + k] = v; }
    }
    int off = h0 * (d_in + 1), prev = h0;
#pragma unroll
    for (int l = 1; l < NH; ++l) {
        const int hl = a.h[l];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            const float b = rg(aBh[l - 1][ob]) * unscale[l];
            if (g == 0 && ob * 16 + r < hl) dst[off + ob * 16 + r] = b;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int o = ob * 16 + 4 * g + s, k = kb * 16 + r;
                    if (o < hl && k < prev) dst[off + hl + o * prev + k] = aW[l - 1][ob][kb][s] * unscale[l];
                }
        }
        off += hl * (prev + 1);
        prev = hl;
    }
    // output layer: sums over the 16 points (lanes) of a row; the four lane rows hold the point sums of one 16-block each in
    // turn (every block's points sit in all four rows g with different FEATURES), so a row sum is the whole sum for its features
    auto rs = [](float v) {
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        return v;
    };
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const float b = rs(aBoT[o]);                       // (the same in every row: g(point r) does not depend on g)
        if (lane == 0 && o < nt.d_out) dst[off + o] = b;
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float v = rs(aWoT[o][fb][s]);
                const int k = fb * 16 + 4 * g + s;
                if (r == 0 && o < nt.d_out && k < prev) dst[off + nt.d_out + o * prev + k] = v;
            }
    }
    if (bwd_wg_slab(NH)) {
        __syncthreads();
        float* out = nt.slab + ((long)blockIdx.x * a.P + p) * nt.D_net;
        for (int e = threadIdx.x; e < nt.D_net; e += 256)
            out[e] = (lds[e] + lds[nt.D_net + e]) + (lds[2 * nt.D_net + e] + lds[3 * nt.D_net + e]);
    }
}

// out[p, w] (+)= sum_c in[c, p, w] for up to two networks in one launch (blockIdx.y): 8 lanes per output element split the
// slabs, fixed order -> deterministic
struct SlabReduce { const float* in; float* out; int Wd; int col0; };    // col0: the network block's first column in the parameter row
// (blockIdx.y == nets, tail_blocks > 0: the step's hyper-parameter reduction rides in this launch -- hyper_tail.h)
// LSH: log2 of the lanes that share an output element -- 3 (8 lanes) behind the fused backward kernel (a few dozen slabs); 5 behind the
// task-fused PACOH-MAP kernel, whose 256 slabs (one per task) cost 8 lanes 32 dependent-latency loads each: 12 us for a 5 us kernel
template <int LSH>
__global__ void __launch_bounds__(256) fused_reduce_slab_kernel(SlabReduce s0, SlabReduce s1, long out_stride, int accumulate, int C, int P,
                                                                int nets, HyperBwdArgs<float> tail, int tail_blocks, float* img_th, const int* img_map) {
    constexpr int LANES = 1 << LSH;
    if ((int)blockIdx.y == nets) {
        __shared__ float red[4];
        if ((int)blockIdx.x < tail_blocks) hyper_tail_block<float>(tail, blockIdx.x, red);
        return;
    }
    const SlabReduce& sr = blockIdx.y ? s1 : s0;
    const long tot = (long)P * sr.Wd;
    const long idx = ((long)blockIdx.x * 256 + threadIdx.x) >> LSH;
    const int part = threadIdx.x & (LANES - 1);
    float s = 0;
    if (idx < tot) for (int c = part; c < C; c += LANES) s += sr.in[(long)c * tot + idx];
#pragma unroll
    for (int m = 1; m < LANES; m <<= 1) s += __shfl_xor(s, m, 64);
    if (idx < tot && part == 0) {
        const int p = (int)(idx / sr.Wd), w = (int)(idx - (long)p * sr.Wd);
        float* o = sr.out + (long)p * out_stride + w;
        const float gv = accumulate ? *o + s : s;
        *o = gv;
        if (P == 1) {                                                      // PACOH-MAP: the AdamW step on this entry (hyper_tail.h)
            const float now = adam_inline<float>(tail.opt, sr.col0 + w, gv);
            // (behind the task-fused kernel, map_task.hip: the entry's copy in the parameter image that kernel's prologue reads)
            if (img_th && tail.opt.param) { const int li = img_map[sr.col0 + w]; if (li >= 0) img_th[li] = now; }
        }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------

bool mlp_fused_applicable(int d_in, const int32_t* hidden, int n_hidden, int d_out) {
    const bool on = g_sw.fused_mlp;
    if (!on || n_hidden < 1 || n_hidden > FMAXNH || d_in < 1 || d_in > 4 || d_out < 1 || d_out > 2) return false;
    for (int l = 0; l < n_hidden; ++l) if (hidden[l] < 1 || hidden[l] > 32) return false;
    return true;
}

static int fused_dnet(int d_in, const int32_t* hidden, int n_hidden, int d_out) {
    int prev = d_in, c = 0;
    for (int l = 0; l < n_hidden; ++l) { c += hidden[l] * (prev + 1); prev = hidden[l]; }
    return c + d_out * (prev + 1);
}

// point blocks (of 16) per tile.  Small launches -- fewer 64-point tiles than the chip has wave slots, e.g. PACOH-MAP's 256 tasks x 32
// points x one parameter row -- are one tile per wave either way and pure latency: 32-point tiles halve that latency
static bool fused_small(int R, int P, int nets) { return (long)((R + 63) / 64) * P * nets <= 512; }
static int fused_bwd_pb(int n_hidden, int R, int P, int nets) {
    return g_sw.fused_bwd_pb > 0 ? g_sw.fused_bwd_pb : ((n_hidden <= 2 && !fused_small(R, P, nets)) ? 4 : 2);
}
static int fused_fwd_pb(int, int R, int P, int nets) { return g_sw.fused_fwd_pb > 0 ? g_sw.fused_fwd_pb : (fused_small(R, P, nets) ? 2 : 4); }

// resident workgroups of a kernel on the whole chip (occupancy query, cached per kernel)
template <typename K> static int resident_wgs(K kern) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v; }
    return per_cu * cus;
}

// Workgroups per (particle, network) for the backward: a grid slightly above a multiple of what is resident costs a whole
// extra round; pick the tiles per workgroup (a wave takes every 4th tile) that minimises rounds x (tiles per wave + overhead)
static int fused_chunks(int R, int P, int nets, int tp, int resident, double overhead) {
    const int tiles = (R + tp - 1) / tp;
    int best_tpw = 4;
    double best = 1e30;
    for (int tpw = 4; tpw <= 1024; tpw += 4) {
        const long wgs = (long)((tiles + tpw - 1) / tpw) * P * nets;
        const long rounds = (wgs + resident - 1) / resident;
        const double cost = (double)rounds * (tpw / 4 + overhead);
        if (cost < best - 1e-9) { best = cost; best_tpw = tpw; }
        if (tpw >= tiles) break;
    }
    return (tiles + best_tpw - 1) / best_tpw;
}

// Hidden layers (counted from the top) whose activations the forward parks in HBM for the backward.  Default: all but the first
// (whose recomputation is one MFMA per block on d_in <= 4 inputs) -- measured at the cfg #3 shape (1024 tasks x 20 particles x
// 64 points, both networks; tools/mlp_time.py): 2 x 32: forward 88 -> 100 us, backward 207 -> 171 us (stashing the first layer too:
// forward 141 us); 4 x 32: forward + backward 801 us without, 694 / 665 / 641 / 663 us with 1 / 2 / 3 / 4 layers stashed.
// PACOH_MLP_STASH=k stashes k layers (0: recompute everything, as rounds 1-2 did).
static int fused_n_stash(int n_hidden) {
    const int want = g_sw.mlp_stash;                               // (PACOH_MLP_STASH, switches.h; -1: the dispatcher's choice)
    if (want < 0) return n_hidden > 1 ? n_hidden - 1 : 0;
    return want > n_hidden ? n_hidden : want;
}
static int fused_nblk(int B, int P, int n) { return 4 * (int)(((long)(B / P) * n + 63) / 64); }

// bytes of activation stash for `nets` networks (0: this build / shape keeps none)
size_t mlp_fused_stash_bytes(int B, int P, int n, int n_hidden, int nets) {
    return (size_t)nets * P * fused_nblk(B, P, n) * fused_n_stash(n_hidden) * 2 * 256 * sizeof(float);
}

static void fused_fill(FusedArgs& a, const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                       const int32_t* hidden, int n_hidden, int B, int n, const void* stash, int nets) {
    a.x = (const float*)x; a.x_div = x_div; a.theta = (const float*)theta; a.theta_stride = theta_stride;
    a.P = P; a.n = n; a.R = (B / P) * n; a.d_in = d_in; a.nh = n_hidden;
    for (int l = 0; l < FMAXNH; ++l) a.h[l] = l < n_hidden ? hidden[l] : 0;
    a.n_stash = stash ? fused_n_stash(n_hidden) : 0;
    a.nblk = fused_nblk(B, P, n);
    const size_t per_net = (size_t)P * a.nblk * a.n_stash * 2 * 256;
#ifdef PACOH_EXP_STASH_NETS      // experiment build (VERDICT r3 #4a): only the first PACOH_EXP_STASH_NETS networks use the stash, the others recompute
    for (int k = 0; k < 2; ++k) a.net[k].stash = (a.n_stash > 0 && k < nets && k < PACOH_EXP_STASH_NETS) ? (float*)stash + k * per_net : nullptr;
#else
    for (int k = 0; k < 2; ++k) a.net[k].stash = (a.n_stash > 0 && k < nets) ? (float*)stash + k * per_net : nullptr;
#endif
}

constexpr int BWD_MINW = 3;      // NH <= 2, 64-point tiles: three waves per SIMD (168 registers; forcing 128 spills)
constexpr int BWD_MINW_PB2 = 2;
// M is applied to the parenthesised kernel instantiation (the commas of the template arguments must not split macro arguments)
#define PACOH_FUSED_DISPATCH(KERNEL, nh, pb, M)                                                          \
    do {                                                                                                 \
        if (pb == 4) {                                                                                   \
            if (nh == 1) { M((KERNEL<1, 4, BWD_MINW>)); } else if (nh == 2) { M((KERNEL<2, 4, BWD_MINW>)); }  \
            else if (nh == 3) { M((KERNEL<3, 4, 1>)); } else { M((KERNEL<4, 4, 1>)); }                    \
        } else {                                                                                         \
            if (nh == 1) { M((KERNEL<1, 2, BWD_MINW_PB2>)); } else if (nh == 2) { M((KERNEL<2, 2, BWD_MINW_PB2>)); }            \
            else if (nh == 3) { M((KERNEL<3, 2, 2>)); } else { M((KERNEL<4, 2, 2>)); }                    \
        }                                                                                                \
    } while (0)

// nets = 1 or 2 networks of the SAME hidden shape at element offsets off[k] of the theta rows
int mlp_fused_fwd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in, const int32_t* hidden,
                  int n_hidden, int nets, const long* off, const int* d_out, void* const* out, void* stash, int B, int n,
                  hipStream_t s, const SvgdDistTail<float>* tail, bool* tail_taken) {
    FusedArgs a = {};
    fused_fill(a, x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, B, n, stash, nets);
    for (int k = 0; k < nets; ++k) { a.net[k].theta_off = off[k]; a.net[k].out = (float*)out[k]; a.net[k].d_out = d_out[k]; }
    const int pb = fused_fwd_pb(n_hidden, a.R, P, nets) == 2 ? 2 : 4;
    const int tiles = (a.R + 16 * pb - 1) / (16 * pb);
    // tiles per workgroup (a wave takes every 4th): 16 at cfg #3 (4 / 8 / 16 / 32: 109 / 100 / 95 / 94 us); small batches -- the 1/8
    // strong-scaling shard -- want fewer, or a SIMD holds two four-tile waves while its neighbour idles: the count that minimises
    // (waves per SIMD) x (tiles per wave + half a tile of fixed cost), larger counts winning ties
    a.tiles_per_wg = g_sw.fused_fwd_tpw;
    if (a.tiles_per_wg <= 0) {
        double best = 1e30;
        for (int tpw = 4; tpw <= 32; tpw *= 2) {
            const long waves = (long)((tiles + tpw - 1) / tpw) * P * nets * 4;
            const double cost = (double)((waves + 1023) / 1024) * (tpw / 4 + 0.5);
            if (cost <= best * 1.02 || tpw == 4) { if (cost < best) best = cost; a.tiles_per_wg = tpw; }
        }
    }
    const int wgs = (tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    // The tail slice of the grid has wgs x P workgroups for the P (P + 1) / 2 particle pairs (+ the snapshot rows).  A thin slice walks
    // several pairs per workgroup one after the other, each a dependent chain of loads and a barrier: at the reference launcher's shape
    // (10 particles of D = 6566, ONE tile workgroup per particle) the forward launch took 51 us, 40 of them the tail's.  Then the
    // distances get a launch of their own (the caller's pacoh_svgd_dist_advance: P x P workgroups, ~5 us).
    if (tail && (long)wgs * P * 2 < (long)tail->P * (tail->P + 1) / 2) tail = nullptr;
    if (tail_taken) *tail_taken = tail != nullptr;
    if (tail) { a.tail_z = nets; a.sv = *tail; }
    const unsigned pad = g_sw.lds_pad_mlp > 0 ? (unsigned)g_sw.lds_pad_mlp : 0u;
#define PACOH_LAUNCH_FWD(K) hipLaunchKernelGGL((K), dim3(wgs, P, nets + (tail ? 1 : 0)), dim3(256), pad, s, a)
    if (pb == 4) {
        if (n_hidden == 1) PACOH_LAUNCH_FWD((mlp_fused_fwd_kernel<1, 4, 4>)); else if (n_hidden == 2) PACOH_LAUNCH_FWD((mlp_fused_fwd_kernel<2, 4, 4>));
        else if (n_hidden == 3) PACOH_LAUNCH_FWD((mlp_fused_fwd_kernel<3, 4, 4>)); else PACOH_LAUNCH_FWD((mlp_fused_fwd_kernel<4, 4, 4>));
    } else {
        if (n_hidden == 1) PACOH_LAUNCH_FWD((mlp_fused_fwd_kernel<1, 2, 4>)); else if (n_hidden == 2) PACOH_LAUNCH_FWD((mlp_fused_fwd_kernel<2, 2, 4>));
        else if (n_hidden == 3) PACOH_LAUNCH_FWD((mlp_fused_fwd_kernel<3, 2, 4>)); else PACOH_LAUNCH_FWD((mlp_fused_fwd_kernel<4, 2, 4>));
    }
#undef PACOH_LAUNCH_FWD
    return launch_status();
}

struct FusedBwdPlan { int pb, chunks, tiles_per_wg; };

static FusedBwdPlan fused_bwd_plan(int R, int P, int nets, int n_hidden) {
    FusedBwdPlan pl;
    pl.pb = fused_bwd_pb(n_hidden, R, P, nets) == 2 ? 2 : 4;
    static int resident[FMAXNH + 1][2] = {};
    int& res = resident[n_hidden][pl.pb == 4];
    if (res == 0) {
#define PACOH_OCC(K) res = resident_wgs(K)
        PACOH_FUSED_DISPATCH(mlp_fused_bwd_kernel, n_hidden, pl.pb, PACOH_OCC);
#undef PACOH_OCC
    }
    const int tp = 16 * pl.pb;
    pl.chunks = fused_chunks(R, P, nets, tp, res, 0.4 * 64 / tp);
    const int tiles = (R + tp - 1) / tp;
    pl.tiles_per_wg = (tiles + pl.chunks - 1) / pl.chunks;
    return pl;
}

// bytes of slab space for ONE network of a (possibly two-network) backward launch
size_t mlp_fused_bwd_workspace(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out, int nets) {
    const FusedBwdPlan pl = fused_bwd_plan((B / P) * n, P, nets, n_hidden);
    return (size_t)pl.chunks * (bwd_wg_slab(n_hidden) ? 1 : 4) * P * fused_dnet(d_in, hidden, n_hidden, d_out) * sizeof(float);
}

int mlp_fused_bwd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in, const int32_t* hidden,
                  int n_hidden, int nets, const long* off, const int* d_out, const void* const* g_out, void* d_theta,
                  long d_theta_stride, int accumulate, void* workspace, const void* stash, int B, int n, hipStream_t s,
                  const HyperBwdArgs<float>* tail, long col_base) {
    FusedArgs a = {};
    fused_fill(a, x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, B, n, stash, nets);
    const FusedBwdPlan pl = fused_bwd_plan(a.R, P, nets, n_hidden);
    a.tiles_per_wg = pl.tiles_per_wg;
    float* ws = (float*)workspace;
    SlabReduce sr[2] = {};
    int wmax = 0;
    for (int k = 0; k < nets; ++k) {
        a.net[k].theta_off = off[k]; a.net[k].g_out = (const float*)g_out[k]; a.net[k].d_out = d_out[k];
        a.net[k].D_net = fused_dnet(d_in, hidden, n_hidden, d_out[k]);
        a.net[k].slab = ws;
        sr[k].in = ws; sr[k].out = (float*)d_theta + off[k]; sr[k].Wd = a.net[k].D_net; sr[k].col0 = (int)(col_base + off[k]);
        ws += (size_t)pl.chunks * (bwd_wg_slab(n_hidden) ? 1 : 4) * P * a.net[k].D_net;
        if (a.net[k].D_net > wmax) wmax = a.net[k].D_net;
    }
    // the step's SVGD bandwidth (requested with the hyper-parameter tail) is computed by one workgroup of THIS launch, fully hidden
    // behind the networks' workgroups; the tail handed to the reduction no longer carries it
    HyperBwdArgs<float> rtail = tail ? *tail : HyperBwdArgs<float>{};
    const bool bw_here = tail && tail->sv_bw, adv_here = tail && tail->nx.counter;
    if (bw_here) { a.bw_d2 = tail->sv_d2; a.bw_P = tail->sv_P; a.bw_out = tail->sv_bw; rtail.sv_bw = nullptr; }
    if (adv_here) a.adv_counter = const_cast<long*>(tail->nx.counter);
    if (bw_here || adv_here) a.tail_z = nets;
    const unsigned pad = g_sw.lds_pad_mlp > 0 ? (unsigned)g_sw.lds_pad_mlp : 0u;
#define PACOH_LAUNCH_BWD(K) hipLaunchKernelGGL(K, dim3(pl.chunks, P, nets + ((bw_here || adv_here) ? 1 : 0)), dim3(256), pad, s, a)
    PACOH_FUSED_DISPATCH(mlp_fused_bwd_kernel, n_hidden, pl.pb, PACOH_LAUNCH_BWD);
#undef PACOH_LAUNCH_BWD
    const long tot = (long)P * wmax;
    unsigned gx = (unsigned)((tot * 8 + 255) / 256);
    const int tail_blocks = tail ? hyper_tail_blocks(rtail) : 0;
    if ((unsigned)tail_blocks > gx) gx = (unsigned)tail_blocks;
    hipLaunchKernelGGL(fused_reduce_slab_kernel<3>, dim3(gx, nets + (tail ? 1 : 0)), dim3(256), 0, s,
                       sr[0], sr[1], d_theta_stride, accumulate, pl.chunks * (bwd_wg_slab(n_hidden) ? 1 : 4), P, nets,
                       rtail, tail_blocks, (float*)nullptr, (const int*)nullptr);
    return launch_status();
}

// The slab reduction as a launch of its own: the task-fused PACOH-MAP kernel (map_task.hip) writes one slab per workgroup and
// network in theta's layout; this sums them into d_theta and runs the step's tail (hyper-parameter reduction, AdamW, next batch).
// P parameter rows (PACOH-SVGD / VI behind the same kernel): slab c holds the rows' blocks one after the other, [c][p][w]; the lanes
// that share an output element follow the slab count (2 task groups per step at the reference launchers' defaults: 2 lanes).
int fused_reduce_launch(const float* slab0, int wd0, long off0, const float* slab1, int wd1, long off1, int nets, float* d_theta,
                        long d_theta_stride, int slabs, const HyperBwdArgs<float>* tail, float* img_th, const int* img_map, hipStream_t s, int P) {
    SlabReduce sr[2] = {{slab0, d_theta + off0, wd0, (int)off0}, {slab1, d_theta + off1, wd1, (int)off1}};
    const int wmax = wd0 > wd1 ? wd0 : wd1;
    const int lsh = slabs <= 2 ? 1 : (slabs <= 16 ? 3 : 5);
    unsigned gx = (unsigned)((((long)P * wmax << lsh) + 255) / 256);
    const int tail_blocks = tail ? hyper_tail_blocks(*tail) : 0;
    if ((unsigned)tail_blocks > gx) gx = (unsigned)tail_blocks;
#define PACOH_REDUCE(LSH) hipLaunchKernelGGL(fused_reduce_slab_kernel<LSH>, dim3(gx, nets + (tail ? 1 : 0)), dim3(256), 0, s, sr[0], sr[1], \
                                             d_theta_stride, 0, slabs, P, nets, tail ? *tail : HyperBwdArgs<float>{}, tail_blocks, img_th, img_map)
    if (lsh == 1) PACOH_REDUCE(1); else if (lsh == 3) PACOH_REDUCE(3); else PACOH_REDUCE(5);
#undef PACOH_REDUCE
    return launch_status();
}

}  // namespace pacoh
