// MFMA per-particle MLP (fp32, hidden widths <= 32, 1-2 hidden layers, d_in <= 16, d_out <= 8):
// forward and backward with all activations REGISTER-resident, weights in LDS.
//
// A wave owns a tile of 64 data points of one particle.  Activations are carried transposed,
// H^T[feature, point], as 16x16 blocks in the v_mfma_f32_16x16x4_f32 accumulator layout
// (lane (r = l&15, g = l>>4), reg s  <->  H^T[feature 4g+s][point r]).  With the k index of a block
// product permuted as k = 4g+s that layout IS the B operand of the next layer's product
//     H_l^T = tanh(W_l * H_{l-1}^T + b_l)            (A operand: one ds_read_b128 of W_l per 4 MFMAs)
// and of the delta chain  dH_{l-1}^T = (W_l^T * dH_l^T) .* (1 - H_{l-1}^T^2), so neither the forward
// nor the delta recursion touches LDS or HBM for activations.  Weight gradients contract over the
// point index, which needs the plain layout H[point, feature]; a block is transposed on the matrix
// core itself (P = S^T * I: 4 MFMAs, no memory traffic), after which
//     dW_l += dH_l^T-blocks x H_{l-1}-blocks          (both operands straight from registers)
// accumulates in registers across all tiles of the workgroup.  Bias gradients are per-lane partial sums
// reduced once at the end.  Per-workgroup partial gradients go to a slab summed in fixed order.
//
// Replaces LinearVectorized / NeuralNetworkVectorized forward (meta_learn/models.py:295-317,343-349; the
// torch.bmm at :313) and its autograd backward; P = 1 is NeuralNetwork.forward (models.py:211-217).
#include "common.h"
#include <stdlib.h>

namespace pacoh {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct MlpMfmaArgs {
    const float* x; int x_div;
    const float* theta; long theta_stride;
    float* out;                // fwd
    const float* g_out;        // bwd
    float* slab;               // bwd: [n_chunks][P][D_net]
    int P, n, R;               // R = rows (points) per particle
    int d_in, d_out, nh, h0, h1;
    int tiles_per_wg;
    int D_net;
};

__device__ __forceinline__ f32x4 mfma4x(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// LDS weight image (floats): W1p [32][16+4] | b1p [32] | W2p [32][32+4] | b2p [32] | W3p [16][32+4] | b3p[16]
constexpr int LW1 = 20, LW2 = 36;
constexpr int OFF_W1 = 0, OFF_B1 = OFF_W1 + 32 * LW1, OFF_W2 = OFF_B1 + 32, OFF_B2 = OFF_W2 + 32 * LW2,
              OFF_W3 = OFF_B2 + 32, OFF_B3 = OFF_W3 + 16 * LW2, W_ELEMS = OFF_B3 + 16;

// zero-padded weights of one particle -> LDS.  NH == 1: layer "2" is absent (output reads H1).
template <int NH>
__device__ void load_weights_mfma(float* wl, const float* __restrict__ th, const MlpMfmaArgs& a) {
    for (int q = threadIdx.x; q < W_ELEMS; q += blockDim.x) wl[q] = 0.0f;
    __syncthreads();
    int src = 0;
    // layer 1: [h0][d_in]
    for (int q = threadIdx.x; q < a.h0; q += blockDim.x) wl[OFF_B1 + q] = th[src + q];
    for (int q = threadIdx.x; q < a.h0 * a.d_in; q += blockDim.x) { int o = q / a.d_in, k = q - o * a.d_in; wl[OFF_W1 + o * LW1 + k] = th[src + a.h0 + q]; }
    src += a.h0 * (a.d_in + 1);
    int prev = a.h0;
    if (NH == 2) {
        for (int q = threadIdx.x; q < a.h1; q += blockDim.x) wl[OFF_B2 + q] = th[src + q];
        for (int q = threadIdx.x; q < a.h1 * prev; q += blockDim.x) { int o = q / prev, k = q - o * prev; wl[OFF_W2 + o * LW2 + k] = th[src + a.h1 + q]; }
        src += a.h1 * (prev + 1);
        prev = a.h1;
    }
    for (int q = threadIdx.x; q < a.d_out; q += blockDim.x) wl[OFF_B3 + q] = th[src + q];
    for (int q = threadIdx.x; q < a.d_out * prev; q += blockDim.x) { int o = q / prev, k = q - o * prev; wl[OFF_W3 + o * LW2 + k] = th[src + a.d_out + q]; }
    __syncthreads();
}

// (task, point) of tile row row0+q given the tile base (t0, i0): one division per tile (scalar); per row
// a conditional wrap when n >= 64 (a 64-row tile then crosses at most one task boundary)
__device__ __forceinline__ void locate(const MlpMfmaArgs& a, int t0, int i0, int q, int& t, int& i) {
    i = i0 + q; t = t0;
    if (a.n >= 64) { if (i >= a.n) { i -= a.n; ++t; } }
    else { const int w = (int)((unsigned)i / (unsigned)a.n); t += w; i -= w * a.n; }
}

// output row index (problem*n + point) of tile row row0+q, or -1 past the end
__device__ __forceinline__ long out_row(const MlpMfmaArgs& a, int p, int row0, int t0, int i0, int q) {
    if (row0 + q >= a.R) return -1;
    int t, i; locate(a, t0, i0, q, t, i);
    return (long)(t * a.P + p) * a.n + i;
}

// address of the input row of tile row row0+q for particle p (nullptr past the end)
__device__ __forceinline__ const float* xrow(const MlpMfmaArgs& a, int p, int row0, int t0, int i0, int q) {
    if (row0 + q >= a.R) return nullptr;
    int t, i; locate(a, t0, i0, q, t, i);
    const int bi = t * a.P + p;
    const int xb = a.x_div == 1 ? bi : (int)((unsigned)bi / (unsigned)a.x_div);
    return a.x + ((long)xb * a.n + i) * (long)a.d_in;
}

// H1^T blocks [2 feature blocks][4 point blocks] of one tile
__device__ __forceinline__ void layer1(const float* wl, const MlpMfmaArgs& a, const float* const (&xp)[4], int r, int g, f32x4 (&H)[2][4]) {
    const int chunks = (a.d_in + 3) >> 2;
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
        f32x4 bias;
#pragma unroll
        for (int s = 0; s < 4; ++s) bias[s] = wl[OFF_B1 + fb * 16 + 4 * g + s];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) H[fb][pb] = bias;
    }
    for (int c = 0; c < chunks; ++c) {
        const int k = 4 * c + g;                       // natural k mapping: one MFMA per chunk
        float bx[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) bx[pb] = (xp[pb] && k < a.d_in) ? xp[pb][k] : 0.0f;
#pragma unroll
        for (int fb = 0; fb < 2; ++fb) {
            const float aw = wl[OFF_W1 + (fb * 16 + r) * LW1 + k];
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) H[fb][pb] = mfma4x(aw, bx[pb], H[fb][pb]);
        }
    }
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int s = 0; s < 4; ++s) H[fb][pb][s] = act_tanh<float>(H[fb][pb][s]);
}

// OUT^T[OB feature blocks][4] = bias + W (rows r of block ob, LD LW2) * IN^T[2][4]
template <int OB, bool ACT>
__device__ __forceinline__ void layer_xs(const float* W, const float* B, int r, int g, const f32x4 (&IN)[2][4], f32x4 (&OUT)[OB][4]) {
#pragma unroll
    for (int ob = 0; ob < OB; ++ob) {
        f32x4 bias;
#pragma unroll
        for (int s = 0; s < 4; ++s) bias[s] = B[ob * 16 + 4 * g + s];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) OUT[ob][pb] = bias;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const float4 aw = *reinterpret_cast<const float4*>(W + (ob * 16 + r) * LW2 + kb * 16 + 4 * g);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                f32x4 acc = OUT[ob][pb];
                acc = mfma4x(aw.x, IN[kb][pb][0], acc); acc = mfma4x(aw.y, IN[kb][pb][1], acc);
                acc = mfma4x(aw.z, IN[kb][pb][2], acc); acc = mfma4x(aw.w, IN[kb][pb][3], acc);
                OUT[ob][pb] = acc;
            }
        }
        if (ACT) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                for (int s = 0; s < 4; ++s) OUT[ob][pb][s] = act_tanh<float>(OUT[ob][pb][s]);
        }
    }
}

// DIN^T[2][4] = W^T (W: [OB*16 rows][LD LW2]) * DOUT^T[OB][4]
template <int OB>
__device__ __forceinline__ void layer_xTs(const float* W, int r, int g, const f32x4 (&DOUT)[OB][4], f32x4 (&DIN)[2][4]) {
#pragma unroll
    for (int fb = 0; fb < 2; ++fb) {
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) DIN[fb][pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) {
            const float* wp = W + (ob * 16 + 4 * g) * LW2 + fb * 16 + r;
            const float a0 = wp[0], a1 = wp[LW2], a2 = wp[2 * LW2], a3 = wp[3 * LW2];
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                f32x4 acc = DIN[fb][pb];
                acc = mfma4x(a0, DOUT[ob][pb][0], acc); acc = mfma4x(a1, DOUT[ob][pb][1], acc);
                acc = mfma4x(a2, DOUT[ob][pb][2], acc); acc = mfma4x(a3, DOUT[ob][pb][3], acc);
                DIN[fb][pb] = acc;
            }
        }
    }
}

// transpose a 16x16 block on the matrix core: P = S^T * I
__device__ __forceinline__ f32x4 tr_block(const f32x4& S, int r, int g) {
    f32x4 P = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s) P = mfma4x(S[s], (4 * g + s == r) ? 1.0f : 0.0f, P);
    return P;
}

// acc[ob][kb] += sum over point blocks of  A_plain[pb][ob]^T-contraction  B_plain[pb][kb]
__device__ __forceinline__ void wgrad(f32x4& acc, const f32x4& Ap, const f32x4& Bp) {
    acc = mfma4x(Ap[0], Bp[0], acc); acc = mfma4x(Ap[1], Bp[1], acc);
    acc = mfma4x(Ap[2], Bp[2], acc); acc = mfma4x(Ap[3], Bp[3], acc);
}

template <int NH, bool NARROW>
// NARROW (d_out <= 2): the output layer runs on the VALU (8 FMAs per point and output + two cross-lane adds) instead of a
// 16-row MFMA layer of which 14 rows would be padding (32 of the tile's 104 MFMAs).
// (256, 4): the allocator then settles at 96 VGPRs without spills = five waves per SIMD; the default allocation took 140
// registers (three waves per SIMD, 4 % slower -- the forward pass is latency-bound, extra waves hide it) and asking for five
// waves outright made it spill 11 registers (36 MB of scratch traffic per launch in the PMC counters)
__global__ void __launch_bounds__(256, 4) mlp_mfma_fwd_kernel(MlpMfmaArgs a) {
    __shared__ __attribute__((aligned(16))) float wl[W_ELEMS];
    const int p = blockIdx.y;
    load_weights_mfma<NH>(wl, a.theta + (long)p * a.theta_stride, a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    float w3r[2][2][4], b3r[2];                          // NARROW: W3[o][feature fb*16+4g+s] (rows >= d_out are zero), b3[o]
    if (NARROW) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            b3r[o] = wl[OFF_B3 + o];
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int s = 0; s < 4; ++s) w3r[o][fb][s] = wl[OFF_W3 + o * LW2 + fb * 16 + 4 * g + s];
        }
    }
    for (int tl = 0; tl < a.tiles_per_wg; tl += 4) {
        const int tile = blockIdx.x * a.tiles_per_wg + tl + wave;
        const int row0 = tile * 64;
        if (tl + wave >= a.tiles_per_wg || row0 >= a.R) continue;
        const int t0 = (int)((unsigned)row0 / (unsigned)a.n), i0 = row0 - t0 * a.n;
        const float* xp[4]; long orow[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) { xp[pb] = xrow(a, p, row0, t0, i0, pb * 16 + r); orow[pb] = out_row(a, p, row0, t0, i0, pb * 16 + r); }
        f32x4 H1[2][4];
        layer1(wl, a, xp, r, g, H1);
        f32x4 H2[2][4];
        if (NH == 2) layer_xs<2, true>(wl + OFF_W2, wl + OFF_B2, r, g, H1, H2);
        const f32x4 (&HL)[2][4] = (NH == 2) ? H2 : H1;
        if (NARROW) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                float v0 = 0.f, v1 = 0.f;
#pragma unroll
                for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                    for (int s = 0; s < 4; ++s) { v0 = fmaf(w3r[0][fb][s], HL[fb][pb][s], v0); v1 = fmaf(w3r[1][fb][s], HL[fb][pb][s], v1); }
                v0 += __shfl_xor(v0, 16, 64); v0 += __shfl_xor(v0, 32, 64);           // sum over the four feature groups g
                v1 += __shfl_xor(v1, 16, 64); v1 += __shfl_xor(v1, 32, 64);
                if (g == 0 && orow[pb] >= 0) {
                    a.out[orow[pb] * a.d_out] = v0 + b3r[0];
                    if (a.d_out > 1) a.out[orow[pb] * a.d_out + 1] = v1 + b3r[1];
                }
            }
        } else {
            f32x4 O[1][4];
            layer_xs<1, false>(wl + OFF_W3, wl + OFF_B3, r, g, HL, O);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                if (orow[pb] >= 0) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) { const int o = 4 * g + s; if (o < a.d_out) a.out[orow[pb] * a.d_out + o] = O[0][pb][s]; }
                }
            }
        }
    }
}

template <int NH>
__global__ void __launch_bounds__(256) mlp_mfma_bwd_kernel(MlpMfmaArgs a) {
    __shared__ __attribute__((aligned(16))) float wl[W_ELEMS];
    __shared__ float red[4 * 64];
    const int p = blockIdx.y;
    load_weights_mfma<NH>(wl, a.theta + (long)p * a.theta_stride, a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;

    f32x4 aW1[2] = {}, aW2[2][2] = {}, aW3[2] = {};       // dW1[ob][0], dW2[ob][kb], dW3[0][kb]
    f32x4 aB1[2] = {}, aB2[2] = {}, aB3 = {};             // per-lane partial bias sums (feature 4g+s)

    for (int tl = 0; tl < a.tiles_per_wg; tl += 4) {
        const int tile = blockIdx.x * a.tiles_per_wg + tl + wave;
        const int row0 = tile * 64;
        if (tl + wave >= a.tiles_per_wg || row0 >= a.R) continue;
        const int t0 = (int)((unsigned)row0 / (unsigned)a.n), i0 = row0 - t0 * a.n;
        const float* xp[4]; long orow[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) { xp[pb] = xrow(a, p, row0, t0, i0, pb * 16 + r); orow[pb] = out_row(a, p, row0, t0, i0, pb * 16 + r); }
        // ---- forward recompute ----------------------------------------------------------------
        f32x4 H1[2][4], H2[2][4];
        layer1(wl, a, xp, r, g, H1);
        if (NH == 2) layer_xs<2, true>(wl + OFF_W2, wl + OFF_B2, r, g, H1, H2);
        f32x4 (&HL)[2][4] = (NH == 2) ? H2 : H1;            // last hidden activation
        // ---- G^T (transposed layout) and G (plain layout) straight from HBM ---------------------
        f32x4 GT[1][4], Gp[4];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int o = 4 * g + s;
                GT[0][pb][s] = (orow[pb] >= 0 && o < a.d_out) ? a.g_out[orow[pb] * a.d_out + o] : 0.0f;
            }
            aB3 += GT[0][pb];
        }
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const long orw = out_row(a, p, row0, t0, i0, pb * 16 + 4 * g + s);
                Gp[pb][s] = (orw >= 0 && r < a.d_out) ? a.g_out[orw * a.d_out + r] : 0.0f;
            }
        }
        // ---- output layer: dW3 += G^T-contraction H_last ; dHL^T = W3^T G^T .* (1 - HL^2) ---------
        f32x4 DL[2][4];
        layer_xTs<1>(wl + OFF_W3, r, g, GT, DL);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                const f32x4 Hp = tr_block(HL[kb][pb], r, g);          // H_last plain [pt][feat]
                wgrad(aW3[kb], Gp[pb], Hp);
#pragma unroll
                for (int s = 0; s < 4; ++s) DL[kb][pb][s] *= (1.0f - HL[kb][pb][s] * HL[kb][pb][s]);
            }
        }
        if (NH == 2) {
            // ---- hidden layer 2: dW2 += dH2-contraction H1 ; db2 ; dH1^T = W2^T dH2^T .* (1 - H1^2) ----
            f32x4 D1[2][4];
            layer_xTs<2>(wl + OFF_W2, r, g, DL, D1);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                f32x4 H1p[2], D2p[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) { H1p[kb] = tr_block(H1[kb][pb], r, g); D2p[kb] = tr_block(DL[kb][pb], r, g); aB2[kb] += DL[kb][pb]; }
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) wgrad(aW2[ob][kb], D2p[ob], H1p[kb]);
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int s = 0; s < 4; ++s) D1[kb][pb][s] *= (1.0f - H1[kb][pb][s] * H1[kb][pb][s]);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) DL[kb][pb] = D1[kb][pb];
        }
        // ---- first layer: dW1 += dH1-contraction X ; db1 ---------------------------------------------
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            f32x4 Xp;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float* xq = xrow(a, p, row0, t0, i0, pb * 16 + 4 * g + s);
                Xp[s] = (xq && r < a.d_in) ? xq[r] : 0.0f;
            }
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const f32x4 Dp = tr_block(DL[ob][pb], r, g);
                wgrad(aW1[ob], Dp, Xp);
                aB1[ob] += DL[ob][pb];
            }
        }
    }
    // ---- bias partials: sum over the 16 point-lanes r of each lane group ------------------------------
    auto red16 = [&](f32x4& v) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float x = v[s];
            x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64);
            v[s] = x;
        }
    };
    red16(aB1[0]); red16(aB1[1]); red16(aB2[0]); red16(aB2[1]); red16(aB3);
    // ---- cross-wave sum (fixed order) and store in the reference's flattened layout ----------------------
    float* dst = a.slab + ((long)blockIdx.x * a.P + p) * a.D_net;
    auto xsum = [&](float v) -> float {        // sum of v over the 4 waves, returned to every wave
        __syncthreads();
        red[wave * 64 + lane] = v;
        __syncthreads();
        return (red[lane] + red[64 + lane]) + (red[128 + lane] + red[192 + lane]);
    };
    const int off1 = 0;
    const int off2 = a.h0 * (a.d_in + 1);
    const int off3 = off2 + (NH == 2 ? a.h1 * (a.h0 + 1) : 0);
    const int hl = NH == 2 ? a.h1 : a.h0;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int o = ob * 16 + 4 * g + s;
            float v = xsum(aW1[ob][s]);
            if (wave == 0 && o < a.h0 && r < a.d_in) dst[off1 + a.h0 + o * a.d_in + r] = v;
            v = xsum(aB1[ob][s]);
            if (wave == 0 && o < a.h0 && r == 0) dst[off1 + o] = v;
            if (NH == 2) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    v = xsum(aW2[ob][kb][s]);
                    const int k = kb * 16 + r;
                    if (wave == 0 && o < a.h1 && k < a.h0) dst[off2 + a.h1 + o * a.h0 + k] = v;
                }
                v = xsum(aB2[ob][s]);
                if (wave == 0 && o < a.h1 && r == 0) dst[off2 + o] = v;
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int o = 4 * g + s;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float v = xsum(aW3[kb][s]);
            const int k = kb * 16 + r;
            if (wave == 0 && o < a.d_out && k < hl) dst[off3 + a.d_out + o * hl + k] = v;
        }
        float v = xsum(aB3[s]);
        if (wave == 0 && o < a.d_out && r == 0) dst[off3 + o] = v;
    }
}

// ---- backward specialised for narrow input/output (d_in <= 4, d_out <= 2: the reference's mean /
// feature networks).  fp32 MFMA runs at the fp32 VALU rate on gfx950, so padding a 2-row output layer or
// a 4-column input layer to 16x16 blocks wastes matrix-core time; here only the hidden 32x32 layer uses
// MFMA (forward, delta and weight-gradient products + the two block transposes the latter needs) while
// the first/output layers' deltas and weight gradients are per-lane VALU partial sums (reduced over the
// 16 point-lanes once per wave at the end).  Every wave writes its own partial slab (no barriers).
template <int NH>
__global__ void __launch_bounds__(256, 2) mlp_mfma_bwd_small_kernel(MlpMfmaArgs a) {
    __shared__ __attribute__((aligned(16))) float wl[W_ELEMS];
    // per-lane partial sums of the narrow layers' gradients live in lane-private LDS slots ([slot][256 lanes],
    // conflict-free, deterministic) instead of ~66 registers: that is what lets two waves share a SIMD
    __shared__ float pacc[66 * 256];
    // per-wave scratch for 16x16 block transposes through LDS (row stride 17 floats -- all that fits beside two resident
    // workgroups' accumulators); the matrix cores did these transposes before (64 of a tile's 256 MFMAs)
    __shared__ float tscr[4 * 16 * 17];
    float* my = pacc + threadIdx.x;
#define PW1(fb, s, k) my[(((fb) * 4 + (s)) * 4 + (k)) * 256]
#define PW3(o, fb, s) my[(32 + ((o) * 2 + (fb)) * 4 + (s)) * 256]
#define PB1(fb, s) my[(48 + (fb) * 4 + (s)) * 256]
#define PB2(fb, s) my[(56 + (fb) * 4 + (s)) * 256]
#define PB3(o) my[(64 + (o)) * 256]
#pragma unroll
    for (int q = 0; q < 66; ++q) my[q * 256] = 0.0f;
    const int p = blockIdx.y;
    load_weights_mfma<NH>(wl, a.theta + (long)p * a.theta_stride, a);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int d_in = a.d_in, d_out = a.d_out;

    float w3r[2][2][4];                                   // W3[o][feature fb*16+4g+s] (rows >= d_out are zero)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int s = 0; s < 4; ++s) w3r[o][fb][s] = wl[OFF_W3 + o * LW2 + fb * 16 + 4 * g + s];
    float idn[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) idn[s] = (4 * g + s == r) ? 1.0f : 0.0f;

    f32x4 aW2[2][2] = {};

    for (int tl = wave; tl < a.tiles_per_wg; tl += 4) {
        const int row0 = (blockIdx.x * a.tiles_per_wg + tl) * 64;
        if (row0 >= a.R) break;
        const int t0 = (int)((unsigned)row0 / (unsigned)a.n), i0 = row0 - t0 * a.n;
        // ---- branch-free loads: clamp the row, mask afterwards -----------------------------------
        float xr[4][4], gr[4][2];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const int q = pb * 16 + r;
            const bool valid = row0 + q < a.R;
            int t, i; locate(a, t0, i0, valid ? q : 0, t, i);
            const int bi = t * a.P + p;
            const int xb = a.x_div == 1 ? bi : (int)((unsigned)bi / (unsigned)a.x_div);
            const float* xq = a.x + ((long)xb * a.n + i) * (long)d_in;
            const float* gq = a.g_out + ((long)bi * a.n + i) * (long)d_out;
            const float m = valid ? 1.0f : 0.0f;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float v = xq[k < d_in ? k : d_in - 1]; xr[pb][k] = (k < d_in) ? v * m : 0.0f; }
#pragma unroll
            for (int o = 0; o < 2; ++o) { const float v = gq[o < d_out ? o : d_out - 1]; gr[pb][o] = (o < d_out) ? v * m : 0.0f; }
        }
        // ---- forward recompute -------------------------------------------------------------------
        f32x4 H1[2][4], H2[2][4];
        {
            float bx[4];
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) bx[pb] = g == 0 ? xr[pb][0] : (g == 1 ? xr[pb][1] : (g == 2 ? xr[pb][2] : xr[pb][3]));
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
                f32x4 bias;
#pragma unroll
                for (int s = 0; s < 4; ++s) bias[s] = wl[OFF_B1 + fb * 16 + 4 * g + s];
                const float aw = wl[OFF_W1 + (fb * 16 + r) * LW1 + g];
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    f32x4 acc = mfma4x(aw, bx[pb], bias);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[s] = act_tanh<float>(acc[s]);
                    H1[fb][pb] = acc;
                }
            }
        }
        if (NH == 2) layer_xs<2, true>(wl + OFF_W2, wl + OFF_B2, r, g, H1, H2);
        f32x4 (&HL)[2][4] = (NH == 2) ? H2 : H1;
        // ---- output layer on the VALU: dW3/db3 partials (summed over the tile's 4 point blocks in registers,
        //      then ONE read-modify-write of the lane's LDS slot), dHL^T in place ------------------------------
        {
            float g0 = 0.f, g1 = 0.f;
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) { g0 += gr[pb][0]; g1 += gr[pb][1]; }
            PB3(0) += g0; PB3(1) += g1;
#pragma unroll
            for (int fb = 0; fb < 2; ++fb)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float t0 = 0.f, t1 = 0.f;
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) {
                        const float h = HL[fb][pb][s];
                        t0 = fmaf(gr[pb][0], h, t0);
                        t1 = fmaf(gr[pb][1], h, t1);
                        const float d = fmaf(w3r[1][fb][s], gr[pb][1], w3r[0][fb][s] * gr[pb][0]);
                        HL[fb][pb][s] = d * (1.0f - h * h);
                    }
                    PW3(0, fb, s) += t0;
                    PW3(1, fb, s) += t1;
                }
        }
        if (NH == 2) {
            // ---- hidden layer: dW2 += dH2-contraction H1 (MFMA, operands transposed on the matrix core) ----
            f32x4 tB2[2] = {};
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) {
                f32x4 H1p[2], D2p[2];
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
                    // block transpose: lane (r,g) holds X[feature 4g+s][point r] in register s and needs X[feature r][point 4g+q]
                    float* tw_ = tscr + (threadIdx.x >> 6) * 272;
                    f32x4 P1, P2;
#pragma unroll
                    for (int s = 0; s < 4; ++s) tw_[(4 * g + s) * 17 + r] = H1[kb][pb][s];
                    asm volatile("" ::: "memory");
                    for (int q = 0; q < 4; ++q) P1[q] = tw_[r * 17 + 4 * g + q];
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int s = 0; s < 4; ++s) tw_[(4 * g + s) * 17 + r] = H2[kb][pb][s];
                    asm volatile("" ::: "memory");
                    for (int q = 0; q < 4; ++q) P2[q] = tw_[r * 17 + 4 * g + q];
                    asm volatile("" ::: "memory");
                    H1p[kb] = P1; D2p[kb] = P2;
                    tB2[kb] += H2[kb][pb];
                }
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) wgrad(aW2[ob][kb], D2p[ob], H1p[kb]);
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int s = 0; s < 4; ++s) PB2(kb, s) += tB2[kb][s];
            // ---- dH1^T = (W2^T dH2^T) .* (1 - H1^2), written over H1 --------------------------------------
#pragma unroll
            for (int fb = 0; fb < 2; ++fb) {
                f32x4 acc[4];
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) acc[pb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ob = 0; ob < 2; ++ob) {
                    const float* wp = wl + OFF_W2 + (ob * 16 + 4 * g) * LW2 + fb * 16 + r;
                    const float a0 = wp[0], a1 = wp[LW2], a2 = wp[2 * LW2], a3 = wp[3 * LW2];
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) {
                        acc[pb] = mfma4x(a0, H2[ob][pb][0], acc[pb]); acc[pb] = mfma4x(a1, H2[ob][pb][1], acc[pb]);
                        acc[pb] = mfma4x(a2, H2[ob][pb][2], acc[pb]); acc[pb] = mfma4x(a3, H2[ob][pb][3], acc[pb]);
                    }
                }
#pragma unroll
                for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                    for (int s = 0; s < 4; ++s) { const float h = H1[fb][pb][s]; H1[fb][pb][s] = acc[pb][s] * (1.0f - h * h); }
            }
        }
        // ---- first layer on the VALU: H1 now holds dH1^T; tile sums in registers, one LDS update per slot ------
#pragma unroll
        for (int fb = 0; fb < 2; ++fb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float tb = 0.f, tw[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int pb = 0; pb < 4; ++pb) {
                    const float dh = H1[fb][pb][s];
                    tb += dh;
#pragma unroll
                    for (int k = 0; k < 4; ++k) tw[k] = fmaf(dh, xr[pb][k], tw[k]);
                }
                PB1(fb, s) += tb;
#pragma unroll
                for (int k = 0; k < 4; ++k) PW1(fb, s, k) += tw[k];
            }
    }
    // ---- per-lane partials: sum over the 16 point lanes of each lane group ---------------------------------
    auto r16 = [](float x) {
        x += __shfl_xor(x, 1, 64); x += __shfl_xor(x, 2, 64); x += __shfl_xor(x, 4, 64); x += __shfl_xor(x, 8, 64);
        return x;
    };
    float* dst = a.slab + ((long)(blockIdx.x * 4 + wave) * a.P + p) * a.D_net;
    const int off2 = a.h0 * (d_in + 1);
    const int off3 = off2 + (NH == 2 ? a.h1 * (a.h0 + 1) : 0);
    const int hl = NH == 2 ? a.h1 : a.h0;
#pragma unroll
    for (int fb = 0; fb < 2; ++fb)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int o = fb * 16 + 4 * g + s;
            const float b1 = r16(PB1(fb, s));
            if (r == 0 && o < a.h0) dst[o] = b1;
#pragma unroll
            for (int k = 0; k < 4; ++k) { const float v = r16(PW1(fb, s, k)); if (r == 0 && o < a.h0 && k < d_in) dst[a.h0 + o * d_in + k] = v; }
            if (NH == 2) {
                const float b2 = r16(PB2(fb, s));
                if (r == 0 && o < a.h1) dst[off2 + o] = b2;
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) { const int k = kb * 16 + r; if (o < a.h1 && k < a.h0) dst[off2 + a.h1 + o * a.h0 + k] = aW2[fb][kb][s]; }
            }
#pragma unroll
            for (int oo = 0; oo < 2; ++oo) { const float v = r16(PW3(oo, fb, s)); if (r == 0 && oo < d_out && o < hl) dst[off3 + d_out + oo * hl + o] = v; }
        }
#pragma unroll
    for (int oo = 0; oo < 2; ++oo) { const float v = r16(PB3(oo)); if (lane == 0 && oo < d_out) dst[off3 + oo] = v; }
}
#undef PW1
#undef PW3
#undef PB1
#undef PB2
#undef PB3

// out[p, w] (+)= sum_c in[c, p, w]: 8 lanes per output element split the slabs, fixed order -> deterministic
template <typename T>
__global__ void __launch_bounds__(256) reduce_slab_kernel(const T* __restrict__ in, T* __restrict__ out, long out_stride, int accumulate,
                                                          int C, int P, int Wd) {
    const long tot = (long)P * Wd;
    const long idx = ((long)blockIdx.x * 256 + threadIdx.x) >> 3;
    const int part = threadIdx.x & 7;
    T s = 0;
    if (idx < tot) for (int c = part; c < C; c += 8) s += in[(long)c * tot + idx];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (idx < tot && part == 0) {
        const int p = (int)(idx / Wd), w = (int)(idx - (long)p * Wd);
        T* o = out + (long)p * out_stride + w;
        *o = accumulate ? *o + s : s;
    }
}

bool mlp_mfma_applicable(int d_in, const int32_t* hidden, int n_hidden, int d_out) {
    const bool on = g_sw.mfma;
    if (!on || n_hidden < 1 || n_hidden > 2 || d_in > 16 || d_out > 8) return false;
    for (int l = 0; l < n_hidden; ++l) if (hidden[l] > 32) return false;
    return true;
}

static void fill_args(MlpMfmaArgs& a, const void* x, int x_div, const void* theta, long theta_stride, int P,
                      int d_in, const int32_t* hidden, int n_hidden, int d_out, int B, int n) {
    a.x = (const float*)x; a.x_div = x_div; a.theta = (const float*)theta; a.theta_stride = theta_stride;
    a.P = P; a.n = n; a.R = (B / P) * n; a.d_in = d_in; a.d_out = d_out; a.nh = n_hidden;
    a.h0 = hidden[0]; a.h1 = n_hidden > 1 ? hidden[1] : 0;
    a.D_net = a.h0 * (d_in + 1) + (n_hidden > 1 ? a.h1 * (a.h0 + 1) : 0) + d_out * ((n_hidden > 1 ? a.h1 : a.h0) + 1);
}

// Number of workgroups per particle for the backward kernels.  The kernel keeps 2 workgroups (8 waves) per
// CU resident, i.e. 512 at a time on the 256 CUs; a grid slightly above a multiple of that (e.g. 1040) costs
// a whole extra round.  Pick the tiles-per-workgroup (a wave takes every 4th tile) that minimises
// rounds x (tiles per wave + fixed per-workgroup overhead ~0.4 tile).
static int mfma_bwd_chunks(int R, int P) {
    const int tiles = (R + 63) / 64;
    const int resident = 512;
    int best_tpw = 4;
    double best = 1e30;
    for (int tpw = 4; tpw <= 256; tpw += 4) {
        const long wgs = (long)((tiles + tpw - 1) / tpw) * P;
        const long rounds = (wgs + resident - 1) / resident;
        const double cost = (double)rounds * (tpw / 4 + 0.4);
        if (cost < best - 1e-9) { best = cost; best_tpw = tpw; }
        if (tpw >= tiles) break;
    }
    return (tiles + best_tpw - 1) / best_tpw;
}

// returns 1 if the MFMA path does not apply
int mlp_mfma_fwd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in, const int32_t* hidden,
                 int n_hidden, int d_out, void* out, int B, int n, hipStream_t s) {
    if (!mlp_mfma_applicable(d_in, hidden, n_hidden, d_out)) return 1;
    MlpMfmaArgs a = {};
    fill_args(a, x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, B, n);
    a.out = (float*)out;
    const int tiles = (a.R + 63) / 64;
    a.tiles_per_wg = 8;            // 2 tiles per wave; measured faster than fewer, larger workgroups
    const int wgs = (tiles + a.tiles_per_wg - 1) / a.tiles_per_wg;
    if (d_out <= 2) {
        if (n_hidden == 2) hipLaunchKernelGGL((mlp_mfma_fwd_kernel<2, true>), dim3(wgs, P), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((mlp_mfma_fwd_kernel<1, true>), dim3(wgs, P), dim3(256), 0, s, a);
    } else {
        if (n_hidden == 2) hipLaunchKernelGGL((mlp_mfma_fwd_kernel<2, false>), dim3(wgs, P), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((mlp_mfma_fwd_kernel<1, false>), dim3(wgs, P), dim3(256), 0, s, a);
    }
    return launch_status();
}

size_t mlp_mfma_bwd_workspace(int B, int P, int n, int d_in, const int32_t* hidden, int n_hidden, int d_out) {
    if (!mlp_mfma_applicable(d_in, hidden, n_hidden, d_out)) return 0;
    MlpMfmaArgs a = {};
    fill_args(a, nullptr, 1, nullptr, 0, P, d_in, hidden, n_hidden, d_out, B, n);
    return (size_t)mfma_bwd_chunks(a.R, P) * 4 * P * a.D_net * sizeof(float);     // x4: per-wave slabs of the narrow-io kernel
}

int mlp_mfma_bwd(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in, const int32_t* hidden,
                 int n_hidden, int d_out, const void* g_out, void* d_theta, long d_theta_stride, int accumulate,
                 void* workspace, int B, int n, hipStream_t s) {
    if (!mlp_mfma_applicable(d_in, hidden, n_hidden, d_out)) return 1;
    MlpMfmaArgs a = {};
    fill_args(a, x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, B, n);
    a.g_out = (const float*)g_out; a.slab = (float*)workspace;
    const int tiles = (a.R + 63) / 64;
    const int chunks = mfma_bwd_chunks(a.R, P);
    a.tiles_per_wg = (tiles + chunks - 1) / chunks;
    int slabs = chunks;
    if (d_in <= 4 && d_out <= 2) {              // narrow io: VALU first/output layers, one slab per WAVE
        slabs = chunks * 4;
        if (n_hidden == 2) hipLaunchKernelGGL(mlp_mfma_bwd_small_kernel<2>, dim3(chunks, P), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(mlp_mfma_bwd_small_kernel<1>, dim3(chunks, P), dim3(256), 0, s, a);
    } else if (n_hidden == 2) hipLaunchKernelGGL(mlp_mfma_bwd_kernel<2>, dim3(chunks, P), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(mlp_mfma_bwd_kernel<1>, dim3(chunks, P), dim3(256), 0, s, a);
    long tot = (long)P * a.D_net;
    hipLaunchKernelGGL(reduce_slab_kernel<float>, dim3((unsigned)((tot * 8 + 255) / 256)), dim3(256), 0, s,
                       (const float*)workspace, (float*)d_theta, d_theta_stride, accumulate, slabs, P, a.D_net);
    return launch_status();
}

}  // namespace pacoh
