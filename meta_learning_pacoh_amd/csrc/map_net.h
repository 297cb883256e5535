// Shared pieces of the two PACOH-MAP iteration kernels (round 5): the persistent one-workgroup kernel that runs K whole iterations
// per launch (map_persist.hip) and the task-fused kernel that runs forward + GP + backward of a LARGE task batch in one launch,
// one workgroup per task (map_task.hip).  Both keep a network as an LDS "parameter image", run it on the matrix cores as
// in-register chains per 16-point tile, and take the GP from gp_reg_body.h.  Here: the layer table, the LDS plan, the task
// descriptors, the chains and the weight-gradient tile.  Reference lines: models.py:206-217 (NeuralNetwork.forward),
// models.py:505-519, GPR_meta_mll.py:104-117.
#pragma once
#include "gp_reg_body.h"
#include "gp8_body.h"
#include "hyper_tail.h"
#include <stdlib.h>
#include <string.h>

namespace pacoh {
namespace {

#ifndef PACOH_MP_NT
#define PACOH_MP_NT 1024
#endif
constexpr int MP_NT = PACOH_MP_NT;   // threads per workgroup of the persistent kernel (16 waves: up to 16 tasks per iteration, one wave each)
constexpr int MP_MAXL = 5;           // up to 4 hidden layers + the output layer
constexpr int MP_MAXQ = 9;           // quads per row: in <= 32 -> S <= 36
constexpr int MP_LDS_BYTES = 158 * 1024;   // dynamic LDS the plan may use (the CU has 160 KB; the kernel's static LDS is < 1 KB)

struct MpLayer {
    int in, out, S;                  // S: row stride of the weight-and-bias matrix = row stride of the input activations
    int w_flat, b_flat;              // offsets in theta: weight [out][in] row-major, bias [out] (models.py:319-323: bias before weight)
    int w_lds;                       // offset in the parameter image
    int a_in;                        // input activations [pts][S] (layer 0: buffer 0 of the double-buffered A0)
    int a_out, s_out;                // outputs [pts][s_out]
    int d_out, s_d;                  // gradient w.r.t. the outputs (before the tanh derivative is applied: see delta_step) [pts][s_d]
    int d_in;                        // gradient w.r.t. the previous layer's pre-activation outputs [pts][32] (-1: layer 0)
    int hidden;                      // tanh on the outputs
    int units;                       // out * S / 4 weight-gradient units
    int pad0, pad1;                  // (64 bytes per row)
};

struct MpArgs {
    float* theta; float* m; float* v;
    const float* x; const float* y; const int32_t* n_valid;
    const int64_t* idx_all; const float* sc_all;
    float* loss_last; float* loss_cum; int32_t* fail_flag;
    int D, n, d, tb, n_sc, K;
    int nets, nl[2];                 // networks present (0: mean, 1: kernel features; a lone network sits in slot 0) and their layer counts
    MpLayer L[2][MP_MAXL];
    int f, mean_mode, kernel_nn;     // GP: feature count, PACOH_MEAN_*, features from the kernel network (else the raw inputs)
    int off_ls, off_os, off_noise, off_const;
    float noise_floor, one_minus_b1, b2, one_minus_b2;
    int nseg, lo[4], hi[4];
    // LDS plan (float offsets)
    int DP;                          // size of the parameter image
    int o_th, o_m, o_v, o_flat, o_a0, a0_sz, S0, o_xs, xs_sz, o_y, y_sz, o_nv, o_mn, o_zk, o_dmn, o_dzk;
    int o_lml, o_info, o_dls, o_dos, o_dnz, o_dc, o_gl, o_hp, o_gp, gpw, total;
    int pts;
    int gp8;                         // n <= 8: the GP runs gp8_body (one matrix entry per lane) instead of the 16 x 16-block body (PACOH_GP8=0: never)
    int o_tasks, slots;              // the task table: [3 phases][slots] descriptors of 16 ints (mp_plan)
};

// hp block: [0..3] ls, [4] os, [5] noise, [6] const (transformed values the GP reads); [8..14] raw, [16..22] m, [24..30] v of the
// entries e = 0..f-1 (lengthscales), 4 (outputscale), 5 (noise), 6 (constant mean)
constexpr int HP_RAW = 8, HP_M = 16, HP_V = 24, HP_SIZE = 32;

__device__ __forceinline__ bool mp_trained(const MpArgs& a, int q) {
    bool in = false;
    for (int s = 0; s < a.nseg; ++s) in |= q >= a.lo[s] && q < a.hi[s];
    return in;
}

// a wave-uniform value / pointer into scalar registers (what comes out of an LDS read sits in a vector register per lane; the GP
// body below holds 17 pointers for its whole length)
__device__ __forceinline__ int sg(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T>
__device__ __forceinline__ T* sp(T* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// Row (k, l) of the layer table.  The prologue reads it from the kernel-argument segment (mp_layer_karg: scalar loads with a
// computed offset -- indexing the by-value argument struct dynamically would make the compiler keep all of it in registers or copy
// it to scratch) and parks the table in LDS; inside the iteration loop a row comes from LDS in ONE round trip (four broadcast
// ds_read_b128 + readfirstlane): as scalar loads the row's fields took 4 dependent round trips of ~300 cycles per use, 1 200-7 000
// cycles per phase -- most of the first MFMA version's time.
__device__ __forceinline__ MpLayer mp_layer_karg(int k, int l) {
    typedef const char __attribute__((address_space(4))) * kptr_t;
    kptr_t kp = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr() + __builtin_offsetof(MpArgs, L) + (unsigned)(k * MP_MAXL + l) * (unsigned)sizeof(MpLayer);
    const int __attribute__((address_space(4)))* ki = (const int __attribute__((address_space(4)))*)kp;
    MpLayer r;
    int* ri = reinterpret_cast<int*>(&r);
#pragma unroll
    for (int q = 0; q < (int)(sizeof(MpLayer) / sizeof(int)); ++q) ri[q] = ki[q];
    return r;
}
static_assert(sizeof(MpLayer) == 64, "the LDS copy of the layer table is read as four 16-byte words per row");
__device__ __forceinline__ MpLayer mp_layer_lds(const int* __restrict__ table, int k, int l) {
    const int4* row = reinterpret_cast<const int4*>(table + (k * MP_MAXL + l) * 16);
    const int4 q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
    MpLayer r;
    r.in = __builtin_amdgcn_readfirstlane(q0.x); r.out = __builtin_amdgcn_readfirstlane(q0.y); r.S = __builtin_amdgcn_readfirstlane(q0.z);
    r.w_flat = __builtin_amdgcn_readfirstlane(q0.w); r.b_flat = __builtin_amdgcn_readfirstlane(q1.x); r.w_lds = __builtin_amdgcn_readfirstlane(q1.y);
    r.a_in = __builtin_amdgcn_readfirstlane(q1.z); r.a_out = __builtin_amdgcn_readfirstlane(q1.w); r.s_out = __builtin_amdgcn_readfirstlane(q2.x);
    r.d_out = __builtin_amdgcn_readfirstlane(q2.y); r.s_d = __builtin_amdgcn_readfirstlane(q2.z); r.d_in = __builtin_amdgcn_readfirstlane(q2.w);
    r.hidden = __builtin_amdgcn_readfirstlane(q3.x); r.units = __builtin_amdgcn_readfirstlane(q3.y); r.pad0 = 0; r.pad1 = 0;
    return r;
}

using gpreg::f32x4;
using gpreg::mfma_;

#ifdef PACOH_MP_STAMPS
__shared__ long long mp_st[2][48];
__shared__ int mp_st_on, mp_st_n[2];
__device__ __forceinline__ void mp_stamp_() {           // (diagnostic build: waves 0 and 15, last iteration of a launch)
    const int t = threadIdx.x;
    if ((t == 0 || t == MP_NT - 64) && mp_st_on) { const int w = t ? 1 : 0; if (mp_st_n[w] < 48) mp_st[w][mp_st_n[w]++] = (long long)__builtin_readcyclecounter(); }
}
#define MP_STAMP() mp_stamp_()
#else
#define MP_STAMP() do {} while (0)
#endif

// The networks run on the matrix cores, one 16x16 output tile per wave and v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains).
//  * The first version of this kernel gave every thread one (point, unit) dot product with both operands read from LDS per FMA:
//    16 waves x 27 ds_read_b128 per layer = 3 100 cycles per layer and phase, LDS-bandwidth bound.  A tile's operands are read
//    ONCE per wave.
//  * A wave issues one instruction per ~4.5 cycles whatever its kind, so what a phase costs is the INSTRUCTIONS its slowest wave
//    runs: the second version walked the (network, layer, tile) loops in every wave to find its tile -- ~700 instructions = 3 000
//    cycles per phase for a 10-instruction tile.  Now wave 0 walks those loops ONCE per launch (mp_plan) and leaves one 64-byte
//    descriptor per (phase, task) in LDS with every offset resolved; in the loop a wave reads its descriptor and runs the tile.
// Lane (r, g) = (lane & 15, lane >> 4) supplies A[i = r][k] and B[k][j = r] of an MFMA step for ITS k of the step's four: the k
// index of a product is free to permute, so where an operand is a row read along k the four lanes g take four consecutive QUADS of
// the row (one ds_read_b128 each, four MFMA steps per read), and a remainder quad is one more step with element g per lane.
// Rows / columns outside a matrix (tile padding) read whatever follows in LDS: they only reach output rows / columns that are not
// stored -- except in the weight-gradient tile, where padding POINTS would enter every sum and are masked to zero.
enum { MP_NONE = 0, MP_FWD = 1, MP_DELTA = 2, MP_WGRAD = 3 };
struct MpTask {                      // 16 ints; offsets in floats from the start of dynamic LDS unless stated
    int kind, S;
    int w;                           // FWD / DELTA chain: the network; WGRAD: parameter-image offset (from th / m / v / flat) of entry (16 J, 16 I); BIAS: of column `in`
    int src, s_src;                  // FWD / DELTA chain: the point tile; WGRAD / BIAS: d_out (column 16 J) and its row stride
    int aux;                         // WGRAD: input activations, column 16 I
    int dst, s_dst;
    int lim_a, lim_b;                // WGRAD / BIAS: valid rows j, columns i of the tile
    int n1, n2;                      // FWD / DELTA chain: layers of the network; WGRAD: MFMA steps = ceil(pts / 4), valid d_out columns
    int flags;                       // bit 1: aux (WGRAD) is relative to the current A0 buffer; bit 2: 32-wide chains; bit 3: the tile sums the bias too
    int kmax;
    int pad0, pad1;                  // pad0: mp_plan's sort key
};
static_assert(sizeof(MpTask) == 64, "a task descriptor is read as four 16-byte words");

__device__ __forceinline__ int sgi(int v) { return __builtin_amdgcn_readfirstlane(v); }

#define MP_WSYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)   // in-order LDS within one wave

// Forward pass of ONE network for the 16 points of one tile, by one wave, layer after layer without leaving the wave: the
// accumulator layout of a layer's output tile -- lane (r, g) holds units 16 U + 4 g + s of point r -- IS the B operand of the next
// layer under the quad permutation above (quad 4 c + g of the row = units 16 c + 4 g ..+3), so the activations go from MFMA to MFMA
// in registers.  Only what the registers do not hold in full groups of 16 units comes from LDS: the first layer's inputs, the bias
// quad (the constant-1 column) and the odd quads of widths that are not a multiple of 16.  Every activation is also stored to LDS
// for the backward pass.  out[p][u] = act(sum_k Wb[u][k] a[p][k]), bias folded: a[p][in] = 1.
__device__ __forceinline__ void mp_fwd_chain(const int* __restrict__ ltab, int k, int nl, int Pt, int pts, const float* __restrict__ th,
                                             float* __restrict__ lds, int a0_off, int r, int g) {
    f32x4 hprev[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int p = 16 * Pt + r;
    for (int l = 0; l < nl; ++l) {
        const MpLayer L = mp_layer_lds(ltab, k, l);
        const int S = L.S, nq = S >> 2;
        const int ngr = l == 0 ? 0 : (L.in >> 4);            // quad groups (16 units) the registers hold in full
        const int nrem = nq - 4 * ngr;                       // quads read from LDS, the bias quad among them
        const float* arow = lds + (l == 0 ? a0_off : L.a_in) + p * S;
        const int nU = (L.out + 15) >> 4;
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int U = 0; U < 2; ++U) {
            if (U < nU) {
                const float* wrow = th + L.w_lds + (16 * U + r) * S;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (c < ngr) {
                        const float4 wq = *reinterpret_cast<const float4*>(wrow + 16 * c + 4 * g);
                        acc[U] = mfma_(wq.x, hprev[c][0], acc[U]); acc[U] = mfma_(wq.y, hprev[c][1], acc[U]);
                        acc[U] = mfma_(wq.z, hprev[c][2], acc[U]); acc[U] = mfma_(wq.w, hprev[c][3], acc[U]);
                    }
                }
                for (int q = 0; q < nrem; ++q) { const int kk = 16 * ngr + 4 * q + g; acc[U] = mfma_(wrow[kk], arow[kk], acc[U]); }
            }
        }
#pragma unroll
        for (int U = 0; U < 2; ++U) {
            if (U < nU) {
                float* dst = lds + L.a_out + p * L.s_out + 16 * U + 4 * g;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float v = L.hidden ? act_tanh<float>(acc[U][s]) : acc[U][s];
                    hprev[U][s] = v;
                    if (p < pts && 16 * U + 4 * g + s < L.out) dst[s] = v;
                }
            }
        }
        MP_WSYNC();
    }
}

// The delta chain of ONE network for one point tile, by one wave, from the GP's gradients down to the first hidden layer:
// d_in[p][i] = (sum_k W[k][i] d_out[p][k]) (1 - h[p][i]^2); the result tile (lane (r, g): inputs 16 I + 4 g + s of point r) is again
// the next step's B operand.  Every delta is stored to LDS for the weight steps.
__device__ __forceinline__ void mp_delta_chain(const int* __restrict__ ltab, int k, int nl, int Pt, int pts, const float* __restrict__ th,
                                               float* __restrict__ lds, int r, int g) {
    f32x4 dprev[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const int p = 16 * Pt + r;
    for (int l = nl - 1; l >= 1; --l) {
        const MpLayer L = mp_layer_lds(ltab, k, l);
        const int S = L.S, nk = (L.out + 3) >> 2;
        const int ngr = l == nl - 1 ? 0 : (L.out >> 4);      // (the GP's gradients come from LDS)
        const int nrem = nk - 4 * ngr;
        const float* drow = lds + L.d_out + p * L.s_d;
        const int nI = (L.in + 15) >> 4;
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            if (I < nI) {
                const float* wcol = th + L.w_lds + 16 * I + r;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (c < ngr) {
                        const int k0 = 16 * c + 4 * g;
                        acc[I] = mfma_(wcol[k0 * S], dprev[c][0], acc[I]); acc[I] = mfma_(wcol[(k0 + 1) * S], dprev[c][1], acc[I]);
                        acc[I] = mfma_(wcol[(k0 + 2) * S], dprev[c][2], acc[I]); acc[I] = mfma_(wcol[(k0 + 3) * S], dprev[c][3], acc[I]);
                    }
                }
                for (int q = 0; q < nrem; ++q) {
                    const int kk = 16 * ngr + 4 * q + g;
                    const bool ok = kk < L.out;
                    acc[I] = mfma_(ok ? wcol[kk * S] : 0.0f, ok ? drow[kk] : 0.0f, acc[I]);
                }
            }
        }
#pragma unroll
        for (int I = 0; I < 2; ++I) {
            if (I < nI) {
                const int i0 = 16 * I + 4 * g;
                const float4 h = *reinterpret_cast<const float4*>(lds + L.a_in + p * S + i0);
                float* dst = lds + L.d_in + p * 32 + i0;
                const float hv[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const float v = acc[I][s] * (1.0f - hv[s] * hv[s]);
                    dprev[I][s] = v;
                    if (p < pts && i0 + s < L.in) dst[s] = v;
                }
            }
        }
        MP_WSYNC();
    }
}

// The same two chains for the common shape -- every hidden layer exactly 32 wide (NeuralNetwork's default (32, 32)) -- with the
// group / remainder / tile counts known at compile time: no predication, no inner loops, whole 16-byte stores.  A wave issues one
// instruction per ~4.5 cycles, so the generic chains' bookkeeping (~300 instructions per layer) was most of their time.
__device__ __forceinline__ void mp_fwd_chain32(const int* __restrict__ ltab, int k, int nl, int Pt, int pts, const float* __restrict__ th,
                                               float* __restrict__ lds, int a0_off, int r, int g) {
    const int p = 16 * Pt + r;
    const bool pok = p < pts;
    f32x4 h0, h1;
    {   // first layer: inputs (and the constant 1) from LDS, S = 4 or 8
        const MpLayer L = mp_layer_lds(ltab, k, 0);
        const int S = L.S;
        const float* arow = lds + a0_off + p * S;
        const float* w0 = th + L.w_lds + r * S;
        const float* w1 = w0 + 16 * S;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        const float x0 = arow[g];
        a0 = mfma_(w0[g], x0, a0); a1 = mfma_(w1[g], x0, a1);
        if (S == 8) { const float x1 = arow[4 + g]; a0 = mfma_(w0[4 + g], x1, a0); a1 = mfma_(w1[4 + g], x1, a1); }
#pragma unroll
        for (int s = 0; s < 4; ++s) { h0[s] = act_tanh<float>(a0[s]); h1[s] = act_tanh<float>(a1[s]); }
        if (pok) {
            float* dst = lds + L.a_out + p * 36 + 4 * g;
            *reinterpret_cast<f32x4*>(dst) = h0; *reinterpret_cast<f32x4*>(dst + 16) = h1;
        }
    }
    for (int l = 1; l < nl - 1; ++l) {   // hidden -> hidden: k = 0..31 from registers, the bias quad's element g from LDS (1, 0, 0, 0)
        const MpLayer L = mp_layer_lds(ltab, k, l);
        const float* w0 = th + L.w_lds + r * 36 + 4 * g;
        const float* w1 = w0 + 16 * 36;
        const float4 wa = *reinterpret_cast<const float4*>(w0), wb = *reinterpret_cast<const float4*>(w0 + 16);
        const float4 wc = *reinterpret_cast<const float4*>(w1), wd = *reinterpret_cast<const float4*>(w1 + 16);
        const float ba = w0[32 - 3 * g], bb = w1[32 - 3 * g];            // element 32 + g of the row
        const float one = g == 0 ? 1.0f : 0.0f;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        a0 = mfma_(wa.x, h0[0], a0); a1 = mfma_(wc.x, h0[0], a1); a0 = mfma_(wa.y, h0[1], a0); a1 = mfma_(wc.y, h0[1], a1);
        a0 = mfma_(wa.z, h0[2], a0); a1 = mfma_(wc.z, h0[2], a1); a0 = mfma_(wa.w, h0[3], a0); a1 = mfma_(wc.w, h0[3], a1);
        a0 = mfma_(wb.x, h1[0], a0); a1 = mfma_(wd.x, h1[0], a1); a0 = mfma_(wb.y, h1[1], a0); a1 = mfma_(wd.y, h1[1], a1);
        a0 = mfma_(wb.z, h1[2], a0); a1 = mfma_(wd.z, h1[2], a1); a0 = mfma_(wb.w, h1[3], a0); a1 = mfma_(wd.w, h1[3], a1);
        a0 = mfma_(ba, one, a0); a1 = mfma_(bb, one, a1);
#pragma unroll
        for (int s = 0; s < 4; ++s) { h0[s] = act_tanh<float>(a0[s]); h1[s] = act_tanh<float>(a1[s]); }
        if (pok) {
            float* dst = lds + L.a_out + p * 36 + 4 * g;
            *reinterpret_cast<f32x4*>(dst) = h0; *reinterpret_cast<f32x4*>(dst + 16) = h1;
        }
    }
    {   // output layer: out <= 4 rows of ONE tile; the lanes g = 0 hold the outputs of point r
        const MpLayer L = mp_layer_lds(ltab, k, nl - 1);
        const float* w0 = th + L.w_lds + r * 36 + 4 * g;
        const float4 wa = *reinterpret_cast<const float4*>(w0), wb = *reinterpret_cast<const float4*>(w0 + 16);
        const float ba = w0[32 - 3 * g];
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f};
        a0 = mfma_(wa.x, h0[0], a0); a0 = mfma_(wa.y, h0[1], a0); a0 = mfma_(wa.z, h0[2], a0); a0 = mfma_(wa.w, h0[3], a0);
        a0 = mfma_(wb.x, h1[0], a0); a0 = mfma_(wb.y, h1[1], a0); a0 = mfma_(wb.z, h1[2], a0); a0 = mfma_(wb.w, h1[3], a0);
        a0 = mfma_(ba, g == 0 ? 1.0f : 0.0f, a0);
        if (pok && g == 0) {
            float* dst = lds + L.a_out + p * L.s_out;
#pragma unroll
            for (int s = 0; s < 4; ++s) if (s < L.out) dst[s] = a0[s];
        }
    }
}

__device__ __forceinline__ void mp_delta_chain32(const int* __restrict__ ltab, int k, int nl, int Pt, int pts, const float* __restrict__ th,
                                                 float* __restrict__ lds, int r, int g) {
    const int p = 16 * Pt + r;
    const bool pok = p < pts;
    f32x4 d0, d1;
    {   // output layer: d_out = the GP's gradients [p][out <= 4] from LDS; W rows k < out
        const MpLayer L = mp_layer_lds(ltab, k, nl - 1);
        const bool ok = g < L.out;
        const float dv = ok ? lds[L.d_out + p * L.s_d + g] : 0.0f;
        const float* wcol = th + L.w_lds + g * 36 + r;
        const float wa = ok ? wcol[0] : 0.0f, wb = ok ? wcol[16] : 0.0f;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        a0 = mfma_(wa, dv, a0); a1 = mfma_(wb, dv, a1);
        const float* hrow = lds + L.a_in + p * 36 + 4 * g;
        const f32x4 ha = *reinterpret_cast<const f32x4*>(hrow), hb = *reinterpret_cast<const f32x4*>(hrow + 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) { d0[s] = a0[s] * (1.0f - ha[s] * ha[s]); d1[s] = a1[s] * (1.0f - hb[s] * hb[s]); }
        if (pok) {
            float* dst = lds + L.d_in + p * 32 + 4 * g;
            *reinterpret_cast<f32x4*>(dst) = d0; *reinterpret_cast<f32x4*>(dst + 16) = d1;
        }
    }
    for (int l = nl - 2; l >= 1; --l) {   // hidden layer l: 32 x 32, d_out from registers
        const MpLayer L = mp_layer_lds(ltab, k, l);
        const float* wcol = th + L.w_lds + 4 * g * 36 + r;              // rows k = 4 g + e (+ 16 c), columns r / 16 + r
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            a0 = mfma_(wcol[e * 36], d0[e], a0); a1 = mfma_(wcol[e * 36 + 16], d0[e], a1);
            a0 = mfma_(wcol[(16 + e) * 36], d1[e], a0); a1 = mfma_(wcol[(16 + e) * 36 + 16], d1[e], a1);
        }
        const float* hrow = lds + L.a_in + p * 36 + 4 * g;
        const f32x4 ha = *reinterpret_cast<const f32x4*>(hrow), hb = *reinterpret_cast<const f32x4*>(hrow + 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) { d0[s] = a0[s] * (1.0f - ha[s] * ha[s]); d1[s] = a1[s] * (1.0f - hb[s] * hb[s]); }
        if (pok) {
            float* dst = lds + L.d_in + p * 32 + 4 * g;
            *reinterpret_cast<f32x4*>(dst) = d0; *reinterpret_cast<f32x4*>(dst + 16) = d1;
        }
    }
}

struct MpAdam { float dm, one_minus_b1, b2, one_minus_b2, ss, rbc2, eps; };     // rbc2 = 1 / sqrt(1 - beta2^t)

// One AdamW update with the arithmetic of torch.optim.AdamW (decoupled decay, bias corrections as step scalars) on the hardware's
// reciprocal and square root (1 ulp each) instead of the IEEE division / square-root sequences of adam_update (hyper_tail.h): 12
// instructions instead of ~47 per entry, 4 entries per lane and tile.  The relative difference, ~1e-7 of a step of size lr, is far
// below what the summation order of the gradient itself moves.
__device__ __forceinline__ void mp_adam_entry(float* __restrict__ th, float* __restrict__ mm, float* __restrict__ vv, const int* __restrict__ flat,
                                              int li, float gv, const MpAdam& o) {
    if (flat[li] >= 0) {
        const float pw = th[li], pm = mm[li], pv = vv[li];
        const float mq = fmaf(gv - pm, o.one_minus_b1, pm);
        const float vq = fmaf(pv, o.b2, o.one_minus_b2 * gv * gv);
        const float denom = fmaf(__builtin_amdgcn_sqrtf(vq), o.rbc2, o.eps);
        th[li] = fmaf(-o.ss * mq, __builtin_amdgcn_rcpf(denom), pw * o.dm);
        mm[li] = mq; vv[li] = vq;
    }
}

// weight step: dWb[j][i] = sum_p d_out[p][j] a[p][i] for 16 rows x 16 columns (the bias column `in` included where the tile covers
// it), then AdamW on the tile's trained entries in place.  The product is formed TRANSPOSED (A = activations, B = deltas): lane
// (r, g) then holds columns 4 g .. 4 g + 3 of row r -- four consecutive entries of the parameter image, so that parameters, both
// moments and the trained-flags move as one 16-byte LDS access each and the update runs without a branch per entry (a tile's
// AdamW was 28 dword accesses and four exec-mask branches per lane in the row-per-register orientation).
// bias (out): flags bit 3 -- this tile also sums the layer's bias gradient of its 16 rows, sum_p d_out[p][j], from the B operands it
// loads anyway: every lane (r, .) ends up holding the sum of row j = r.  (A task of its own per layer -- a wave reading the column
// again, one entry per lane -- made 20 tasks of 16 at demo.py's shape: a second round.)
__device__ __forceinline__ f32x4 mp_wgrad_acc(const int4 d0, const int4 d1, const int4 d2, const int4 d3, const float* __restrict__ lds,
                                              int a0_off, int pts, int r, int g, float& bias) {
    const int S = d0.y, steps = (pts + 3) >> 2, s_d = d1.x;      // (pts: the caller's actual point count, wave-uniform)
    const bool jok = r < d2.w;
    const bool with_bias = sgi(d3.x) & 8;
    const float* dp = lds + d0.w + (jok ? r : 0) + g * s_d;                              // d_out[4 ks + g][16 J + r]
    const float* ap = lds + d1.y + ((d3.x & 2) ? a0_off : 0) + r + g * S;               // a[4 ks + g][16 I + r]
    const int dstep = 4 * s_d, astep = 4 * S;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    MP_STAMP();
    // every step but the last has four valid points; the last is clamped and masked (padding points would enter the sums)
    // (the bias sums: this phase is bound by the instruction issue of ONE compute unit -- 16 waves in step -- and its matrix pipes; a
    //  second product per step with the constant 1 as A operand cost more of both than a vector add per step and two cross-row adds)
    float bsum = 0.0f;
    if (with_bias) {
#pragma unroll 4
        for (int ks = 0; ks + 1 < steps; ++ks) {
            const float dv = jok ? *dp : 0.0f;
            acc = mfma_(*ap, dv, acc); bsum += dv;
            ap += astep; dp += dstep;
        }
    } else {
#pragma unroll 4
        for (int ks = 0; ks + 1 < steps; ++ks) { acc = mfma_(*ap, jok ? *dp : 0.0f, acc); ap += astep; dp += dstep; }
    }
    {
        const int back = 4 * (steps - 1) + g - (pts - 1);                                  // > 0: this lane's point does not exist
        const bool ok = back <= 0;
        const float av = ap[ok ? 0 : -back * S], dv = dp[ok ? 0 : -back * s_d];
        const float dm = ok && jok ? dv : 0.0f;
        acc = mfma_(ok ? av : 0.0f, dm, acc);
        bsum += dm;
    }
    MP_STAMP();
    bsum += __shfl_xor(bsum, 16, 64); bsum += __shfl_xor(bsum, 32, 64);                    // over the four lane rows: every lane (r, .) holds row r's sum
    bias = bsum;
    return acc;
}

__device__ __forceinline__ void mp_wgrad_tile(const int4 d0, const int4 d1, const int4 d2, const int4 d3, float* __restrict__ th,
                                              float* __restrict__ mm, float* __restrict__ vv, const int* __restrict__ flat,
                                              const float* __restrict__ lds, int a0_off, int pts, int r, int g, const MpAdam& o) {
    const int S = d0.y;
    float bias;
    const f32x4 acc = mp_wgrad_acc(d0, d1, d2, d3, lds, a0_off, pts, r, g, bias);
    if ((d3.x & 8) && g == 0 && r < d2.x) mp_adam_entry(th, mm, vv, flat, d3.y + r * S, bias, o);      // (kmax: the bias entry of row 0 of the tile)
    if (r < d2.x && 4 * g < d2.y) {
        const int li = d0.z + r * S + 4 * g;
        const int4 fq = *reinterpret_cast<const int4*>(flat + li);
        f32x4 pw = *reinterpret_cast<const f32x4*>(th + li), pm = *reinterpret_cast<const f32x4*>(mm + li), pv = *reinterpret_cast<const f32x4*>(vv + li);
        const int fqv[4] = {fq.x, fq.y, fq.z, fq.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float gv = acc[s];
            const float mq = fmaf(gv - pm[s], o.one_minus_b1, pm[s]);
            const float vq = fmaf(pv[s], o.b2, o.one_minus_b2 * gv * gv);
            const float denom = fmaf(__builtin_amdgcn_sqrtf(vq), o.rbc2, o.eps);
            const float pn = fmaf(-o.ss * mq, __builtin_amdgcn_rcpf(denom), pw[s] * o.dm);
            const bool tr = fqv[s] >= 0;
            pw[s] = tr ? pn : pw[s]; pm[s] = tr ? mq : pm[s]; pv[s] = tr ? vq : pv[s];
        }
        *reinterpret_cast<f32x4*>(th + li) = pw; *reinterpret_cast<f32x4*>(mm + li) = pm; *reinterpret_cast<f32x4*>(vv + li) = pv;
    }
}

// The launch's task table, written by ONE thread.  Phase 0: forward chains (network, point tile); phase 1: delta chains; phase 2: the
// weight steps of every layer -- the tiles in descending order of their AdamW entries, so that a wave with two tasks gets a small
// second one.  ntask[phase] = tasks of the phase; task q of it belongs to wave q % 16.
__device__ void mp_plan(const MpArgs& a, const int* __restrict__ ltab, MpTask* __restrict__ tasks, int* __restrict__ ntask) {
    const int pts = a.pts, nPt = (pts + 15) >> 4, slots = a.slots;
    const MpLayer* lt = reinterpret_cast<const MpLayer*>(ltab);
    ntask[0] = ntask[1] = ntask[2] = 0;
    auto put = [&](int ph, const MpTask& tk) { tasks[ph * slots + ntask[ph]] = tk; ntask[ph] += 1; };
    for (int k = 0; k < a.nets; ++k)
        for (int Pt = 0; Pt < nPt; ++Pt) {
            MpTask tk = {};
            tk.kind = MP_FWD; tk.w = k; tk.src = Pt; tk.n1 = a.nl[k];
            bool w32 = lt[k * MP_MAXL].S <= 8 && lt[k * MP_MAXL + a.nl[k] - 1].out <= 4;
            for (int l = 0; l + 1 < a.nl[k]; ++l) w32 = w32 && lt[k * MP_MAXL + l].out == 32;
            tk.flags = w32 ? 4 : 0;            // every hidden layer 32 wide: the compile-time chains
            put(0, tk);
            tk.kind = MP_DELTA;
            put(1, tk);
        }
    for (int k = 0; k < a.nets; ++k)
        for (int l = 0; l < a.nl[k]; ++l) {
            const MpLayer& L = lt[k * MP_MAXL + l];
            for (int J = 0; J < (L.out + 15) >> 4; ++J)
                for (int I = 0; I < (L.in + 15) >> 4; ++I) {
                    MpTask tk = {};
                    tk.kind = MP_WGRAD; tk.S = L.S; tk.w = L.w_lds + 16 * J * L.S + 16 * I;
                    tk.src = L.d_out + 16 * J; tk.s_src = L.s_d; tk.aux = (l == 0 ? 0 : L.a_in) + 16 * I; tk.flags = l == 0 ? 2 : 0;
                    tk.lim_a = L.out - 16 * J; tk.lim_b = L.S - 16 * I; tk.n1 = (pts + 3) >> 2; tk.n2 = L.s_d - 16 * J;
                    if (I == 0 && (L.in & 15) == 0) { tk.flags |= 8; tk.kmax = L.w_lds + 16 * J * L.S + L.in; }     // (no tile covers the bias column)
                    const int ja = tk.lim_a < 16 ? tk.lim_a : 16, ib = tk.lim_b < 16 ? tk.lim_b : 16;
                    tk.pad0 = ja * ib;                       // (sort key)
                    put(2, tk);
                }
        }
    MpTask* w = tasks + 2 * slots;
    for (int i = 0; i < ntask[2]; ++i) {
        int best = i;
        for (int j = i + 1; j < ntask[2]; ++j) if (w[j].pad0 > w[best].pad0) best = j;
        if (best != i) { const MpTask tmp = w[i]; w[i] = w[best]; w[best] = tmp; }
    }
}

// softplus on the hardware's exp2 / log2 (relative error ~2e-7; log(1 + e) = e below the point where 1 + e rounds to 1)
__device__ __forceinline__ float mp_softplus(float x) {
    const float e = __builtin_amdgcn_exp2f(1.4426950408889634f * x);
    return x > 20.0f ? x : (x < -15.0f ? e : 0.6931471805599453f * __builtin_amdgcn_logf(1.0f + e));
}

int round4(int v) { return (v + 3) & ~3; }

// network `k` of the plan: hidden[0..nh) -> d_out outputs, parameters at theta[off ..) in the reference's order (per layer: bias, weight)
void plan_net(MpArgs& a, int k, int off, int d_in, const int32_t* hidden, int nh, int d_out, int out_off, int out_stride, int dout_off,
              int& lds_top, int& dp_top) {
    a.nl[k] = nh + 1;
    int prev = d_in, q = off;
    for (int l = 0; l <= nh; ++l) {
        MpLayer& L = a.L[k][l];
        L.in = prev; L.out = l < nh ? hidden[l] : d_out; L.S = round4(prev + 1);
        L.b_flat = q; L.w_flat = q + L.out; q += L.out + L.out * prev;
        L.w_lds = dp_top; dp_top += L.out * L.S;
        L.hidden = l < nh;
        L.units = L.out * (L.S / 4);
        prev = L.out;
    }
    // activations: layer l's outputs are layer l+1's inputs, row stride S of layer l+1
    for (int l = 0; l <= nh; ++l) {
        MpLayer& L = a.L[k][l];
        if (l == 0) { L.a_in = a.o_a0; L.d_in = -1; }
        if (l < nh) {
            MpLayer& N = a.L[k][l + 1];
            // (sized for whole 16-point tiles: the chains and weight tiles read the last tile's padding rows, which must be the plan's own
            //  zeroed memory, never whatever follows it -- ADVICE r5)
            const int pts16 = (a.pts + 15) & ~15;
            L.a_out = lds_top; L.s_out = N.S; N.a_in = lds_top; lds_top += pts16 * N.S;
            L.d_out = lds_top; L.s_d = 32; N.d_in = lds_top; lds_top += pts16 * 32;
        } else {
            L.a_out = out_off; L.s_out = out_stride; L.d_out = dout_off; L.s_d = out_stride;
        }
    }
}

}  // namespace


// -> 0 and the filled plan, or PACOH_ELIMIT when the shape is outside what the persistent kernel takes (the caller then runs the
// four-launch iteration)
static int map_persist_plan(MpArgs& a, int n, int d, int tb, int K, int mean_mode, int off_mean, const int32_t* mean_hidden, int n_mean_hidden,
                            int kernel_nn, int off_kernel, const int32_t* kernel_hidden, int n_kernel_hidden, int f, int* nb_out, int* fp_out,
                            bool with_moments = true) {
    // with_moments: the persistent kernel keeps both Adam moments and the image-to-theta index beside the parameter image (four
    // copies of its size); the task-fused kernels only read the image -- with four copies two 4 x 32 networks (the reference
    // launchers' default) did not fit the plan
    memset(&a, 0, sizeof(a));
    if (n < 1 || n > 32 || d < 1 || d > 4 || tb < 1 || tb > MP_NT / 64 || K < 1 || K > 1024 || f < 1 || f > 4) return PACOH_ELIMIT;
    if (tb * n * (d + 1) > MP_NT) return PACOH_ELIMIT;
    if (n_mean_hidden > MP_MAXL - 1 || n_kernel_hidden > MP_MAXL - 1) return PACOH_ELIMIT;
    if (mean_mode == PACOH_MEAN_VECTOR && n_mean_hidden < 1) return PACOH_ELIMIT;
    if (kernel_nn && n_kernel_hidden < 1) return PACOH_ELIMIT;
    for (int l = 0; l < n_mean_hidden; ++l) if (mean_mode == PACOH_MEAN_VECTOR && (mean_hidden[l] < 1 || mean_hidden[l] > 32)) return PACOH_ELIMIT;
    for (int l = 0; l < n_kernel_hidden; ++l) if (kernel_nn && (kernel_hidden[l] < 1 || kernel_hidden[l] > 32)) return PACOH_ELIMIT;
    if (!kernel_nn && f != d) return PACOH_EINVAL;
    int NB = (n + 15) / 16; const int FP = f <= 2 ? 2 : 4;
    *nb_out = NB; *fp_out = FP;
    a.n = n; a.d = d; a.tb = tb; a.K = K; a.f = f; a.mean_mode = mean_mode; a.kernel_nn = kernel_nn;
    a.pts = tb * n;
    a.gp8 = (n <= 8 && g_sw.gp8) ? 1 : 0;
    int top = 0;
    auto take = [&](int count) { const int o = top; top += round4(count); return o; };
    a.S0 = round4(d + 1);
    a.o_hp = take(HP_SIZE);
    const int pts16 = (a.pts + 15) & ~15;              // whole 16-point tiles: padding rows are read (never stored) by the MFMA tiles
    a.a0_sz = round4(pts16 * a.S0); a.o_a0 = take(2 * a.a0_sz);
    a.xs_sz = round4(a.pts * d); a.o_xs = take(2 * a.xs_sz);
    a.y_sz = round4(a.pts); a.o_y = take(2 * a.y_sz);
    a.o_nv = take(32);
    a.o_mn = take(pts16); a.o_zk = take(pts16 * f); a.o_dmn = take(pts16); a.o_dzk = take(pts16 * f);
    a.o_lml = take(16); a.o_info = take(16); a.o_dls = take(16 * 4); a.o_dos = take(16); a.o_dnz = take(16); a.o_dc = take(16); a.o_gl = take(16);
    int dp = 0;
    a.nets = 0;
    if (mean_mode == PACOH_MEAN_VECTOR) { plan_net(a, a.nets, off_mean, d, mean_hidden, n_mean_hidden, 1, a.o_mn, 1, a.o_dmn, top, dp); a.nets++; }
    if (kernel_nn) { plan_net(a, a.nets, off_kernel, d, kernel_hidden, n_kernel_hidden, f, a.o_zk, f, a.o_dzk, top, dp); a.nets++; }
    a.DP = round4(dp > 0 ? dp : 4);
    a.o_th = take(a.DP);
    if (with_moments) { a.o_m = take(a.DP); a.o_v = take(a.DP); a.o_flat = take(a.DP); }
    else a.o_m = a.o_v = a.o_flat = a.o_th;
    for (int k = 0; k < a.nets; ++k) for (int l = 0; l < a.nl[k]; ++l) a.L[k][l].w_lds += 0;      // (image offsets are relative to o_th / o_m / o_v)
    const int NP = 16 * NB, NU = NB * (NB + 1) / 2;
    a.gpw = round4(2 * NP * FP + 2 * NP + gpreg::GPR_SCR + (NB > 1 ? (NU - NB) * 256 : 4));
    a.o_gp = take(a.gpw * tb);
    {   // tasks per phase as mp_plan emits them -> slots (the table's row length)
        const int nPt = (a.pts + 15) / 16;
        int chains = a.nets * nPt, wt = 0;
        for (int k = 0; k < a.nets; ++k)
            for (int l = 0; l < a.nl[k]; ++l) {
                const MpLayer& L = a.L[k][l];
                wt += ((L.out + 15) / 16) * ((L.in + 15) / 16) + ((L.in & 15) == 0 ? 1 : 0);
            }
        a.slots = chains > wt ? chains : (wt > 0 ? wt : 1);
        a.o_tasks = take(3 * a.slots * 16);
    }
    a.total = top;
    if ((size_t)top * sizeof(float) > (size_t)MP_LDS_BYTES) return PACOH_ELIMIT;
    return PACOH_OK;
}

}  // namespace pacoh
