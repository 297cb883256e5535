// gp8_body: the task GP of a context of n <= 8 points -- LML / n, its gradients, the psd_safe_cholesky jitter ladder -- by ONE wave with
// ONE MATRIX ENTRY PER LANE: lane l holds entry (i, j) = (l >> 3, l & 7) of the 8 x 8 matrix (rows / columns >= n_valid: identity).
// Round 6: the reference's own demo runs 5-point tasks (demo.py:14-26), for which gp_reg_body's 16 x 16 MFMA blocks,
// four-column elimination steps and block transposes are fixed costs of ~1 850 instructions per wave -- 8 400 + 1 300 of the persistent
// PACOH-MAP kernel's 23 000 cycles per iteration.  Here the matrix is inverted in place by eight Gauss-Jordan
// sweeps (no pivoting: the matrix is symmetric positive definite or the ladder adds jitter), each one rank-1 update of all 64 entries:
//     p = A_kk, d = 1 / p;   A_ij -= A_ik A_kj d (i, j != k);   A_kj *= d;   A_ik *= -d;   A_kk = d
// whose pivots are the squares of the Cholesky factor's diagonal (log det = sum_k log p_k; a pivot <= 0 is exactly the Cholesky's failure),
// then alpha = K^-1 r by row sums over the 8 lanes of a row (three DPP adds), and every gradient is a masked sum over the entries
// H_ij = (alpha_i alpha_j - K^-1_ij) / 2 -- no LDS scratch, ~300 vector instructions.
// Same arguments, outputs and conventions as gp_reg_body (gp_reg_body.h; reference: random_gp.py:54-89, models.py:418-446,
// gpytorch ExactMarginalLogLikelihood / psd_safe_cholesky [gpytorch-upstream]): lml[b] = LML / n_valid, gradients of g_lml[b] * lml[b],
// info[b] = ladder rung 0..3 | -1 (outputs NaN), padding rows ignored with exact-zero gradients.
#pragma once
#include "gp_reg_body.h"

namespace pacoh {
namespace gpreg {

// sum over the 8 lanes of a matrix row (lanes 8 i .. 8 i + 7), in every lane of the row
__device__ __forceinline__ float row8_sum_(float v) {
    v = dpp_add_<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]
    v = dpp_add_<0x4E, 0xF>(v);       // quad_perm [2,3,0,1]
    v = dpp_add_<0x141, 0xF>(v);      // row_half_mirror: the other quad of the 8
    return v;
}

// Gauss-Jordan sweep K of the in-place inversion (the rows behind n_valid are rows of the identity: the caller skips their sweeps).
// A[i][K]: lane K of the own 8-lane row -- ds_swizzle in bit-mask mode (lane id & 0x18 | K inside each half of the wave), no address
// register; A[K][j]: row K may sit in the other half of the wave -- ds_bpermute.
template <int K>
__device__ __forceinline__ void gp8_sweep_(float& A, float& l2, bool& ok, int i, int j) {
    const float pk = readlane_(A, 9 * K);
    ok = ok && pk > 0.0f && pk < __builtin_huge_valf();
    const float d = rcp_(pk);
    const float aik = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(A), 0x18 | (K << 5)));
    const float akj = __shfl(A, (K << 3) | j, 64);
    const float upd = fmaf(-aik * d, akj, A);
    A = (i == K) ? (j == K ? d : akj * d) : (j == K ? -aik * d : upd);
    l2 += __builtin_amdgcn_logf(pk);
}

template <int FP, class Ctx>
__device__ __forceinline__ void gp8_body(const GpMfmaArgs& a, const Ctx& cx) {
    const int lane = cx.lane();
    const int i = lane >> 3, j = lane & 7;
    const unsigned blk = cx.block();
    const long b = blk;
    const int n = a.n, f = a.f;
    const int p = (int)(blk % (unsigned)a.P);
    const long ty = blk / (unsigned)a.y_div;
    // every operand is requested before the first is used (the pointers are generic: flat loads of ~500 cycles each when they point
    // into LDS) -- rows clamped to n - 1, masked afterwards, so that no address waits for n_valid
    const int ic = i < n ? i : n - 1, jc = j < n ? j : n - 1;
    const float* zb = a.z + (long)(blk / (unsigned)a.z_div) * n * (long)f;
    float zi[FP], zj[FP], lsv[FP];
#pragma unroll
    for (int c = 0; c < FP; ++c) {
        const int cc = c < f ? c : 0;
        zi[c] = zb[(long)ic * f + cc]; zj[c] = zb[(long)jc * f + cc]; lsv[c] = a.ls[(long)p * f + cc];
    }
    const float yj = a.y[ty * n + jc];
    const float mj = a.mean_mode == PACOH_MEAN_VECTOR ? a.mean[b * n + jc] : (a.mean_mode == PACOH_MEAN_CONST ? a.mean[p] : 0.0f);
    const float os = a.os ? a.os[p] : 1.0f;
    const float noise = a.noise[p];
    int nv = a.n_valid ? a.n_valid[ty] : n;
    nv = nv < n ? nv : n; nv = nv < 0 ? 0 : nv;
    const bool vi = i < nv, vj = j < nv, vij = vi && vj;
    const int nvu = __builtin_amdgcn_readfirstlane(nv);

    float ils[FP];                                            // 1 / lengthscale
#pragma unroll
    for (int c = 0; c < FP; ++c) ils[c] = c < f ? rcp_(lsv[c]) : 0.0f;
    // scaled coordinate differences and the kernel entry of this lane's pair
    float dz[FP];
    float q = 0.0f;
#pragma unroll
    for (int c = 0; c < FP; ++c) {
        dz[c] = (vij && c < f) ? (zi[c] - zj[c]) * ils[c] : 0.0f;
        q = fmaf(dz[c], dz[c], q);
    }
    const float e = vij ? __builtin_amdgcn_exp2f(-0.7213475204444817f * q) : 0.0f;      // exp(-q / 2)
    const float rj = vj ? yj - mj : 0.0f;

    float A = 0.0f, logdet = 0.0f;                            // logdet: log det K = sum_k log p_k (padding rows: pivot 1)
    int my_info = -1;
    float jitter = 0.0f;
    for (int attempt = 0; attempt < 4; ++attempt) {
        A = vij ? fmaf(os, e, i == j ? noise + jitter : 0.0f) : (i == j ? 1.0f : 0.0f);
        float l2 = 0.0f;                                      // sum of log2 of the pivots (wave-uniform operands: eight cheap v_log)
        bool ok = true;
        if (nvu > 0) gp8_sweep_<0>(A, l2, ok, i, j);
        if (nvu > 1) gp8_sweep_<1>(A, l2, ok, i, j);
        if (nvu > 2) gp8_sweep_<2>(A, l2, ok, i, j);
        if (nvu > 3) gp8_sweep_<3>(A, l2, ok, i, j);
        if (nvu > 4) gp8_sweep_<4>(A, l2, ok, i, j);
        if (nvu > 5) gp8_sweep_<5>(A, l2, ok, i, j);
        if (nvu > 6) gp8_sweep_<6>(A, l2, ok, i, j);
        if (nvu > 7) gp8_sweep_<7>(A, l2, ok, i, j);
        if (ok) { my_info = attempt; logdet = 0.6931471805599453f * l2; break; }
        jitter = 1e-6f;
        for (int t = 0; t < attempt; ++t) jitter *= 10.0f;
    }
    const bool okf = my_info >= 0;
    const float bad = okf ? 0.0f : NAN;

    // alpha = K^-1 r: alpha_i in every lane of row i, alpha_j by the symmetric sum over the lanes of column j
    const float ai = row8_sum_(A * rj);
    float aj = A * __shfl(rj, (i << 3) | i, 64);              // K^-1_ij r_i (r_i sits in the diagonal lane of row i)
    aj += __shfl_xor(aj, 8, 64); aj += __shfl_xor(aj, 16, 64); aj += __shfl_xor(aj, 32, 64);
    const float quad = wave_sum_(i == j ? rj * aj : 0.0f);
    const float inv_nv = nv > 0 ? rcp_((float)nv) : 0.0f;
    float lml = -0.5f * (quad + logdet + (float)nv * 1.8378770664093453f) * inv_nv;
    if (!okf) lml = NAN;

    // gradients: H_ij = (alpha_i alpha_j - K^-1_ij) / 2 over the valid pairs, W_ij = H_ij e_ij
    const float gup = a.g_lml ? a.g_lml[b] : 1.0f;
    const float sc = gup * inv_nv;
    const float H = vij ? 0.5f * fmaf(ai, aj, -A) : 0.0f;
    const float W = H * e;
    const float s_noise = wave_sum_(i == j ? H : 0.0f);
    const float s_os = wave_sum_(W);
    // d K_ij / d ls_c = os e_ij dz_c^2 / ls_c (dz already divided by ls_c);  d K_ij / d z_ic = -os e_ij dz_c / ls_c, and (i, j), (j, i) both move
    float s_ls[FP], rz[FP];
#pragma unroll
    for (int c = 0; c < FP; ++c) { s_ls[c] = wave_sum_(W * dz[c] * dz[c]); rz[c] = row8_sum_(W * dz[c]); }
    const float sa = a.mean_mode == PACOH_MEAN_CONST ? wave_sum_((j == 0 && vi) ? ai : 0.0f) : 0.0f;

    // ---- stores, two exec-masked regions in all (a store of its own per output was eight branches): lane L of the first eight writes
    //      per-problem output L -- 0 lml, 1 d_noise, 2 d_os, 3 d_mean (constant mean), 4 info, 5.. d_ls[c] --, lane (i, c) entry c of
    //      point i's d_z and lane (i, f) its d_mean (f <= 4 < 8 columns) ----------------------------------------------------------------
    {
        uint32_t* ptr = nullptr; uint32_t bits = 0;
        const int c = lane - 5;
        float lsv_c = 0.0f, ils_c = 0.0f;
#pragma unroll
        for (int q = 0; q < FP; ++q) if (q == c) { lsv_c = s_ls[q]; ils_c = ils[q]; }
        if (lane == 0) { ptr = (uint32_t*)(a.lml + b); bits = __float_as_uint(lml); }
        else if (lane == 1) { ptr = (uint32_t*)(a.d_noise + b); bits = __float_as_uint(sc * s_noise + bad); }
        else if (lane == 2) { if (a.d_os) ptr = (uint32_t*)(a.d_os + b); bits = __float_as_uint(sc * s_os + bad); }
        else if (lane == 3) { if (a.mean_mode == PACOH_MEAN_CONST && a.d_mean) ptr = (uint32_t*)(a.d_mean + b); bits = __float_as_uint(sc * sa + bad); }
        else if (lane == 4) { if (a.info) ptr = (uint32_t*)(a.info + b); bits = (uint32_t)my_info; }
        else if (c < f) { ptr = (uint32_t*)(a.d_ls + b * f + c); bits = __float_as_uint(sc * os * lsv_c * ils_c + bad); }
        if (ptr) *ptr = bits;
    }
    {
        float* ptr = nullptr; float val = 0.0f;
        float rz_j = 0.0f, ils_j = 0.0f;
#pragma unroll
        for (int q = 0; q < FP; ++q) if (q == j) { rz_j = rz[q]; ils_j = ils[q]; }
        if (i < n) {
            if (j < f) { if (a.d_z) ptr = a.d_z + (b * n + i) * (long)f + j; val = vi ? -2.0f * sc * os * rz_j * ils_j + bad : 0.0f; }
            else if (j == f && a.mean_mode == PACOH_MEAN_VECTOR) { if (a.d_mean) ptr = a.d_mean + b * n + i; val = vi ? sc * ai + bad : 0.0f; }
        }
        if (ptr) *ptr = val;
    }
}

}  // namespace gpreg
}  // namespace pacoh
