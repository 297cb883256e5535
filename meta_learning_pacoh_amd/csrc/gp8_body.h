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

template <int FP, class Ctx>
__device__ __forceinline__ void gp8_body(const GpMfmaArgs& a, const Ctx& cx) {
    const int lane = cx.lane();
    const int i = lane >> 3, j = lane & 7;
    const unsigned blk = cx.block();
    const long b = blk;
    const int n = a.n, f = a.f;
    const int p = (int)(blk % (unsigned)a.P);
    const long ty = blk / (unsigned)a.y_div;
    int nv = a.n_valid ? a.n_valid[ty] : n;
    nv = nv < n ? nv : n; nv = nv < 0 ? 0 : nv;
    const bool vi = i < nv, vj = j < nv, vij = vi && vj;

    float ils[FP];                                            // 1 / lengthscale
#pragma unroll
    for (int c = 0; c < FP; ++c) ils[c] = c < f ? rcp_(a.ls[(long)p * f + c]) : 0.0f;
    const float os = a.os ? a.os[p] : 1.0f;
    const float noise = a.noise[p];

    // scaled coordinate differences and the kernel entry of this lane's pair
    const float* zb = a.z + (long)(blk / (unsigned)a.z_div) * n * (long)f;
    float dz[FP];
    float q = 0.0f;
#pragma unroll
    for (int c = 0; c < FP; ++c) {
        dz[c] = (vij && c < f) ? (zb[(long)i * f + c] - zb[(long)j * f + c]) * ils[c] : 0.0f;
        q = fmaf(dz[c], dz[c], q);
    }
    const float e = vij ? __builtin_amdgcn_exp2f(-0.7213475204444817f * q) : 0.0f;      // exp(-q / 2)
    auto resid = [&](int t) -> float {
        if (t >= nv) return 0.0f;
        float m = 0.0f;
        if (a.mean_mode == PACOH_MEAN_VECTOR) m = a.mean[b * n + t];
        else if (a.mean_mode == PACOH_MEAN_CONST) m = a.mean[p];
        return a.y[ty * n + t] - m;
    };
    const float rj = resid(j);

    float A = 0.0f, logdet = 0.0f;                            // logdet: log det K = sum_k log p_k (padding rows: pivot 1)
    int my_info = -1;
    float jitter = 0.0f;
    for (int attempt = 0; attempt < 4; ++attempt) {
        A = vij ? fmaf(os, e, i == j ? noise + jitter : 0.0f) : (i == j ? 1.0f : 0.0f);
        float l2 = 0.0f;                                      // sum of log2 of the pivots (wave-uniform operands: eight cheap v_log)
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float pk = readlane_(A, 9 * k);
            ok = ok && pk > 0.0f && pk < __builtin_huge_valf();
            const float d = rcp_(pk);
            const float aik = __shfl(A, (lane & 0x38) | k, 64);       // A[i][k]
            const float akj = __shfl(A, (k << 3) | j, 64);            // A[k][j]
            const float upd = fmaf(-aik * d, akj, A);
            A = (i == k) ? (j == k ? d : akj * d) : (j == k ? -aik * d : upd);
            l2 += __builtin_amdgcn_logf(pk);
        }
        if (ok) { my_info = attempt; logdet = 0.6931471805599453f * l2; break; }
        jitter = 1e-6f;
        for (int t = 0; t < attempt; ++t) jitter *= 10.0f;
    }
    const bool okf = my_info >= 0;
    if (lane == 0 && a.info) a.info[b] = my_info;
    const float bad = okf ? 0.0f : NAN;

    // alpha = K^-1 r: alpha_i in every lane of row i, alpha_j by the symmetric sum over the lanes of column j
    const float ai = row8_sum_(A * rj);
    float aj = A * __shfl(rj, (i << 3) | i, 64);              // K^-1_ij r_i (r_i sits in the diagonal lane of row i)
    aj += __shfl_xor(aj, 8, 64); aj += __shfl_xor(aj, 16, 64); aj += __shfl_xor(aj, 32, 64);
    const float quad = wave_sum_(i == j ? rj * aj : 0.0f);
    const float inv_nv = nv > 0 ? rcp_((float)nv) : 0.0f;
    float lml = -0.5f * (quad + logdet + (float)nv * 1.8378770664093453f) * inv_nv;
    if (!okf) lml = NAN;
    if (lane == 0) a.lml[b] = lml;

    // gradients: H_ij = (alpha_i alpha_j - K^-1_ij) / 2 over the valid pairs, W_ij = H_ij e_ij
    const float gup = a.g_lml ? a.g_lml[b] : 1.0f;
    const float sc = gup * inv_nv;
    const float H = vij ? 0.5f * fmaf(ai, aj, -A) : 0.0f;
    const float W = H * e;
    const float s_noise = wave_sum_(i == j ? H : 0.0f);
    const float s_os = wave_sum_(W);
    if (lane == 0) {
        if (a.d_os) a.d_os[b] = sc * s_os + bad;
        a.d_noise[b] = sc * s_noise + bad;
    }
#pragma unroll
    for (int c = 0; c < FP; ++c) {
        if (c < f) {
            // d K_ij / d ls_c = os e_ij dz_c^2 / ls_c (dz already divided by ls_c);  d K_ij / d z_ic = -os e_ij dz_c / ls_c, and (i, j), (j, i) both move
            const float s_ls = wave_sum_(W * dz[c] * dz[c]);
            if (lane == 0) a.d_ls[b * f + c] = sc * os * s_ls * ils[c] + bad;
            const float rz = row8_sum_(W * dz[c]);
            if (a.d_z && j == 0 && i < n) a.d_z[(b * n + i) * (long)f + c] = vi ? -2.0f * sc * os * rz * ils[c] + bad : 0.0f;
        }
    }
    if (a.mean_mode == PACOH_MEAN_VECTOR) {
        if (a.d_mean && j == 0 && i < n) a.d_mean[b * n + i] = vi ? sc * ai + bad : 0.0f;
    } else if (a.mean_mode == PACOH_MEAN_CONST) {
        const float sa = wave_sum_((j == 0 && vi) ? ai : 0.0f);
        if (a.d_mean && lane == 0) a.d_mean[b] = sc * sa + bad;
    }
}

}  // namespace gpreg
}  // namespace pacoh
