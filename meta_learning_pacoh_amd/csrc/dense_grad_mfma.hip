// Gradient contractions of the large-context path on the matrix cores, ARD-RBF, f <= 8 (round 4: fp64; round 5: fp32 as well).
//
//   G = (alpha alpha^T - W) / 2n,  M = G o (os K),   d_z_i = sum_j M_ij (z_j - z_i),   d_os = sum_ij G_ij K_ij / os,  d_noise = sum_i G_ii
//
// dense_grad_cols_kernel evaluated every ORDERED pair (i, j) on the vector units: eight coordinate differences, an fp64 exp of
// ~25 instructions and eight more fmas per pair -- 0.33 ms per 256 x 512^2 launch, bound by the fp64 vector rate.  Here
//   * M is symmetric, so only the tiles on and below the block diagonal are evaluated (half the exps, half of W read): a 64 x 64
//     tile (I, J), J < I, feeds the rows of I with M Z_J and the rows of J with M^T Z_I;
//   * the squared distances come from ONE MFMA product per 16 x 16 block (|z_i|^2 + |z_j|^2 - 2 z_i . z_j; fp64: the
//     cancellation costs ~1e-16 |z|^2, far below the path's 1e-8 parity bar -- fp32 keeps the direct differences of the old kernel),
//     and both contractions with the coordinates are MFMA products as well: the accumulator block of M is directly the A operand
//     of M^T Z_I (register index = k), and goes through a 16 x 17 LDS scratch once for M Z_J.  A column of ones appended to the
//     coordinate images delivers the row / column sums of M (the -z_i sum_j M_ij term and d_os) from the same products.
// Per tile and wave: 40 MFMAs and 64 exps per lane, where the vector kernel spent ~3500 fp64 instructions per lane.
// One workgroup per (problem, row block I), four waves = the four 16-column strips of a tile.  The column-side sums of the tiles
// J < I land in a partial buffer [problem][tile][64][9] and are added in fixed order by dense_grad_combine_kernel (deterministic).
// Reference: the backward of ExactMarginalLogLikelihood through the RBF kernel (meta_learn/random_gp.py:83-85, models.py:428-446).
#include "common.h"
#include "dense_diag.h"

namespace pacoh {
namespace {

constexpr int GT = 64;               // tile edge
constexpr int GZL = 17;              // leading dimension of the coordinate images [64][16 (+1)]: 8 coordinates | 1 | zeros


// a - b * c with the product ROUNDED first (no fused multiply-add): sum_j M_ij z_j - z_i sum_j M_ij is exactly zero for a task of one
// point (and for coincident points) only if both products are rounded alike -- the direct-difference kernel returned exact zeros there
template <typename T>
__device__ __forceinline__ T sub_rounded_product(T a, T b, T c) {
#pragma clang fp contract(off)
    const T p = b * c;
    return a - p;
}

__device__ __forceinline__ int clamp_nv2(const int32_t* n_valid, long ty, int n) {
    int nv = n_valid ? n_valid[ty] : n;
    nv = nv < n ? nv : n;
    return nv < 0 ? 0 : nv;
}

// number of tiles (I, J), J < I, in front of row block I
__host__ __device__ inline int tiles_before(int I) { return I * (I - 1) / 2; }

template <typename T>
__global__ void __launch_bounds__(256, 3) dense_grad_tile_kernel(const T* __restrict__ zs, const T* __restrict__ osp,
                                                              const int32_t* __restrict__ n_valid, int y_div,
                                                              const T* __restrict__ alpha, const T* __restrict__ Wm,
                                                              const int32_t* __restrict__ info, T* __restrict__ rowside,
                                                              T* __restrict__ colpart, T* __restrict__ gdiag, int P, int n,
                                                              int f) {
    using Acc = typename Mf<T>::acc;
    auto gm_row = [](int g_, int q_) { return Mf<T>::row(g_, q_); };       // accumulator row of register q = the k index of step q
    __shared__ T ZI[GT][GZL], ZJ[2][GT][GZL], n2I[GT], n2J[2][GT], aI[GT], aJ[2][GT];
    __shared__ T Tr[4][16][17];
    __shared__ T Rows[GT][GZL];
    // grid (problems, row blocks), LONGEST row block first.  Workgroups go to the eight XCDs round-robin in linear-id order: with the
    // row block as the fast index and eight row blocks (n = 512) XCD k received every workgroup of row block k -- 1 tile each on XCD
    // 0, 8 tiles each on XCD 7, which then ran 1.8 x the balanced time (267 us; this order: see profiles/r05_dense_kernel_stats_fp64.csv).
    const long b = blockIdx.x;
    const int nI = (n + GT - 1) / GT;
    const int I = nI - 1 - (int)blockIdx.y, I0 = I * GT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    if (info[b] < 0) return;                               // (the combine kernel writes the NaNs)
    const int nv = clamp_nv2(n_valid, b / y_div, n);
    const T os = osp ? osp[b % P] : 1.0;
    const T inv2n = nv > 0 ? T(0.5) / (T)nv : 0.0;
    const T* zb = zs + b * (long)n * f;
    const T* ab = alpha + b * (long)n;
    const T* Wb = Wm + b * (long)n * n;

    auto stage = [&](T (*Z)[GZL], T* n2, T* al, int R0) {
        // rows R0 .. R0 + 64 of the scaled coordinates: [c < f] coordinates, [8] = 1, rest 0; rows beyond nv: all zero (no weight)
        for (int e = tid; e < GT * 16; e += 256) {
            const int i = e >> 4, c = e & 15;
            const int row = R0 + i;
            T v = 0.0;
            if (row < nv) v = c < f ? zb[(long)row * f + c] : (c == 8 ? 1.0 : 0.0);
            Z[i][c] = v;
        }
        if (tid < GT) {
            const int row = R0 + tid;
            T s = 0.0;
            if (row < nv) for (int c = 0; c < f; ++c) { const T v = zb[(long)row * f + c]; s = fma(v, v, s); }
            n2[tid] = s;
            al[tid] = row < nv ? ab[row] : 0.0;
        }
    };
    stage(ZI, n2I, aI, I0);
    stage(ZJ[0], n2J[0], aJ[0], 0);
    Acc rowacc[4];
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) rowacc[ib] = Acc{0, 0, 0, 0};
    const int j = 16 * w + r;                              // this lane's column inside a tile
    // this lane's entries of W for one 16 x 16 block: unconditional loads at clamped addresses, requested one block ahead
    auto load_w = [&](int J0, int ib, T (&wv)[4]) __attribute__((always_inline)) {
        const int jc = J0 + j < n ? J0 + j : n - 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ic = I0 + 16 * ib + gm_row(g, q);
            wv[q] = Wb[(long)(ic < n ? ic : n - 1) * n + jc];
        }
    };

    for (int J = 0; J <= I; ++J) {
        const int J0 = J * GT, cur = J & 1;
        __syncthreads();                                   // tile J's coordinates are staged; the other buffer's readers are done
        if (J < I) stage(ZJ[cur ^ 1], n2J[cur ^ 1], aJ[cur ^ 1], J0 + GT);      // next tile's coordinates fly under this tile's work
        T (*Zj)[GZL] = ZJ[cur];
        const T* n2j = n2J[cur];
        const T* aj = aJ[cur];
        const bool diag = J == I;
        Acc oj = {0, 0, 0, 0};
        T wv[2][4];
        load_w(J0, 0, wv[0]);
        // (Also measured and dropped: eight waves per workgroup sharing an LDS image of the M tile -- one accumulator per side and wave,
        //  128 registers, four waves per SIMD, two workgroup barriers per tile instead of eight wave-level waits: 336 us.)
        // one 16 x 16 block of the tile at a time, W requested one block ahead.  (Requesting the WHOLE next tile a tile ahead -- 16 loads
        // per lane in flight instead of 4 -- needs 223 registers, two waves per SIMD instead of three: 268 -> 306 us.  The kernel
        // waits on its own LDS / MFMA chain per block, not on HBM.)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            if (ib < 3) load_w(J0, ib + 1, wv[(ib + 1) & 1]);
            Acc s = {0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 2; ++st)
                s = Mf<T>::mma(ZI[16 * ib + r][4 * st + g], Zj[16 * w + r][4 * st + g], s);
            Acc M;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * ib + gm_row(g, q);
                const bool ok = I0 + i < nv && J0 + j < nv;
                T d2 = n2I[i] + n2j[j] - 2.0 * s[q];
                d2 = d2 > 0.0 ? d2 : 0.0;
                const T e = rbf_exp<T>(-0.5 * d2);
                const T G = ok ? (aI[i] * aj[j] - wv[ib & 1][q]) * inv2n : 0.0;
                if (diag && i == j && ok) gdiag[b * (long)n + I0 + i] = G;
                M[q] = G * os * e;
            }
            // rows of J <- M^T [Z_I | 1] (tiles below the block diagonal only): the accumulator block of M as A operand = its transpose
            if (!diag) {
#pragma unroll
                for (int st = 0; st < 4; ++st)
                    oj = Mf<T>::mma(M[st], ZI[16 * ib + gm_row(g, st)][r], oj);
            }
            // rows of I <- M [Z_J | 1]: M's block through the wave's 16 x 17 scratch (A operand = M itself)
#pragma unroll
            for (int q = 0; q < 4; ++q) Tr[w][gm_row(g, q)][r] = M[q];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int st = 0; st < 4; ++st)
                rowacc[ib] = Mf<T>::mma(Tr[w][r][gm_row(g, st)], Zj[16 * w + gm_row(g, st)][r], rowacc[ib]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
        }
        if (!diag) {
            // oj: lane (r, g) register q = sum_i M[i][jj] [Z_I | 1][i][c = r], jj = 16 w + row(g, q)
            T* cp = colpart + ((b * (long)(nI * (nI - 1) / 2) + tiles_before(I) + J) * GT) * 9;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jj = 16 * w + gm_row(g, q);
                const T cs = __shfl(oj[q], 16 * g + 8, 64);          // column sum of M (the ones column, c = 8)
                if (r < 8) cp[(long)jj * 9 + r] = sub_rounded_product<T>(oj[q], Zj[jj][r], cs);
                else if (r == 8) cp[(long)jj * 9 + 8] = cs;
            }
        }
    }
    // the four strips' row sums, added in wave order (fixed: deterministic)
    for (int ww = 0; ww < 4; ++ww) {
        __syncthreads();
        if (w == ww) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    T* p = &Rows[16 * ib + gm_row(g, q)][r];
                    *p = (ww == 0 ? 0.0 : *p) + rowacc[ib][q];
                }
        }
    }
    __syncthreads();
    for (int e = tid; e < GT * 9; e += 256) {
        const int i = e / 9, c = e - i * 9;
        if (I0 + i < n) rowside[(b * (long)n + I0 + i) * 9 + c] = c < 8 ? sub_rounded_product<T>(Rows[i][c], ZI[i][c], Rows[i][8]) : Rows[i][8];
    }
}

// per row: the row-side sums + the column-side partial sums of the tiles below it (fixed order), then the outputs of
// dense_grad_cols_kernel: d_z, d_mean, rowpart = {lengthscale terms, d_os term, G_ii, alpha_i}.
// One thread per (row, sum c = 0..8): consecutive threads read consecutive doubles of rowside [row][9] and of a tile's partials
// [64][9] (round 5; one thread per row walking its nine sums read 72-byte strides and took 36.5 us per 256 x 512 launch).
constexpr int GC_ROWS = 28;          // rows per 256-thread workgroup (252 threads busy)
template <typename T>
__global__ void __launch_bounds__(256) dense_grad_combine_kernel(const T* __restrict__ zs, const T* __restrict__ lsp,
                                                                 const T* __restrict__ osp, const int32_t* __restrict__ n_valid,
                                                                 int y_div, const T* __restrict__ g_lml,
                                                                 const T* __restrict__ alpha, const int32_t* __restrict__ info,
                                                                 const T* __restrict__ rowside, const T* __restrict__ colpart,
                                                                 const T* __restrict__ gdiag, T* __restrict__ d_z,
                                                                 T* __restrict__ d_mean, int mean_mode, T* __restrict__ rowpart,
                                                                 int P, int n, int f) {
    const long b = blockIdx.y;
    const int t = threadIdx.x, lr = t / 9, c = t - 9 * lr;
    const int i = blockIdx.x * GC_ROWS + lr;
    if (lr >= GC_ROWS || i >= n) return;
    const int p = (int)(b % P);
    const int nv = clamp_nv2(n_valid, b / y_div, n);
    const bool failed = info[b] < 0;
    const T gup = g_lml ? g_lml[b] : 1.0;
    const int W3 = f + 3;
    T* rp = rowpart + (b * n + i) * (long)W3;
    if (failed || i >= nv) {
        const T v = failed ? T(NAN) : 0.0;
        if (c < f) { if (d_z) d_z[(b * n + i) * (long)f + c] = v; rp[c] = v; }
        if (c == 8) {
            if (d_mean && mean_mode == PACOH_MEAN_VECTOR) d_mean[b * n + i] = v;
            rp[f] = v; rp[f + 1] = v; rp[f + 2] = v;
        }
        return;
    }
    const int nI = (n + GT - 1) / GT, Ji = i / GT, il = i - Ji * GT;
    T acc = rowside[(b * (long)n + i) * 9 + c];
    for (int I = Ji + 1; I < nI; ++I)
        acc += colpart[((b * (long)(nI * (nI - 1) / 2) + tiles_before(I) + Ji) * GT + il) * 9 + c];
    if (c < f) {
        const T* zb = zs + b * (long)n * f;
        if (d_z) d_z[(b * n + i) * (long)f + c] = 2.0 * gup * acc / lsp[(long)p * f + c];
        rp[c] = -2.0 * (zb[(long)i * f + c] - zb[c]) * acc;            // (lengthscale sums from the finished d_z sums, as dense_grad_cols_kernel)
    } else if (c == 8) {
        const T os = osp ? osp[p] : 1.0;
        const T ai = alpha[b * (long)n + i];
        rp[f] = acc / os;                                  // sum_j G_ij K_ij / os = sum_j M_ij / os
        rp[f + 1] = gdiag[b * (long)n + i];
        rp[f + 2] = ai;
        if (d_mean && mean_mode == PACOH_MEAN_VECTOR) d_mean[b * n + i] = gup * ai / (T)nv;
    }
}

// ---- the Gram matrix for the factorisation, same idea -----------------------------------------------------------------------------
// A = os exp(-|u_i - u_j|^2 / 2) + noise I, u = z / lengthscale, tiles (I, J) with J <= I only (every Cholesky kernel of the path reads
// the lower triangle; diagonal tiles are written whole).  gram_kernel spends 8 subtractions + 8 fmas + the 20-instruction exp per
// entry on the fp64 vector units and is bound by them at d = 8 (241 us for the full 256 x 512^2 matrix against 113 us at d = 2, the same
// bytes); here the distances are |u_i|^2 + |u_j|^2 - 2 u_i . u_j with the product on the matrix cores.  The coordinates are taken
// relative to the problem's first point before scaling, which keeps |u|^2 -- and with it the cancellation error ~1e-16 |u|^2 -- small;
// the diagonal is exact by construction (d2 = 0), the result is symmetric bit for bit (S_ij and S_ji are the same products).
// MIRROR (pacoh_gram_rbf_ard on one point set): the tiles below the block diagonal are also written transposed, through a 16 x 17
// LDS scratch per wave so that both stores are 128-byte row segments -- half the exps of the full matrix, all of its bytes.
template <bool MIRROR>
__global__ void __launch_bounds__(256, 4) dense_gram_tile_kernel(const double* __restrict__ z, int z_div, const double* __restrict__ lsp,
                                                                 const double* __restrict__ osp, const double* __restrict__ noisep,
                                                                 double* __restrict__ K, int P, int n, int f) {
    using Acc = f64x4_t;
    auto gm_row = [](int g_, int q_) { return Mf<double>::row(g_, q_); };
    __shared__ double ZI[GT][GZL], ZJ[GT][GZL], n2I[GT], n2J[GT];
    __shared__ double Tr[MIRROR ? 4 : 1][16][17];
    const long b = blockIdx.y;
    int I = (int)((sqrtf(8.0f * (float)blockIdx.x + 1.0f) - 1.0f) * 0.5f);
    while (I * (I + 1) / 2 > (int)blockIdx.x) --I;
    while ((I + 1) * (I + 2) / 2 <= (int)blockIdx.x) ++I;
    const int J = blockIdx.x - I * (I + 1) / 2;
    const int I0 = I * GT, J0 = J * GT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int p = (int)(b % P);
    const double* zb = z + (b / z_div) * (long)n * f;
    const double* ls = lsp + (long)p * f;
    auto stage = [&](double (*Z)[GZL], double* n2, int R0) {
        for (int e = tid; e < GT * 8; e += 256) {
            const int i = e >> 3, c = e & 7;
            const int row = R0 + i;
            Z[i][c] = (row < n && c < f) ? (zb[(long)row * f + c] - zb[c]) / ls[c] : 0.0;
        }
        __syncthreads();
        if (tid < GT) {
            double s = 0.0;
            for (int c = 0; c < 8; ++c) s = fma(Z[tid][c], Z[tid][c], s);
            n2[tid] = s;
        }
    };
    stage(ZI, n2I, I0);
    stage(ZJ, n2J, J0);
    __syncthreads();
    const double os = osp ? osp[p] : 1.0, noise = noisep ? noisep[p] : 0.0;
    const int j = 16 * w + r, gj = J0 + j;
    double* Kb = K + b * (long)n * n;
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) {
        Acc s = {0, 0, 0, 0};
#pragma unroll
        for (int st = 0; st < 2; ++st)
            s = __builtin_amdgcn_mfma_f64_16x16x4f64(ZI[16 * ib + r][4 * st + g], ZJ[j][4 * st + g], s, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = 16 * ib + gm_row(g, q), gi = I0 + i;
            double d2 = n2I[i] + n2J[j] - 2.0 * s[q];
            d2 = (d2 > 0.0 && gi != gj) ? d2 : 0.0;
            double k = os * rbf_exp<double>(-0.5 * d2);
            if (gi == gj) k += noise;
            if (gi < n && gj < n) Kb[(long)gi * n + gj] = k;
            if constexpr (MIRROR) Tr[w][gm_row(g, q)][r] = k;
        }
        if constexpr (MIRROR) {
            if (I != J) {                                          // (wave-uniform) K[J0 + 16 w + jj][I0 + 16 ib + ii], ii = r: rows of 128 bytes
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int jj = gm_row(g, q), mj = J0 + 16 * w + jj, mi = I0 + 16 * ib + r;
                    if (mj < n && mi < n) Kb[(long)mj * n + mi] = Tr[w][r][jj];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// The same tiles for the factorisation's Gram matrix (lower block triangle, no mirror), a ROW of tiles per workgroup (round 5): row
// block I's coordinates are staged once and the tiles J = 0 .. I walked with tile J + 1's coordinates staged under tile J's work --
// one barrier per tile instead of a whole prologue (two global round trips and three barriers) per tile; grid (problem, row block),
// longest row first (the XCD lesson of dense_grad_tile_kernel).
__global__ void __launch_bounds__(256, 4) dense_gram_row_kernel(const double* __restrict__ z, int z_div, const double* __restrict__ lsp,
                                                                const double* __restrict__ osp, const double* __restrict__ noisep,
                                                                double* __restrict__ K, int P, int n, int f) {
    using Acc = f64x4_t;
    auto gm_row = [](int g_, int q_) { return Mf<double>::row(g_, q_); };
    __shared__ double ZI[GT][GZL], ZJ[2][GT][GZL], n2I[GT], n2J[2][GT];
    const long b = blockIdx.x;
    const int nI = (n + GT - 1) / GT;
    const int I = nI - 1 - (int)blockIdx.y, I0 = I * GT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int p = (int)(b % P);
    const double* zb = z + (b / z_div) * (long)n * f;
    const double* ls = lsp + (long)p * f;
    // scaled coordinates relative to the problem's first point (columns 0..7, zero beyond f) and their squared norms: a thread owns
    // whole rows' quarter -- 64 rows x 4 threads -- so that the norm is a sum over its own registers and one lane exchange
    auto stage = [&](double (*Z)[GZL], double* n2, int R0) {
        const int i = tid >> 2, c0 = (tid & 3) * 2;
        const int row = R0 + i;
        double v0 = 0.0, v1 = 0.0;
        if (row < n) {
            if (c0 < f) v0 = (zb[(long)row * f + c0] - zb[c0]) / ls[c0];
            if (c0 + 1 < f) v1 = (zb[(long)row * f + c0 + 1] - zb[c0 + 1]) / ls[c0 + 1];
        }
        Z[i][c0] = v0; Z[i][c0 + 1] = v1;
        double s = fma(v0, v0, v1 * v1);
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
        if ((tid & 3) == 0) n2[i] = s;
    };
    stage(ZI, n2I, I0);
    stage(ZJ[0], n2J[0], 0);
    const double os = osp ? osp[p] : 1.0, noise = noisep ? noisep[p] : 0.0;
    double* Kb = K + b * (long)n * n;
    for (int J = 0; J <= I; ++J) {
        const int J0 = J * GT, cur = J & 1;
        __syncthreads();                                   // tile J's coordinates are staged; the other buffer's readers are done
        if (J < I) stage(ZJ[cur ^ 1], n2J[cur ^ 1], J0 + GT);
        const int j = 16 * w + r, gj = J0 + j;
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) {
            Acc s = {0, 0, 0, 0};
#pragma unroll
            for (int st = 0; st < 2; ++st)
                s = __builtin_amdgcn_mfma_f64_16x16x4f64(ZI[16 * ib + r][4 * st + g], ZJ[cur][j][4 * st + g], s, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = 16 * ib + gm_row(g, q), gi = I0 + i;
                double d2 = n2I[i] + n2J[cur][j] - 2.0 * s[q];
                d2 = (d2 > 0.0 && gi != gj) ? d2 : 0.0;
                double k = os * rbf_exp<double>(-0.5 * d2);
                if (gi == gj) k += noise;
                if (gi < n && gj < n) Kb[(long)gi * n + gj] = k;
            }
        }
    }
}

}  // namespace

// fp64 ARD-RBF Gram + noise for the factorisation, lower block triangle only; returns 1 when outside its plan (caller: gram_kernel)
int dense_gram_mfma_try(const void* z, int z_div, const void* ls, const void* os, const void* noise, void* K, int B, int P, int n, int f,
                        int dtype, hipStream_t s) {
    const bool on = g_sw.gram_mfma;
    if (!on || dtype != PACOH_F64 || f > 8 || n < GT || !noise) return 1;
    const int nI = (n + GT - 1) / GT;
    // (one workgroup per tile -- dense_gram_tile_kernel<false> -- took 91 us per 256 x 512^2 launch, a row of tiles per workgroup 78)
    hipLaunchKernelGGL(dense_gram_row_kernel, dim3(B, nI), dim3(256), 0, s, (const double*)z, z_div, (const double*)ls,
                       (const double*)os, (const double*)noise, (double*)K, P, n, f);
    return launch_status();
}

// pacoh_gram_rbf_ard with z1 == z2 (one point set, square, fp64, f <= 8, n >= 64): the whole symmetric matrix from its lower tiles
int dense_gram_mfma_full(const void* z, int z_div, const void* ls, const void* os, const void* noise, void* K, int B, int P, int n, int f,
                         hipStream_t s) {
    const bool on = g_sw.gram_mfma;
    if (!on || f > 8 || n < GT) return 1;
    const int nI = (n + GT - 1) / GT;
    hipLaunchKernelGGL(dense_gram_tile_kernel<true>, dim3(nI * (nI + 1) / 2, B), dim3(256), 0, s, (const double*)z, z_div, (const double*)ls,
                       (const double*)os, (const double*)noise, (double*)K, P, n, f);
    return launch_status();
}

// scratch the two kernels need (bytes): rowside [B][n][9] | column-side partials [B][tiles][64][9] | G_ii [B][n]
size_t dense_grad_mfma_scratch(int B, int n, int dtype) {
    const size_t nI = (n + GT - 1) / GT;
    return ((size_t)B * n * 9 + (size_t)B * (nI * (nI - 1) / 2) * GT * 9 + (size_t)B * n) * (dtype == PACOH_F64 ? 8 : 4);
}

// does the MFMA contraction take this call?  (the caller asks before it builds W: the tile kernel reads W's lower 64-tiles only)
// fp32 since round 5: the squared distances from one product cost ~1e-7 |z|^2 there -- harmless in the contraction, whose kernel
// entries only weight sums (the Gram matrix that is FACTORED keeps its direct differences in fp32: dense_gram_mfma_try)
bool dense_grad_mfma_plan(int B, int n, int f, int kind, int dtype, size_t scratch_bytes) {
    const bool f32_on = g_sw.grad_mfma_f32;
    return (dtype == PACOH_F64 || f32_on) && kind == PACOH_KERNEL_RBF && f <= 8 && n >= GT && dense_grad_mfma_scratch(B, n, dtype) <= scratch_bytes;
}

template <typename T>
static int grad_mfma_launch(const void* zs, const void* ls, const void* os, const int32_t* n_valid, int y_div, const void* g_lml,
                            const void* alpha, const void* Wm, const int32_t* info, void* d_z, void* d_mean, int mean_mode, void* rowpart,
                            void* scratch, int B, int P, int n, int f, hipStream_t s) {
    const int nI = (n + GT - 1) / GT;
    T* rowside = (T*)scratch;
    T* colpart = rowside + (size_t)B * n * 9;
    T* gdiag = colpart + (size_t)B * ((size_t)nI * (nI - 1) / 2) * GT * 9;
    hipLaunchKernelGGL(dense_grad_tile_kernel<T>, dim3(B, nI), dim3(256), 0, s, (const T*)zs, (const T*)os, n_valid, y_div,
                       (const T*)alpha, (const T*)Wm, info, rowside, colpart, gdiag, P, n, f);
    hipLaunchKernelGGL(dense_grad_combine_kernel<T>, dim3((n + GC_ROWS - 1) / GC_ROWS, B), dim3(256), 0, s, (const T*)zs, (const T*)ls,
                       (const T*)os, n_valid, y_div, (const T*)g_lml, (const T*)alpha, info, rowside, colpart, gdiag,
                       (T*)d_z, (T*)d_mean, mean_mode, (T*)rowpart, P, n, f);
    return launch_status();
}

// ARD-RBF, f <= 8: the MFMA contraction; returns 1 when outside its plan (caller: dense_grad_cols / rows kernels)
int dense_grad_mfma_try(const void* zs, const void* ls, const void* os, const int32_t* n_valid, int y_div, const void* g_lml,
                        const void* alpha, const void* Wm, const int32_t* info, void* d_z, void* d_mean, int mean_mode, void* rowpart,
                        void* scratch, size_t scratch_bytes, int B, int P, int n, int f, int kind, int dtype, hipStream_t s) {
    if (!dense_grad_mfma_plan(B, n, f, kind, dtype, scratch_bytes)) return 1;
    return dtype == PACOH_F64 ? grad_mfma_launch<double>(zs, ls, os, n_valid, y_div, g_lml, alpha, Wm, info, d_z, d_mean, mean_mode, rowpart, scratch, B, P, n, f, s)
                              : grad_mfma_launch<float>(zs, ls, os, n_valid, y_div, g_lml, alpha, Wm, info, d_z, d_mean, mean_mode, rowpart, scratch, B, P, n, f, s);
}

}  // namespace pacoh
