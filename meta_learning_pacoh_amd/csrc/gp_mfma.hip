// MFMA-blocked fused task-GP kernel for n <= 64, fp32: one 64-lane wavefront per (task, particle)
// problem, the n x n matrix lives in LDS (17 KB at n = 64 -> 8 problems per CU) and every O(n^3)
// phase runs on the matrix cores as 16x16 block products (v_mfma_f32_16x16x4_f32, exact fp32):
//
//   A = os*K(z) + (noise+jitter) I        row-per-lane Gram build straight into LDS (lower blocks)
//   blocked right-looking Cholesky        diagonal 16x16 block: factor + invert in registers
//                                         (row per lane, broadcasts by v_readlane), panel and trailing
//                                         update as X*Y^T block products
//   Z = L^-1 (in place, block forward substitution), W = K^-1 = Z^T Z (in place, mirrored to a full
//   symmetric matrix), u = Z r, alpha = W r, then the gradient sums with lane i owning row i of W.
//
// The k index of a block product is permuted (lane group g, chunk s -> k = 4g + s) so that one
// ds_read_b128 per operand feeds the four MFMAs of a 16x16x16 product, and so that a product held in
// the accumulator layout is directly the B operand of the next product (no LDS round trip).
//
// Same arithmetic as gp_small.hip (which remains the general path: any n <= 128, fp64, predict);
// reference lines replaced: random_gp.py:54-89, GPR_meta_mll.py:104-117 (through gpytorch).
#include "common.h"
#include <stdlib.h>

namespace pacoh {

using f32x4 = __attribute__((ext_vector_type(4))) float;

struct GpMfmaArgs {
    const float* z; int z_div;
    const float* mean; int mean_mode;
    const float* y; int y_div;
    const float* ls; const float* os; const float* noise;
    const int32_t* n_valid;
    const float* g_lml;
    float* lml; int32_t* info;
    float* d_z; float* d_mean; float* d_ls; float* d_os; float* d_noise;
    int B, P, n, f;
};

#ifdef PACOH_GP_STAMPS
// diagnostic build (python -m meta_learning_pacoh_amd._build --variant stamps -DPACOH_GP_STAMPS=1, tools/gp_stamps.py): s_memtime at
// the phase boundaries of every problem's wave, kept in registers and written once at the end to a slot of a device array of its
// own (no output depends on it, no atomics); summed by pacoh_debug_gp_stamps()
constexpr int STAMP_SLOTS = 32768;
__device__ unsigned int g_gp_stamps[STAMP_SLOTS][12];
#define STAMP(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); st_[k] += (unsigned int)(t_ - t_prev); t_prev = t_; } while (0)
#else
#define STAMP(k) do {} while (0)
#endif

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// acc += sgn * X * Y^T, X at (xr,xc), Y at (yr,yc) in the LDS matrix
__device__ __forceinline__ void mm_xyT(f32x4& acc, const float* A, int LD, int xr, int xc, int yr, int yc, int r, int g, float sgn) {
    const float4 a = *reinterpret_cast<const float4*>(A + (xr + r) * LD + xc + 4 * g);
    const float4 b = *reinterpret_cast<const float4*>(A + (yr + r) * LD + yc + 4 * g);
    acc = mfma4(sgn * a.x, b.x, acc); acc = mfma4(sgn * a.y, b.y, acc);
    acc = mfma4(sgn * a.z, b.z, acc); acc = mfma4(sgn * a.w, b.w, acc);
}

// acc += X * Y, both from LDS
__device__ __forceinline__ void mm_xy(f32x4& acc, const float* A, int LD, int xr, int xc, int yr, int yc, int r, int g) {
    const float4 a = *reinterpret_cast<const float4*>(A + (xr + r) * LD + xc + 4 * g);
    const float* yp = A + (yr + 4 * g) * LD + yc + r;
    acc = mfma4(a.x, yp[0], acc); acc = mfma4(a.y, yp[LD], acc);
    acc = mfma4(a.z, yp[2 * LD], acc); acc = mfma4(a.w, yp[3 * LD], acc);
}

// acc += sgn * X * S, X from LDS, S a 16x16 block held in accumulator layout
__device__ __forceinline__ void mm_xs(f32x4& acc, const float* A, int LD, int xr, int xc, const f32x4& S, int r, int g, float sgn) {
    const float4 a = *reinterpret_cast<const float4*>(A + (xr + r) * LD + xc + 4 * g);
    acc = mfma4(sgn * a.x, S[0], acc); acc = mfma4(sgn * a.y, S[1], acc);
    acc = mfma4(sgn * a.z, S[2], acc); acc = mfma4(sgn * a.w, S[3], acc);
}

// acc += X^T * Y, both from LDS
__device__ __forceinline__ void mm_xTy(f32x4& acc, const float* A, int LD, int xr, int xc, int yr, int yc, int r, int g) {
    const float* xp = A + (xr + 4 * g) * LD + xc + r;
    const float* yp = A + (yr + 4 * g) * LD + yc + r;
    acc = mfma4(xp[0], yp[0], acc); acc = mfma4(xp[LD], yp[LD], acc);
    acc = mfma4(xp[2 * LD], yp[2 * LD], acc); acc = mfma4(xp[3 * LD], yp[3 * LD], acc);
}

__device__ __forceinline__ f32x4 load_c(const float* A, int LD, int br, int bc, int r, int g) {
    const float* p = A + (br + 4 * g) * LD + bc + r;
    f32x4 c; c[0] = p[0]; c[1] = p[LD]; c[2] = p[2 * LD]; c[3] = p[3 * LD];
    return c;
}

__device__ __forceinline__ void store_c(float* A, int LD, int br, int bc, int r, int g, const f32x4& c) {
    float* p = A + (br + 4 * g) * LD + bc + r;
    p[0] = c[0]; p[LD] = c[1]; p[2 * LD] = c[2]; p[3 * LD] = c[3];
}

// store the TRANSPOSE of the accumulator block at (br, bc): element (4g+s, r) -> A[br + r][bc + 4g + s]
__device__ __forceinline__ void store_ct(float* A, int LD, int br, int bc, int r, int g, const f32x4& c) {
    float4 v; v.x = c[0]; v.y = c[1]; v.z = c[2]; v.w = c[3];
    *reinterpret_cast<float4*>(A + (br + r) * LD + bc + 4 * g) = v;
}

// v_readlane_b32 of a float (the builtin is declared on int: pass the BITS, not the value)
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// v + (v of the lane a DPP control selects); rows a row_mask leaves out contribute 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF));
}
// sum over the 64 lanes, returned in every lane (as a scalar): six DPP adds and one v_readlane instead of six dependent
// ds_bpermute round trips (__shfl_xor)
__device__ __forceinline__ float wave_sum(float v) {
    v = dpp_add<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]
    v = dpp_add<0x4E, 0xF>(v);       // quad_perm [2,3,0,1]
    v = dpp_add<0x141, 0xF>(v);      // row_half_mirror
    v = dpp_add<0x140, 0xF>(v);      // row_mirror: every lane holds the sum of its 16-lane row
    v = dpp_add<0x142, 0xA>(v);      // row_bcast15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v);      // row_bcast31 into rows 2 and 3: lane 63 holds the total
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Factor the 16x16 diagonal block at (d0,d0) and invert its factor; a pivot that is not positive raises the sticky flag scr[40]
// (the block then fills with NaNs, which the retry with more jitter rebuilds from scratch).
// On exit the block holds L11^-1 (lower triangular, zeros above) and invd[d0..d0+15] = 1/diag(L11).
//
// The block S is held in the MFMA accumulator layout (lane (r,g), register s <-> S[4g+s][r]) and eliminated FOUR columns at a
// time.  Step k: the pivot rows 4k..4k+3 sit in the registers of lane row g = k.  The 4x4 pivot block reaches every lane as
// ten v_readlane broadcasts (wave-uniform values), the four pivot-row entries of a lane's own column as four ds_bpermute
// (off the critical path).  Every lane then runs the 4x4 Cholesky Lp in its own registers -- a chain of 4 x (rsq, mul, fma),
// no cross-lane traffic -- solves its row of the 16x4 panel X Lp^T = S[:,k] by forward substitution, and the rank-4 trailing
// update S -= X X^T is ONE v_mfma_f32_16x16x4_f32 whose A and B operand are the same register.  The inverse L11^-1 is built
// alongside by block forward substitution: rows 4k..4k+3 of Z solve Lp Z_k = E_k - (L11 Z)[k], the product L11 Z being four
// MFMAs with Z (accumulator layout = B operand) straight from registers.  hardware rsq (1 ulp) is used as is.
// This replaces a column-by-column elimination whose 240 + 240 cross-lane broadcasts (DPP row_newbcast, one per update) formed
// 16 + 16 dependent steps per block: 5.6 k cycles per block = 35 % of the kernel, and took 100 more registers.
template <int NW>
__device__ __forceinline__ void factor_diag_block(float* A, int LD, int d0, float* invd, float* scr, int r, int g, bool active) {
    // `active` = this wave does the work (wave 0); the caller brackets the call with SYNC() for the other waves.
    if (!active) return;
    float* blk = A + d0 * LD + d0;
    f32x4 C = load_c(A, LD, d0, d0, r, g);                  // S[4g+s][r]
    f32x4 Z = {0.f, 0.f, 0.f, 0.f};                         // rows 4g+s of L11^-1, filled block row by block row
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        // pivot block P[c][j] = S[4k+c][4k+j] = register c of lane (r = 4k+j, g = k): wave-uniform
        const int l0 = 20 * k;
        const float p00 = readlane_f(C[0], l0), p10 = readlane_f(C[1], l0), p20 = readlane_f(C[2], l0), p30 = readlane_f(C[3], l0);
        const float p11 = readlane_f(C[1], l0 + 1), p21 = readlane_f(C[2], l0 + 1), p31 = readlane_f(C[3], l0 + 1);
        const float p22 = readlane_f(C[2], l0 + 2), p32 = readlane_f(C[3], l0 + 2), p33 = readlane_f(C[3], l0 + 3);
        // rt[c] = S[4k+c][r] = register c of lane (r, g = k)
        const int src = (16 * k + r) * 4;
        // (elements copied to scalars first: __builtin_bit_cast applied to a vector ELEMENT reads element 0 whatever the index)
        const float c0 = C[0], c1 = C[1], c2 = C[2], c3 = C[3];
        const float rt0 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(c0)));
        const float rt1 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(c1)));
        const float rt2 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(c2)));
        const float rt3 = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(c3)));
        // ---- 4x4 Cholesky of the pivot block, in every lane: the chain is rsq -> mul -> fma per pivot -----------------------
        const float r0 = __builtin_amdgcn_rsqf(p00);
        const float l10 = p10 * r0, l20 = p20 * r0, l30 = p30 * r0;
        const float q11 = fmaf(-l10, l10, p11);
        const float r1 = __builtin_amdgcn_rsqf(q11);
        const float l21 = fmaf(-l20, l10, p21) * r1, l31 = fmaf(-l30, l10, p31) * r1;
        const float q22 = fmaf(-l21, l21, fmaf(-l20, l20, p22));
        const float r2 = __builtin_amdgcn_rsqf(q22);
        const float l32 = fmaf(-l31, l21, fmaf(-l30, l20, p32)) * r2;
        const float q33 = fmaf(-l32, l32, fmaf(-l31, l31, fmaf(-l30, l30, p33)));
        const float r3 = __builtin_amdgcn_rsqf(q33);
        ok = ok && (p00 > 0.0f) && (q11 > 0.0f) && (q22 > 0.0f) && (q33 > 0.0f);        // (off the chain; NaNs compare false)
        // ---- this lane's row of the panel: X Lp^T = S[:, 4k..4k+3] by forward substitution = column block k of L11 -----------
        float x0 = rt0 * r0;
        float x1 = fmaf(-x0, l10, rt1) * r1;
        float x2 = fmaf(-x1, l21, fmaf(-x0, l20, rt2)) * r2;
        float x3 = fmaf(-x2, l32, fmaf(-x1, l31, fmaf(-x0, l30, rt3))) * r3;
        const int rr = r - 4 * k;                            // row inside (0..3) / below (>= 4) / above (< 0) the pivot block
        if (rr < 0) x0 = 0.0f;
        if (rr < 1) x1 = 0.0f;
        if (rr < 2) x2 = 0.0f;
        if (rr < 3) x3 = 0.0f;
        if (g == 0) {
            float4 v; v.x = x0; v.y = x1; v.z = x2; v.w = x3;
            *reinterpret_cast<float4*>(blk + r * LD + 4 * k) = v;       // L11[:, 4k..4k+3], zeros above the diagonal
            if (rr >= 0 && rr < 4) invd[d0 + r] = rr == 0 ? r0 : (rr == 1 ? r1 : (rr == 2 ? r2 : r3));
        }
        // ---- trailing update S[i][j] -= sum_c X[i][c] X[j][c] for i, j >= 4k+4: one MFMA, A operand == B operand ----------
        if (k < 3) {
            const float xg = g == 0 ? x0 : (g == 1 ? x1 : (g == 2 ? x2 : x3));
            const float am = rr >= 4 ? xg : 0.0f;
            C = mfma4(-am, am, C);
        }
        // ---- rows 4k..4k+3 of L11^-1: Lp Z_k = E_k - (L11 Z)[k-th block row] -------------------------------------------------
        f32x4 Y = {0.f, 0.f, 0.f, 0.f};
        if (k > 0) {
            asm volatile("" ::: "memory");
            const float4 lw = *reinterpret_cast<const float4*>(blk + r * LD + 4 * g);   // (columns >= 4k meet zero rows of Z)
            Y = mfma4(lw.x, Z[0], Y); Y = mfma4(lw.y, Z[1], Y); Y = mfma4(lw.z, Z[2], Y); Y = mfma4(lw.w, Z[3], Y);
        }
        if (g == k) {
            const float t0 = (rr == 0 ? 1.0f : 0.0f) - Y[0], t1 = (rr == 1 ? 1.0f : 0.0f) - Y[1];
            const float t2 = (rr == 2 ? 1.0f : 0.0f) - Y[2], t3 = (rr == 3 ? 1.0f : 0.0f) - Y[3];
            const float z0 = t0 * r0;
            const float z1 = fmaf(-z0, l10, t1) * r1;
            const float z2 = fmaf(-z1, l21, fmaf(-z0, l20, t2)) * r2;
            const float z3 = fmaf(-z2, l32, fmaf(-z1, l31, fmaf(-z0, l30, t3))) * r3;
            Z[0] = z0; Z[1] = z1; Z[2] = z2; Z[3] = z3;
        }
    }
    if (!ok && threadIdx.x == 0) scr[40] = 1.0f;
    asm volatile("" ::: "memory");
    store_c(A, LD, d0, d0, r, g, Z);                         // every read of L11 was issued before (in-order LDS)
}

// sum over the whole workgroup (NW waves); red = NW floats of LDS scratch
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if (NW > 1) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        v = red[0];
#pragma unroll
        for (int q = 1; q < NW; ++q) v += red[q];
    }
    return v;
}

// NB 16x16 block rows/columns (n <= 16*NB), NW = 1 wave (NB <= 4) or 2 waves (NB <= 8) per problem.
// Storage: the lower block triangle holds A -> L (off-diagonal) and L11^-1 (diagonal blocks); Z = L^-1 is
// produced TRANSPOSED into the free upper triangle (block (jb,ib) = Z[ib][jb]^T), which makes its block
// columns independent (no in-place hazard -> split over the waves, no barriers inside) and turns every
// operand of the later products into a contiguous ds_read_b128; W = K^-1 then overwrites the dead lower
// triangle row by row and is mirrored into the upper one.
// Occupancy: LDS admits 8 problems per CU (2 waves per SIMD) at n = 64 and the kernel takes the 189 VGPRs that allows; up to
// n = 48 LDS admits 3 waves per SIMD, which is worth squeezing into 168 VGPRs (a few spilled dwords): -7 % at n = 48, -10 % at n = 32.
template <int NB, int NW, int FP, bool BWD>
__global__ void __launch_bounds__(64 * NW, (NB <= 3 && FP <= 4 ? 3 : 2)) gp_mfma_kernel(GpMfmaArgs a) {
    constexpr int NP = 16 * NB;              // padded problem size
    constexpr int LD = NP + 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* A = reinterpret_cast<float*>(smem_raw);        // [NP][LD]
    float* zf = A + NP * LD;                               // [NP][FP]
    float* rv = zf + NP * FP;                              // [NP] residual
    float* av = rv + NP;                                   // [NP] alpha
    float* invd = av + NP;                                 // [NP]
    float* scr = invd + NP;                                // [64] broadcast scratch / cross-wave sums

    // One wave per problem (NW == 1): LDS instructions of a wave execute in issue order, so cross-lane hand-offs
    // through LDS need no s_barrier (and no full lgkmcnt drain); only the compiler must not reorder them (the empty asm
    // memory clobber pins it; wavefront-scope fences would do too but cost a full counter drain each).
#define SYNC() do { if (NW > 1) __syncthreads(); else { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } } while (0)
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const long b = blockIdx.x;
    const int n = a.n, f = a.f;
    // (32-bit unsigned divisions: the 64-bit ones cost ~100 instructions each, three of them per problem)
    const int p = (int)(blockIdx.x % (unsigned)a.P);
    const long ty = blockIdx.x / (unsigned)a.y_div;
    int nv = a.n_valid ? a.n_valid[ty] : n;
    nv = nv < n ? nv : n; nv = nv < 0 ? 0 : nv;

#ifdef PACOH_GP_STAMPS
    unsigned int st_[12] = {};
    unsigned long long t_prev = __builtin_readcyclecounter();
#endif
    // features are kept as z * KAPPA / lengthscale, KAPPA^2 = log2(e) / 2: a kernel entry is exp2(-|dz|^2), one v_exp_f32 on the
    // squared distance (as in gp_reg.hip; the compensated exp was six instructions for each of the 3 n / 2 entries a thread visits)
    constexpr float KAPPA = 0.8493218002880191f, INV_KAPPA2 = 1.3862943611198906f;
    float kls[FP];
#pragma unroll
    for (int c = 0; c < FP; ++c) kls[c] = (c < f) ? KAPPA / a.ls[(long)p * f + c] : 1.0f;
    const float os = a.os ? a.os[p] : 1.0f;
    const float noise = a.noise[p];

    // ---- features (pre-divided by the lengthscale) and residual, thread i = row i -----------------
    const int i = tid;
    float zs[FP];
#pragma unroll
    for (int c = 0; c < FP; ++c) zs[c] = 0.0f;
    float ri = 0.0f;
    if (i < nv) {
        const float* zp = a.z + ((long)(blockIdx.x / (unsigned)a.z_div) * n + i) * (long)f;
#pragma unroll
        for (int c = 0; c < FP; ++c) if (c < f) zs[c] = zp[c] * kls[c];
        float mi = 0.0f;
        if (a.mean_mode == PACOH_MEAN_VECTOR) mi = a.mean[b * n + i];
        else if (a.mean_mode == PACOH_MEAN_CONST) mi = a.mean[p];
        ri = a.y[ty * n + i] - mi;
    }
    if (i < NP) {
#pragma unroll
        for (int c = 0; c < FP; ++c) zf[i * FP + c] = zs[c];
        rv[i] = ri;
    }
    SYNC();
    STAMP(0);                                  // loads

    // ---- Gram build + blocked Cholesky with the psd_safe_cholesky jitter ladder -----------------
    int my_info = -1;
    float jitter = 0.0f;
    for (int attempt = 0; attempt < 4; ++attempt) {
        if constexpr (NB == 4 && NW == 1) {
            // balanced build of the lower block triangle (10 blocks = 40 entries per lane): a lane of block row 0 / 1 also
            // fills the first 24 / 8 columns of row i+48 / i+16, so nobody computes the 64 entries of the longest rows
            // (the row-per-lane loop below runs as long as its busiest lane: 64 entries, 37 % of them idle on average)
            const int ib = i >> 4;
            const int nown = ib == 0 ? 4 : (ib == 1 ? 8 : 10);           // quads of the lane's own row
            const int hoff = ib == 0 ? 48 : (ib == 1 ? 16 : 0);          // helper row offset
            const int c0 = ib == 2 ? 8 : (ib == 3 ? 24 : 0);             // first own column
            if (nv == NP) {
                // every row is a real data point (the usual case): no validity masks, and the diagonal term is added afterwards by
                // one read-modify-write per lane instead of a test in each of the 40 entries (11 instead of ~20 instructions each)
#pragma unroll 2
                for (int qq = 0; qq < 10; ++qq) {
                    const bool help = qq >= nown;
                    const int row = help ? i + hoff : i;
                    const int col = help ? 4 * (qq - nown) : c0 + 4 * qq;
                    float zr[FP];
#pragma unroll
                    for (int c = 0; c < FP; ++c) zr[c] = zf[row * FP + c];
                    float kv[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        float s = 0.0f;
#pragma unroll
                        for (int c = 0; c < FP; ++c) { const float d = zr[c] - zf[(col + v) * FP + c]; s = fmaf(d, d, s); }
                        kv[v] = os * __builtin_amdgcn_exp2f(-s);
                    }
                    float4 o; o.x = kv[0]; o.y = kv[1]; o.z = kv[2]; o.w = kv[3];
                    *reinterpret_cast<float4*>(A + row * LD + col) = o;
                }
                asm volatile("" ::: "memory");                           // (own row: written by this lane just above, in-order LDS)
                A[i * LD + i] += noise + jitter;
            } else {
#pragma unroll 2
            for (int qq = 0; qq < 10; ++qq) {
                const bool help = qq >= nown;
                const int row = help ? i + hoff : i;
                const int col = help ? 4 * (qq - nown) : c0 + 4 * qq;
                float zr[FP];
#pragma unroll
                for (int c = 0; c < FP; ++c) zr[c] = zf[row * FP + c];
                float kv[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int j = col + v;
                    float s = 0.0f;
#pragma unroll
                    for (int c = 0; c < FP; ++c) { const float d = zr[c] - zf[j * FP + c]; s = fmaf(d, d, s); }
                    float k = os * __builtin_amdgcn_exp2f(-s);
                    if (!(row < nv && j < nv)) k = 0.0f;
                    if (row == j) k = (row < nv) ? k + noise + jitter : 1.0f;
                    kv[v] = k;
                }
                float4 o; o.x = kv[0]; o.y = kv[1]; o.z = kv[2]; o.w = kv[3];
                *reinterpret_cast<float4*>(A + row * LD + col) = o;
            }
            }
        } else if (i < NP) {
            const int ib = i >> 4;
            for (int jb = 0; jb <= ib; ++jb) {                 // lower block triangle only
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float kv[4];
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        const int j = jb * 16 + q * 4 + v;
                        float s = 0.0f;
#pragma unroll
                        for (int c = 0; c < FP; ++c) { const float d = zs[c] - zf[j * FP + c]; s = fmaf(d, d, s); }
                        float k = os * __builtin_amdgcn_exp2f(-s);
                        if (!(i < nv && j < nv)) k = 0.0f;
                        if (i == j) k = (i < nv) ? k + noise + jitter : 1.0f;
                        kv[v] = k;
                    }
                    float4 o; o.x = kv[0]; o.y = kv[1]; o.z = kv[2]; o.w = kv[3];
                    *reinterpret_cast<float4*>(A + i * LD + jb * 16 + q * 4) = o;
                }
            }
        }
        if (tid == 0) scr[40] = 0.0f;
        SYNC();
        STAMP(1);                              // Gram build
#pragma unroll
        for (int kb = 0; kb < NB; ++kb) {
            const int k0 = kb * 16;
            factor_diag_block<NW>(A, LD, k0, invd, scr, r, g, wave == 0);      // wave 0 works, the others join its barriers
            SYNC();
            STAMP(2);                          // diagonal blocks
            // panel: L[ib][kb] = A[ib][kb] * Linv^T   (block rows dealt to the waves)
#pragma unroll
            for (int ib = kb + 1; ib < NB; ++ib) {
                if ((ib - kb - 1) % NW == wave) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    mm_xyT(acc, A, LD, ib * 16, k0, k0, k0, r, g, 1.0f);
                    store_c(A, LD, ib * 16, k0, r, g, acc);   // same wave read it just before: in-order LDS
                }
            }
            SYNC();
            // trailing update: A[ib][jb] -= L[ib][kb] L[jb][kb]^T
            int q = 0;
#pragma unroll
            for (int ib = kb + 1; ib < NB; ++ib) {
#pragma unroll
                for (int jb = kb + 1; jb <= ib; ++jb, ++q) {
                    if (q % NW == wave) {
                        f32x4 acc = load_c(A, LD, ib * 16, jb * 16, r, g);
                        mm_xyT(acc, A, LD, ib * 16, k0, jb * 16, k0, r, g, -1.0f);
                        store_c(A, LD, ib * 16, jb * 16, r, g, acc);
                    }
                }
            }
            SYNC();
            STAMP(3);                          // panel + trailing update
        }
        const bool ok = scr[40] == 0.0f;
        SYNC();                                                   // flag read by all before the next attempt clears it
        if (ok) { my_info = attempt; break; }
        jitter = 1e-6f;
        for (int q = 0; q < attempt; ++q) jitter *= 10.0f;
    }
    const bool okf = my_info >= 0;
    if (tid == 0 && a.info) a.info[b] = my_info;

    // ---- Z = L^-1: block column jb (dealt to the waves), written transposed into the upper triangle ----
#pragma unroll
    for (int jb = 0; jb < NB - 1; ++jb) {
        if (jb % NW == wave) {
#pragma unroll
            for (int ib = jb + 1; ib < NB; ++ib) {
                f32x4 S = {0.f, 0.f, 0.f, 0.f};
                mm_xy(S, A, LD, ib * 16, jb * 16, jb * 16, jb * 16, r, g);               // L[ib][jb] * Linv_jb
#pragma unroll
                for (int kb = jb + 1; kb < ib; ++kb) {                                    // L[ib][kb] * Z[kb][jb]
                    const float4 aw = *reinterpret_cast<const float4*>(A + (ib * 16 + r) * LD + kb * 16 + 4 * g);
                    const float4 bz = *reinterpret_cast<const float4*>(A + (jb * 16 + r) * LD + kb * 16 + 4 * g);
                    S = mfma4(aw.x, bz.x, S); S = mfma4(aw.y, bz.y, S); S = mfma4(aw.z, bz.z, S); S = mfma4(aw.w, bz.w, S);
                }
                f32x4 Zb = {0.f, 0.f, 0.f, 0.f};
                mm_xs(Zb, A, LD, ib * 16, ib * 16, S, r, g, -1.0f);                       // -Linv_ib * S
                store_ct(A, LD, jb * 16, ib * 16, r, g, Zb);
            }
        }
    }
    SYNC();
    STAMP(4);                                  // Z = L^-1
    // ---- u = Z r (thread i = row i of Z = column i of the upper storage + its diagonal-block row) -------
    float ui = 0.0f;
    if (i < NP) {
        const int d0 = (i >> 4) * 16;
        float u1 = 0.0f, u2 = 0.0f, u3 = 0.0f;                      // four FMA chains
        for (int j = 0; j < d0; j += 4) {
            const float4 rq = *reinterpret_cast<const float4*>(rv + j);
            ui = fmaf(A[j * LD + i], rq.x, ui); u1 = fmaf(A[(j + 1) * LD + i], rq.y, u1);
            u2 = fmaf(A[(j + 2) * LD + i], rq.z, u2); u3 = fmaf(A[(j + 3) * LD + i], rq.w, u3);
        }
        const float4* zr = reinterpret_cast<const float4*>(A + i * LD + d0);
        const float4* rr = reinterpret_cast<const float4*>(rv + d0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 zq = zr[v], rq = rr[v];
            ui = fmaf(zq.x, rq.x, ui); u1 = fmaf(zq.y, rq.y, u1); u2 = fmaf(zq.z, rq.z, u2); u3 = fmaf(zq.w, rq.w, u3);
        }
        ui = (ui + u1) + (u2 + u3);
    }
    const float quad = block_sum<NW>((i < nv) ? ui * ui : 0.0f, scr);
    const float logdet = block_sum<NW>((i < nv) ? -logf(invd[i < NP ? i : 0]) : 0.0f, scr);
    float lml = nv > 0 ? -0.5f * (quad + 2.0f * logdet + (float)nv * 1.8378770664093453f) / (float)nv : 0.0f;
    if (!okf) lml = NAN;
    if (tid == 0) a.lml[b] = lml;
    STAMP(5);                                  // u, quadratic form, log-det, lml
    if (!BWD) return;

    // ---- W = Z^T Z: block rows dealt to the waves, results overwrite the (dead) lower triangle --------
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
        if (ib % NW == wave) {
            f32x4 Wb[NB];
#pragma unroll
            for (int jb = 0; jb <= ib; ++jb) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = ib + 1; kb < NB; ++kb) mm_xyT(acc, A, LD, ib * 16, kb * 16, jb * 16, kb * 16, r, g, 1.0f);
                // kb == ib term: Linv_ib^T * Z[ib][jb]
                const float* xp = A + (ib * 16 + 4 * g) * LD + ib * 16 + r;
                const float a0 = xp[0], a1 = xp[LD], a2 = xp[2 * LD], a3 = xp[3 * LD];
                if (jb < ib) {
                    const float4 bz = *reinterpret_cast<const float4*>(A + (jb * 16 + r) * LD + ib * 16 + 4 * g);
                    acc = mfma4(a0, bz.x, acc); acc = mfma4(a1, bz.y, acc); acc = mfma4(a2, bz.z, acc); acc = mfma4(a3, bz.w, acc);
                } else {
                    acc = mfma4(a0, a0, acc); acc = mfma4(a1, a1, acc); acc = mfma4(a2, a2, acc); acc = mfma4(a3, a3, acc);
                }
                Wb[jb] = acc;
            }
#pragma unroll
            for (int jb = 0; jb <= ib; ++jb) store_c(A, LD, ib * 16, jb * 16, r, g, Wb[jb]);
        }
    }
    SYNC();
    {   // mirror the strictly-lower blocks into the upper triangle (Z^T is dead now)
        int q = 0;
#pragma unroll
        for (int ib = 1; ib < NB; ++ib) {
#pragma unroll
            for (int jb = 0; jb < ib; ++jb, ++q) {
                if (q % NW == wave) { const f32x4 c = load_c(A, LD, ib * 16, jb * 16, r, g); store_ct(A, LD, jb * 16, ib * 16, r, g, c); }
            }
        }
    }
    SYNC();
    STAMP(6);                                  // W = Z^T Z + mirror
    // ---- alpha = W r ------------------------------------------------------------------------------
    float ai = 0.0f;
    if (i < NP) {
        const float4* wr = reinterpret_cast<const float4*>(A + i * LD);
        const float4* rr = reinterpret_cast<const float4*>(rv);
        float a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll 4
        for (int v = 0; v < NP / 4; ++v) {
            const float4 wq = wr[v], rq = rr[v];
            ai = fmaf(wq.x, rq.x, ai); a1 = fmaf(wq.y, rq.y, a1); a2 = fmaf(wq.z, rq.z, a2); a3 = fmaf(wq.w, rq.w, a3);
        }
        ai = (ai + a1) + (a2 + a3);
        av[i] = ai;
    }
    SYNC();
    STAMP(7);                                  // alpha
    // ---- gradient sums: thread i owns row i of W ---------------------------------------------------
    const float gup = a.g_lml ? a.g_lml[b] : 1.0f;
    float dz[FP], dls[FP];
#pragma unroll
    for (int c = 0; c < FP; ++c) { dz[c] = 0.0f; dls[c] = 0.0f; }
    float msum = 0.0f, dnz = 0.0f;                        // sum of M_ij = os x (d lml / d os)
    const float inv2n = nv > 0 ? 0.5f / (float)nv : 0.0f;
    const float osn = inv2n * os;                         // the outputscale rides on the 1/(2n) factor
    if (i < nv) {
        // four columns per iteration from 16-byte LDS reads (four independent exp chains in flight).  The loop runs over the
        // padded size: for j >= nv both alpha_j and W_ij are exactly 0 (identity block), so those columns contribute nothing.
        const float* wrow = A + i * LD;
        dnz = (ai * ai - wrow[i]) * inv2n;
#pragma unroll 2
        for (int j4 = 0; j4 < NP; j4 += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(wrow + j4);
            const float4 a4 = *reinterpret_cast<const float4*>(av + j4);
            const float wv[4] = {w4.x, w4.y, w4.z, w4.w}, avv[4] = {a4.x, a4.y, a4.z, a4.w};
            float zj[4][FP];
            if constexpr (FP % 4 == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int c4 = 0; c4 < FP; c4 += 4) {
                        const float4 q = *reinterpret_cast<const float4*>(zf + (j4 + u) * FP + c4);
                        zj[u][c4] = q.x; zj[u][c4 + 1] = q.y; zj[u][c4 + 2] = q.z; zj[u][c4 + 3] = q.w;
                    }
            } else {                                  // FP == 2: two rows per 16-byte read
#pragma unroll
                for (int u = 0; u < 4; u += 2) {
                    const float4 q = *reinterpret_cast<const float4*>(zf + (j4 + u) * FP);
                    zj[u][0] = q.x; zj[u][1] = q.y; zj[u + 1][0] = q.z; zj[u + 1][1] = q.w;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float Gij = (ai * avv[u] - wv[u]) * osn;
                float s = 0.0f, df[FP];
#pragma unroll
                for (int c = 0; c < FP; ++c) { df[c] = zj[u][c] - zs[c]; s = fmaf(df[c], df[c], s); }
                const float M = Gij * __builtin_amdgcn_exp2f(-s);
                msum += M;
#pragma unroll
                for (int c = 0; c < FP; ++c) { const float md = M * df[c]; dz[c] += md; dls[c] = fmaf(md, df[c], dls[c]); }
            }
        }
    }
    STAMP(8);                                  // gradient loop
    const float bad = okf ? 0.0f : NAN;
    if (a.d_z && i < n) {
        for (int c = 0; c < f; ++c) a.d_z[(b * n + i) * (long)f + c] = (i < nv) ? (2.0f * INV_KAPPA2) * gup * dz[c] * kls[c] + bad : 0.0f;
    }
    if (a.mean_mode == PACOH_MEAN_VECTOR) {
        if (a.d_mean && i < n) a.d_mean[b * n + i] = (i < nv) ? gup * ai / (float)nv + bad : 0.0f;
    } else if (a.mean_mode == PACOH_MEAN_CONST) {
        const float sa = block_sum<NW>((i < nv) ? ai : 0.0f, scr);
        if (a.d_mean && tid == 0) a.d_mean[b] = nv > 0 ? gup * sa / (float)nv + bad : 0.0f;
    }
#pragma unroll
    for (int c = 0; c < FP; ++c) {
        if (c < f) {
            const float sc = block_sum<NW>(dls[c], scr);
            if (tid == 0) a.d_ls[b * f + c] = (INV_KAPPA2 / KAPPA) * gup * sc * kls[c] + bad;
        }
    }
    const float sdos = block_sum<NW>(msum, scr) / os, sdnz = block_sum<NW>(dnz, scr);
    if (tid == 0) {
        if (a.d_os) a.d_os[b] = gup * sdos + bad;
        a.d_noise[b] = gup * sdnz + bad;
    }
    STAMP(9);                                  // reductions + stores
#ifdef PACOH_GP_STAMPS
    if (lane == 0) {
        unsigned int* dst = g_gp_stamps[blockIdx.x % STAMP_SLOTS];
#pragma unroll
        for (int k = 0; k < 12; ++k) dst[k] += st_[k];
    }
#endif
}

#undef SYNC

template <int NB, int NW, bool BWD>
static int launch_nb_nw(const GpMfmaArgs& a, int FP, hipStream_t s) {
    constexpr int NP = 16 * NB, LD = NP + 4;
    size_t lds = (size_t)(NP * LD + NP * FP + 3 * NP + 64) * sizeof(float);
#define PACOH_GPM_CASE(fp) case fp: { auto kern = gp_mfma_kernel<NB, NW, fp, BWD>; \
        if (lds > 64u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PACOH_ELIMIT; \
        hipLaunchKernelGGL(kern, dim3((unsigned)a.B), dim3(64 * NW), lds, s, a); } break;
    switch (FP) { PACOH_GPM_CASE(2) PACOH_GPM_CASE(4) PACOH_GPM_CASE(8) default: PACOH_GPM_CASE(16) }
#undef PACOH_GPM_CASE
    return launch_status();
}

template <int NB, bool BWD>
static int launch_nb(const GpMfmaArgs& a, int FP, hipStream_t s) {
    return launch_nb_nw<NB, (NB <= 4 ? 1 : 2), BWD>(a, FP, s);
}

#ifdef PACOH_GP_STAMPS
extern "C" int pacoh_debug_gp_stamps(unsigned long long* out16, int reset) {
    static unsigned int host[STAMP_SLOTS][12];
    if (out16) {
        if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_gp_stamps), sizeof(host)) != hipSuccess) return -1;
        for (int k = 0; k < 16; ++k) out16[k] = 0;
        for (int q = 0; q < STAMP_SLOTS; ++q) for (int k = 0; k < 12; ++k) out16[k] += host[q][k];
    }
    if (reset) { void* p = nullptr; if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_gp_stamps)) != hipSuccess || hipMemset(p, 0, sizeof(host)) != hipSuccess) return -1; }
    return 0;
}
#endif

// entry used by gp_small.hip's C-ABI functions; returns 1 if this path does not apply
int gp_mfma_try(const GpMfmaArgs& a, bool bwd, hipStream_t s) {
    if (a.n > 128 || a.f > PACOH_MAX_FEATURES) return 1;
    const int NB = (a.n + 15) / 16;
    const int FP = a.f <= 2 ? 2 : (a.f <= 4 ? 4 : (a.f <= 8 ? 8 : 16));
#define PACOH_GPM_NB(nb) case nb: return bwd ? launch_nb<nb, true>(a, FP, s) : launch_nb<nb, false>(a, FP, s);
    switch (NB) { PACOH_GPM_NB(1) PACOH_GPM_NB(2) PACOH_GPM_NB(3) PACOH_GPM_NB(4) PACOH_GPM_NB(5) PACOH_GPM_NB(6) PACOH_GPM_NB(7) default: PACOH_GPM_NB(8) }
#undef PACOH_GPM_NB
}

}  // namespace pacoh
