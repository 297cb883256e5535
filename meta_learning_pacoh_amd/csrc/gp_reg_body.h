// Register-resident fused task-GP kernel, fp32, n <= 128, f <= 4: one 64-lane wavefront per (task, particle) problem, and the
// n x n matrix never leaves the VECTOR REGISTERS: every 16x16 block is held in the v_mfma_f32_16x16x4_f32 accumulator layout
// (lane (r = l&15, g = l>>4), register s  <->  X[4g+s][r]; four registers per block, the upper block triangle = 40 registers).
// LDS holds the vectors (features, residual, alpha) and, between the matrix-core phase and the gradient loop, the six strictly
// upper blocks of K^-1: 7.7 KB per problem; with 122 registers that is four waves per SIMD.
//
// fp32 MFMAs and vector instructions share one issue budget on gfx950 (tools/mfma_valu_overlap.hip, tools/valu_rates.hip): the
// kernel takes 32 cycles x MFMAs + the sum of its vector instructions, so everything below is about doing the work in few of both.
//
// One primitive does the O(n^3) work.  With the k index of a 16x16x16 product permuted as k = 4g+s, an accumulator-layout
// block is directly an MFMA operand: as B it stands for itself, as A for its TRANSPOSE.  So  mmT(X, Y) = X^T Y  maps two
// register blocks to a register block (4 MFMAs, no memory traffic), and a block is transposed by mmT(X, I).  In terms of the
// upper factor R = L^T (K = R^T R); the not-yet-eliminated blocks are kept NEGATED so that no operand needs negating:
//   Gram build        U[I][J] = -(os k(z_i, z_j) + (noise, jitter on the diagonal)), I <= J, straight into accumulator layout;
//                     features pre-scaled so that k = exp2(-|dz|^2): one v_exp_f32 per entry; padding rows = far-away points
//   Cholesky          diagonal block: factor16() (4x4 pivot blocks, see below) -> Z_K = L_KK^-1;  V_K = -Z_K^T = mmT(Z_K, -I)
//                     panel R[K][J] = L_KK^-1 A[K][J] = mmT(V_K, U[K][J]);  trailing U[I][J] += mmT(R[K][I], R[K][J])
//   u = L^-1 r        t = -r_K + sum_m R[m][K]^T u_m on the vector units (mvT_, four fmas per block), u_K = mmT(V_K, t)
//   L^-1 (backward)   G[I][J] = -L_II^-1 sum_m L[I][m] G[m][J] = mmT(V_I, sum_m mmT(R[m][I], G[m][J]))
//   W = K^-1          W[I][J] = sum_m G[m][I]^T G[m][J];   alpha = L^-T u = sum_I G[I][K]^T u_I on the vector units
//   gradient sums     every ordered pair (i, j), column block by column block: what is destined for point j accumulates in the
//                     lane and needs two lane exchanges per block; diagonal blocks of W are consumed where they are produced
// factor16(): the 16x16 diagonal block is eliminated four columns at a time -- the 4x4 pivot block reaches every lane by ten
// v_readlane broadcasts, every lane runs the 4x4 Cholesky in its own registers (rsq -> mul -> fma per pivot), solves its row of
// the 16x4 panel, the rank-4 trailing update is one MFMA (A operand == B operand), and L_KK^-1 is built alongside by block
// forward substitution, its right-hand side -E + L Z likewise by one rank-4 MFMA per step.
//
// Same arithmetic as gp_mfma.hip / gp_small.hip; reference lines replaced: random_gp.py:54-89, GPR_meta_mll.py:104-117
// (ExactMarginalLogLikelihood + autograd through gpytorch).
#pragma once
#include "common.h"

namespace pacoh {

struct GpMfmaArgs {            // (same struct as in gp_small.hip / gp_mfma.hip)
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

// the test side of a posterior-predictive call (gp_reg_predict_kernel): test inputs [B / zt_div, m, f], the prior mean at them
// (mean_mode of GpMfmaArgs: [B, m] or [P]), outputs mu / var [B, m]
struct GpPredArgs {
    const float* zt; int zt_div;
    const float* mean_tst;
    float* mu; float* var;
    int m;
    float* V_out;              // optional [B, m, n]: V = L^-1 K_xs, test point major (what gp_predict_cov_kernel builds the covariance from)
};

namespace gpreg {


using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ f32x4 mfma_(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// acc + X^T Y  /  acc - X^T Y   (X, Y, acc: 16x16 blocks in accumulator layout)
__device__ __forceinline__ f32x4 mmT(const f32x4& X, const f32x4& Y, f32x4 acc) {
    acc = mfma_(X[0], Y[0], acc); acc = mfma_(X[1], Y[1], acc); acc = mfma_(X[2], Y[2], acc); acc = mfma_(X[3], Y[3], acc);
    return acc;
}
__device__ __forceinline__ f32x4 mmT_neg(const f32x4& X, const f32x4& Y, f32x4 acc) {
    acc = mfma_(-X[0], Y[0], acc); acc = mfma_(-X[1], Y[1], acc); acc = mfma_(-X[2], Y[2], acc); acc = mfma_(-X[3], Y[3], acc);
    return acc;
}

__device__ __forceinline__ float readlane_(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add_(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF));
}
// sum over the 16 lanes of the own lane row, in every lane of the row
__device__ __forceinline__ float row_sum_(float v) {
    v = dpp_add_<0xB1, 0xF>(v);       // quad_perm [1,0,3,2]
    v = dpp_add_<0x4E, 0xF>(v);       // quad_perm [2,3,0,1]
    v = dpp_add_<0x141, 0xF>(v);      // row_half_mirror
    v = dpp_add_<0x140, 0xF>(v);      // row_mirror
    return v;
}
// sum over the four lane rows (lanes of equal r), in every lane
__device__ __forceinline__ float xg_sum_(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// this lane's share of (X^T v)[r] = sum_{g,s} X[4g+s][r] v[4g+s]  (X in accumulator layout, v replicated: register s of lane
// (r, g) = v[4g+s]); xg_sum_() of it is the product, in "column layout" (lane (r, .) holds entry r).  A matrix-vector product
// on the matrix cores costs a full 16x16x16 block product (128 issue cycles); this is four fmas (+ 1/4 of the row exchange)
__device__ __forceinline__ float mvT_(const f32x4& X, const f32x4& v, float acc) {
    acc = fmaf(X[0], v[0], acc); acc = fmaf(X[1], v[1], acc); acc = fmaf(X[2], v[2], acc); acc = fmaf(X[3], v[3], acc);
    return acc;
}
// sum over the 64 lanes, as a wave-uniform value
__device__ __forceinline__ float wave_sum_(float v) {
    v = row_sum_(v);
    v = dpp_add_<0x142, 0xA>(v);      // row_bcast15 into rows 1 and 3
    v = dpp_add_<0x143, 0xC>(v);      // row_bcast31 into rows 2 and 3: lane 63 holds the total
    return readlane_(v, 63);
}

// Cholesky of the 16x16 block C = -Cn (accumulator layout, symmetric; handed over NEGATED, as the kernel stores the matrix) and
// the inverse of its factor: on exit Z = L^-1 (accumulator layout, zeros above the diagonal).  nId = -identity.  The diagonal of L
// is multiplied into `dprod` as its RECIPROCAL, read off the finished inverse (1 / L[r][r] = Z[r][r]: register r & 3 of lane
// (r, r >> 2); the other lanes multiply by 1) once per block -- round 3; rounds 1-2 picked L[r][r] out of the panel row with three
// selects under exec-mask branches in each of the four elimination steps.  A pivot that is not positive turns its lane's product
// into NaN (rsq), a zero pivot into infinity, which is also how the caller notices the failure.
// What is NOT computed: the entries of L above the 4-column panel being eliminated are left as they fall out of the
// substitution (garbage): they only ever produce rows of L Z that have been consumed already.
// fs: per-wave LDS (GPR_FSC floats, see below).  The four registers of the lane row g == k (rows 4k..4k+3 of the block, one column per lane) have
// to reach all four lane rows twice per step (the panel rows of C, then the fresh rows of Z).  As four ds_bpermute each that is
// 8 x 24 issue cycles per step (tools/valu_rates.hip); as one 16-lane ds_write_b128 + one ds_read_b128 (the four lanes of equal r
// read one address: a broadcast) it is 2 x (13 + 4) -- and the pivot block comes out of the same 256 bytes by four uniform
// ds_read_b128 instead of ten v_readlane (4.7 cycles each, and their scalar results make every instruction of the 4x4 Cholesky
// a scalar-operand instruction: 4.7 instead of 3.2 cycles).
// SKIP (round 5; n <= 32, where the last block is often mostly padding -- the 5-point tasks of cfg #1 pay for a 16 x 16 block; in the
// 6- / 8-block kernels, which also serve 5 and 7 blocks, the branches cost 12 registers and the n = 128 kernel its second wave per
// SIMD): `rows` = rows of this block inside the problem (wave-uniform).  A step
// whose four columns are all padding is not run: the padding rows are rows of the identity, exactly (their off-diagonal kernel
// entries are exp2(-1e20) = 0), so the step would produce x = 0, Z rows = identity rows and a pivot of 1 -- which is what is set.
// Round 6 (groups 1 and 2 of profiles/r05_gp_reg_isa_histogram.txt), -DPACOH_F16_BRANCHFREE=1: the two stores of a step
// are `if (g == k)` regions (s_and_saveexec / s_cbranch_execz / s_or per region, 32 regions per block); in the branch-free form EVERY
// lane row stores its registers to a slot of its own (fs + 64 g, fs + 256 + 64 g) and the readers address lane row k's slot -- no
// exec-mask code around a store -- and the Z entry a lane needs for the rank-4 update is read as ONE float (fs[.. + 4 r + g]) instead
// of a 16-byte read and three selects.  Same bits; measured on the cfg #3 step, same box, two runs each
// (profiles/r06_gp_factor16_ab.txt): 0.1449 / 0.1461 ms against 0.1452 / 0.1466 -- nothing, like round 5's permlane swap: the kernel
// does not respond to a 3 % instruction diet.  It does respond to occupancy: the same build with the slots in a scratch array of
// their own (10.75 KB of LDS per problem: 14 problems per CU instead of 16) ran 0.157 ms, +11 %.  Not the default.
// GPR_FSC: floats of per-wave LDS factor16() uses; GPR_SCR: the scratch it shares with the 16x16 transpose (tsc, 320 floats: one
// wave, in-order LDS, and no transpose is in flight across a factor16() call).
#ifdef PACOH_F16_BRANCHFREE
constexpr int GPR_FSC = 512;
#else
constexpr int GPR_FSC = 128;
#endif
constexpr int GPR_SCR = GPR_FSC > 320 ? GPR_FSC : 320;
template <bool SKIP = false>
__device__ __forceinline__ void factor16(f32x4 Cn, f32x4& Z, float& dprod, const f32x4& nId, int r, int g, float* fs, int rows = 16) {
    f32x4 Tn = nId;                                         // -E + L Z, built up by one rank-4 MFMA per step (see below)
    Z = f32x4{0.f, 0.f, 0.f, 0.f};
    const int rows_u = SKIP ? __builtin_amdgcn_readfirstlane(rows) : 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (SKIP && 4 * k >= rows_u) {
            if (g == k) { Z[0] = -nId[0]; Z[1] = -nId[1]; Z[2] = -nId[2]; Z[3] = -nId[3]; }
            continue;
        }
        // pivot block P[c][j] = C[4k+c][4k+j] = register c of lane (r = 4k+j, g = k): wave-uniform
        const float c0 = Cn[0], c1 = Cn[1], c2 = Cn[2], c3 = Cn[3];
        const int src = (16 * k + r) * 4;
#ifndef PACOH_F16_BRANCHFREE
        if (g == k) *reinterpret_cast<f32x4*>(fs + 4 * r) = Cn;
        const float* fk = fs;
#else
        *reinterpret_cast<f32x4*>(fs + 64 * g + 4 * r) = Cn;
        const float* fk = fs + 64 * k;
#endif
        asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier();
        const f32x4 rtv = *reinterpret_cast<const f32x4*>(fk + 4 * r);
        const float rt0 = -rtv[0], rt1 = -rtv[1], rt2 = -rtv[2], rt3 = -rtv[3];
        const f32x4 pc0 = *reinterpret_cast<const f32x4*>(fk + 16 * k), pc1 = *reinterpret_cast<const f32x4*>(fk + 16 * k + 4);
        const f32x4 pc2 = *reinterpret_cast<const f32x4*>(fk + 16 * k + 8), pc3 = *reinterpret_cast<const f32x4*>(fk + 16 * k + 12);
        const float p00 = -pc0[0], p10 = -pc0[1], p20 = -pc0[2], p30 = -pc0[3];
        const float p11 = -pc1[1], p21 = -pc1[2], p31 = -pc1[3];
        const float p22 = -pc2[2], p32 = -pc2[3], p33 = -pc3[3];
        asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier();
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
        // this lane's row of the panel: X Lp^T = C[:, 4k..4k+3] by forward substitution = L[r][4k..4k+3]
        const float x0 = rt0 * r0;
        const float x1 = fmaf(-x0, l10, rt1) * r1;
        const float x2 = fmaf(-x1, l21, fmaf(-x0, l20, rt2)) * r2;
        const float x3 = fmaf(-x2, l32, fmaf(-x1, l31, fmaf(-x0, l30, rt3))) * r3;
        const float xg = g == 0 ? x0 : (g == 1 ? x1 : (g == 2 ? x2 : x3));      // L[r][4k+g]
        if (k < 3) {                                         // Cn[i][j] += sum_c X[i][c] X[j][c] for i, j >= 4k+4: one MFMA
            const float am = r >= 4 * k + 4 ? xg : 0.0f;
            Cn = mfma_(am, am, Cn);
        }
        // rows 4k..4k+3 of L^-1: Lp Z_k = E_k - (L Z)[k-th block row]; the lanes g == k hold that block row of Tn = -E + L Z
        if (g == k) {
            const float z0 = -Tn[0] * r0;
            const float z1 = fmaf(-z0, l10, -Tn[1]) * r1;
            const float z2 = fmaf(-z1, l21, fmaf(-z0, l20, -Tn[2])) * r2;
            const float z3 = fmaf(-z2, l32, fmaf(-z1, l31, fmaf(-z0, l30, -Tn[3]))) * r3;
            Z[0] = z0; Z[1] = z1; Z[2] = z2; Z[3] = z3;
        }
        if (k < 3) {
            // Tn += L[:, 4k..4k+3] Z[4k..4k+3, :]: ONE MFMA (k index = the four new columns) -- A[i][kk] = L[i][4k+kk] is xg of lane
            // (i, kk); B[kk][j] = Z[4k+kk][j] lives in register kk of lane (j, k) and reaches lane (j, kk) by four lane reads.  (Forming
            // the block row of L Z as a full product with L^T kept in registers took four MFMAs per step for a 4-row result.)
#ifndef PACOH_F16_BRANCHFREE
            if (g == k) *reinterpret_cast<f32x4*>(fs + 64 + 4 * r) = Z;
            asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier();
            const f32x4 wv = *reinterpret_cast<const f32x4*>(fs + 64 + 4 * r);
            asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier();
            const float w0 = wv[0], w1 = wv[1], w2 = wv[2], w3 = wv[3];
            const float zb = g == 0 ? w0 : (g == 1 ? w1 : (g == 2 ? w2 : w3));
#else
            *reinterpret_cast<f32x4*>(fs + 256 + 64 * g + 4 * r) = Z;      // (lane rows g != k store what they hold: nobody reads it)
            asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier();
            const float zb = fs[256 + 64 * k + 4 * r + g];
            asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier();
#endif
            Tn = mfma_(xg, zb, Tn);
        }
    }
    // 1 / L[r][r] = Z[r][r] sits in register r & 3 of lane (r, r >> 2)
    const int c = r & 3;
    const float zd = c == 0 ? Z[0] : (c == 1 ? Z[1] : (c == 2 ? Z[2] : Z[3]));
    if (g == (r >> 2)) dprod *= zd;
}

// The standalone kernel's context: one problem per workgroup; a kernel argument is read where it is needed.  The compiler loads the
// whole argument struct into scalar registers at kernel entry (one s_load_dwordx16 tuple among others) and, this kernel being
// short of scalar registers, spills it and reloads all sixteen words at every use of one of them: ~230 v_writelane / v_readlane
// per problem.  The pointers only needed for the final stores are fetched from the kernel-argument segment at that point
// instead (a scalar-cache hit).
struct KernelCtx {
    __device__ __forceinline__ int lane() const { return threadIdx.x; }
    __device__ __forceinline__ unsigned block() const { return blockIdx.x; }
    template <typename T>
    __device__ __forceinline__ T late(T, unsigned offset) const {
        typedef const char __attribute__((address_space(4))) * kptr_t;
        kptr_t kp = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        return *(const volatile T __attribute__((address_space(4)))*)(kp + offset);
    }
};
// A wave of a larger workgroup working on problem `b` of an argument block held by the caller (map_persist.hip)
struct WaveCtx {
    unsigned b;
    __device__ __forceinline__ int lane() const { return threadIdx.x & 63; }
    __device__ __forceinline__ unsigned block() const { return b; }
    template <typename T>
    __device__ __forceinline__ T late(T value, unsigned) const { return value; }
};

// 1 / x on the hardware reciprocal plus one Newton step (3 instructions, <= 1 ulp) instead of the IEEE division sequence (11): round 5's
// part of the instruction diet -- six divisions and one logf per problem were ~85 of the kernel's ~2 200 vector instructions
__device__ __forceinline__ float rcp_(float x) { const float r = __builtin_amdgcn_rcpf(x); return r * fmaf(-x, r, 2.0f); }

__host__ __device__ constexpr int uidx(int NB, int K, int J) { return K * NB - K * (K - 1) / 2 + (J - K); }   // upper block (K <= J)


// HAS_OS = false: the caller has no outputscale (SVGD / VI: SEKernelLight, models.py:418-446) -- os == 1 at compile time: the 40
// multiplies of the Gram build and the 64 additions of the gradient loop that only feed d lml / d outputscale are not compiled in
// (instantiated for the shapes of cfg #3 / #4 only: every instantiation costs build time)
//
// The body of the kernel as a device function of ONE wavefront (round 5), so that two callers share it: gp_reg_kernel (gp_reg.hip:
// one problem per 64-thread workgroup, arguments read late from the kernel-argument segment) and the persistent PACOH-MAP
// iteration kernel (map_persist.hip: one wave per task of the batch inside a 1024-thread workgroup, every operand in LDS).
// Ctx: lane() = lane of the wave, block() = problem index b, late<T>(a.field, offset of it) = that field, read late.
// zf [16 NB FP], rv / av [16 NB], fsc / tsc [GPR_SCR, shared], dzc [BWD ? 16 NB FP : 1], Wl [BWD && NB > 1 ? (NU - NB) 256 : 4]: per-wave
// LDS scratch (16-byte aligned).
// PRED (round 5): the posterior predictive instead of the gradients -- mu_s = m_s + k_s^T alpha, var_s = os + noise - |L^-1 k_s|^2 for
// the m test points of *pa, 16 at a time: the kernel entries K_xs of a test block against every context block (accumulator layout:
// context rows x test columns), V = L^-1 K_xs as block products with the blocks of L^-1 the backward instantiation keeps in registers,
// the contractions with K_xs / V on the vector units.  The LDS-resident general kernel it replaces for these shapes ran 20 480
// problems of n = m = 64 in 1.85 ms; the LML + gradient kernel takes 0.14 for the same batch.
template <int NB, int FP, bool BWD, bool HAS_OS, class Ctx, bool PRED = false>
__device__ __forceinline__ void gp_reg_body(const GpMfmaArgs& a, const Ctx& cx, float* zf, float* rv,
                                            float* av, float* fsc, float* tsc,
                                            float* dzc, float* Wl, const GpPredArgs* pa = nullptr) {
    static_assert(!PRED || BWD, "the predictive runs on the backward instantiation (it needs the blocks of L^-1)");
    constexpr int NP = 16 * NB;
    constexpr int NU = NB * (NB + 1) / 2;
#define GPR_LATE(field) cx.template late<decltype(GpMfmaArgs::field)>(a.field, (unsigned)__builtin_offsetof(GpMfmaArgs, field))
#define WSYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)   // in-order LDS within one wave
    // The kernel is one long unrolled instruction stream of mutually independent block computations; left alone, the scheduler
    // interleaves dozens of them (40 exp chains of the Gram build at once) and the register file overflows.  Fences between the
    // blocks keep the live set at what the algorithm needs.
#define SCHED_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    const int lane = cx.lane();
    const int r = lane & 15, g = lane >> 4;
    const unsigned blk = cx.block();
    const long b = blk;
    const int n = a.n, f = a.f;
    const int p = (int)(blk % (unsigned)a.P);
    const long ty = blk / (unsigned)a.y_div;
    int nv = a.n_valid ? a.n_valid[ty] : n;
    nv = nv < n ? nv : n; nv = nv < 0 ? 0 : nv;

    // The features are kept as z * KAPPA / lengthscale, KAPPA^2 = log2(e) / 2: a kernel entry is then exp2(-|dz|^2), ONE
    // instruction on top of the squared distance (104 entries per lane pass through it: 40 of the upper block triangle in the
    // Gram build, 64 in the gradient loop), and the constant comes back out in the chain-rule factors at the very end.
    constexpr float KAPPA = 0.8493218002880191f, INV_KAPPA2 = 1.3862943611198906f;
    float kls[FP];                                            // KAPPA / lengthscale
#pragma unroll
    for (int c = 0; c < FP; ++c) kls[c] = (c < f) ? KAPPA * rcp_(a.ls[(long)p * f + c]) : 1.0f;
    const float os = HAS_OS ? (a.os ? a.os[p] : 1.0f) : 1.0f;
    const float noise = a.noise[p];

    // ---- features (pre-divided by the lengthscale) and residual, lane l = rows l, l + 64, ... --------------------------------
    constexpr int RPL = (NP + 63) / 64;                       // rows per lane (1 up to n = 64)
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
        const int i = lane + 64 * rr;
        float zs[FP];
#pragma unroll
        for (int c = 0; c < FP; ++c) zs[c] = 0.0f;
        float ri = 0.0f;
        // Padding rows (nv <= i < 16 NB) must come out as rows of the identity.  They are placed far away from every other point --
        // each at its own distance, so that exp2(-|dz|^2) is exactly 0 against anything else -- instead of masking 104 entries per
        // lane with compares the compiler hoists out of the retry loop into scalar registers it does not have.
        zs[0] = 1e10f * (float)(i + 1);
        if (i < nv) {
            zs[0] = 0.0f;
            const float* zp = a.z + ((long)(blk / (unsigned)a.z_div) * n + i) * (long)f;
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
    }
    WSYNC();

    f32x4 nId;                                                // -identity block in accumulator layout
#pragma unroll
    for (int s = 0; s < 4; ++s) nId[s] = (4 * g + s == r) ? -1.0f : 0.0f;

    // ---- Gram build + blocked Cholesky (upper factor R = L^T) with the psd_safe_cholesky jitter ladder ------------------------
    f32x4 U[NU];                                              // U[uidx(K,J)], K <= J: block (K,J) of the matrix -> R[K][J]
    f32x4 Zd[NB];                                             // L_KK^-1
    f32x4 uB[NB];                                             // u = L^-1 r, replicated: register s of lane (r,g) = u[16K + 4g+s]
    f32x4 G[NB][NB];                                          // strictly-lower blocks of L^-1 (backward only)
    float dprod = 1.0f;                                       // this lane's share of prod_i L_ii (padding rows: 1)
    int my_info = -1;
    float jitter = 0.0f;
    for (int attempt = 0; attempt < 4; ++attempt) {
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            float zr[4][FP];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int c = 0; c < FP; ++c) zr[s][c] = zf[(16 * I + 4 * g + s) * FP + c];
#pragma unroll
            for (int J = I; J < NB; ++J) {
                float zc[FP];
#pragma unroll
                for (int c = 0; c < FP; ++c) zc[c] = zf[(16 * J + r) * FP + c];
                // the diagonal gets noise + jitter (a padding row: 1 - os, its kernel entry being os) through the -identity block
                const float dadd = (I == J) ? ((16 * J + r < nv) ? noise + jitter : 1.0f - os) : 0.0f;
                f32x4 blk;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float q = 0.0f;
#pragma unroll
                    for (int c = 0; c < FP; ++c) { const float d = zr[s][c] - zc[c]; q = fmaf(d, d, q); }
                    const float k = os * __builtin_amdgcn_exp2f(-q);
                    blk[s] = (I == J) ? fmaf(nId[s], dadd, -k) : -k;     // (the NEGATED matrix is stored: see the trailing update)
                }
                U[uidx(NB, I, J)] = blk;
                SCHED_FENCE();
            }
        }
        dprod = 1.0f;
        // Step K also finishes everything that only needs block rows <= K of R: u_K and (backward) block row K of L^-1, so that
        // V_K is a temporary and column K of R is dead afterwards -- the matrix drains out of the register file as the loop advances.
#pragma unroll
        for (int K = 0; K < NB; ++K) {
            SCHED_FENCE();
            // Signs: the matrix cores only accumulate (D = C + A B), and negating an operand costs four moves plus four more live
            // registers per product.  So the blocks not yet eliminated are kept NEGATED (Un = -A): the trailing update becomes
            // Un[I][J] += R[K][I]^T R[K][J] with both operands as they are, and every other product of the step takes the one
            // negated operand Vn = -L_KK^-T.
            factor16<(NB <= 2)>(U[uidx(NB, K, K)], Zd[K], dprod, nId, r, g, fsc, nv - 16 * K);
            SCHED_FENCE();
            // -L_KK^-T: the block transposed through 1.25 KB of LDS (4 dword writes + 4 dword reads, the conflict-free skewed
            // stride-17 layout of mlp_fused.hip's f_turn) instead of a product with the -identity block (4 MFMAs = 128 of the
            // issue cycles the matrix cores and the vector units share)
            f32x4 Vn = {0.f, 0.f, 0.f, 0.f};
            if constexpr (NB > 1) {                             // (one block: no panel and no off-diagonal block of L^-1 reads it)
                const int twr = 68 * g + 12 * (g & 1) + 32 * (g >> 1) + r, trd = 17 * r + 12 * ((r >> 2) & 1) + 32 * (r >> 3) + 4 * g;
#pragma unroll
                for (int s = 0; s < 4; ++s) tsc[twr + 17 * s] = -Zd[K][s];
                WSYNC();
#pragma unroll
                for (int q = 0; q < 4; ++q) Vn[q] = tsc[trd + q];
                WSYNC();
            }
#pragma unroll
            for (int J = K + 1; J < NB; ++J) U[uidx(NB, K, J)] = mmT(Vn, U[uidx(NB, K, J)], f32x4{0.f, 0.f, 0.f, 0.f});   // R[K][J] = L_KK^-1 A[K][J]
            SCHED_FENCE();
            {   // u_K = L_KK^-1 (r_K - sum_m L[K][m] u_m), L[K][m] = R[m][K]^T, all on the vector units: t = r_K - sum_m R[m][K]^T u_m
                // comes out of mvT_ / xg_sum_ in column layout (lane (r, .) holds entry r), and row 4g+s of L_KK^-1 t is a sum over the
                // 16 lanes of a lane row of Z[4g+s][r] t[r] -- four multiplies and 16 DPP adds, which leave u_K in the replicated layout
                // the later products want.  (Rounds 1-2: a 16x16x16 product on the matrix cores for this one vector, 128 issue cycles,
                // fed through an LDS round trip that turned t into the replicated layout.)
                float tc = rv[16 * K + r];
                if (K > 0) {
                    float tp = 0.0f;
#pragma unroll
                    for (int m = 0; m < K; ++m) tp = mvT_(U[uidx(NB, m, K)], uB[m], tp);
                    tc -= xg_sum_(tp);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) uB[K][s] = row_sum_(Zd[K][s] * tc);
            }
            SCHED_FENCE();
            if (BWD) {
#pragma unroll
                for (int J = 0; J < K; ++J) {                                          // G[K][J] = -L_KK^-1 sum_m L[K][m] Linv[m][J]
                    f32x4 S = mmT(U[uidx(NB, J, K)], Zd[J], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
                    for (int m = J + 1; m < K; ++m) S = mmT(U[uidx(NB, m, K)], G[m][J], S);
                    G[K][J] = mmT(Vn, S, f32x4{0.f, 0.f, 0.f, 0.f});
                    SCHED_FENCE();
                }
            }
#pragma unroll
            for (int I = K + 1; I < NB; ++I)
#pragma unroll
                for (int J = I; J < NB; ++J) U[uidx(NB, I, J)] = mmT(U[uidx(NB, K, I)], U[uidx(NB, K, J)], U[uidx(NB, I, J)]);
        }
        if (__builtin_amdgcn_ballot_w64(dprod > 0.0f && dprod < __builtin_huge_valf()) == ~0ull) { my_info = attempt; break; }   // every pivot positive
        jitter = 1e-6f;
        for (int q = 0; q < attempt; ++q) jitter *= 10.0f;
    }
    const bool okf = my_info >= 0;
    { int32_t* info_p = GPR_LATE(info); if (lane == 0 && info_p) info_p[b] = my_info; }

    float q2 = 0.0f;
#pragma unroll
    for (int K = 0; K < NB; ++K) q2 += (uB[K][0] * uB[K][0] + uB[K][1] * uB[K][1]) + (uB[K][2] * uB[K][2] + uB[K][3] * uB[K][3]);
    const float quad = wave_sum_(r == 0 ? q2 : 0.0f);
    // (hardware log2: its ~1e-7 relative error on a lane's share is far below what the sum's own rounding moves)
    const float logdet = -0.6931471805599453f * wave_sum_(__builtin_amdgcn_logf(dprod));   // log det = 2 sum log L_ii (dprod: 1 / L_ii); padding rows have pivot 1
    const float inv_nv = nv > 0 ? rcp_((float)nv) : 0.0f;
    float lml = -0.5f * (quad + 2.0f * logdet + (float)nv * 1.8378770664093453f) * inv_nv;
    if (!okf) lml = NAN;
    if (!PRED && lane == 0) GPR_LATE(lml)[b] = lml;
    if (!BWD) return;

    SCHED_FENCE();
    // ---- alpha = L^-T u: alpha_K = sum_{I >= K} Linv[I][K]^T u_I, on the vector units, published to LDS ---------------------------
#pragma unroll
    for (int K = 0; K < NB; ++K) {
        float ap = mvT_(Zd[K], uB[K], 0.0f);
#pragma unroll
        for (int I = K + 1; I < NB; ++I) ap = mvT_(G[I][K], uB[I], ap);
        ap = xg_sum_(ap);
        if (g == 0) av[16 * K + r] = ap;
    }
    SCHED_FENCE();
    if constexpr (PRED) {
        // ---- V = L^-1 K_xs block by block from registers -- var_s = os + noise - |V_s|^2 -- with the blocks of L^-1 TRANSPOSED once, in
        //      place (mmT(X, Y) = X^T Y is the only product the layout offers: L^-1[I][J] K_J = mmT(L^-1[I][J]^T, K_J)).  No K^-1 at all:
        //      NB (NB + 1) / 2 block products per test block instead of NB^2 with K^-1, none to form it, nothing parked in LDS (the first
        //      version, through K^-1 parked in LDS, took 0.26 ms for 20 480 problems of n = m = 64; this one: see gp_reg.hip)
        const int twr = 68 * g + 12 * (g & 1) + 32 * (g >> 1) + r, trd = 17 * r + 12 * ((r >> 2) & 1) + 32 * (r >> 3) + 4 * g;
        auto turn = [&](f32x4& X) {
#pragma unroll
            for (int s = 0; s < 4; ++s) tsc[twr + 17 * s] = X[s];
            WSYNC();
#pragma unroll
            for (int q = 0; q < 4; ++q) X[q] = tsc[trd + q];
            WSYNC();
        };
#pragma unroll
        for (int I = 0; I < NB; ++I) {
            turn(Zd[I]);
#pragma unroll
            for (int J = 0; J < I; ++J) turn(G[I][J]);
            SCHED_FENCE();
        }
        const int m_tst = pa->m;
        const float* ztp = pa->zt + (long)(blk / (unsigned)pa->zt_div) * m_tst * (long)f;
        const float bad = okf ? 0.0f : NAN;
        const float prior_var = os + noise;
#pragma unroll 1
        for (int s0 = 0; s0 < m_tst; s0 += 16) {
            const int sidx = s0 + r;
            float zt[FP];
#pragma unroll
            for (int c = 0; c < FP; ++c) zt[c] = (sidx < m_tst && c < f) ? ztp[(long)sidx * f + c] * kls[c] : 0.0f;
            f32x4 Ks[NB];
            float mu_p = 0.0f;
#pragma unroll
            for (int I = 0; I < NB; ++I) {
                const f32x4 ai4 = *reinterpret_cast<const f32x4*>(av + 16 * I + 4 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float q = 0.0f;
#pragma unroll
                    for (int c = 0; c < FP; ++c) { const float d = zf[(16 * I + 4 * g + s) * FP + c] - zt[c]; q = fmaf(d, d, q); }
                    const float k = os * __builtin_amdgcn_exp2f(-q);
                    Ks[I][s] = k;
                    mu_p = fmaf(k, ai4[s], mu_p);
                }
                SCHED_FENCE();
            }
            float vv = 0.0f;
#pragma unroll
            for (int I = 0; I < NB; ++I) {
                f32x4 Vi = {0.f, 0.f, 0.f, 0.f};
                asm volatile("" : "+v"(Vi));                     // (keeps block row I's products behind block row I - 1's sums: see the W loop)
#pragma unroll
                for (int J = 0; J < I; ++J) Vi = mmT(G[I][J], Ks[J], Vi);
                Vi = mmT(Zd[I], Ks[I], Vi);
                if (pa->V_out && sidx < m_tst) {                 // rows 16 I + 4 g .. + 3 of column sidx: four consecutive floats
                    float* vp = pa->V_out + (b * m_tst + sidx) * (long)n + 16 * I + 4 * g;
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (16 * I + 4 * g + q < n) vp[q] = okf ? Vi[q] : NAN;
                }
                vv = fmaf(Vi[0], Vi[0], vv); vv = fmaf(Vi[1], Vi[1], vv); vv = fmaf(Vi[2], Vi[2], vv); vv = fmaf(Vi[3], Vi[3], vv);
                asm volatile("" : "+v"(vv));
                SCHED_FENCE();
            }
            mu_p = xg_sum_(mu_p);
            vv = xg_sum_(vv);
            if (g == 0 && sidx < m_tst) {
                float mt = 0.0f;
                if (a.mean_mode == PACOH_MEAN_VECTOR) mt = pa->mean_tst[b * m_tst + sidx];
                else if (a.mean_mode == PACOH_MEAN_CONST) mt = pa->mean_tst[p];
                pa->mu[b * m_tst + sidx] = mt + mu_p + bad;
                pa->var[b * m_tst + sidx] = prior_var - vv + bad;
            }
        }
        return;
    }
    // ---- gradient sums ---------------------------------------------------------------------------------------------------------------
    const float* g_lml_p = GPR_LATE(g_lml);
    const float gup = g_lml_p ? g_lml_p[b] : 1.0f;
    const float osn = 0.5f * os * inv_nv;   // the outputscale rides on the 1/(2 n) factor: M_ij = G_ij os e_ij
    float msum = 0.0f, dnz = 0.0f;                              // sum of M (= os d lml/d os), os x trace part (= os d lml/d noise); x osn at the end
    // Every ordered pair (i, j) is visited, column block by column block: lane (r, g) holds the entries (i = 16I + 4g+s, j = 16J + r),
    // so everything destined for point j -- d_z[j] = sum_i M_ij (z_i - z_j) -- accumulates in the lane over s and I and needs only
    // two lane exchanges (over g) per column block at the end.  (Using the symmetry instead -- upper blocks only, each entry feeding
    // the row sum of i as well -- saves 24 of the 64 exponentials per lane but needs sums over the 16 lanes of a row: 128 DPP adds,
    // and the compiler kept every block row's partial sums alive to the end of the kernel, 100 registers over budget.)
    // the four entries of one block in this lane: M_ij (z_i - z_j) into colacc (point j), M_ij (z_i - z_j)^2 into dls, M_ij into msum
    // Instruction diet of round 3 (64 entries per lane pass through here): the factor os / (2 n) is applied to the finished sums
    // instead of every entry, and the lengthscale gradient sum_ij M_ij (z_i - z_j)^2 is not accumulated at all -- M being symmetric
    // it equals -2 sum_j (z_j - c) . colsum_j for any constant c (sum_j colsum_j = 0), i.e. it falls out of the finished d_z sums
    // with one multiply per point: 10 instead of 15 vector instructions per entry at f = 2.
    auto block_entries = [&](const f32x4& Wb, const int I, const float (&zc)[FP], const float aj, float (&colacc)[FP], const bool diag) {
        const f32x4 ai4 = *reinterpret_cast<const f32x4*>(av + 16 * I + 4 * g);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float zi[FP];
#pragma unroll
            for (int c = 0; c < FP; ++c) zi[c] = zf[(16 * I + 4 * g + s) * FP + c];
            const float Gij = fmaf(ai4[s], aj, -Wb[s]);
            if (diag) dnz = fmaf(nId[s], Gij, dnz);           // minus the trace part (padding rows: taken out again below)
            float q = 0.0f, df[FP];
#pragma unroll
            for (int c = 0; c < FP; ++c) { df[c] = zi[c] - zc[c]; q = fmaf(df[c], df[c], q); }
            const float M = Gij * __builtin_amdgcn_exp2f(-q);
            if (HAS_OS) msum += M;
#pragma unroll
            for (int c = 0; c < FP; ++c) colacc[c] = fmaf(M, df[c], colacc[c]);
        }
    };
    // ---- W = K^-1, upper block triangle: W[I][J] = sum_{m >= J} Linv[m][I]^T Linv[m][J] ------------------------------------------------
    // n <= 64 (PARK): a diagonal block is consumed where it is produced; the strictly upper blocks are parked in LDS for the loop
    // below, which needs each of them twice (as block (I,J) and, transposed, as block (J,I)) -- the two phases do not share the
    // register file, which is what keeps the n = 64 kernel at four waves per SIMD.
    // n > 64: parking costs 1 KB per block -- 28 KB per problem at n = 128, which alone held the kernel at ONE wave per SIMD (four
    // problems per CU), where a lone wave issues a vector instruction every ~7 cycles instead of every 4.  There every block is
    // consumed where it is produced, once as it is and once transposed through the 1.25 KB transpose scratch, with the column sums
    // of all NB column blocks live (NB x FP registers): 4.8 KB of LDS per problem, 2 waves per SIMD.
    constexpr bool PARK = NB <= 4;
    if constexpr (!PARK) {
        float colacc[NB][FP];
#pragma unroll
        for (int J = 0; J < NB; ++J)
#pragma unroll
            for (int c = 0; c < FP; ++c) colacc[J][c] = 0.0f;
        const int twr = 68 * g + 12 * (g & 1) + 32 * (g >> 1) + r, trd = 17 * r + 12 * ((r >> 2) & 1) + 32 * (r >> 3) + 4 * g;
#pragma unroll
        for (int I = 0; I < NB; ++I) {
#pragma unroll
            for (int J = I; J < NB; ++J) {
                // (pure arithmetic floats freely across SCHED_FENCE when the instruction stream is first laid out: left alone, all 36
                //  block products come first -- their results spilled -- and the 4 608 entries after them.  The volatile asm
                //  statements pass the accumulator's zero and the running sums through: block (I,J)'s products start behind the
                //  previous block's last entry.)
                f32x4 Wb = {0.f, 0.f, 0.f, 0.f};
                asm volatile("" : "+v"(Wb));
                Wb = mmT(I == J ? Zd[J] : G[J][I], Zd[J], Wb);
#pragma unroll
                for (int m = J + 1; m < NB; ++m) Wb = mmT(G[m][I], G[m][J], Wb);
                {
                    float zc[FP];
#pragma unroll
                    for (int c = 0; c < FP; ++c) zc[c] = zf[(16 * J + r) * FP + c];
                    block_entries(Wb, I, zc, av[16 * J + r], colacc[J], I == J);
                }
                if (I != J) {                                  // ... and as block (J, I): the transpose
                    f32x4 Wt;
#pragma unroll
                    for (int s = 0; s < 4; ++s) tsc[twr + 17 * s] = Wb[s];
                    WSYNC();
#pragma unroll
                    for (int q = 0; q < 4; ++q) Wt[q] = tsc[trd + q];
                    WSYNC();
                    float zc[FP];
#pragma unroll
                    for (int c = 0; c < FP; ++c) zc[c] = zf[(16 * I + r) * FP + c];
                    block_entries(Wt, J, zc, av[16 * I + r], colacc[I], false);
#pragma unroll
                    for (int c = 0; c < FP; ++c) asm volatile("" : "+v"(colacc[I][c]));
                }
#pragma unroll
                for (int c = 0; c < FP; ++c) asm volatile("" : "+v"(colacc[J][c]));
                SCHED_FENCE();
            }
        }
#pragma unroll
        for (int J = 0; J < NB; ++J)
#pragma unroll
            for (int c = 0; c < FP; ++c) {
                float v = colacc[J][c];
                v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                if (g == 0) dzc[(16 * J + r) * FP + c] = v;
            }
    } else {
#pragma unroll
    for (int I = 0; I < NB; ++I) {
#pragma unroll
        for (int J = I; J < NB; ++J) {
            f32x4 Wb = mmT(I == J ? Zd[J] : G[J][I], Zd[J], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int m = J + 1; m < NB; ++m) Wb = mmT(G[m][I], G[m][J], Wb);
            if (I == J) {
                float zc[FP], colacc[FP];
#pragma unroll
                for (int c = 0; c < FP; ++c) { zc[c] = zf[(16 * J + r) * FP + c]; colacc[c] = 0.0f; }
                block_entries(Wb, I, zc, av[16 * J + r], colacc, true);
#pragma unroll
                for (int c = 0; c < FP; ++c) {
                    float v = colacc[c];
                    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                    if (g == 0) dzc[(16 * J + r) * FP + c] = v;
                }
            } else {
                *reinterpret_cast<f32x4*>(Wl + (uidx(NB, I, J) - (I + 1)) * 256 + lane * 4) = Wb;
            }
            SCHED_FENCE();
        }
    }
    WSYNC();
    // (real loops, all operands from LDS: fully unrolled, the compiler hoists the loads of every iteration to the top and spills)
#pragma unroll 1
    for (int J = 0; J < NB; ++J) {
        float zc[FP], colacc[FP];
#pragma unroll
        for (int c = 0; c < FP; ++c) { zc[c] = zf[(16 * J + r) * FP + c]; colacc[c] = 0.0f; }
        const float aj = av[16 * J + r];
#pragma unroll 1
        for (int I = 0; I < NB; ++I) {
            if (I == J) continue;
            f32x4 Wb;
            if (I < J) {
                Wb = *reinterpret_cast<const f32x4*>(Wl + (I * NB - I * (I - 1) / 2 + (J - I) - (I + 1)) * 256 + lane * 4);
            } else {                                          // transpose of the stored block (J, I): element (r, 4g+s) of it
                const float* wt = Wl + (J * NB - J * (J - 1) / 2 + (I - J) - (J + 1)) * 256 + (16 * (r >> 2)) * 4 + (r & 3);
#pragma unroll
                for (int s = 0; s < 4; ++s) Wb[s] = wt[(4 * g + s) * 4];
            }
            block_entries(Wb, I, zc, aj, colacc, false);
        }
#pragma unroll
        for (int c = 0; c < FP; ++c) {
            float v = colacc[c];
            v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
            if (g == 0) dzc[(16 * J + r) * FP + c] += v;
        }
    }
    }
    WSYNC();
    const float bad = okf ? 0.0f : NAN;
    float* d_z_p = GPR_LATE(d_z);
    float asum = 0.0f;
    float dls[FP];                                              // sum_j (z_j - z_0) colsum_j (see block_entries)
#pragma unroll
    for (int c = 0; c < FP; ++c) dls[c] = 0.0f;
#pragma unroll
    for (int rr = 0; rr < RPL; ++rr) {
        const int i = lane + 64 * rr;
        const float ai = i < NP ? av[i] : 0.0f;
        float dzi[FP];
#pragma unroll
        for (int c = 0; c < FP; ++c) {
            dzi[c] = i < NP ? dzc[i * FP + c] : 0.0f;
            if (i < nv) dls[c] = fmaf(zf[i * FP + c] - zf[c], dzi[c], dls[c]);
        }
        if (d_z_p && i < n) {
#pragma unroll
            for (int c = 0; c < FP; ++c)
                if (c < f) d_z_p[(b * n + i) * (long)f + c] = (i < nv) ? (2.0f * INV_KAPPA2) * osn * gup * dzi[c] * kls[c] + bad : 0.0f;
        }
        if (a.mean_mode == PACOH_MEAN_VECTOR) {
            float* d_mean_p = GPR_LATE(d_mean);
            if (d_mean_p && i < n) d_mean_p[b * n + i] = (i < nv) ? gup * ai * inv_nv + bad : 0.0f;
        }
        asum += (i < nv) ? ai : 0.0f;
    }
    if (a.mean_mode == PACOH_MEAN_CONST) {
        const float sa = wave_sum_(asum);
        float* d_mean_p = GPR_LATE(d_mean);
        if (d_mean_p && lane == 0) d_mean_p[b] = gup * sa * inv_nv + bad;
    }
#pragma unroll
    for (int c = 0; c < FP; ++c) {
        if (c < f) {
            const float sc = -2.0f * osn * wave_sum_(dls[c]);
            if (lane == 0) GPR_LATE(d_ls)[b * f + c] = (INV_KAPPA2 / KAPPA) * gup * sc * kls[c] + bad;
        }
    }
    // a padding row's diagonal entry is G_ii = (0 - 1) osn exactly, with kernel entry 1: out of both sums again
    const float padc = (float)(NP - nv) * osn;
    const float sdos = osn * wave_sum_(msum) + padc, sdnz = padc - osn * wave_sum_(dnz);
    if (lane == 0) {
        float* d_os_p = GPR_LATE(d_os);
        const float inv_os = rcp_(os);
        if (d_os_p) d_os_p[b] = gup * sdos * inv_os + bad;
        GPR_LATE(d_noise)[b] = gup * sdnz * inv_os + bad;
    }
#undef WSYNC
#undef SCHED_FENCE
#undef GPR_LATE
}

}  // namespace gpreg
}  // namespace pacoh
