// Dense (large-n) Cholesky / Gaussian log-density, LEFT-LOOKING version (round 4): one 1024-thread workgroup per matrix.
//
// Why: the right-looking kernel of dense_mfma.hip streams the whole trailing matrix through HBM/L2 once per 32-wide panel
// (read + write: 10 MB per 512 x 512 fp64 matrix, 2.6 GB per 256-matrix launch -- the kernel's bound).  Here a 64-wide block
// column j is produced from the finished columns to its left in ONE streaming read of L[c0:, 0:c0] (c0 = 64 j): the
// (n - c0) x 64 result lives in MFMA accumulators spread over twelve wavefronts (<= 64 of a wave's 128 registers), the left
// panels pass through LDS in k-slabs of 16 / 32 columns filled by LDS-DMA (global_load_lds_dwordx4, per-lane source
// addresses, XOR-swizzled 128 / 256-byte rows: conflict-free ds_read_b64 operand reads, no staging registers), and
// nothing is written inside the accumulation loop.  Traffic per 512 x 512 fp64 matrix: 2.75 MB of slabs + one read of the
// lower triangle + one write of L = 4.8 MB.
//
// Wave specialisation (the 16 waves of the workgroup; waves w and w + 4 share a SIMD, MI355X_MICROARCH.md "LDS"):
//   wave 0            the CHAIN: factors and inverts the 64 x 64 diagonal block of column j -- two register-resident 32 x 32
//                     eliminations (dense_diag.h) plus 48 MFMAs -- in 26 quanta interleaved with the workgroup's barriers,
//                     WHILE the bulk waves accumulate column j.  The diagonal block itself was finished one column earlier:
//                     the bulk waves carry the next diagonal block's ten 16 x 16 blocks as extra "look-ahead" accumulators.
//   waves 4, 8, 12    helpers on the chain's SIMD (an MFMA-saturated SIMD would starve the chain's dependent fp64 chain):
//                     they issue the LDS-DMA of every slab; wave 4 also runs the forward solve u = L^-1 r left-looking
//                     (u_j = Z_jj (r_j - L[j, 0:j] u), the dot products taken from the slabs' top 64 rows).
//   other 12 waves    BULK: each owns <= 2 row blocks (16 rows x 64 columns = 4 accumulator blocks each) and <= 2 look-ahead
//                     blocks.  All accumulators are held TRANSPOSED (C^T: MFMA A operand = the column's top rows, B operand =
//                     the block's own rows), so that an accumulator block is directly the B operand of the next product
//                     (k index = its register index): the panel solve L21^T = Z_jj A21^T runs from registers with Z_jj, the
//                     explicit inverse of the diagonal block, as LDS operand images written by the chain.
// Storage format on exit = dense_mfma.hip's: L in the lower triangle, the strictly lower part of every 32 x 32 diagonal
// block's inverse transposed into that block's strictly upper part (read by trtri_dense_kernel and the backward solve).
// Replaces torch/gpytorch MultivariateNormal.log_prob -> LAPACK potrf/potrs on the reference's CPU path
// (random_gp.py:83-85 at the large-context configuration; joint test log-likelihood of abstract.py:134-163).
#include "common.h"
#include "dense_diag.h"
#include <type_traits>

namespace pacoh {
namespace {

constexpr int LLW = 64;              // block-column width
constexpr int LL_NT = 1024;
constexpr int LL_NBULK = 12;
constexpr int LL_NQ = 29;            // quanta of the chain per block column
constexpr int LL_QCAP = 12;          // ... of which in front of P0 (none of them touches Zimg; about what the bulk waves' pass 1 takes)            // quanta of the chain per block column

__host__ __device__ constexpr int tri_idx(int a, int b) { return a * (a + 1) / 2 + b; }

struct ColCfg { int c0, m, mb, R, RB, KS, ns, nb, bsz, mrows; };

template <typename T>
__host__ __device__ inline ColCfg col_cfg(int j, int n, int S0, int S1) {
    ColCfg c;
    c.c0 = j * LLW; c.m = n - c.c0; c.mb = (c.m + 15) >> 4;
    c.R = c.mb > 4 ? c.mb - 4 : 0;
    // the slabs of a column go through a ring of up to four buffers in the S0 + S1 bytes behind the fixed LDS areas (prefetch
    // distance = buffers - 1); 256-byte rows (half as many slabs and barriers) when three of those fit, else 128-byte rows.
    // A DMA instruction fills 1 KiB: 4 rows of 256 bytes or 8 of 128.
    const int tot = S0 + S1;
    const int m256 = (c.m + 3) & ~3, m128 = (c.m + 7) & ~7;
    int nb256 = tot / (m256 * 256), nb128 = tot / (m128 * 128);
    nb256 = nb256 > 4 ? 4 : nb256; nb128 = nb128 > 4 ? 4 : nb128;
    if (nb256 >= 3) { c.RB = 256; c.mrows = m256; c.nb = nb256; } else { c.RB = 128; c.mrows = m128; c.nb = nb128; }
    c.bsz = c.mrows * c.RB;
    c.KS = c.RB / (int)sizeof(T);
    c.ns = c.R > 0 ? c.c0 / c.KS : 0;                         // (a last column with nothing below its diagonal block: no slabs)
    return c;
}

template <typename T> __host__ __device__ constexpr size_t ll_fixed_bytes(int n) {
    return (size_t)20 * 256 * sizeof(T) + ((size_t)((n + 63) & ~63) + 64 + 64 + 32) * sizeof(T);
}

// byte offset of element (c15, r) inside a 16x16 block image: register q_of(c15) of lane 16 g_of(c15) + r
template <typename T> __device__ __forceinline__ constexpr int img_off(int c15, int r) {
    return (Mf<T>::q_of(c15) * 64 + 16 * Mf<T>::g_of(c15) + r) * (int)sizeof(T);
}

__device__ __forceinline__ void glds16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// s_waitcnt vmcnt(n) for a run-time n (the instruction takes an immediate); n > 63 waits for less than asked: never here
__device__ __forceinline__ void wait_vmcnt(int n) {
#define LL_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        LL_VMC(0) LL_VMC(1) LL_VMC(2) LL_VMC(3) LL_VMC(4) LL_VMC(5) LL_VMC(6) LL_VMC(7) LL_VMC(8) LL_VMC(9) LL_VMC(10) LL_VMC(11) LL_VMC(12)
        LL_VMC(13) LL_VMC(14) LL_VMC(15) LL_VMC(16) LL_VMC(17) LL_VMC(18) LL_VMC(19) LL_VMC(20) LL_VMC(21) LL_VMC(22) LL_VMC(23) LL_VMC(24)
        LL_VMC(25) LL_VMC(26) LL_VMC(27) LL_VMC(28) LL_VMC(29) LL_VMC(30) LL_VMC(31) LL_VMC(32) LL_VMC(33) LL_VMC(34) LL_VMC(35) LL_VMC(36)
        LL_VMC(37) LL_VMC(38) LL_VMC(39) LL_VMC(40) LL_VMC(41) LL_VMC(42) LL_VMC(43) LL_VMC(44) LL_VMC(45) LL_VMC(46) LL_VMC(47) LL_VMC(48)
        LL_VMC(49) LL_VMC(50) LL_VMC(51) LL_VMC(52) LL_VMC(53) LL_VMC(54) LL_VMC(55) LL_VMC(56) LL_VMC(57) LL_VMC(58) LL_VMC(59) LL_VMC(60)
        LL_VMC(61) LL_VMC(62)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef LL_VMC
}

#define LL_BAR() __syncthreads()
#define LL_LDSWAIT() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

template <typename T>
__device__ __forceinline__ void chol_ll_body(T* __restrict__ A, const T* __restrict__ resid, T* __restrict__ logp,
                                             T* __restrict__ alpha_out, int32_t* __restrict__ info, T scale, int n,
                                             int attempt, int u_only, int S0, int S1) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    constexpr int ES = sizeof(T), BLK = 256 * ES;
    using Acc = typename Mf<T>::acc;
    unsigned char* const Dimg = sm;                               // next / current diagonal block, 10 block images (C^T layout)
    unsigned char* const Zimg = Dimg + 10 * BLK;                  // Z00 | -L10 | Z11 as MFMA A-operand images
    const int npad = (n + 63) & ~63;
    T* const uvec = reinterpret_cast<T*>(Zimg + 10 * BLK);        // [npad]  u = L^-1 r
    T* const tmpv = uvec + npad;                                  // [64]
    T* const invd = tmpv + 64;                                    // [64]   1 / diag of the 32-block being eliminated
    T* const flg = invd + 64;                                     // [32]   0: fail, 1: logdet, 2..17: per-wave partials
    unsigned char* const slab0 = reinterpret_cast<unsigned char*>(flg + 32);
    unsigned char* const slab1 = slab0 + S0;
    unsigned char* const Limg = slab1;                            // 16 block images of L21^T (rows 64..127 of the column), phase B only

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    T* const Ab = A + (size_t)blockIdx.x * n * n;
#ifdef PACOH_LL_STAMPS
    if (blockIdx.x == 0 && lane == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        printf("wave %d: HW_ID %08x  simd %u  cu %u  wave_slot %u\n", wave, hwid, (hwid >> 4) & 3, (hwid >> 8) & 15, hwid & 15);
    }
#endif
    const int ncol = (n + LLW - 1) / LLW;
    if (tid < 32) flg[tid] = T(0);
    for (int q = tid; q < npad; q += LL_NT) uvec[q] = T(0);

    // ---- the initial diagonal block -> Dimg (bulk waves 2..11 hold the ten look-ahead blocks, one each) ----------------------
    const bool is_chain = wave == 0;
    const bool is_helper = !is_chain && (wave & 3) == 0;
    const int hidx = (wave >> 2) - 1;                                                  // helpers 0..2
    const int bw = (wave >> 2) * 3 + (wave & 3) - 1;                                   // bulk waves 0..11
    // look-ahead block t (0..9) = (cbr, ibr), cbr <= ibr, of the next diagonal block
    auto la_coords = [](int t, int& cbr, int& ibr) __attribute__((always_inline)) {
        ibr = t >= 6 ? 3 : (t >= 3 ? 2 : (t >= 1 ? 1 : 0));
        cbr = t - tri_idx(ibr, 0);
    };
    // value of the (identity padded) matrix at (row, col)
    auto a_padded = [&](int row, int col) __attribute__((always_inline)) -> T {
        return (row < n && col < n) ? Ab[(size_t)row * n + col] : (row == col ? T(1) : T(0));
    };
    if (!is_chain && !is_helper && bw >= 2) {
        int cbr, ibr;
        la_coords(bw - 2, cbr, ibr);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<T*>(Dimg + tri_idx(ibr, cbr) * BLK + (q * 64 + lane) * ES) = a_padded(16 * ibr + r, 16 * cbr + Mf<T>::row(g, q));
    }
    LL_BAR();

    // ---- pieces shared by the bulk waves and (first column) the helpers -------------------------------------------------------
    // accumulator blocks are TRANSPOSED (register q of lane (r, g) = element [row 16 ib + r][column 16 cb + row(g, q)]); they start
    // as A -- loads that nothing waits for until the first MFMA: they fly under the first slab's fetch -- and collect - L L^T
    // (the top-row operand is negated on its way into the MFMA: four sign flips per eight MFMAs)
    auto load_a = [&](Acc (&X)[4], int c0, int ib) __attribute__((always_inline)) {
        const int row = c0 + 16 * ib + r;
        const int rowc = row < n ? row : n - 1;                   // (rows beyond the matrix: a copy of the last row, never stored -- a
                                                                  //  select on the loaded value would make every load wait where it is issued)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) X[cb][q] = Ab[(size_t)rowc * n + c0 + 16 * cb + Mf<T>::row(g, q)];
    };
    auto zmul = [&](int av, int b, const Acc& X, Acc o) __attribute__((always_inline)) -> Acc {
        const T* zp = reinterpret_cast<const T*>(Zimg + tri_idx(av, b) * BLK) + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) o = Mf<T>::mma(zp[s * 64], X[s], o);
        return o;
    };
    // panel solve (4 blocks = 64 columns x 16 rows, transposed): X := L21^T = Z_jj X, through the 2 x 2 structure of Z_jj with the
    // images Z00 | -L10 | Z11
    auto solve = [&](Acc (&X)[4]) __attribute__((always_inline)) {
        const Acc z = {0, 0, 0, 0};
        Acc t1 = zmul(1, 0, X[0], z);
        t1 = zmul(1, 1, X[1], t1);
        const Acc t0 = zmul(0, 0, X[0], z);
        X[0] = t0; X[1] = t1;
        X[2] = zmul(2, 0, X[0], X[2]); X[2] = zmul(2, 1, X[1], X[2]);      // panel_hi - L10 Y_lo
        X[3] = zmul(3, 0, X[0], X[3]); X[3] = zmul(3, 1, X[1], X[3]);
        Acc t3 = zmul(3, 2, X[2], z);
        t3 = zmul(3, 3, X[3], t3);
        const Acc t2 = zmul(2, 2, X[2], z);
        X[2] = t2; X[3] = t3;
    };
    auto store_l = [&](const Acc (&X)[4], int c0, int ib) __attribute__((always_inline)) {
        const int row = c0 + 16 * ib + r;
        if (row < n) {
#pragma unroll
            for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) Ab[(size_t)row * n + c0 + 16 * cb + Mf<T>::row(g, q)] = X[cb][q];
        }
    };

#ifdef LL_X_CHAIN
    if (false) {
#else
    if (is_chain) {
#endif
        // =========================================================== CHAIN ===========================================================
        T a[DNB];
        bool bad = false;
        T logdet_part = 0;
#ifdef PACOH_LL_STAMPS
        long long st_q = 0, st_t0 = wall_clock64(), st_qc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        for (int j = 0; j < ncol; ++j) {
            const ColCfg cf = col_cfg<T>(j, n, S0, S1);
            const int c0 = cf.c0;
            const int kbw = cf.m < LLW ? cf.m : LLW;
            const bool two = kbw > 32;
            // Pieces of the factorisation of the 64 x 64 block in Dimg.  h = 0: the 64 x 32 PANEL [D00; D10], a row per lane (all 64 lanes
            // carry distinct rows: the elimination of D00 yields L10 = D10 L00^-T in the same instructions); h = 1: the 32 x 32 block
            // D11 - L10 L10^T (both half-waves carry the same rows).
            auto load_rows = [&](int h, int lane) __attribute__((always_inline)) {          // h = 2: L00 back from Dimg, for its inverse
                const int row = h == 1 ? 32 + (lane & 31) : (h == 2 ? (lane & 31) : lane);
                if (h == 2) h = 0;
                const int ibv = row >> 4;
                const int base = tri_idx(ibv, 0) * BLK;
#pragma unroll
                for (int cc = 0; cc < DNB; ++cc) {
                    const int cb = 2 * h + (cc >> 4);
                    T v = T(0);
                    if (cb <= ibv) v = *reinterpret_cast<const T*>(Dimg + base + cb * BLK + img_off<T>(cc & 15, row & 15));
                    a[cc] = v;
                }
            };
            // The chain writes nothing to global memory (its scattered 8-byte stores queued behind the bulk waves' panel stores and
            // stalled the elimination): L(h,h) goes back into the Dimg images its rows came from, -L10 into its Zimg operand images,
            // and helper waves 1 / 2 copy all of it out behind X3.
            auto after_elim = [&](int h, int lane, int n, int c0) __attribute__((always_inline)) {
                const int row = h ? 32 + (lane & 31) : lane;
                const int ibv = row >> 4;
                const bool is_l10 = h == 0 && lane >= 32;         // (-L10 goes into the Dimg slots D10 came from: Zimg still belongs to the
                unsigned char* const dst = Dimg + tri_idx(ibv, 0) * BLK;      //  previous column's panel solve when this runs in front of P0)
                if (h == 0 || lane < 32) {
#pragma unroll
                    for (int cc = 0; cc < DNB; ++cc) {
                        const int cb = 2 * h + (cc >> 4);
                        if (cb <= ibv) *reinterpret_cast<T*>(dst + cb * BLK + img_off<T>(cc & 15, row & 15)) = is_l10 ? -a[cc] : a[cc];
                    }
                }
                if (lane < 32 && c0 + 32 * h + lane < n) logdet_part -= t_log<T>(invd[32 * h + lane]);
            };
            auto after_inv = [&](int h, int rr, int lane, int n, int c0) __attribute__((always_inline)) {       // Z(h,h) -> operand images
                if (lane < 32) {
                    const int av = 2 * h + (rr >> 4);
#pragma unroll
                    for (int cc = 0; cc < DNB; ++cc) {
                        const int b = 2 * h + (cc >> 4);
                        if (b <= av) *reinterpret_cast<T*>(Zimg + tri_idx(av, b) * BLK + img_off<T>(cc & 15, rr & 15)) = a[cc];
                    }
                }
            };
            auto upd11 = [&](int lane) __attribute__((always_inline)) {      // D11 -= L10 L10^T, both operands from the -L10 images
                LL_LDSWAIT();
#pragma unroll
                for (int ibr = 0; ibr < 2; ++ibr)
#pragma unroll
                    for (int cbr = 0; cbr <= ibr; ++cbr) {
                        Acc sacc = {0, 0, 0, 0};
#pragma unroll
                        for (int av = 0; av < 2; ++av)
#pragma unroll
                            for (int s = 0; s < 4; ++s)
                                sacc = Mf<T>::mma(*reinterpret_cast<const T*>(Dimg + tri_idx(2 + cbr, av) * BLK + (s * 64 + lane) * ES),
                                                  *reinterpret_cast<const T*>(Dimg + tri_idx(2 + ibr, av) * BLK + (s * 64 + lane) * ES), sacc);
                        T* dp = reinterpret_cast<T*>(Dimg + tri_idx(2 + ibr, 2 + cbr) * BLK);
#pragma unroll
                        for (int q = 0; q < 4; ++q) dp[q * 64 + lane] -= sacc[q];
                    }
                LL_LDSWAIT();
            };
            // The factorisation is straight-line code (one definition chain for the 32 row registers: as a switch inside a loop
            // every quantum boundary was a 27-way merge of all of them and the allocator spilled); after each quantum the barriers
            // of the slots that end there are executed.  Slot 0 ends at P0 (quanta 0..11: Zimg still belongs to the previous
            // column's panel solve, and none of them touches it), slot 1 at P1, slot 2 + s at slab s's barrier(s).
            const int nslots = cf.ns + 2;
            int slot = 0;
#ifdef PACOH_LL_STAMPS
            long long s0_ = wall_clock64();
#endif
            auto sync_point = [&](int qdone) __attribute__((always_inline)) {
#ifdef PACOH_LL_STAMPS
                { const long long d_ = wall_clock64() - s0_; st_q += d_; st_qc[j & 7] += d_; }
#endif
                while (slot < nslots) {
                    // the chain is the kernel's critical path: everything Zimg's hand-over allows goes in front of P0 (the bulk waves
                    // are in the previous column's panel solve meanwhile), the rest is spread over P1 and the slabs
                    const int qe = slot <= 1 ? LL_QCAP : LL_QCAP + (slot - 1) * (LL_NQ - LL_QCAP) / (nslots - 2);
                    if (qe > qdone) break;
                    LL_BAR();
                    if (slot >= 2 && cf.nb == 1) LL_BAR();
                    ++slot;
                }
#ifdef PACOH_LL_STAMPS
                s0_ = wall_clock64();
#endif
            };
            // (lane ids and n, c0 made opaque HERE, inside the column loop: loop-invariant, every lane mask and row address of the
            //  unrolled steps below would otherwise be hoisted out of the loop and kept in ~180 scalar + ~280 vector registers)
#define LL_OPAQUE() int lane = threadIdx.x & 63; asm volatile("" : "+v"(lane)); const int rr = lane & 31; int nq = n, c0q = c0; \
            asm volatile("" : "+s"(nq), "+s"(c0q)); (void)rr; (void)nq; (void)c0q
            sync_point(0);
            // quanta 0..7: elimination of the panel, 8: L00 / -L10 images (in Dimg), 9: update of D11, 10..17: elimination of D11,
            // 18: -L10 into Zimg, 19..22: inverse of L11, 23: images, 24..28: inverse of L00 and images
            { LL_OPAQUE(); load_rows(0, lane); ElimRange<T, 0, 4>::run(a, invd, bad, lane); } sync_point(1);
            { LL_OPAQUE(); ElimRange<T, 4, 8>::run(a, invd, bad, lane); } sync_point(2);
            { LL_OPAQUE(); ElimRange<T, 8, 12>::run(a, invd, bad, lane); } sync_point(3);
            { LL_OPAQUE(); ElimRange<T, 12, 16>::run(a, invd, bad, lane); } sync_point(4);
            { LL_OPAQUE(); ElimRange<T, 16, 20>::run(a, invd, bad, lane); } sync_point(5);
            { LL_OPAQUE(); ElimRange<T, 20, 24>::run(a, invd, bad, lane); } sync_point(6);
            { LL_OPAQUE(); ElimRange<T, 24, 28>::run(a, invd, bad, lane); } sync_point(7);
            { LL_OPAQUE(); ElimRange<T, 28, 32>::run(a, invd, bad, lane); } sync_point(8);
            { LL_OPAQUE(); LL_LDSWAIT(); after_elim(0, lane, nq, c0q); LL_LDSWAIT(); } sync_point(9);
            if (two) { LL_OPAQUE(); upd11(lane); } sync_point(10);
            if (two) { LL_OPAQUE(); load_rows(1, lane); ElimRange<T, 0, 4>::run(a, invd + 32, bad, lane); } sync_point(11);
            if (two) { LL_OPAQUE(); ElimRange<T, 4, 8>::run(a, invd + 32, bad, lane); } sync_point(12);
            if (two) { LL_OPAQUE(); ElimRange<T, 8, 12>::run(a, invd + 32, bad, lane); } sync_point(13);
            if (two) { LL_OPAQUE(); ElimRange<T, 12, 16>::run(a, invd + 32, bad, lane); } sync_point(14);
            if (two) { LL_OPAQUE(); ElimRange<T, 16, 20>::run(a, invd + 32, bad, lane); } sync_point(15);
            if (two) { LL_OPAQUE(); ElimRange<T, 20, 24>::run(a, invd + 32, bad, lane); } sync_point(16);
            if (two) { LL_OPAQUE(); ElimRange<T, 24, 28>::run(a, invd + 32, bad, lane); } sync_point(17);
            if (two) { LL_OPAQUE(); ElimRange<T, 28, 32>::run(a, invd + 32, bad, lane); } sync_point(18);
            {   // the -L10 images move from Dimg into their Zimg operand slots (always behind P0: quantum 18 >= the slot-0 cap)
                LL_OPAQUE();
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) {
                    const int off = tri_idx(2 + (blk >> 1), blk & 1) * BLK;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<T*>(Zimg + off + (q * 64 + lane) * ES) = *reinterpret_cast<const T*>(Dimg + off + (q * 64 + lane) * ES);
                }
            } sync_point(19);
            if (two) { LL_OPAQUE(); LL_LDSWAIT(); after_elim(1, lane, nq, c0q); InvRange<T, 31, 24>::run(a, invd + 32, rr); } sync_point(20);
            if (two) { LL_OPAQUE(); InvRange<T, 23, 16>::run(a, invd + 32, rr); } sync_point(21);
            if (two) { LL_OPAQUE(); InvRange<T, 15, 8>::run(a, invd + 32, rr); } sync_point(22);
            if (two) { LL_OPAQUE(); InvRange<T, 7, 0>::run(a, invd + 32, rr); } sync_point(23);
            if (two) { LL_OPAQUE(); after_inv(1, rr, lane, nq, c0q); } sync_point(24);
            // the inverse of L00 last: nothing on this wave's path needs it (L10 came out of the panel elimination), the bulk waves need
            // it at X1 -- and since the panel elimination this wave is no longer the kernel's critical path
            { LL_OPAQUE(); load_rows(2, lane); InvRange<T, 31, 24>::run(a, invd, rr); } sync_point(25);
            { LL_OPAQUE(); InvRange<T, 23, 16>::run(a, invd, rr); } sync_point(26);
            { LL_OPAQUE(); InvRange<T, 15, 8>::run(a, invd, rr); } sync_point(27);
            { LL_OPAQUE(); InvRange<T, 7, 0>::run(a, invd, rr); } sync_point(28);
            { LL_OPAQUE(); after_inv(0, rr, lane, nq, c0q); } sync_point(29);
#undef LL_OPAQUE
            LL_BAR();                                             // X1: Zimg complete
            LL_BAR();                                             // X2
            LL_BAR();                                             // X3: Dimg holds the next diagonal block
        }
        logdet_part = subwave_sum<T>(logdet_part, 64);
        if (lane == 0) { flg[1] = logdet_part; if (bad) flg[0] = T(1); }
#ifdef PACOH_LL_STAMPS
        if (lane == 0 && blockIdx.x == 0)
            printf("chain (us): total %.1f | in quanta %.1f | per column %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f\n", (wall_clock64() - st_t0) * 0.01, st_q * 0.01,
                   st_qc[0] * 0.01, st_qc[1] * 0.01, st_qc[2] * 0.01, st_qc[3] * 0.01, st_qc[4] * 0.01, st_qc[5] * 0.01, st_qc[6] * 0.01, st_qc[7] * 0.01);
#endif
#ifdef LL_X_HELPER
    } else if (false) {
#else
    } else if (is_helper) {
#endif
        // =========================================================== HELPERS =========================================================
        T sacc = 0;                                               // wave 4: (L[j, 0:j] u)_lane
#ifdef PACOH_LL_STAMPS
        long long sh_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sh_t = wall_clock64();
#define HST(k) do { const long long t_ = wall_clock64(); sh_[k] += t_ - sh_t; sh_t = t_; } while (0)
#else
#define HST(k) do {} while (0)
#endif
        for (int j = 0; j < ncol; ++j) {
            const ColCfg cf = col_cfg<T>(j, n, S0, S1);
            const int c0 = cf.c0;
            const int ppr_sh = cf.RB == 128 ? 3 : 4;              // log2(pieces per row)
            // LDS-DMA of slab s: rows c0.., columns s KS .. + KS.  Instruction v fills rows v * rpi .. + rpi (rpi = 8 at 128-byte rows, 4 at
            // 256).  Its per-lane source address = a wave-uniform base (row block, slab column) + a lane offset that depends on v only
            // through the low bits of v in the XOR swizzle: one v_xor and a 64-bit add per instruction (the general index arithmetic
            // cost as much again as the instructions' own issue, 60-185 cycles each: MI355X_MICROARCH.md).  Dealing the instructions to
            // all fifteen waves beside the chain was measured and lost: inside the MFMA loop an issue costs the bulk waves more than
            // the helpers' waiting costs them (slab loop 245 -> 270 us).
            const int rpi = 64 >> ppr_sh;
            const int lrow = lane >> ppr_sh, lpp = lane & ((1 << ppr_sh) - 1);
            const int lsig = cf.RB == 128 ? (lrow >> 1) : lrow;
            const unsigned vrow0 = (unsigned)lrow * (unsigned)n * ES, vpc0 = (unsigned)((lpp ^ lsig) << 4);
            const int ni_all = (cf.mrows << ppr_sh) >> 6;
            const bool ragged = (cf.m & (rpi - 1)) != 0;          // last instruction reaches beyond the matrix: clamp its rows
            auto issue_cfg = [&](int s, unsigned char* buf, int cc0, int KS, int ni, int mm, bool rag, unsigned vr0, unsigned vp0, int rb) __attribute__((always_inline)) {
                const unsigned char* const colb = reinterpret_cast<const unsigned char*>(Ab + (size_t)cc0 * n + (size_t)s * KS);
                const int vmask = rb == 128 ? 1 : 3;
                const int rp = rb == 128 ? 8 : 4;
                for (int v = hidx; v < ni; v += 3) {
                    unsigned vo = vr0 + (vp0 ^ (unsigned)((v & vmask) << 6));
                    if (rag && v == ni - 1) {                     // rows beyond the matrix read the last row (never stored)
                        const int i = v * rp + (rb == 128 ? (lane >> 3) : (lane >> 4));
                        if (i >= mm) vo -= (unsigned)(i - (mm - 1)) * (unsigned)n * ES;
                    }
                    glds16(colb + (size_t)v * rp * n * ES + vo, buf + v * 1024);
                }
            };
            auto issue = [&](int s, unsigned char* buf) __attribute__((always_inline)) { issue_cfg(s, buf, c0, cf.KS, ni_all, cf.m, ragged, vrow0, vpc0, cf.RB); };
            auto dot = [&](int s, const unsigned char* buf) __attribute__((always_inline)) {     // wave 4: sacc += slab row `lane` . u[k0 ..]
                if (hidx != 0) return;
                const int sig = cf.RB == 128 ? ((lane >> 1) & 7) : (lane & 15);
                const unsigned char* rowp = buf + lane * cf.RB;
                const int k0 = s * cf.KS;
                const int np = cf.RB >> 4;
                for (int p = 0; p < np; ++p) {
                    const T* pc = reinterpret_cast<const T*>(rowp + ((p ^ sig) << 4));
#pragma unroll
                    for (int e = 0; e < 16 / ES; ++e) sacc = fma(pc[e], uvec[k0 + p * (16 / ES) + e], sacc);
                }
            };
            sacc = 0;
            LL_BAR();                                             // P0
            // ring of cf.nb slab buffers, prefetch distance D = nb - 1.  The helpers' barrier inside the loop is a RAW s_barrier behind a
            // COUNTED vmcnt: __syncthreads() would drain every DMA in flight, i.e. wait out the fetch latency of the slab issued a moment
            // ago in every iteration (cdna_hip_programming.md, "Pipelining across barriers") -- at the thin late columns an iteration's
            // MFMA work is shorter than that latency
            const int D = cf.nb > 1 ? cf.nb - 1 : 0;
            const int per = ni_all > hidx ? (ni_all - hidx + 2) / 3 : 0;        // this helper's DMA instructions per slab
            for (int s = (j < 2 ? 0 : 1); s < D && s < cf.ns; ++s) issue(s, slab0 + s * cf.bsz);   // (slab 0 of columns >= 2: behind the previous X1)
            if (D == 0 && cf.ns > 0 && j < 2) issue(0, slab0);
            LL_BAR();                                             // P1
            for (int s = 0; s < cf.ns; ++s) {
                unsigned char* cur = slab0 + (s % cf.nb) * cf.bsz;
                if (cf.nb > 1) {
                    HST(0);
                    if (s + D < cf.ns) issue(s + D, slab0 + ((s + D) % cf.nb) * cf.bsz);
                    HST(1);
                    dot(s, cur);
                    HST(2);
                    int keep = cf.ns - 2 - s;                     // slabs s + 2 .. s + D may stay in flight across the barrier
                    keep = keep > D - 1 ? D - 1 : keep;
                    wait_vmcnt(keep > 0 ? keep * per : 0);
                    HST(3);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    HST(4);
                } else {
                    dot(s, cur);
                    LL_BAR();
                    if (s + 1 < cf.ns) issue(s + 1, slab0);
                    LL_BAR();
                }
            }
            LL_BAR();                                             // X1
            if (j >= 1 && j + 1 < ncol) {                         // slab 0 of column j + 1 (slab0 is free from here on; its rows: a subset)
                const ColCfg cn = col_cfg<T>(j + 1, n, S0, S1);
                if (cn.ns > 0) {
                    const int sh = cn.RB == 128 ? 3 : 4, rp = 64 >> sh;
                    const int lr = lane >> sh, lp = lane & ((1 << sh) - 1), ls_ = cn.RB == 128 ? (lr >> 1) : lr;
                    issue_cfg(0, slab0, cn.c0, cn.KS, (cn.mrows << sh) >> 6, cn.m, (cn.m & (rp - 1)) != 0,
                              (unsigned)lr * (unsigned)n * ES, (unsigned)((lp ^ ls_) << 4), cn.RB);
                }
            }
            // Only what must leave Dimg before the look-ahead owners overwrite it (behind X2) happens in front of X2 -- the priority
            // row blocks' solve is 2 us, and every wave of the workgroup waits here: helper h' = hidx - 1 takes the three blocks of
            // L(h',h') into registers.  Everything else (copy-out of Z and L10, forward solve, first-column row blocks) reads Zimg,
            // which the chain rewrites only behind the next P0.
            // (lane ids made opaque inside the column loop, as in the chain: the ~100 LDS addresses of the unrolled code below are
            //  loop invariant and would be hoisted and spilled -- 87 registers, reloaded next to every slab barrier)
            int lane = threadIdx.x & 63;
            asm volatile("" : "+v"(lane));
            const int r = lane & 15, g = lane >> 4;
            T lreg[3][4];
            if (hidx != 0) {
                const int hh = hidx - 1;
#pragma unroll
                for (int av = 0; av < 2; ++av)
#pragma unroll
                    for (int b = 0; b <= av; ++b) {
                        const T* lp = reinterpret_cast<const T*>(Dimg + tri_idx(2 * hh + av, 2 * hh + b) * BLK) + lane;
#pragma unroll
                        for (int q = 0; q < 4; ++q) lreg[av + b][q] = lp[q * 64];
                    }
            }
            LL_BAR();                                             // X2
            LL_BAR();                                             // X3
            if (hidx != 0) {
                // L(h',h'), the strictly lower part of Z(h',h') transposed into the block's upper triangle, half of L10
                const int hh = hidx - 1;
                const int k0 = c0 + 32 * hh;
#pragma unroll
                for (int av = 0; av < 2; ++av)
#pragma unroll
                    for (int b = 0; b <= av; ++b) {
                        const T* zp = reinterpret_cast<const T*>(Zimg + tri_idx(2 * hh + av, 2 * hh + b) * BLK) + lane;
                        const int i = 16 * av + r;                // row inside the 32-block
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int c = 16 * b + Mf<T>::row(g, q);
                            if (c <= i && k0 + i < n) Ab[(size_t)(k0 + i) * n + k0 + c] = lreg[av + b][q];
                            if (c < i && k0 + i < n) Ab[(size_t)(k0 + c) * n + k0 + i] = zp[q * 64];
                        }
                    }
#pragma unroll
                for (int b = 0; b < 2; ++b) {                     // L10 block (hh, b): rows c0 + 32 + 16 hh + r
                    const T* lp = reinterpret_cast<const T*>(Zimg + tri_idx(2 + hh, b) * BLK) + lane;
                    const int row = c0 + 32 + 16 * hh + r;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (row < n) Ab[(size_t)row * n + c0 + 16 * b + Mf<T>::row(g, q)] = -lp[q * 64];
                }
            }
            if (hidx == 0 && cf.ns == 0 && c0 > 0) {              // no slabs in this column: L[j, 0:j] u straight from the rows in global memory
                const int row = c0 + lane;
                const T* lp = Ab + (size_t)(row < n ? row : n - 1) * n;
                T s0 = 0, s1 = 0, s2 = 0, s3 = 0;
                for (int k = 0; k < c0; k += 4) {
                    s0 = fma(lp[k], uvec[k], s0); s1 = fma(lp[k + 1], uvec[k + 1], s1);
                    s2 = fma(lp[k + 2], uvec[k + 2], s2); s3 = fma(lp[k + 3], uvec[k + 3], s3);
                }
                sacc = (s0 + s1) + (s2 + s3);
            }
            if (hidx == 0) {                                      // u_j = Z_jj (r_j - L[j, 0:j] u)
                const int row = c0 + lane;
                const T rhs = row < n ? resid[(size_t)blockIdx.x * n + row] - sacc : T(0);
                tmpv[lane] = rhs;
                LL_LDSWAIT();
                const int li = lane & 31, av = li >> 4;
                T u = 0;
                if (lane < 32) {
#pragma unroll
                    for (int c = 0; c < 32; ++c)
                        if ((c >> 4) <= av) u = fma(*reinterpret_cast<const T*>(Zimg + tri_idx(av, c >> 4) * BLK + img_off<T>(c & 15, li & 15)), tmpv[c], u);
                    uvec[c0 + lane] = u;
                }
                LL_LDSWAIT();
                if (lane >= 32) {
                    T t = tmpv[lane];                             // r_hi - L10 u_lo
#pragma unroll
                    for (int c = 0; c < 32; ++c)
                        t = fma(*reinterpret_cast<const T*>(Zimg + tri_idx(2 + av, c >> 4) * BLK + img_off<T>(c & 15, li & 15)), uvec[c0 + c], t);
                    tmpv[lane] = t;
                }
                LL_LDSWAIT();
                if (lane >= 32) {
#pragma unroll
                    for (int c = 0; c < 32; ++c)
                        if ((c >> 4) <= av) u = fma(*reinterpret_cast<const T*>(Zimg + tri_idx(2 + av, 2 + (c >> 4)) * BLK + img_off<T>(c & 15, li & 15)), tmpv[32 + c], u);
                    uvec[c0 + lane] = row < n ? u : T(0);
                }
            }
        }
#ifdef PACOH_LL_STAMPS
        HST(0);
        if (lane == 0 && blockIdx.x == 0)
            printf("helper %d (us): outside slab loop %.1f | DMA issue %.1f | dot %.1f | vmcnt wait %.1f | barrier wait %.1f\n", hidx, sh_[0] * 0.01, sh_[1] * 0.01, sh_[2] * 0.01, sh_[3] * 0.01, sh_[4] * 0.01);
#endif
#undef HST
#ifdef LL_X_BULK
    } else if (false) {
#else
    } else {
#endif
        // =========================================================== BULK ============================================================
        Acc acc0[4], acc1[4], la;
        const bool la_owner = bw >= 2;
        int lacb = 4, laib = 4;
        if (la_owner) { int cbr, ibr; la_coords(bw - 2, cbr, ibr); lacb = 4 + cbr; laib = 4 + ibr; }
#ifdef PACOH_LL_STAMPS
        long long sb_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sb_t = wall_clock64();
#define BST(k) do { const long long t_ = wall_clock64(); sb_[k] += t_ - sb_t; sb_t = t_; } while (0)
#else
#define BST(k) do {} while (0)
#endif
        for (int j = 0; j < ncol; ++j) {
            const ColCfg cf = col_cfg<T>(j, n, S0, S1);
            const int c0 = cf.c0;
            const int ib0 = 4 + bw, ib1 = 4 + bw + LL_NBULK;      // the wave's row blocks (of 16 rows) in this column
            const bool v0 = bw < cf.R, v1 = bw + LL_NBULK < cf.R;
            const bool has_next = cf.m > LLW;
            const bool do_la = la_owner && has_next;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) { acc0[cb] = Acc{0, 0, 0, 0}; acc1[cb] = Acc{0, 0, 0, 0}; }
            la = Acc{0, 0, 0, 0};
            // A into the accumulators: the loads fly while the column's first slab is fetched (P0 / P1)
            if (v0) load_a(acc0, c0, ib0);
            if (v1) load_a(acc1, c0, ib1);
            if (do_la) {
                const int row = c0 + 16 * laib + r;
#pragma unroll
                for (int q = 0; q < 4; ++q) {                   // (clamped, not padded: the padding is applied when the block is handed on)
                    const int col = c0 + 16 * lacb + Mf<T>::row(g, q);
                    la[q] = Ab[(size_t)(row < n ? row : n - 1) * n + (col < n ? col : n - 1)];
                }
            }
            // accumulators -= (top rows) (own rows)^T over one slab.  The wave's shape (second row block? look-ahead block?) is a pair of
            // compile-time flags chosen once per slab: as wave-uniform branches inside the k loop they cut every step's ds_read / MFMA
            // stream into three basic blocks, and the waves with a look-ahead block ran 30 % longer for 12 % more MFMAs
            auto slab_body = [&](auto v1c, auto lac, const unsigned char* buf) __attribute__((always_inline)) {
                constexpr bool V1 = decltype(v1c)::value, LA = decltype(lac)::value;
                const int RB = cf.RB;
                const int sig = RB == 128 ? ((r >> 1) & 7) : r;
                // piece index of element kk = 16 t + row(g, s) is const(t, s) | lane bits; see the file header
                const int lbits = (ES == 8) ? ((g >> 1) ^ sig) : (g ^ sig);
                const int lx = (lbits << 4) | (ES == 8 ? (g & 1) * 8 : 0);
                const unsigned char* const pa = buf + r * RB;                     // top rows (A operands): + cb * 16 * RB
                const unsigned char* const pb0 = pa + ib0 * 16 * RB;
                const unsigned char* const pb1 = pa + ib1 * 16 * RB;
                const unsigned char* const pla = pa + lacb * 16 * RB;
                const unsigned char* const plb = pa + laib * 16 * RB;
                const int nchunk = cf.KS >> 4;
                for (int t = 0; t < nchunk; ++t) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int cst = (ES == 8) ? ((8 * t + 2 * s) << 4) : ((4 * t) << 4);
                        const int xo = (lx ^ cst) + ((ES == 8) ? 0 : 4 * s);
                        T av[4];
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) av[cb] = -*reinterpret_cast<const T*>(pa + cb * 16 * RB + xo);
                        const T b0 = *reinterpret_cast<const T*>(pb0 + xo);
                        T b1 = 0, aa = 0, bb = 0;
                        if constexpr (V1) b1 = *reinterpret_cast<const T*>(pb1 + xo);
                        if constexpr (LA) { aa = -*reinterpret_cast<const T*>(pla + xo); bb = *reinterpret_cast<const T*>(plb + xo); }
#pragma unroll
                        for (int cb = 0; cb < 4; ++cb) acc0[cb] = Mf<T>::mma(av[cb], b0, acc0[cb]);
                        if constexpr (V1) {
#pragma unroll
                            for (int cb = 0; cb < 4; ++cb) acc1[cb] = Mf<T>::mma(av[cb], b1, acc1[cb]);
                        }
                        if constexpr (LA) la = Mf<T>::mma(aa, bb, la);
                    }
                }
            };
            auto slab_la_only = [&](const unsigned char* buf) __attribute__((always_inline)) {      // (a wave without row blocks: thin columns)
                const int RB = cf.RB;
                const int sig = RB == 128 ? ((r >> 1) & 7) : r;
                const int lbits = (ES == 8) ? ((g >> 1) ^ sig) : (g ^ sig);
                const int lx = (lbits << 4) | (ES == 8 ? (g & 1) * 8 : 0);
                const unsigned char* const pla = buf + (lacb * 16 + r) * RB;
                const unsigned char* const plb = buf + (laib * 16 + r) * RB;
                const int nchunk = cf.KS >> 4;
                for (int t = 0; t < nchunk; ++t) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int cst = (ES == 8) ? ((8 * t + 2 * s) << 4) : ((4 * t) << 4);
                        const int xo = (lx ^ cst) + ((ES == 8) ? 0 : 4 * s);
                        la = Mf<T>::mma(-*reinterpret_cast<const T*>(pla + xo), *reinterpret_cast<const T*>(plb + xo), la);
                    }
                }
            };
            auto slab_mma = [&](const unsigned char* buf) __attribute__((always_inline)) {
                if (v0) {
                    if (v1) { if (do_la) slab_body(std::true_type{}, std::true_type{}, buf); else slab_body(std::true_type{}, std::false_type{}, buf); }
                    else { if (do_la) slab_body(std::false_type{}, std::true_type{}, buf); else slab_body(std::false_type{}, std::false_type{}, buf); }
                } else if (do_la) slab_la_only(buf);
            };
            BST(5);
            LL_BAR();                                             // P0
            BST(0);
            LL_BAR();                                             // P1
            BST(6);
            for (int s = 0; s < cf.ns; ++s) {
                slab_mma(slab0 + (s % cf.nb) * cf.bsz);
                BST(1);
                LL_BAR();
                if (cf.nb == 1) LL_BAR();
                BST(2);
            }
            BST(3);
            LL_BAR();                                             // X1: the chain's Zimg is complete
            BST(4);
            auto publish = [&](const Acc (&X)[4], int ib) __attribute__((always_inline)) {       // rows 64..127 of the column: operand images for the look-ahead update
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<T*>(Limg + ((ib - 4) * 4 + cb) * BLK + (q * 64 + lane) * ES) = X[cb][q];
            };
            // pass 0: the rows of the next diagonal block (they gate the chain), pass 1: everything else
            const bool pr0 = v0 && ib0 < 8;                       // (ib1 >= 16: never a priority block)
            if (pr0) { solve(acc0); publish(acc0, ib0); }
            BST(7);
            LL_BAR();                                             // X2: Limg complete
            BST(8);
            if (do_la) {                                          // next diagonal block = la - L21 L21^T over this column's 64 columns
                Acc sacc = {0, 0, 0, 0};
                const T* pa = reinterpret_cast<const T*>(Limg + (lacb - 4) * 4 * BLK) + lane;
                const T* pb = reinterpret_cast<const T*>(Limg + (laib - 4) * 4 * BLK) + lane;
#pragma unroll
                for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                    for (int s = 0; s < 4; ++s) sacc = Mf<T>::mma(pa[kb * 256 + s * 64], pb[kb * 256 + s * 64], sacc);
                const int row = c0 + 16 * laib + r;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = c0 + 16 * lacb + Mf<T>::row(g, q);
                    const T v = (row < n && col < n) ? la[q] - sacc[q] : (row == col ? T(1) : T(0));
                    *reinterpret_cast<T*>(Dimg + tri_idx(laib - 4, lacb - 4) * BLK + (q * 64 + lane) * ES) = v;
                }
            }
            BST(9);
            LL_BAR();                                             // X3: Dimg holds the next diagonal block
            BST(10);
            if (pr0) store_l(acc0, c0, ib0);
            if (v0 && !pr0) { solve(acc0); store_l(acc0, c0, ib0); }
            if (v1) { solve(acc1); store_l(acc1, c0, ib1); }
            // first column (nothing accumulated: 28 row blocks at n = 512 against 24 slots): a third row block for the first waves
            if (cf.ns == 0 && bw + 2 * LL_NBULK < cf.R) {
                load_a(acc0, c0, ib1 + LL_NBULK);
                solve(acc0);
                store_l(acc0, c0, ib1 + LL_NBULK);
            }
            BST(11);
        }
#ifdef PACOH_LL_STAMPS
        BST(5);
        if ((bw == 0 || bw == 11) && lane == 0 && blockIdx.x == 0)
            printf("bulk %d (us): load -A issue %.1f | P0 wait %.1f | P1 wait %.1f | slab mma %.1f | slab barrier %.1f | X1 wait %.1f | pass 0 %.1f | X2 wait %.1f | LA %.1f | X3 wait %.1f | pass 1 + stores %.1f\n", bw,
                   sb_[5] * 0.01, sb_[0] * 0.01, sb_[6] * 0.01, sb_[1] * 0.01, sb_[2] * 0.01, (sb_[3] + sb_[4]) * 0.01, sb_[7] * 0.01, sb_[8] * 0.01, sb_[9] * 0.01, sb_[10] * 0.01, sb_[11] * 0.01);
#endif
#undef BST
    }
#ifdef PACOH_LL_STAMPS
    if (wave == 1 && lane == 0 && blockIdx.x == 0) {
        // (bulk wave 0's stamps live in its branch: re-read through LDS is not worth it -- printed there)
    }
#endif
    LL_BAR();

    // ---- log-density, u / alpha ------------------------------------------------------------------------------------------------
    T* const rv = uvec;
    T quad_part = 0;
    for (int q = tid; q < n; q += LL_NT) quad_part = fma(rv[q], rv[q], quad_part);
    quad_part = subwave_sum<T>(quad_part, 64);
    if (lane == 0) flg[2 + wave] = quad_part;
    LL_BAR();
    T quad = 0;
    for (int w = 0; w < LL_NT / 64; ++w) quad += flg[2 + w];
    const T logdet = flg[1];
    const bool ok = flg[0] == T(0);
    if (tid == 0) {
        const T LOG2PI = T(1.8378770664093453);
        const T lp = T(-0.5) * (quad + T(2) * logdet + T(n) * LOG2PI) * scale;
        logp[blockIdx.x] = ok ? lp : T(NAN);
        if (info) info[blockIdx.x] = ok ? attempt : -1;
    }
    if (!alpha_out) return;
    if (u_only) {      // the caller goes on to Z = L^-1 and takes alpha = Z^T u from there (dense_alpha_kernel)
        for (int q = tid; q < n; q += LL_NT) alpha_out[(size_t)blockIdx.x * n + q] = ok ? rv[q] : T(NAN);
        return;
    }
    // ---- backward solve L^T alpha = u, blocked from the bottom with the saved inverse diagonal blocks (as dense_mfma.hip) ------
    T (*Db[2])[DNB + 1] = {reinterpret_cast<T (*)[DNB + 1]>(slab0), reinterpret_cast<T (*)[DNB + 1]>(slab0 + DNB * (DNB + 1) * ES)};
    T* const red = reinterpret_cast<T*>(slab0 + 2 * DNB * (DNB + 1) * ES);      // [32]
    const int nblk = (n + DNB - 1) / DNB;
    auto fetch_diag = [&](int kbk, T (*D)[DNB + 1]) __attribute__((always_inline)) {
        const int k0 = kbk * DNB;
        const int kb = (n - k0 < DNB) ? (n - k0) : DNB;
        for (int q = tid; q < DNB * DNB; q += LL_NT) {
            const int i = q / DNB, c = q - i * DNB;
            D[i][c] = (i < kb && c < kb) ? Ab[(size_t)(k0 + i) * n + k0 + c] : T(0);
        }
    };
    LL_BAR();
    fetch_diag(nblk - 1, Db[(nblk - 1) & 1]);
    for (int kbk = nblk - 1; kbk >= 0; --kbk) {
        const int k0 = kbk * DNB;
        const int kb = (n - k0 < DNB) ? (n - k0) : DNB;
        T (*D)[DNB + 1] = Db[kbk & 1];
        LL_BAR();
        {
            const int t = tid >> 5, c = tid & 31;                 // one product per thread, summed over the 32 lanes of a row
            T pr = 0;
            if (t < kb && c < kb) {
                if (c == t) pr = rv[k0 + t] / D[t][t];
                else if (c > t) pr = D[t][c] * rv[k0 + c];
            }
            pr = subwave_sum<T>(pr, 32);
            if (c == 0 && t < kb) red[t] = pr;
        }
        LL_BAR();
        if (tid < kb) rv[k0 + tid] = red[tid];
        LL_BAR();
        if (kbk > 0) fetch_diag(kbk - 1, Db[(kbk - 1) & 1]);
        for (int i = tid; i < k0; i += LL_NT) {
            T sacc = rv[i];
            if (kb == DNB) {
#pragma unroll
                for (int c = 0; c < DNB; ++c) sacc = fma(-Ab[(size_t)(k0 + c) * n + i], rv[k0 + c], sacc);
            } else {
                for (int c = 0; c < kb; ++c) sacc = fma(-Ab[(size_t)(k0 + c) * n + i], rv[k0 + c], sacc);
            }
            rv[i] = sacc;
        }
    }
    LL_BAR();
    for (int q = tid; q < n; q += LL_NT) alpha_out[(size_t)blockIdx.x * n + q] = ok ? rv[q] : T(NAN);
}

// LDS plan for matrices of size n: slab areas S0 / S1 (S1 also holds the 16 look-ahead operand images).  false: not eligible.
template <typename T>
__global__ void __launch_bounds__(LL_NT) chol_ll_kernel(T* __restrict__ A, const T* __restrict__ resid, T* __restrict__ logp,
                                                         T* __restrict__ alpha_out, int32_t* __restrict__ info, T scale, int n,
                                                         int attempt, int u_only, int S0, int S1) {
    if (attempt > 0 && info && info[blockIdx.x] >= 0) return;      // jitter-ladder retry: only the failed problems
    chol_ll_body<T>(A, resid, logp, alpha_out, info, scale, n, attempt, u_only, S0, S1);
}

// ---- the WHOLE jitter ladder in one launch (the large-context GP path; dense_gp.hip) -----------------------------------------------
// gpytorch's psd_safe_cholesky retries a failed factorisation with 1e-6 / 1e-8 * 10^k added to the diagonal.  As separate launches
// that is, per call, three re-Gram launches and three factorisation launches that exit at once when nothing failed -- 6 x 4.5-5.9 us.
// Here a failed problem's workgroup rebuilds its own matrix (what regram_failed_kernel + dense_mask_kernel wrote: direct
// differences, kern_val) and factors it again, rung after rung; a problem that did not fail costs one early exit.
struct LLRegen {
    const void* z; const void* ls; const void* os; const void* noise; const int32_t* n_valid;
    double jitter_base;
    int z_div, y_div, P, f, kind;
};

template <typename T>
__global__ void __launch_bounds__(LL_NT) chol_ll_retry_kernel(T* __restrict__ A, const T* __restrict__ resid, T* __restrict__ logp,
                                                               T* __restrict__ alpha_out, int32_t* __restrict__ info, T scale, int n,
                                                               int att_lo, int att_hi, int u_only, int S0, int S1, LLRegen rg) {
    const long b = blockIdx.x;
    // the usual case first, in front of everything the compiler hoists out of the rung loop (its spill stores ran before the loop's
    // own test: 91 MB of scratch writes and 15 us per launch with nothing to do)
    if (info[b] >= 0) return;
    asm volatile("" ::: "memory");
    for (int attempt = att_lo; attempt <= att_hi; ++attempt) {
        // (info[b] of the previous rung was written by this workgroup a moment ago: read it past the L1)
        if (__hip_atomic_load(&info[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 0) return;
        {
            const int p = (int)(b % rg.P), f = rg.f;
            const T* zb = (const T*)rg.z + (b / rg.z_div) * (long)n * f;
            const T* ls = (const T*)rg.ls + (long)p * f;
            const T osv = rg.os ? ((const T*)rg.os)[p] : T(1);
            double jit = rg.jitter_base;
            for (int q = 1; q < attempt; ++q) jit *= 10.0;
            const T dg = ((const T*)rg.noise)[p] + (T)jit;
            int nv = n;
            if (rg.n_valid) { nv = rg.n_valid[b / rg.y_div]; nv = nv < 0 ? 0 : (nv > n ? n : nv); }
            T* const Ab = A + (size_t)b * n * n;
            for (int i = threadIdx.x >> 6; i < n; i += LL_NT / 64) {
                T* row = Ab + (size_t)i * n;
                for (int j = threadIdx.x & 63; j < n; j += 64) {
                    T s = 0;
                    for (int c = 0; c < f; ++c) { const T d = zb[(long)i * f + c] / ls[c] - zb[(long)j * f + c] / ls[c]; s = fma(d, d, s); }
                    T v = osv * kern_val<T>(rg.kind, s) + (i == j ? dg : T(0));
                    if (i >= nv || j >= nv) v = (i == j) ? T(1) : T(0);
                    row[j] = v;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");         // the rebuilt matrix is in L2, this CU's L1 holds no stale lines of it
        __syncthreads();
        // (the body's loop-invariant addresses must not be hoisted out of the rung loop: 134 spilled registers, and a kernel with a
        //  large scratch segment takes 16 us to launch and exit where this one should take 4)
        int nq = n, s0q = S0, s1q = S1, uq = u_only;
        asm volatile("" : "+s"(nq), "+s"(s0q), "+s"(s1q), "+s"(uq));
        chol_ll_body<T>(A, resid, logp, alpha_out, info, scale, nq, attempt, uq, s0q, s1q);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
        __syncthreads();
    }
}

template <typename T>
bool ll_plan(int n, int* S0, int* S1, size_t* lds) {
    const int ES = sizeof(T), BLK = 256 * ES;
    if (n < 1 || ((size_t)n * ES) % 16 != 0) return false;          // 16-byte DMA pieces: rows must start 16-byte aligned
    const int mb1 = n > LLW ? (n - LLW + 15) / 16 : 0;               // row blocks of column 1, the tallest accumulated column
    if (mb1 - 4 > 2 * LL_NBULK) return false;                        // two row blocks per bulk wave
    const size_t fixed = ll_fixed_bytes<T>(n);
    const size_t cap = 160u * 1024u;
    const int m1 = n > LLW ? n - LLW : 0;
    int s0 = ((m1 + 7) & ~7) * 128;                                  // one slab of 128-byte rows
    if (s0 < 2 * DNB * (DNB + 1) * ES + 64 * ES) s0 = 2 * DNB * (DNB + 1) * ES + 64 * ES;   // (backward-solve scratch)
    int s1 = s0 < 16 * BLK ? 16 * BLK : s0;
    if (fixed + s0 + s1 > cap) {
        s1 = 16 * BLK;                                               // single-buffered slabs
        if (fixed + s0 + s1 > cap) return false;
    }
    *S0 = s0; *S1 = s1; *lds = fixed + s0 + s1;
    return true;
}

template <typename T>
int launch_ll(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n, int attempt,
              int u_only, hipStream_t s) {
    int S0 = 0, S1 = 0;
    size_t lds = 0;
    if (!ll_plan<T>(n, &S0, &S1, &lds)) return 1;
    auto kern = chol_ll_kernel<T>;
    if (lds > 64u * 1024u &&
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        // NOT "outside the plan" (1): the caller has already routed on dense_chol_saves_inverse(), which answers from ll_plan alone --
        // the fallback kernels ignore u_only and leave no inverse diagonal blocks, the gradients would be silently wrong
        (void)hipGetLastError();
        return PACOH_ELIMIT;
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(LL_NT), lds, s, (T*)A, (const T*)resid, (T*)logp, (T*)alpha_out, info, (T)scale, n, attempt,
                       u_only, S0, S1);
    return launch_status();
}

}  // namespace

bool dense_ll_fits(int n, int dtype) {
    int S0, S1;
    size_t lds;
    return dtype == PACOH_F32 ? ll_plan<float>(n, &S0, &S1, &lds) : ll_plan<double>(n, &S0, &S1, &lds);
}

// the rungs att_lo .. att_hi of the jitter ladder in ONE launch (see chol_ll_retry_kernel); 1: outside the plan
int dense_ll_retry_try(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n, int dtype,
                       int att_lo, int att_hi, int u_only, const void* z, int z_div, const void* ls, const void* os, const void* noise,
                       const int32_t* n_valid, int y_div, double jitter_base, int P, int f, int kind, hipStream_t s) {
    if (!info) return 1;
    const LLRegen rg = {z, ls, os, noise, n_valid, jitter_base, z_div, y_div, P, f, kind};
    int S0 = 0, S1 = 0;
    size_t lds = 0;
    if (dtype == PACOH_F32) {
        if (!ll_plan<float>(n, &S0, &S1, &lds)) return 1;
        auto kern = chol_ll_retry_kernel<float>;
        if (lds > 64u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return PACOH_ELIMIT; }
        hipLaunchKernelGGL(kern, dim3(B), dim3(LL_NT), lds, s, (float*)A, (const float*)resid, (float*)logp, (float*)alpha_out, info, (float)scale, n,
                           att_lo, att_hi, u_only, S0, S1, rg);
    } else {
        if (!ll_plan<double>(n, &S0, &S1, &lds)) return 1;
        auto kern = chol_ll_retry_kernel<double>;
        if (lds > 64u * 1024u && hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { (void)hipGetLastError(); return PACOH_ELIMIT; }
        hipLaunchKernelGGL(kern, dim3(B), dim3(LL_NT), lds, s, (double*)A, (const double*)resid, (double*)logp, (double*)alpha_out, info, scale, n,
                           att_lo, att_hi, u_only, S0, S1, rg);
    }
    return launch_status();
}

// returns 1 when the matrix size is outside this kernel's plan (caller falls back to dense_mfma.hip / dense.hip)
int dense_ll_try(void* A, const void* resid, void* logp, void* alpha_out, int32_t* info, double scale, int B, int n, int dtype,
                 int attempt, int u_only, hipStream_t s) {
    return dtype == PACOH_F32 ? launch_ll<float>(A, resid, logp, alpha_out, info, scale, B, n, attempt, u_only, s)
                              : launch_ll<double>(A, resid, logp, alpha_out, info, scale, B, n, attempt, u_only, s);
}

}  // namespace pacoh
