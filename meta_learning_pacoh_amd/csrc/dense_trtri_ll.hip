// Z = L^-1 in place (lower triangle) for the large-context path, LEFT-LOOKING by 64-row panels (round 4): one 1024-thread
// workgroup per matrix, the companion of dense_ll.hip (which leaves the inverse of every 32 x 32 diagonal block of L transposed
// in that block's upper triangle).
//
//   panel I (rows c0 = 64 I .. + 64):   Z[I, 0:I] = - Z_II * ( L[I, 0:I] * Z[0:I, 0:I] ),      Z[I, I] = Z_II = L[I,I]^-1
//
// The 64 x c0 product lives in MFMA accumulators (<= 8 blocks of 16 x 16 per wave, 14 waves) and is taken in slabs of 16 along k:
//   * the A operand of a slab -- the 64 x 16 piece of the panel's own L rows, which EVERY wave multiplies with -- goes through a ring
//     of fixed LDS slots by LDS-DMA from the two helper waves (L is pure input: one pipeline over all panels' slabs);
//   * the B operand -- a 16 x 16 block of a finished panel of Z -- is needed by exactly ONE wave, which loads its fragments straight
//     into registers a slab ahead (one register set, refreshed in place behind the MFMAs that read it).
// (The first version streamed the finished rows of Z through LDS too; LDS-DMA costs its issuing wave 85-750 ns per instruction, and
// the 3 200 instructions per matrix that took bounded the kernel: 511 us against 392 now.)  Nothing is written inside the loop.  There
// is no dependency chain at all -- Z_II is assembled from what the Cholesky left behind (two 32 x 32 inverses + L10) -- so every
// SIMD runs MFMAs.  The right-looking kernel this replaces (trtri_dense_kernel) re-read the inverted trailing matrix from
// L2 / HBM for every 32-column panel with per-lane 8-byte loads: 1.9 GB per 256 x 512^2 fp64 launch, 0.46 ms.
// Slabs are taken in DESCENDING k order, the previous panel's four blocks last (those rows were stored a moment ago).
// Roles: waves 0 and 1 = helpers (LDS-DMA issue, the operand images of Z_II, the diagonal block itself),
// waves 2..15 = MFMA waves; MFMA wave w owns column blocks w and NB - 1 - w of the panel (Z is lower triangular: column block
// q only meets the slabs k >= q, so the pair's work is the same for every w).
// Reference semantics: the explicit inverse gpytorch's inv_quad_logdet / torch.cholesky_inverse produce on the way to K^-1
// (meta_learn/random_gp.py:83-85 backward at the large-context configuration).
#include "common.h"
#include "dense_diag.h"
#include <type_traits>

namespace pacoh {
namespace {

constexpr int TL_NT = 1024;
constexpr int TL_NMMA = 14;
constexpr int TL_G = 2;              // slabs per barrier of the slab loop (1: 409 us, 2: 392 us, 4: 493 us at 256 x 512^2 fp64)

__host__ __device__ constexpr int tl_tri(int a, int b) { return a * (a + 1) / 2 + b; }

__device__ __forceinline__ void tl_glds16(const void* src, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}
__device__ __forceinline__ void tl_wait_vmcnt(int n) {
#define TL_VMC(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        TL_VMC(0) TL_VMC(1) TL_VMC(2) TL_VMC(3) TL_VMC(4) TL_VMC(5) TL_VMC(6) TL_VMC(7) TL_VMC(8) TL_VMC(9) TL_VMC(10) TL_VMC(11) TL_VMC(12)
        TL_VMC(13) TL_VMC(14) TL_VMC(15) TL_VMC(16) TL_VMC(17) TL_VMC(18) TL_VMC(19) TL_VMC(20) TL_VMC(21) TL_VMC(22) TL_VMC(23) TL_VMC(24)
        TL_VMC(25) TL_VMC(26) TL_VMC(27) TL_VMC(28) TL_VMC(29) TL_VMC(30) TL_VMC(31) TL_VMC(32) TL_VMC(33) TL_VMC(34) TL_VMC(35) TL_VMC(36)
        TL_VMC(37) TL_VMC(38) TL_VMC(39) TL_VMC(40)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef TL_VMC
}

template <typename T> __device__ __forceinline__ constexpr int tl_img_off(int c15, int r) {
    return (Mf<T>::q_of(c15) * 64 + 16 * Mf<T>::g_of(c15) + r) * (int)sizeof(T);
}

// LDS: Zd images (10 blocks) | u, alpha partial sums | ring of <= 16 L slabs: 64 rows x 128 bytes (fp32: 64) each, XOR-swizzled pieces
template <typename T>
__global__ void __launch_bounds__(TL_NT) trtri_ll_kernel(T* __restrict__ A, const int32_t* __restrict__ info, int n, int ring_bytes,
                                                         const T* __restrict__ u, T* __restrict__ alpha) {
    // u / alpha (optional): alpha = Z^T u = A^-1 r on the fly -- every finished row block of Z is in registers once; the separate
    // pass over Z that computed it (dense_alpha_kernel) read the whole inverse again: 0.28 GB and 84 us per 256 x 512^2 launch
    if (info && info[blockIdx.x] < 0) {
        if (alpha) for (int q = threadIdx.x; q < n; q += TL_NT) alpha[(size_t)blockIdx.x * n + q] = T(NAN);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    constexpr int ES = sizeof(T), BLK = 256 * ES;
    constexpr int KS = 16;                                        // rows of a Z slab = columns of an L slab
    constexpr int LRB = KS * ES;                                  // bytes per L slab row (128 fp64 / 64 fp32)
    using Acc = typename Mf<T>::acc;
    unsigned char* const Zdimg = sm;                              // -Z00 | +L10 | -Z11 operand images of the panel's diagonal block
    const int npad = (n + 63) & ~63;
    T* const uv = reinterpret_cast<T*>(sm + 10 * BLK);            // [npad] u (zero beyond n)
    T* const al_a = uv + npad;                                    // [npad] alpha: the panels' off-diagonal rows and the inverse 32-blocks
    T* const al_b = al_a + npad;                                  // [npad] ... the Z10 blocks (a writer of its own per entry: no races)
    unsigned char* const ring = reinterpret_cast<unsigned char*>(al_b + npad);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    T* const Ab = A + (size_t)blockIdx.x * n * n;
    const int npan = (n + 63) >> 6;
    const bool is_helper = wave < 2;
    const int mw = wave - 2;                                      // MFMA wave 0..13
    for (int q = tid; q < npad; q += TL_NT) {
        uv[q] = (u && q < n) ? u[(size_t)blockIdx.x * n + q] : T(0);
        al_a[q] = T(0); al_b[q] = T(0);
    }

    // Waves 0 / 1 assemble the operand images of Z_II = [Z00 0; Z10 Z11] from what the Cholesky left (two inverse 32-blocks,
    // transposed, in the upper triangles; 1 / diagonal; L10): -Z00 | +L10 | -Z11, one 32-block and one row block of L10 each.  The
    // loads for panel I + 1 are issued at the start of panel I's loop and sit in registers until that panel's epilogue has read the
    // images of panel I (these waves have no accumulators): loaded where they are needed they cost 20 us per panel, with every
    // other wave waiting at the barrier.
    auto run = [&](auto imgc) __attribute__((always_inline)) {
    constexpr bool IMG = decltype(imgc)::value;
    T pre[IMG ? 5 : 1][4];
    const int hw = wave & 1;                                      // helper wave hw: inverse block Z(hw,hw), row block hw of L10, column block hw of Z10
    auto img_load = [&](int I) __attribute__((always_inline)) {
        const int c0 = I << 6;
        const int last = n - 1;
        {
            const int k0 = c0 + 32 * hw;
#pragma unroll
            for (int av = 0; av < 2; ++av)
#pragma unroll
                for (int b = 0; b <= av; ++b) {
                    const int i = 16 * av + r;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = 16 * b + Mf<T>::row(g, q);
                        int row = k0 + (c < i ? c : i), col = k0 + i;             // (c < i: the transposed entry; else the diagonal)
                        row = row < last ? row : last; col = col < last ? col : last;
                        pre[av + b][q] = Ab[(size_t)row * n + col];
                    }
                }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            int row = c0 + 32 + 16 * hw + r;
            row = row < last ? row : last;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int col = c0 + 16 * b + Mf<T>::row(g, q);
                col = col < last ? col : last;
                pre[3 + b][q] = Ab[(size_t)row * n + col];
            }
        }
    };
    auto img_write = [&](int I) __attribute__((always_inline)) {
        const int c0 = I << 6;
        {
            const int k0 = c0 + 32 * hw;
#pragma unroll
            for (int av = 0; av < 2; ++av)
#pragma unroll
                for (int b = 0; b <= av; ++b) {
                    T* const zp = reinterpret_cast<T*>(Zdimg + tl_tri(2 * hw + av, 2 * hw + b) * BLK) + lane;
                    const int i = 16 * av + r;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = 16 * b + Mf<T>::row(g, q);
                        const T x = pre[av + b][q];
                        T v = T(0);
                        if (k0 + i < n) v = c < i ? x : (c == i ? T(1) / x : T(0));
                        else if (c == i) v = T(1);                // (rows beyond the matrix: identity)
                        zp[q * 64] = -v;
                    }
                }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            T* const lp = reinterpret_cast<T*>(Zdimg + tl_tri(2 + hw, b) * BLK) + lane;
            const int row = c0 + 32 + 16 * hw + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) lp[q * 64] = row < n ? pre[3 + b][q] : T(0);
        }
    };
    if constexpr (IMG) img_load(0);
#ifdef PACOH_LL_STAMPS
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = wall_clock64();
#define TST(k) do { const long long t_ = wall_clock64(); st_[k] += t_ - st_t; st_t = t_; } while (0)
#else
#define TST(k) do {} while (0)
#endif
    // Slab order of a panel with ns k-blocks: the blocks of the panels before the previous one first (descending), then the previous
    // panel's four -- those rows were stored a moment ago.
    auto kb_of = [](int ns, int t) -> int { return ns <= 4 ? ns - 1 - t : (t < ns - 4 ? ns - 5 - t : 2 * ns - 5 - t); };
    // ---- Operands.  The A operand of a slab -- the 64 x 16 piece L[c0 .. c0 + 64, 16 kb .. + 16] every wave multiplies with -- goes
    // through LDS: NLI LDS-DMA instructions per slab, issued by the two helper waves into a ring of NSLOT fixed slots; L is pure input
    // (a panel overwrites only its own rows), so the helpers run ONE continuous pipeline over all panels' slabs.  The B operand --
    // the 16 x 16 block Z[16 kb .. + 16, 16 qb .. + 16] of a finished panel -- is needed by exactly ONE wave (the owner of column
    // block qb): it loads its fragments straight into registers, one slab ahead, with plain 8-byte loads (16 lanes = 128 bytes of a
    // row).  The first left-looking version streamed the Z rows through LDS as well: 3 200 of its 4 096 DMA instructions per matrix,
    // and LDS-DMA turned out to cost 85-750 ns PER INSTRUCTION on the issuing wave (stamps: 77-196 us of "DMA issue" per wave, twice
    // the time the waves spent on MFMAs) -- the kernel was bound by the DMA path, not by the matrix cores.
    constexpr int PPR = LRB / 16;                                 // L slab: 16-byte pieces per row, 8 (fp64) / 4 (fp32)
    constexpr int RPI = 64 / PPR;                                 // ... rows per instruction
    constexpr int NLI = 64 / RPI;                                 // ... instructions per slab
    constexpr int LSB = 64 * LRB;                                 // ... bytes
    constexpr int HPI = NLI / 2;                                  // instructions per slab and helper wave
    const int nslot = (ring_bytes / LSB) < 16 ? (ring_bytes / LSB) : 16;
    // issue pointer over the global slab sequence (panel iI, slab it, global index gi); consume pointer gs
    int iI = 1, it = 0, gi = 0, gs = 0;
    auto issue_upto = [&](int lim) __attribute__((always_inline)) {          // helpers: bring slabs gi .. lim - 1
        while (gi < lim && iI < npan) {
            const int c0_ = iI << 6, ns_ = c0_ >> 4;
            const int pm_ = (n - c0_) < 64 ? (n - c0_) : 64;
            const int kb = kb_of(ns_, it);
            unsigned char* const dst = ring + (gi % nslot) * LSB;
#pragma unroll
            for (int h = 0; h < HPI; ++h) {
                const int v = 2 * h + hw;
                const int i = v * RPI + lane / PPR, pp = lane % PPR;
                const int sig = (ES == 8) ? ((i >> 1) & 7) : ((i >> 2) & 3);
                const int ic = i < pm_ ? i : pm_ - 1;
                tl_glds16(reinterpret_cast<const unsigned char*>(Ab + (size_t)(c0_ + ic) * n + 16 * kb) + ((pp ^ sig) << 4), dst + v * 1024);
            }
            ++gi;
            if (++it == ns_) { it = 0; ++iI; }
        }
    };
    for (int I = 0; I < npan; ++I) {
        const int c0 = I << 6;
        const int NBq = c0 >> 4;                                  // column blocks of the panel's product
        const int ns = NBq;                                       // slabs (16 columns of L x 16 rows of Z each)
        const int pm = (n - c0) < 64 ? (n - c0) : 64;             // rows of this panel inside the matrix

        TST(0);
        __syncthreads();                                          // previous panel's stores are visible, its image reads are over
        TST(1);
        if constexpr (IMG) { issue_upto(gs + nslot); img_write(I); }
        TST(2);
        __syncthreads();                                          // images written; (vmcnt(0): the slabs issued so far have landed)
        TST(3);
        if constexpr (IMG) { if (I + 1 < npan) img_load(I + 1); }           // (rows of L / inverse blocks no panel before I + 1 writes)

        Acc acc0[IMG ? 1 : 4], acc1[IMG ? 1 : 4];
        if constexpr (!IMG) {
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) { acc0[ib] = Acc{0, 0, 0, 0}; acc1[ib] = Acc{0, 0, 0, 0}; }
        }
        const int qb0 = mw, qb1 = NBq - 1 - mw;                   // this wave's column blocks (qb1 > qb0 when it has two)
        const bool v0 = !is_helper && mw < (NBq + 1) / 2, v1 = v0 && qb1 > qb0;
        // B fragments of slab kb for this wave's column blocks: unconditional loads at clamped (always valid) addresses -- whether a
        // block takes part is decided where the values are used.  ONE register set: fragment s of the NEXT slab is requested into the
        // register of fragment s of this slab right behind the MFMAs that read it, so every fragment has a whole slab's time to arrive
        // (a second set for ping-pong spilled 52 registers at the 128 this 1024-thread kernel has: the "load issue" phase then read
        // 249 us on the stamps -- scratch traffic).
        T bf0[IMG ? 1 : 4], bf1[IMG ? 1 : 4];
        auto frag_off = [&](int kb, int s, int& o0, int& o1) __attribute__((always_inline)) {
            const int q0 = qb0 <= kb ? qb0 : kb, q1 = (v1 && qb1 <= kb) ? qb1 : kb;
            const int orow = (16 * kb + Mf<T>::row(g, s)) * n + r;           // (32-bit element offsets from the wave-uniform base)
            o0 = orow + 16 * q0; o1 = orow + 16 * q1;
        };
        if constexpr (!IMG) {
            if (ns > 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s) { int o0, o1; frag_off(kb_of(ns, 0), s, o0, o1); bf0[s] = Ab[o0]; bf1[s] = Ab[o1]; }
            }
        }
        for (int t = 0; t < ns; ++t) {
            const int kb = kb_of(ns, t);
            const int kbn = t + 1 < ns ? kb_of(ns, t + 1) : kb;   // (last slab: a harmless reload of itself)
            const unsigned char* const lb = ring + (gs % nslot) * LSB;
            // (slabs are handed over in groups of TL_G: one barrier per group -- the ring is sixteen slabs deep)
            if constexpr (IMG) { if ((t & (TL_G - 1)) == 0) issue_upto(gs + nslot); }
            TST(7);
            // out[i][q] += L[c0 + i][16 kb + k] Z[16 kb + k][q]: column block qb takes part iff qb <= kb; in the diagonal block
            // (qb == kb) the entries above Z's diagonal are not Z (the Cholesky's inverse blocks live there): masked to zero
            const bool d0 = v0 && qb0 <= kb, d1 = v1 && qb1 <= kb;
            if constexpr (!IMG) {
                auto body = [&](auto c0c, auto c1c) __attribute__((always_inline)) {
                    constexpr bool C0 = decltype(c0c)::value, C1 = decltype(c1c)::value;
                    const int lsig = (ES == 8) ? ((r >> 1) & 7) : ((r >> 2) & 3);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int kk = Mf<T>::row(g, s);          // k inside the slab for this lane at step s
                        if constexpr (C0 || C1) {
                            const int piece = (kk * ES) >> 4, lo = (kk * ES) & 15;
                            T av[4];
#pragma unroll
                            for (int ib = 0; ib < 4; ++ib)
                                av[ib] = *reinterpret_cast<const T*>(lb + (16 * ib + r) * LRB + ((piece ^ lsig) << 4) + lo);
                            if constexpr (C0) {
                                const T bv = bf0[s] * ((qb0 == kb && r > kk) ? T(0) : T(1));
#pragma unroll
                                for (int ib = 0; ib < 4; ++ib) acc0[ib] = Mf<T>::mma(av[ib], bv, acc0[ib]);
                            }
                            if constexpr (C1) {
                                const T bv = bf1[s] * ((qb1 == kb && r > kk) ? T(0) : T(1));
#pragma unroll
                                for (int ib = 0; ib < 4; ++ib) acc1[ib] = Mf<T>::mma(av[ib], bv, acc1[ib]);
                            }
                        }
                        int o0, o1;
                        frag_off(kbn, s, o0, o1);
                        bf0[s] = Ab[o0]; bf1[s] = Ab[o1];
                    }
                };
                if (d0 && d1) body(std::true_type{}, std::true_type{});
                else if (d0) body(std::true_type{}, std::false_type{});
                else if (d1) body(std::false_type{}, std::true_type{});
                else body(std::false_type{}, std::false_type{});
            }
            TST(4);
            // the loop's barrier: a RAW s_barrier; the helpers wait for THEIR instructions of the next slab first (counted: the slabs
            // beyond it stay in flight), the MFMA waves' fragment loads stay in flight across it
            if (((t + 1) & (TL_G - 1)) == 0 || t + 1 == ns) {
                if constexpr (IMG) {
                    const int later = gi - (gs + 1 + TL_G);       // slabs issued behind the next group
                    tl_wait_vmcnt(later > 0 ? later * HPI : 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
            ++gs;
            TST(5);
        }
        // ---- epilogue: Z[I, q-block] = -Z_II out, through the 2 x 2 structure of Z_II (images -Z00 | L10 | -Z11), from registers
        auto zmul = [&](int av, int b, const Acc& X, Acc o) __attribute__((always_inline)) -> Acc {
            const T* zp = reinterpret_cast<const T*>(Zdimg + tl_tri(av, b) * BLK) + lane;
#pragma unroll
            for (int s = 0; s < 4; ++s) o = Mf<T>::mma(zp[s * 64], X[s], o);
            return o;
        };
        auto finish = [&](Acc (&X)[IMG ? 1 : 4], int qb) __attribute__((always_inline)) {
            if constexpr (!IMG) {
            const Acc z = {0, 0, 0, 0};
            Acc t1 = zmul(1, 0, X[0], z);
            t1 = zmul(1, 1, X[1], t1);
            const Acc t0 = zmul(0, 0, X[0], z);
            X[0] = t0; X[1] = t1;                                 // R_lo = -Z00 X_lo
            X[2] = zmul(2, 0, X[0], X[2]); X[2] = zmul(2, 1, X[1], X[2]);      // X_hi + L10 R_lo
            X[3] = zmul(3, 0, X[0], X[3]); X[3] = zmul(3, 1, X[1], X[3]);
            Acc t3 = zmul(3, 2, X[2], z);
            t3 = zmul(3, 3, X[3], t3);
            const Acc t2 = zmul(2, 2, X[2], z);
            X[2] = t2; X[3] = t3;                                 // R_hi = -Z11 (X_hi + L10 R_lo)
            T part = 0;                                           // sum_i Z[i][16 qb + r] u[i] over the panel's rows held by this lane
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = c0 + 16 * ib + Mf<T>::row(g, q);
                    if (row < n) Ab[(size_t)row * n + 16 * qb + r] = X[ib][q];
                    part = fma(X[ib][q], uv[row], part);
                }
            part += shfl_xor_t<T>(part, 16);
            part += shfl_xor_t<T>(part, 32);
            if (g == 0) al_a[16 * qb + r] += part;                // (this wave is the column block's only writer in this panel)
            }
        };
        if constexpr (!IMG) {
            if (v0) finish(acc0, qb0);
            if (v1) finish(acc1, qb1);
        }
        TST(6);
        // ---- helper 1: the diagonal block itself.  Z00, Z11 (lower parts) from the images; Z10 = (-Z11) (L10 Z00)
        if constexpr (IMG) {
            {
                const int h = hw;
                const int k0 = c0 + 32 * h;
#pragma unroll
                for (int av = 0; av < 2; ++av)
#pragma unroll
                    for (int b = 0; b <= av; ++b) {
                        const T* const zp = reinterpret_cast<const T*>(Zdimg + tl_tri(2 * h + av, 2 * h + b) * BLK) + lane;
                        const int i = 16 * av + r;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int c = 16 * b + Mf<T>::row(g, q);
                            if (c <= i && k0 + i < n) Ab[(size_t)(k0 + i) * n + k0 + c] = -zp[q * 64];
                        }
                    }
                // alpha[k0 + c] += sum_i Z(h,h)[i][c] u[k0 + i]: per column block b the two row blocks, then over the 16 rows (lanes r)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        T sacc = 0;
#pragma unroll
                        for (int av = b; av < 2; ++av)
                            sacc = fma(-*(reinterpret_cast<const T*>(Zdimg + tl_tri(2 * h + av, 2 * h + b) * BLK) + q * 64 + lane), uv[k0 + 16 * av + r], sacc);
                        sacc += shfl_xor_t<T>(sacc, 1); sacc += shfl_xor_t<T>(sacc, 2);
                        sacc += shfl_xor_t<T>(sacc, 4); sacc += shfl_xor_t<T>(sacc, 8);
                        if (r == 0) al_a[k0 + 16 * b + Mf<T>::row(g, q)] += sacc;
                    }
            }
            if (pm > 32) {
                // T = L10 Z00 (A = the L10 image; B[k][c] = Z00[k][c] read out of the image of -Z00, whose element (i', c') sits at
                // img_off(c', i')), then Z10 = (-Z11) T with T's accumulator blocks as B operands
                {
                    const int cb = hw;
                    T zpart = 0;
                    Acc tt[2];
#pragma unroll
                    for (int ibr = 0; ibr < 2; ++ibr) {
                        Acc o = {0, 0, 0, 0};
#pragma unroll
                        for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                if (kb2 < cb) continue;
                                const T a = *(reinterpret_cast<const T*>(Zdimg + tl_tri(2 + ibr, kb2) * BLK) + s * 64 + lane);
                                const T b = -*reinterpret_cast<const T*>(Zdimg + tl_tri(kb2, cb) * BLK + tl_img_off<T>(r, Mf<T>::row(g, s)));
                                o = Mf<T>::mma(a, b, o);
                            }
                        tt[ibr] = o;
                    }
#pragma unroll
                    for (int ibp = 0; ibp < 2; ++ibp) {
                        Acc zz = {0, 0, 0, 0};
#pragma unroll
                        for (int ibr = 0; ibr <= ibp; ++ibr)
#pragma unroll
                            for (int s = 0; s < 4; ++s)
                                zz = Mf<T>::mma(*(reinterpret_cast<const T*>(Zdimg + tl_tri(2 + ibp, 2 + ibr) * BLK) + s * 64 + lane), tt[ibr][s], zz);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int row = c0 + 32 + 16 * ibp + Mf<T>::row(g, q);
                            if (row < n) Ab[(size_t)row * n + c0 + 16 * cb + r] = zz[q];
                            zpart = fma(zz[q], uv[row], zpart);
                        }
                    }
                    zpart += shfl_xor_t<T>(zpart, 16);
                    zpart += shfl_xor_t<T>(zpart, 32);
                    if (g == 0) al_b[c0 + 16 * cb + r] = zpart;
                }
            }
        }
    }
#ifdef PACOH_LL_STAMPS
    TST(0);
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 1 || wave == 2 || wave == 15))
        printf("trtri wave %d (us): tail+diag %.1f | panel barrier %.1f | prologue dma / images %.1f | barrier %.1f | loop work %.1f | loop barrier %.1f | finish %.1f | dma issue %.1f\n",
               wave, st_[0] * 0.01, st_[1] * 0.01, st_[2] * 0.01, st_[3] * 0.01, st_[4] * 0.01, st_[5] * 0.01, st_[6] * 0.01, st_[7] * 0.01);
#endif
    };
    // (two instantiations of the panel loop: waves 0 / 1 hold twenty prefetched image entries each in registers and no accumulators,
    //  the other waves hold accumulators and no image entries -- one allocation for the union of both would spill 165 registers)
    if (wave < 2) run(std::true_type{}); else run(std::false_type{});
    if (alpha) {
        __syncthreads();
        for (int q = tid; q < n; q += TL_NT) alpha[(size_t)blockIdx.x * n + q] = al_a[q] + al_b[q];
    }
#undef TST
}

template <typename T>
bool tl_plan(int n, int* ring_bytes, size_t* lds) {
    const int ES = sizeof(T), BLK = 256 * ES;
    if (n < 1 || n > 512 || ((size_t)n * ES) % 16 != 0) return false;
    const size_t cap = 160u * 1024u;
    const int npan = (n + 63) / 64;
    const int c0max = (npan - 1) * 64;
    (void)c0max;
    size_t ring = cap - 10 * BLK - 3 * (size_t)((n + 63) & ~63) * ES;
    if (ring < (size_t)4 * 64 * 16 * ES) return false;              // at least four L slabs in flight
    *ring_bytes = (int)ring; *lds = cap;
    return true;
}

}  // namespace

bool trtri_ll_fits(int n, int dtype) {
    int rb; size_t lds;
    return dtype == PACOH_F32 ? tl_plan<float>(n, &rb, &lds) : tl_plan<double>(n, &rb, &lds);
}

// returns 1 when n is outside this kernel's plan (caller: trtri_dense_kernel).  u / alpha (optional): alpha = Z^T u as well
int trtri_ll_try(void* A, const int32_t* info, int B, int n, int dtype, hipStream_t s, const void* u, void* alpha) {
    int rb = 0; size_t lds = 0;
    if (dtype == PACOH_F32) {
        if (!tl_plan<float>(n, &rb, &lds)) return 1;
        auto kern = trtri_ll_kernel<float>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        hipLaunchKernelGGL(kern, dim3(B), dim3(TL_NT), lds, s, (float*)A, info, n, rb, (const float*)u, (float*)alpha);
    } else {
        if (!tl_plan<double>(n, &rb, &lds)) return 1;
        auto kern = trtri_ll_kernel<double>;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        hipLaunchKernelGGL(kern, dim3(B), dim3(TL_NT), lds, s, (double*)A, info, n, rb, (const double*)u, (double*)alpha);
    }
    return launch_status();
}

}  // namespace pacoh
