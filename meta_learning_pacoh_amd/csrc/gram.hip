// Standalone (materialised) ARD-RBF Gram build: the HBM-write-bound kernel of the path.
// K[b,i,j] = os_p * exp(-0.5 * sum_k ((z1[b,i,k]-z2[b,j,k])/l_pk)^2) (+ noise_p on the diagonal).
// Replaces SEKernelLight.forward (meta_learn/models.py:428-446) / ScaleKernel(RBFKernel(ard))
// (GPR_meta_mll.py:218,223) when K is needed in memory (large-n dense path, K_xs, K_ss).
//
// Algorithmic bytes per Gram: n*f*s (+ m*f*s) read + n*m*s written.  Each lane produces
// VW consecutive columns of one row and stores them with ONE 16-byte store, so a wave writes 1 KiB
// contiguous; the few input rows it needs come out of L1/L2.
#include "common.h"

namespace pacoh {

// One unit = one problem b and one tile of TI rows x TJ columns (1024 output quads, QPT per
// thread).  The tile's input rows are staged once into LDS already divided by the lengthscale, so
// the inner loop is LDS reads + FMA + exp + one 16-byte store per quad; b, its hyper-parameters and
// all 64-bit index arithmetic are wave-uniform scalars.
//
// Interior tiles (the whole tile inside the matrix, row length a multiple of the vector width) take a
// branch-free body: no bounds tests, no exec-mask regions, so the compiler issues the LDS reads of all
// QPT quads up front and the QPT stores back to back.  With the per-quad bounds tests in place the same
// arithmetic ran at 3.6 TB/s instead of 5+ (each quad became its own basic block: read, wait, compute,
// store, serialised).  Edge tiles keep the checked body.
constexpr int QPT = 4;

template <typename T, int FP, bool FULL, bool NOISE>
__device__ __forceinline__ void gram_tile_body(const T* __restrict__ z1s, const T* __restrict__ z2s, T* __restrict__ Kb,
                                               T osv, T nz, int i0, int j00, int n, int m, int tjq_shift) {
    using V = typename VecOf<T>::type;
    constexpr int VW = VecOf<T>::W;
    const int TJQ = 1 << tjq_shift;
    const bool vec_ok = FULL || (m % VW) == 0;
#pragma unroll
    for (int u = 0; u < QPT; ++u) {
        const int q = u * 256 + threadIdx.x;
        const int il = q >> tjq_shift, jq = q & (TJQ - 1);
        const int i = i0 + il, j0 = j00 + jq * VW;
        T a[FP];
#pragma unroll
        for (int c = 0; c < FP; ++c) a[c] = z1s[il * FP + c];
        // z2 is staged feature-major ([FP][TJ]): one 16-byte read per feature brings that feature for the lane's VW
        // columns and consecutive lanes read consecutive addresses (the point-major layout put the lanes FP*VW elements
        // apart: 128-byte strides = 32-way bank conflicts at f = 8 in fp64)
        T sacc[VW];
#pragma unroll
        for (int v = 0; v < VW; ++v) sacc[v] = 0;
#pragma unroll
        for (int c = 0; c < FP; ++c) {
            const V bq = *reinterpret_cast<const V*>(z2s + c * (VW << tjq_shift) + jq * VW);
            T bvals[VW];
            if constexpr (VW == 4) { bvals[0] = bq.x; bvals[1] = bq.y; bvals[2] = bq.z; bvals[3] = bq.w; } else { bvals[0] = bq.x; bvals[1] = bq.y; }
#pragma unroll
            for (int v = 0; v < VW; ++v) { const T d = a[c] - bvals[v]; sacc[v] = fma(d, d, sacc[v]); }
        }
        T out[VW];
#pragma unroll
        for (int v = 0; v < VW; ++v) {
            T k = osv * rbf_exp<T>(T(-0.5) * sacc[v]);
            if (NOISE) k += (i == j0 + v) ? nz : T(0);
            out[v] = k;
        }
        if (FULL || (i < n && j0 < m)) {
            T* kp = Kb + (long)i * m + j0;
            if (vec_ok) {
                V o;
                if constexpr (VW == 4) { o.x = out[0]; o.y = out[1]; o.z = out[2]; o.w = out[3]; } else { o.x = out[0]; o.y = out[1]; }
                *reinterpret_cast<V*>(kp) = o;
            } else {
#pragma unroll
                for (int v = 0; v < VW; ++v) if (j0 + v < m) kp[v] = out[v];
            }
        }
    }
}

// A workgroup owns G consecutive units (unit = problem x tile).  All G units' inputs are fetched first (G*R loads per
// thread in flight together), divided by the lengthscale and written to LDS behind ONE barrier; the G tile bodies then run
// back to back without further synchronisation (4*G 16-byte stores per thread per prologue).
// ALLFULL (chosen by the host when every tile is interior and every workgroup complete) compiles the kernel WITHOUT any
// bounds-checked code.  Measured at 20480 Grams of 64x64, f=2, fp32 (tools/gram_bench.py): 97 us with per-quad bounds tests,
// 88 us with a branch-free interior body beside the checked one, 80 us after removing the index divisions, 62 us (5.56 TB/s)
// once the checked code is not in the kernel at all -- the never-executed generic paths cost 25 % through the compiler's
// register allocation and conservative s_waitcnt placement at their join points.  A pure 16-byte fill of the same buffer
// reaches 5.6-6.5 TB/s (tools/write_roof.hip).  A persistent software-pipelined variant (prefetch unit u+1 while storing
// unit u) was measured slower: its per-unit barrier and in-order vmcnt waits behind the stores serialise it.
template <typename T, int FP, int R, int G, bool ALLFULL>
__global__ void __launch_bounds__(256) gram_kernel(const T* __restrict__ z1, int z1_div, const T* __restrict__ z2, int z2_div,
                                                   const T* __restrict__ ls, const T* __restrict__ os,
                                                   const T* __restrict__ noise, int add_noise, T* __restrict__ K,
                                                   int P, int n, int m, int f, int tjq_shift, int tiles_i, int tiles_j,
                                                   int total_units, int lower) {
    constexpr int VW = VecOf<T>::W;
    // lower != 0 (square Grams feeding the Cholesky of the dense path, which reads the lower triangle only; G == 1): tiles that lie
    // entirely above the diagonal are neither evaluated nor written -- 48 of 128 tiles at n = 512 in fp64 (16 x 128 tiles)
    // The grid holds only the kept tiles (`lower` = their number per problem; the launch is dispatch-bound -- one 16 KB tile per
    // workgroup -- so skipped workgroups that merely exit early bought nothing): unit index -> (problem, tile row, tile column).
    int unit_remap = -1;
    if (lower) {
        const int TIl = (QPT * 256) >> tjq_shift, TJl = VW << tjq_shift;
        int t = blockIdx.x % lower, ti = 0;
        for (;; ++ti) {
            int cnt = (ti * TIl + TIl - 1) / TJl + 1;
            cnt = cnt > tiles_j ? tiles_j : cnt;
            if (t < cnt) break;
            t -= cnt;
        }
        unit_remap = ((blockIdx.x / lower) * tiles_i + ti) * tiles_j + t;
    }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int TJ = VW << tjq_shift;                  // tile columns
    const int TI = (QPT * 256) >> tjq_shift;         // tile rows
    const int E1 = TI * FP, E = (TI + TJ) * FP;      // staged elements per unit: z1 rows, then z2 rows
    T* lds0 = reinterpret_cast<T*>(smem_raw);
    const bool mvec = (m % VW) == 0;
    const int u0 = lower ? unit_remap : blockIdx.x * G;
    // decompose the first unit once; the others follow by incrementing
    int tjs[G], tis[G], bs[G], ps[G];
    {
        int tj = u0 % tiles_j;
        int rest = u0 / tiles_j;
        int ti = rest % tiles_i;
        int b = rest / tiles_i;
        int p = b % P;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            tjs[g] = tj; tis[g] = ti; bs[g] = b; ps[g] = p;
            if (++tj == tiles_j) { tj = 0; if (++ti == tiles_i) { ti = 0; ++b; if (++p == P) p = 0; } }
        }
    }
    T pre[G][R], lpre[G][R];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const bool live = u0 + g < total_units;
        const int fb = bs[g];
        const T* lp = ls + (long)ps[g] * f;
        const T* z1b = z1 + (long)(z1_div == 1 ? fb : fb / z1_div) * n * f;
        const T* z2b = z2 + (long)(z2_div == 1 ? fb : fb / z2_div) * m * f;
        const int i0 = tis[g] * TI, j00 = tjs[g] * TJ;
        const bool interior = ALLFULL || ((i0 + TI <= n) && (j00 + TJ <= m) && f == FP);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = r * 256 + threadIdx.x;
            T v = 0, l = 1;
            if ((ALLFULL || live) && e < E) {
                const bool first = e < E1;
                const int e2 = first ? e : e - E1;
                if (interior) {                      // contiguous rows, no bounds tests
                    v = (first ? z1b + (long)i0 * FP : z2b + (long)j00 * FP)[e2];
                    l = lp[e2 % FP];
                } else {
                    const int row = e2 / FP, c = e2 - row * FP;
                    const int gr = (first ? i0 : j00) + row;
                    if (c < f && gr < (first ? n : m)) { v = (first ? z1b : z2b)[(long)gr * f + c]; l = lp[c]; }
                }
            }
            pre[g][r] = v; lpre[g][r] = l;
        }
    }
    T osv[G], nz[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { osv[g] = os ? os[ps[g]] : T(1); nz[g] = add_noise ? noise[ps[g]] : T(0); }
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = r * 256 + threadIdx.x;
            if (e < E) {
                // z1 rows stay point-major (one row is broadcast to a wave); z2 goes in feature-major
                const int e2 = e - E1;
                const int dst = e < E1 ? e : E1 + (e2 % FP) * TJ + e2 / FP;
                lds0[g * E + dst] = pre[g][r] / lpre[g][r];
            }
        }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < G; ++g) {
        if (!ALLFULL && u0 + g >= total_units) break;
        const T* z1s = lds0 + g * E;
        const T* z2s = z1s + E1;
        const int i0 = tis[g] * TI, j00 = tjs[g] * TJ;
        T* Kb = K + (long)bs[g] * n * m;
        const bool full = ALLFULL || ((i0 + TI <= n) && (j00 + TJ <= m) && mvec);
        if (full) {
            if (add_noise && i0 < j00 + TJ && j00 < i0 + TI)      // only tiles that touch the diagonal
                gram_tile_body<T, FP, true, true>(z1s, z2s, Kb, osv[g], nz[g], i0, j00, n, m, tjq_shift);
            else
                gram_tile_body<T, FP, true, false>(z1s, z2s, Kb, osv[g], nz[g], i0, j00, n, m, tjq_shift);
        } else if (!ALLFULL) {
            gram_tile_body<T, FP, false, true>(z1s, z2s, Kb, osv[g], nz[g], i0, j00, n, m, tjq_shift);
        }
    }
}

template <typename T>
static int launch_gram(const void* z1, int z1_div, const void* z2, int z2_div, const void* ls, const void* os,
                       const void* noise, int add_noise, void* K, int B, int P, int n, int m, int f, hipStream_t s, int lower = 0) {
    constexpr int VW = VecOf<T>::W;
    const int mq = (m + VW - 1) / VW;
    int tjq_shift = 0;
    while ((1 << tjq_shift) < mq && tjq_shift < 6) ++tjq_shift;       // tile: up to 64 quads wide
    const int TJ = (1 << tjq_shift) * VW, TI = (QPT * 256) >> tjq_shift;
    const int tiles_i = (n + TI - 1) / TI, tiles_j = (m + TJ - 1) / TJ;
    const long units = (long)B * tiles_i * tiles_j;
    if (units > 0x7fffffffL) return PACOH_ELIMIT;
    const int FP = f <= 2 ? 2 : (f <= 4 ? 4 : (f <= 8 ? 8 : 16));
    const int E = (TI + TJ) * FP;
    const int R = (E + 255) / 256;                                    // staged elements per thread and unit: 1..65
    int G = R <= 4 ? 2 : 1;                                           // units per workgroup (measured: 2 best at n=64; 1, 4 within 10 %)
    while (G > 1 && units / G < 2048) G >>= 1;                        // keep >= 8 workgroups per CU
    if (R > 4 || lower) G = 1;
    const size_t lds = (size_t)G * E * sizeof(T);
    long blocks = (units + G - 1) / G;
    if (lower) {                                                      // `lower` becomes the number of kept tiles per problem
        int kept = 0;
        for (int ti = 0; ti < tiles_i; ++ti) { const int c = (ti * TI + TI - 1) / TJ + 1; kept += c > tiles_j ? tiles_j : c; }
        lower = kept;
        blocks = (long)B * kept;
    }
    // every tile interior and every workgroup complete: the kernel variant without any bounds-checked code
    const bool allfull = (n % TI) == 0 && (m % TJ) == 0 && f == FP && (units % G) == 0;      // (G == 1 with `lower`)
#define PACOH_GRAM_LAUNCH(fp, r, g) if (allfull) PACOH_GRAM_LAUNCH2(fp, r, g, true); else PACOH_GRAM_LAUNCH2(fp, r, g, false)
#define PACOH_GRAM_LAUNCH2(fp, r, g, af) hipLaunchKernelGGL((gram_kernel<T, fp, r, g, af>), dim3((unsigned)blocks), dim3(256), lds, s, \
        (const T*)z1, z1_div, (const T*)z2, z2_div, (const T*)ls, (const T*)os, (const T*)noise, add_noise, (T*)K, P, n, m, f, \
        tjq_shift, tiles_i, tiles_j, (int)units, lower)
#define PACOH_GRAM_RG(fp, r) do { if (G >= 2) { PACOH_GRAM_LAUNCH(fp, r, 2); } else { PACOH_GRAM_LAUNCH(fp, r, 1); } } while (0)
#define PACOH_GRAM_CASE(fp) case fp: \
        if (R <= 1) PACOH_GRAM_RG(fp, 1); else if (R <= 2) PACOH_GRAM_RG(fp, 2); else if (R <= 4) PACOH_GRAM_RG(fp, 4); \
        else if (R <= 8) { PACOH_GRAM_LAUNCH(fp, 8, 1); } else if (R <= 17) { PACOH_GRAM_LAUNCH(fp, 17, 1); } else { PACOH_GRAM_LAUNCH(fp, 65, 1); } break;
    switch (FP) { PACOH_GRAM_CASE(2) PACOH_GRAM_CASE(4) PACOH_GRAM_CASE(8) default: PACOH_GRAM_CASE(16) }
#undef PACOH_GRAM_CASE
#undef PACOH_GRAM_RG
#undef PACOH_GRAM_LAUNCH
#undef PACOH_GRAM_LAUNCH2
    return launch_status();
}

// the dense path's own Gram call (dense_gp.hip): lower != 0 -> only the tiles that touch the lower triangle (A = os K + noise I is
// about to be factorised in place, and every Cholesky kernel of the path reads the lower triangle only)
int gram_rbf_for_chol(const void* z, int z_div, const void* ls, const void* os, const void* noise, void* K, int B, int P, int n, int f,
                      int dtype, hipStream_t s, int lower) {
    if (dtype == PACOH_F32) return launch_gram<float>(z, z_div, z, z_div, ls, os, noise, 1, K, B, P, n, n, f, s, lower);
    return launch_gram<double>(z, z_div, z, z_div, ls, os, noise, 1, K, B, P, n, n, f, s, lower);
}

}  // namespace pacoh

using namespace pacoh;

namespace pacoh {
int dense_gram_mfma_full(const void* z, int z_div, const void* ls, const void* os, const void* noise, void* K, int B, int P, int n, int f,
                         hipStream_t s);                                                     // dense_grad_mfma.hip (1: not in its plan)
}

// Any kernel family (common.h: kern_eval), one thread per entry: the families other than ARD-RBF are capability, not hot path
namespace pacoh {
template <typename T>
__global__ void __launch_bounds__(256) gram_family_kernel(const T* __restrict__ z1, int z1_div, const T* __restrict__ z2, int z2_div,
                                                          const T* __restrict__ ls, const T* __restrict__ os, const T* __restrict__ noise,
                                                          int add_noise, T* __restrict__ K, int B, int P, int n, int m, int f, int kind) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * n * m) return;
    const int j = (int)(idx % m), i = (int)((idx / m) % n);
    const long b = idx / ((long)n * m);
    const int p = (int)(b % P);
    const T* a = z1 + ((b / z1_div) * n + i) * (long)f;
    const T* c2 = z2 + ((b / z2_div) * m + j) * (long)f;
    T s = 0;
    for (int c = 0; c < f; ++c) { const T d = a[c] / ls[(long)p * f + c] - c2[c] / ls[(long)p * f + c]; s = fma(d, d, s); }
    T k = (os ? os[p] : T(1)) * kern_val<T>(kind, s);
    if (add_noise && i == j) k += noise[p];
    K[idx] = k;
}
}  // namespace pacoh

extern "C" int pacoh_gram_rbf_ard(const void* z1, int z1_div, const void* z2, int z2_div,
                                  const void* lengthscale, const void* outputscale, const void* noise,
                                  int add_noise_diag, void* K,
                                  int B, int P, int n, int m, int f, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    const int kind = kernel_of(f);
    f = features_of(f);
    if (!z1 || !z2 || !lengthscale || !K || B <= 0 || P <= 0 || n <= 0 || m <= 0 || f <= 0 || z1_div <= 0 || z2_div <= 0)
        return PACOH_EINVAL;
    if (add_noise_diag && !noise) return PACOH_EINVAL;
    if (f > PACOH_MAX_FEATURES || kind < 0 || kind > PACOH_KERNEL_COSINE) return PACOH_ELIMIT;
    if (kind != PACOH_KERNEL_RBF) {
        const long total = (long)B * n * m;
        if (dtype == PACOH_F32)
            hipLaunchKernelGGL(gram_family_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)z1,
                               z1_div, (const float*)z2, z2_div, (const float*)lengthscale, (const float*)outputscale, (const float*)noise,
                               add_noise_diag, (float*)K, B, P, n, m, f, kind);
        else
            hipLaunchKernelGGL(gram_family_kernel<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const double*)z1,
                               z1_div, (const double*)z2, z2_div, (const double*)lengthscale, (const double*)outputscale, (const double*)noise,
                               add_noise_diag, (double*)K, B, P, n, m, f, kind);
        return launch_status();
    }
    if (dtype == PACOH_F32)
        return launch_gram<float>(z1, z1_div, z2, z2_div, lengthscale, outputscale, noise, add_noise_diag, K, B, P, n, m, f, (hipStream_t)stream);
    if (z1 == z2 && z1_div == z2_div && n == m) {            // one point set: symmetric, distances on the matrix cores (dense_grad_mfma.hip)
        const int rc = dense_gram_mfma_full(z1, z1_div, lengthscale, outputscale, add_noise_diag ? noise : nullptr, K, B, P, n, f, (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    return launch_gram<double>(z1, z1_div, z2, z2_div, lengthscale, outputscale, noise, add_noise_diag, K, B, P, n, m, f, (hipStream_t)stream);
}
