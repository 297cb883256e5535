// Fused small-n task-GP kernels: Gram build -> (+noise) -> Cholesky with jitter retry -> solves ->
// log-det -> per-datapoint LML, its gradient, and the exact posterior predictive.  The Gram matrix
// never leaves the CU: every problem (task x particle) lives in LDS for its whole lifetime.
//
// Mapping: one "group" of GS = pow2ceil(n) lanes per GP problem, lane i owns row i.  n <= 64: a
// group is (a slice of) one 64-wide wavefront, 64/GS problems are packed per wave; n > 64: one
// problem per workgroup of GS threads.  All O(n^3) phases are the same primitive -- dot product of
// the lane's own LDS row with one row broadcast to all lanes (ds_read_b128 both, conflict-free by
// the leading dimension chosen in lds_ld()):
//   Cholesky (left-looking)   acc_i = A_ik - <L_i, L_k>          A_ik computed on the fly from z
//   Z = L^-1 (column per lane) z_r  = (d_rc - <Z_c, L_r>)/L_rr
//   W = K^-1 = Z^T Z           W_ij = <Z_i, Z_j>   consumed immediately by the gradient sums
// Reference arithmetic being replaced: random_gp.py:54-89, GPR_meta_mll.py:104-117 (through gpytorch).
#include "common.h"
#include <stdlib.h>

namespace pacoh {

template <typename T>
struct GpArgs {
    const T* z; int z_div;
    const T* mean; int mean_mode;
    const T* y; int y_div;
    const T* ls; const T* os; const T* noise;
    const int32_t* n_valid;
    const T* g_lml;
    T* lml; T* alpha_out; T* L_out; int32_t* info;
    T* d_z; T* d_mean; T* d_ls; T* d_os; T* d_noise;
    const T* z_tst; int zt_div; const T* mean_tst; T* mu; T* var; T* V_out; int m;
    int B, P, n, f, GS, G, LD;
    int kind;             // kernel family (PACOH_KERNEL_*), decoded from the f argument of the entry point
    int nZ;               // rows of the second LDS matrix (Z / test-point tile)
    unsigned per_group;   // LDS elements per group
};

enum { MODE_FWD = 0, MODE_FWDBWD = 1, MODE_PREDICT = 2 };

template <typename T> __host__ __device__ inline unsigned gp_group_elems(int n, int LD, int FP, int nZ) {
    unsigned e = (unsigned)(n + nZ) * LD + (unsigned)((n * FP + 3) & ~3) + 3u * LD + 16u;
    return (e + 3u) & ~3u;
}

template <typename T>
__device__ __forceinline__ T group_sum(T v, int GS, int i, T* red) {
    v = subwave_sum<T>(v, GS < 64 ? GS : 64);
    if (GS > 64) {               // one group per workgroup in this case -> barriers are uniform
        __syncthreads();
        if ((i & 63) == 0) red[i >> 6] = v;
        __syncthreads();
        T s = 0;
        for (int q = 0; q < GS / 64; ++q) s += red[q];
        v = s;
    }
    return v;
}

template <typename T, int FP, int MODE>
__global__ void __launch_bounds__(256) gp_small_kernel(GpArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    using V = typename VecOf<T>::type;
    constexpr int W = VecOf<T>::W;
    T* smem = reinterpret_cast<T*>(smem_raw);

    const int tid = threadIdx.x, GS = a.GS;
    const int g = tid / GS, i = tid - g * GS;
    const int n = a.n, LD = a.LD, f = a.f;
    const long b = (long)blockIdx.x * a.G + g;
    const bool live = b < a.B;
    const int p = live ? (int)(b % a.P) : 0;
    const long ty = live ? b / a.y_div : 0;
    int nv = 0;
    if (live) { nv = a.n_valid ? a.n_valid[ty] : n; nv = nv < n ? nv : n; nv = nv < 0 ? 0 : nv; }

    const int nZ = a.nZ;
    T* Lmat = smem + (size_t)g * a.per_group;
    T* Zmat = Lmat + (size_t)n * LD;
    T* zf = Zmat + (size_t)nZ * LD;
    T* rvec = zf + ((n * FP + 3) & ~3);
    T* avec = rvec + LD;
    T* invd = avec + LD;
    T* red = invd + LD;          // [0..3] cross-wave sums, [8] failure flag

    // ---- hyper-parameters of this problem's set p -------------------------------------------
    T ls[FP];
#pragma unroll
    for (int c = 0; c < FP; ++c) ls[c] = (live && c < f) ? a.ls[(long)p * f + c] : T(1);
    const T os = (live && a.os) ? a.os[p] : T(1);
    const T noise = live ? a.noise[p] : T(1);

    // ---- load features (pre-divided by the lengthscale, models.py:435-436) and residual ------
    T zs[FP];
#pragma unroll
    for (int c = 0; c < FP; ++c) zs[c] = 0;
    T ri = 0;
    if (i < nv) {
        const T* zp = a.z + ((b / a.z_div) * n + i) * (long)f;
#pragma unroll
        for (int c = 0; c < FP; ++c) if (c < f) zs[c] = zp[c] / ls[c];
        T mi = 0;
        if (a.mean_mode == PACOH_MEAN_VECTOR) mi = a.mean[b * n + i];
        else if (a.mean_mode == PACOH_MEAN_CONST) mi = a.mean[p];
        ri = a.y[ty * n + i] - mi;
    }
    if (i < n) {
#pragma unroll
        for (int c = 0; c < FP; ++c) zf[i * FP + c] = zs[c];
    }

    // ---- Cholesky with the psd_safe_cholesky jitter ladder -----------------------------------
    const T jitter_base = sizeof(T) == 4 ? T(1e-6) : T(1e-8);
    int my_info = -1;
    bool active = true;          // uniform per group
    T jitter = 0;
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (active && i < n) {
            V* row = reinterpret_cast<V*>(Lmat + (size_t)i * LD);
            V zero;
            if constexpr (W == 4) { zero.x = 0; zero.y = 0; zero.z = 0; zero.w = 0; } else { zero.x = 0; zero.y = 0; }
            for (int v = 0; v < LD / W; ++v) row[v] = zero;
        }
        if (i == 0) red[8] = 0;
        __syncthreads();
        for (int k = 0; k < n; ++k) {
            T acc = 0;
            if (active && i >= k && i < n) {
                T aik;
                if (i < nv && k < nv) {
                    T s = 0;
#pragma unroll
                    for (int c = 0; c < FP; ++c) { T d = zs[c] - zf[k * FP + c]; s = fma(d, d, s); }
                    aik = os * kern_val<T>(a.kind, s);
                    if (i == k) aik += noise + jitter;
                } else {
                    aik = (i == k) ? T(1) : T(0);
                }
                acc = aik - dot_rows<T>(Lmat + (size_t)i * LD, Lmat + (size_t)k * LD, 0, k);
                if (i == k) {
                    if (!(acc > T(0))) { red[8] = 1; acc = 1; }
                    T d = t_sqrt<T>(acc);
                    invd[k] = T(1) / d;
                    Lmat[(size_t)k * LD + k] = d;
                }
            }
            __syncthreads();
            if (active && i > k && i < n) Lmat[(size_t)i * LD + k] = acc * invd[k];
            __syncthreads();
        }
        bool failed = active && (red[8] != T(0));
        if (active && !failed) { my_info = attempt; active = false; }
        int any = __syncthreads_or(failed ? 1 : 0);
        if (!any) break;
        jitter = jitter_base;
        for (int q = 0; q < attempt; ++q) jitter *= T(10);
    }
    const bool ok = my_info >= 0;
    if (live && i == 0 && a.info) a.info[b] = my_info;

    // ---- forward solve L u = r (column oriented), u_k published in rvec -----------------------
    for (int k = 0; k < n; ++k) {
        if (i == k) { ri *= invd[k]; rvec[k] = ri; }
        __syncthreads();
        if (i > k && i < n) ri = fma(-Lmat[(size_t)i * LD + k], rvec[k], ri);
    }
    const T ui = ri;
    T quad = group_sum<T>((i < nv) ? ui * ui : T(0), GS, i, red);
    T logdet = group_sum<T>((i < nv) ? -t_log<T>(invd[i]) : T(0), GS, i, red);
    const T LOG2PI = T(1.8378770664093453);
    T lml = nv > 0 ? T(-0.5) * (quad + T(2) * logdet + T(nv) * LOG2PI) / T(nv) : T(0);
    if (!ok) lml = T(NAN);

    if (MODE != MODE_PREDICT) {
        if (live && i == 0) a.lml[b] = lml;
        // ---- backward solve L^T alpha = u ---------------------------------------------------
        T ai = 0, acc_u = ui;
        for (int k = n - 1; k >= 0; --k) {
            if (i == k) { ai = acc_u * invd[k]; avec[k] = ai; }
            __syncthreads();
            if (i < k) acc_u = fma(-Lmat[(size_t)k * LD + i], avec[k], acc_u);
        }
        if (live && a.alpha_out && i < n) a.alpha_out[b * n + i] = ok ? ai : T(NAN);
        if (live && a.L_out && i < n) {
            for (int j = 0; j < n; ++j) a.L_out[(b * n + i) * (long)n + j] = (j <= i) ? Lmat[(size_t)i * LD + j] : T(0);
        }

        if (MODE == MODE_FWDBWD) {
            // ---- Z = L^-1, lane c owns column c of Z stored as row c of Zmat ------------------
            if (i < n) {
                V* row = reinterpret_cast<V*>(Zmat + (size_t)i * LD);
                V zero;
                if constexpr (W == 4) { zero.x = 0; zero.y = 0; zero.z = 0; zero.w = 0; } else { zero.x = 0; zero.y = 0; }
                for (int v = 0; v < LD / W; ++v) row[v] = zero;
                for (int r = i; r < n; ++r) {
                    T acc = dot_rows<T>(Zmat + (size_t)i * LD, Lmat + (size_t)r * LD, i, r);
                    Zmat[(size_t)i * LD + r] = ((r == i ? T(1) : T(0)) - acc) * invd[r];
                }
            }
            __syncthreads();
            // ---- W_ij = <Z_i, Z_j>, G = (alpha alpha^T - W)/(2 n), gradient sums ---------------
            const T gup = (live && a.g_lml) ? a.g_lml[b] : T(1);
            T dz[FP], dls[FP];
#pragma unroll
            for (int c = 0; c < FP; ++c) { dz[c] = 0; dls[c] = 0; }
            T dos = 0, dnz = 0;
            const T inv2n = nv > 0 ? T(0.5) / T(nv) : T(0);
            if (i < nv) {
                for (int j = 0; j < nv; ++j) {
                    int lo = i > j ? i : j;
                    T w = dot_rows<T>(Zmat + (size_t)i * LD, Zmat + (size_t)j * LD, lo, n);
                    T Gij = (ai * avec[j] - w) * inv2n;
                    T s = 0;
                    T df[FP];
#pragma unroll
                    for (int c = 0; c < FP; ++c) { df[c] = zf[j * FP + c] - zs[c]; s = fma(df[c], df[c], s); }
                    T e, ed;                           // k / os and the weight of (z_i - z_j) in its derivative (RBF: the same)
                    kern_eval<T>(a.kind, s, e, ed);
                    dos = fma(Gij, e, dos);
                    T M = Gij * os * ed;
#pragma unroll
                    for (int c = 0; c < FP; ++c) { T md = M * df[c]; dz[c] += md; dls[c] = fma(md, df[c], dls[c]); }
                    if (j == i) dnz = Gij;
                }
            }
            const T bad = ok ? T(0) : T(NAN);
            if (live && a.d_z && i < n) {
                for (int c = 0; c < f; ++c)
                    a.d_z[(b * n + i) * (long)f + c] = (i < nv) ? T(2) * gup * dz[c] / ls[c] + bad : T(0);
            }
            if (a.mean_mode == PACOH_MEAN_VECTOR) {
                if (live && a.d_mean && i < n) a.d_mean[b * n + i] = (i < nv) ? gup * ai / T(nv) + bad : T(0);
            } else if (a.mean_mode == PACOH_MEAN_CONST) {
                T sa = group_sum<T>((i < nv) ? ai : T(0), GS, i, red);
                if (live && a.d_mean && i == 0) a.d_mean[b] = nv > 0 ? gup * sa / T(nv) + bad : T(0);
            }
#pragma unroll
            for (int c = 0; c < FP; ++c) {
                if (c < f) {                   // f is uniform: no divergent barrier inside group_sum
                    T sc = group_sum<T>(dls[c], GS, i, red);
                    if (live && i == 0) a.d_ls[b * f + c] = gup * sc / ls[c] + bad;
                }
            }
            T sdos = group_sum<T>(dos, GS, i, red);
            T sdnz = group_sum<T>(dnz, GS, i, red);
            if (live && i == 0) {
                if (a.d_os) a.d_os[b] = gup * sdos + bad;
                a.d_noise[b] = gup * sdnz + bad;
            }
        }
    } else {
        // ---- posterior predictive: lane s solves L v = k_*s (same primitive as the Z phase) ----
        const int m = a.m;
        const T bad = ok ? T(0) : T(NAN);
        for (int s0 = 0; s0 < m; s0 += nZ) {      // nZ test points per pass (tile shrinks if LDS is short)
            const int s = s0 + i;
            const bool has = live && i < nZ && s < m;
            T zt[FP];
#pragma unroll
            for (int c = 0; c < FP; ++c) zt[c] = 0;
            T mt = 0;
            if (has) {
                const T* zp = a.z_tst + ((b / a.zt_div) * m + s) * (long)f;
#pragma unroll
                for (int c = 0; c < FP; ++c) if (c < f) zt[c] = zp[c] / ls[c];
                if (a.mean_mode == PACOH_MEAN_VECTOR) mt = a.mean_tst[b * m + s];
                else if (a.mean_mode == PACOH_MEAN_CONST) mt = a.mean_tst[p];
            }
            if (i < nZ) {
                V* row = reinterpret_cast<V*>(Zmat + (size_t)i * LD);
                V zero;
                if constexpr (W == 4) { zero.x = 0; zero.y = 0; zero.z = 0; zero.w = 0; } else { zero.x = 0; zero.y = 0; }
                for (int v = 0; v < LD / W; ++v) row[v] = zero;
            }
            T vu = 0, vv = 0;
            if (has) {
                for (int r = 0; r < n; ++r) {
                    T ks = 0;
                    if (r < nv) {
                        T sd = 0;
#pragma unroll
                        for (int c = 0; c < FP; ++c) { T d = zt[c] - zf[r * FP + c]; sd = fma(d, d, sd); }
                        ks = os * kern_val<T>(a.kind, sd);
                    }
                    T acc = dot_rows<T>(Zmat + (size_t)i * LD, Lmat + (size_t)r * LD, 0, r);
                    T val = (ks - acc) * invd[r];
                    Zmat[(size_t)i * LD + r] = val;
                    vu = fma(val, rvec[r], vu);
                    vv = fma(val, val, vv);
                    if (a.V_out) a.V_out[(b * m + s) * (long)n + r] = val;
                }
                a.mu[b * m + s] = mt + vu + bad;
                a.var[b * m + s] = os + noise - vv + bad;
            }
        }
    }
}

// predictive covariance from V = L^-1 K_x*:  cov[s,s'] = os*k(z*_s, z*_s') - <V_s, V_s'> + noise*d_ss'
template <typename T>
__global__ void gp_predict_cov_kernel(const T* __restrict__ z_tst, int zt_div, const T* __restrict__ V,
                                      const T* __restrict__ ls, const T* __restrict__ os,
                                      const T* __restrict__ noise, T* __restrict__ cov,
                                      int B, int P, int n, int m, int f, int kind) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * m * m;
    if (idx >= total) return;
    int s2 = (int)(idx % m);
    int s1 = (int)((idx / m) % m);
    long b = idx / ((long)m * m);
    int p = (int)(b % P);
    const T* za = z_tst + ((b / zt_div) * m + s1) * (long)f;
    const T* zb = z_tst + ((b / zt_div) * m + s2) * (long)f;
    T sd = 0;
    for (int c = 0; c < f; ++c) { T d = (za[c] - zb[c]) / ls[(long)p * f + c]; sd = fma(d, d, sd); }
    T osv = os ? os[p] : T(1);
    T k = osv * kern_val<T>(kind, sd);
    const T* va = V + (b * m + s1) * (long)n;
    const T* vb = V + (b * m + s2) * (long)n;
    T acc = 0;
    for (int r = 0; r < n; ++r) acc = fma(va[r], vb[r], acc);
    cov[idx] = k - acc + (s1 == s2 ? noise[p] : T(0));
}

static inline int pow2ceil(int n) { int g = 8; while (g < n) g <<= 1; return g; }

template <typename T, int MODE>
static int launch_gp_small(GpArgs<T> a, hipStream_t stream) {
    if (a.B <= 0 || a.P <= 0 || a.n <= 0 || a.f <= 0 || a.z_div <= 0 || a.y_div <= 0) return PACOH_EINVAL;
    if (a.f > PACOH_MAX_FEATURES || a.kind < 0 || a.kind > PACOH_KERNEL_COSINE) return PACOH_ELIMIT;
    if (!a.z || !a.y || !a.ls || !a.noise) return PACOH_EINVAL;
    if (a.mean_mode != PACOH_MEAN_ZERO && !a.mean) return PACOH_EINVAL;
    const int FP = a.f <= 2 ? 2 : (a.f <= 4 ? 4 : (a.f <= 8 ? 8 : 16));
    a.GS = pow2ceil(a.n);
    if (a.GS > 256) return PACOH_ELIMIT;
    a.G = a.GS >= 64 ? 1 : 64 / a.GS;
    a.LD = lds_ld<T>(a.n);
    a.nZ = (MODE == MODE_FWD) ? 0 : (MODE == MODE_FWDBWD ? a.n : a.GS);
    a.per_group = gp_group_elems<T>(a.n, a.LD, FP, a.nZ);
    size_t lds = (size_t)a.per_group * a.G * sizeof(T);
    while (MODE == MODE_PREDICT && lds > 160u * 1024u && a.nZ > 8) {   // smaller test-point tile
        a.nZ /= 2;
        a.per_group = gp_group_elems<T>(a.n, a.LD, FP, a.nZ);
        lds = (size_t)a.per_group * a.G * sizeof(T);
    }
    if (lds > 160u * 1024u) return PACOH_ELIMIT;
    int threads = a.GS >= 64 ? a.GS : 64;
    long blocks = ((long)a.B + a.G - 1) / a.G;
    void (*kern)(GpArgs<T>) = nullptr;
    switch (FP) {
        case 2: kern = gp_small_kernel<T, 2, MODE>; break;
        case 4: kern = gp_small_kernel<T, 4, MODE>; break;
        case 8: kern = gp_small_kernel<T, 8, MODE>; break;
        default: kern = gp_small_kernel<T, 16, MODE>; break;
    }
    if (lds > 64u * 1024u) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess) return PACOH_ELIMIT;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(threads), lds, stream, a);
    return launch_status();
}

template <typename T>
static int max_n_for(int want_grad) {
    int best = 0;
    for (int n = 1; n <= 256; ++n) {
        int LD = lds_ld<T>(n);
        int GS = pow2ceil(n);
        int G = GS >= 64 ? 1 : 64 / GS;
        size_t lds = (size_t)gp_group_elems<T>(n, LD, 16, want_grad ? n : 0) * G * sizeof(T);
        if (lds <= 160u * 1024u) best = n;
    }
    return best;
}

}  // namespace pacoh

using namespace pacoh;

// MFMA-blocked fp32 path for n <= 128 (gp_mfma.hip); PACOH_DISABLE_MFMA=1 forces the general LDS kernel
namespace pacoh {
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
int gp_mfma_try(const GpMfmaArgs& a, bool bwd, hipStream_t s);
int gp_reg_try(const GpMfmaArgs& a, bool bwd, hipStream_t s);      // register-resident kernel (gp_reg.hip): n <= 64, f <= 4
struct GpPredArgs {            // (same struct as in gp_reg_body.h)
    const float* zt; int zt_div;
    const float* mean_tst;
    float* mu; float* var;
    int m;
    float* V_out;
};
int gp_reg_predict_try(const GpMfmaArgs& a, const GpPredArgs& pa, hipStream_t s);      // ... its predictive form (n <= 64, no covariance)
}
static bool mfma_enabled() {
    const bool on = g_sw.mfma;
    return on;
}
static int try_mfma(const GpArgs<float>& a, bool bwd, hipStream_t s) {
    if (a.kind != PACOH_KERNEL_RBF) return 1;                  // (the register- / LDS-resident MFMA kernels evaluate the RBF family only)
    if (!mfma_enabled() || a.n > 128 || a.B <= 0 || a.P <= 0 || a.f <= 0 || a.f > PACOH_MAX_FEATURES ||
        a.z_div <= 0 || a.y_div <= 0 || !a.z || !a.y || !a.ls || !a.noise || (a.mean_mode != PACOH_MEAN_ZERO && !a.mean))
        return 1;
    GpMfmaArgs m = {a.z, a.z_div, a.mean, a.mean_mode, a.y, a.y_div, a.ls, a.os, a.noise, a.n_valid, a.g_lml,
                    a.lml, a.info, a.d_z, a.d_mean, a.d_ls, a.d_os, a.d_noise, a.B, a.P, a.n, a.f};
    // the forward-only entry points may ask for alpha / L outputs, which only the LDS-resident kernels produce: they do not come here
    const int rc = gp_reg_try(m, bwd, s);
    if (rc != 1) return rc;
    return gp_mfma_try(m, bwd, s);
}

extern "C" int pacoh_gp_small_max_n(int dtype, int want_grad) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    return dtype == PACOH_F32 ? max_n_for<float>(want_grad) : max_n_for<double>(want_grad);
}

template <typename T>
static GpArgs<T> make_args(const void* z, int z_div, const void* mean, int mean_mode, const void* y, int y_div,
                           const void* ls, const void* os, const void* noise, const int32_t* n_valid,
                           int B, int P, int n, int f) {
    GpArgs<T> a = {};
    a.z = (const T*)z; a.z_div = z_div; a.mean = (const T*)mean; a.mean_mode = mean_mode;
    a.y = (const T*)y; a.y_div = y_div; a.ls = (const T*)ls; a.os = (const T*)os; a.noise = (const T*)noise;
    a.n_valid = n_valid; a.B = B; a.P = P; a.n = n; a.f = features_of(f); a.kind = kernel_of(f);
    return a;
}

extern "C" int pacoh_gp_lml_fwd(const void* z, int z_div, const void* mean, int mean_mode,
                                const void* y, int y_div, const void* lengthscale, const void* outputscale,
                                const void* noise, const int32_t* n_valid,
                                void* lml, void* alpha_out, void* L_out, int32_t* info,
                                int B, int P, int n, int f, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!lml) return PACOH_EINVAL;
    if (dtype == PACOH_F32) {
        auto a = make_args<float>(z, z_div, mean, mean_mode, y, y_div, lengthscale, outputscale, noise, n_valid, B, P, n, f);
        a.lml = (float*)lml; a.alpha_out = (float*)alpha_out; a.L_out = (float*)L_out; a.info = info;
        if (!alpha_out && !L_out) { int rc = try_mfma(a, false, (hipStream_t)stream); if (rc != 1) return rc; }
        return launch_gp_small<float, MODE_FWD>(a, (hipStream_t)stream);
    }
    auto a = make_args<double>(z, z_div, mean, mean_mode, y, y_div, lengthscale, outputscale, noise, n_valid, B, P, n, f);
    a.lml = (double*)lml; a.alpha_out = (double*)alpha_out; a.L_out = (double*)L_out; a.info = info;
    return launch_gp_small<double, MODE_FWD>(a, (hipStream_t)stream);
}

extern "C" int pacoh_gp_lml_fwdbwd(const void* z, int z_div, const void* mean, int mean_mode,
                                   const void* y, int y_div, const void* lengthscale, const void* outputscale,
                                   const void* noise, const int32_t* n_valid, const void* g_lml,
                                   void* lml, void* d_z, void* d_mean, void* d_lengthscale,
                                   void* d_outputscale, void* d_noise, int32_t* info,
                                   int B, int P, int n, int f, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!lml || !d_lengthscale || !d_noise) return PACOH_EINVAL;
    if (dtype == PACOH_F32) {
        auto a = make_args<float>(z, z_div, mean, mean_mode, y, y_div, lengthscale, outputscale, noise, n_valid, B, P, n, f);
        a.g_lml = (const float*)g_lml; a.lml = (float*)lml; a.d_z = (float*)d_z; a.d_mean = (float*)d_mean;
        a.d_ls = (float*)d_lengthscale; a.d_os = (float*)d_outputscale; a.d_noise = (float*)d_noise; a.info = info;
        { int rc = try_mfma(a, true, (hipStream_t)stream); if (rc != 1) return rc; }
        return launch_gp_small<float, MODE_FWDBWD>(a, (hipStream_t)stream);
    }
    auto a = make_args<double>(z, z_div, mean, mean_mode, y, y_div, lengthscale, outputscale, noise, n_valid, B, P, n, f);
    a.g_lml = (const double*)g_lml; a.lml = (double*)lml; a.d_z = (double*)d_z; a.d_mean = (double*)d_mean;
    a.d_ls = (double*)d_lengthscale; a.d_os = (double*)d_outputscale; a.d_noise = (double*)d_noise; a.info = info;
    return launch_gp_small<double, MODE_FWDBWD>(a, (hipStream_t)stream);
}

extern "C" size_t pacoh_gp_predict_workspace_bytes(int B, int n, int m, int dtype, int want_cov) {
    if (!want_cov || B <= 0 || n <= 0 || m <= 0) return 0;
    return (size_t)B * m * n * (dtype == PACOH_F64 ? 8 : 4);
}

extern "C" int pacoh_gp_predict(const void* z_ctx, int z_div, const void* mean_ctx, int mean_mode,
                                const void* y, int y_div, const void* z_tst, int zt_div, const void* mean_tst,
                                const void* lengthscale, const void* outputscale, const void* noise,
                                const int32_t* n_valid, void* mu, void* var, void* cov, int32_t* info,
                                void* workspace, int B, int P, int n, int m, int f, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!mu || !var || !z_tst || m <= 0 || zt_div <= 0) return PACOH_EINVAL;
    if (mean_mode != PACOH_MEAN_ZERO && !mean_tst) return PACOH_EINVAL;
    if (cov && !workspace) return PACOH_EINVAL;
    int rc;
    if (dtype == PACOH_F32) {
        auto a = make_args<float>(z_ctx, z_div, mean_ctx, mean_mode, y, y_div, lengthscale, outputscale, noise, n_valid, B, P, n, f);
        a.z_tst = (const float*)z_tst; a.zt_div = zt_div; a.mean_tst = (const float*)mean_tst;
        a.mu = (float*)mu; a.var = (float*)var; a.V_out = cov ? (float*)workspace : nullptr; a.m = m; a.info = info;
        // marginal predictive of an RBF-family GP at n <= 128, f <= 4: the register-resident MFMA kernel (round 5)
        rc = 1;
        if (a.kind == PACOH_KERNEL_RBF && mfma_enabled() && a.n <= 128 && a.f <= 4 && info && B > 0 && P > 0 && z_div > 0 && y_div > 0 && z_ctx && y &&
            lengthscale && noise && (mean_mode == PACOH_MEAN_ZERO || mean_ctx)) {
            GpMfmaArgs ma = {a.z, a.z_div, a.mean, a.mean_mode, a.y, a.y_div, a.ls, a.os, a.noise, a.n_valid, nullptr,
                             nullptr, info, nullptr, nullptr, nullptr, nullptr, nullptr, a.B, a.P, a.n, a.f};
            GpPredArgs pa = {(const float*)z_tst, zt_div, (const float*)mean_tst, (float*)mu, (float*)var, m, a.V_out};
            rc = gp_reg_predict_try(ma, pa, (hipStream_t)stream);         // (1: not its shape)
            if (rc != 0 && rc != 1) return rc;
        }
        if (rc == 1) rc = launch_gp_small<float, MODE_PREDICT>(a, (hipStream_t)stream);
        if (rc == PACOH_OK && cov) {
            long total = (long)B * m * m;
            hipLaunchKernelGGL(gp_predict_cov_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                               (hipStream_t)stream, (const float*)z_tst, zt_div, (const float*)workspace,
                               (const float*)lengthscale, (const float*)outputscale, (const float*)noise,
                               (float*)cov, B, P, n, m, features_of(f), kernel_of(f));
            rc = launch_status();
        }
        return rc;
    }
    auto a = make_args<double>(z_ctx, z_div, mean_ctx, mean_mode, y, y_div, lengthscale, outputscale, noise, n_valid, B, P, n, f);
    a.z_tst = (const double*)z_tst; a.zt_div = zt_div; a.mean_tst = (const double*)mean_tst;
    a.mu = (double*)mu; a.var = (double*)var; a.V_out = cov ? (double*)workspace : nullptr; a.m = m; a.info = info;
    rc = launch_gp_small<double, MODE_PREDICT>(a, (hipStream_t)stream);
    if (rc == PACOH_OK && cov) {
        long total = (long)B * m * m;
        hipLaunchKernelGGL(gp_predict_cov_kernel<double>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, (const double*)z_tst, zt_div, (const double*)workspace,
                           (const double*)lengthscale, (const double*)outputscale, (const double*)noise,
                           (double*)cov, B, P, n, m, features_of(f), kernel_of(f));
        rc = launch_status();
    }
    return rc;
}
