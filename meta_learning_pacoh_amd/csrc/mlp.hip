// C-ABI entry points of the per-particle MLP (kernels: mlp_impl.h, instantiated in mlp_f32/f64.hip).
#include "common.h"

namespace pacoh {
#define PACOH_MLP_DECL(sfx) \
int mlp_fwd_##sfx(const void*, int, const void*, long, int, int, const int32_t*, int, int, void*, int, int, hipStream_t); \
int mlp_bwd_##sfx(const void*, int, const void*, long, int, int, const int32_t*, int, int, const void*, void*, long, int, void*, int, int, hipStream_t);
PACOH_MLP_DECL(f32)
PACOH_MLP_DECL(f64)
#undef PACOH_MLP_DECL
// register-resident MFMA path (mlp_mfma.hip): fp32, <= 2 hidden layers of width <= 32; returns 1 if not applicable
int mlp_mfma_fwd(const void*, int, const void*, long, int, int, const int32_t*, int, int, void*, int, int, hipStream_t);
size_t mlp_mfma_bwd_workspace(int, int, int, int, const int32_t*, int, int);
int mlp_mfma_bwd(const void*, int, const void*, long, int, int, const int32_t*, int, int, const void*, void*, long, int, void*, int, int, hipStream_t);

// out[o] (+)= scale * sum_c in[c, o]; one wavefront per output element, lanes stride over c, fixed order
template <typename T>
__global__ void __launch_bounds__(256) reduce_tasks_kernel(const T* __restrict__ in, T* __restrict__ out, T scale, int accumulate,
                                                           int C, int P, int Wd) {
    const long tot = (long)P * Wd;
    const long idx = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= tot) return;
    const int lane = threadIdx.x & 63;
    T s = 0;
    for (int c = lane; c < C; c += 64) s += in[(long)c * tot + idx];
    s = subwave_sum<T>(s, 64);
    if (lane == 0) out[idx] = accumulate ? out[idx] + scale * s : scale * s;
}
}  // namespace pacoh

using namespace pacoh;

extern "C" int pacoh_mlp_fwd(const void* x, int x_div, const void* theta, long theta_stride, int P,
                             int d_in, const int32_t* hidden, int n_hidden, int d_out, void* out,
                             int B, int n, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!out) return PACOH_EINVAL;
    if (dtype == PACOH_F32 && x && theta && x_div > 0 && P > 0 && B > 0 && n > 0 && B % P == 0 && d_in > 0 && d_out > 0 &&
        n_hidden >= 1 && hidden && (long)B * n <= 0x3fffffffL) {
        int rc = mlp_mfma_fwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, out, B, n, (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    return dtype == PACOH_F32
        ? mlp_fwd_f32(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, out, B, n, (hipStream_t)stream)
        : mlp_fwd_f64(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, out, B, n, (hipStream_t)stream);
}

static int mlp_layout(int d_in, const int32_t* hidden, int n_hidden, int d_out, int dtype, int& params, int& tile) {
    if (d_in <= 0 || d_out <= 0 || n_hidden < 0 || n_hidden > PACOH_MAX_HIDDEN_LAYERS || (n_hidden > 0 && !hidden)) return -1;
    int prev = d_in, mx = 0;
    params = 0;
    for (int l = 0; l < n_hidden; ++l) {
        if (hidden[l] <= 0 || hidden[l] > PACOH_MAX_WIDTH) return -1;
        params += hidden[l] * (prev + 1); prev = hidden[l]; mx = hidden[l] > mx ? hidden[l] : mx;
    }
    params += d_out * (prev + 1);
    tile = (mx <= 32 ? 256 : 128) / (dtype == PACOH_F64 ? 2 : 1);     // = bwd_tile<T>(HP) in mlp_impl.h
    return 0;
}

extern "C" size_t pacoh_mlp_bwd_workspace_bytes(int B, int P, int n, int d_in, const int32_t* hidden,
                                                int n_hidden, int d_out, int dtype) {
    int params, tile;
    if (P <= 0 || B <= 0 || n <= 0 || mlp_layout(d_in, hidden, n_hidden, d_out, dtype, params, tile)) return 0;
    long rows = (long)(B / P) * n;
    long tiles = (rows + tile - 1) / tile;
    long want = (2048 + P - 1) / P;
    long chunks = tiles < want ? tiles : want;
    if (chunks < 1) chunks = 1;
    size_t need = (size_t)chunks * P * params * (dtype == PACOH_F64 ? 8 : 4);
    if (dtype == PACOH_F32 && n_hidden >= 1) {
        size_t m = mlp_mfma_bwd_workspace(B, P, n, d_in, hidden, n_hidden, d_out);
        if (m > need) need = m;
    }
    return need;
}

extern "C" int pacoh_mlp_bwd(const void* x, int x_div, const void* theta, long theta_stride, int P,
                             int d_in, const int32_t* hidden, int n_hidden, int d_out, const void* g_out,
                             void* d_theta, long d_theta_stride, int accumulate, void* workspace,
                             int B, int n, int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!g_out || !d_theta || !workspace) return PACOH_EINVAL;
    if (dtype == PACOH_F32 && x && theta && x_div > 0 && P > 0 && B > 0 && n > 0 && B % P == 0 && d_in > 0 && d_out > 0 &&
        n_hidden >= 1 && hidden && (long)B * n <= 0x3fffffffL) {
        int rc = mlp_mfma_bwd(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, g_out, d_theta, d_theta_stride,
                              accumulate, workspace, B, n, (hipStream_t)stream);
        if (rc != 1) return rc;
    }
    return dtype == PACOH_F32
        ? mlp_bwd_f32(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, g_out, d_theta, d_theta_stride, accumulate, workspace, B, n, (hipStream_t)stream)
        : mlp_bwd_f64(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, g_out, d_theta, d_theta_stride, accumulate, workspace, B, n, (hipStream_t)stream);
}

extern "C" int pacoh_reduce_tasks(const void* in, void* out, double scale, int accumulate, int T_, int P, int Wd,
                                  int dtype, void* stream) {
    if (check_dtype(dtype)) return PACOH_EDTYPE;
    if (!in || !out || T_ <= 0 || P <= 0 || Wd <= 0) return PACOH_EINVAL;
    long tot = (long)P * Wd;
    if (dtype == PACOH_F32)
        hipLaunchKernelGGL(reduce_tasks_kernel<float>, dim3((unsigned)((tot + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)in, (float*)out, (float)scale, accumulate, T_, P, Wd);
    else
        hipLaunchKernelGGL(reduce_tasks_kernel<double>, dim3((unsigned)((tot + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const double*)in, (double*)out, scale, accumulate, T_, P, Wd);
    return launch_status();
}
