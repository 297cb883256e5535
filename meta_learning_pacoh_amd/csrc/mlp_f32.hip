// float instantiation of the per-particle MLP kernels (split per dtype to compile in parallel).
#include "mlp_impl.h"
namespace pacoh {
int mlp_fwd_f32(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                const int32_t* hidden, int n_hidden, int d_out, void* out, int B, int n, hipStream_t s) {
    return mlp_fwd_entry<float>(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, out, B, n, s);
}
int mlp_bwd_f32(const void* x, int x_div, const void* theta, long theta_stride, int P, int d_in,
                const int32_t* hidden, int n_hidden, int d_out, const void* g_out, void* d_theta,
                long d_theta_stride, int accumulate, void* workspace, int B, int n, hipStream_t s) {
    return mlp_bwd_entry<float>(x, x_div, theta, theta_stride, P, d_in, hidden, n_hidden, d_out, g_out, d_theta,
                             d_theta_stride, accumulate, workspace, B, n, s);
}
}  // namespace pacoh
