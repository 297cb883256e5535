// accuracy of rbf_exp<double> (common.h) against libm exp on the device, x in [-760, 0]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../meta_learning_pacoh_amd/csrc/common.h"
__global__ void k(const double* x, double* mine, double* ref, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { mine[i] = pacoh::rbf_exp<double>(x[i]); ref[i] = exp(x[i]); }
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), a(n), b(n);
    for (int i = 0; i < n; ++i) { double u = (i + 0.5) / n; x[i] = (i % 3 == 0) ? -760.0 * u : ((i % 3 == 1) ? -40.0 * u * u : -1e-3 * u); }
    double *dx, *da, *db;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&da, n * 8); (void)hipMalloc(&db, n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, n);
    (void)hipMemcpy(a.data(), da, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(b.data(), db, n * 8, hipMemcpyDeviceToHost);
    double worst = 0, worst_host = 0; int bad = 0;
    for (int i = 0; i < n; ++i) {
        if (b[i] > 1e-300) { double e = std::fabs(a[i] - b[i]) / b[i]; if (e > worst) worst = e; double eh = std::fabs(a[i] - std::exp(x[i])) / std::exp(x[i]); if (eh > worst_host) worst_host = eh; }
        else if (std::fabs(a[i] - b[i]) > 1e-300) ++bad;
    }
    printf("max rel err vs device libm %.3e, vs host libm %.3e (normal range); denormal/underflow mismatches %d\n", worst, worst_host, bad);
    return 0;
}
