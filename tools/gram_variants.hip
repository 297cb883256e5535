// experiment harness: which part of the n=64, f=2 fp32 Gram build costs what?  (stores / exp / staging)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include "../meta_learning_pacoh_amd/csrc/gram.hip"   // the product kernel, timed in the same harness

template <int MODE>   // 0 = product kernel's arithmetic, 1 = plain exp2 (no correction), 2 = prescaled inputs, exp2 only, 3 = no exp at all
__global__ void __launch_bounds__(256) gram64(const float* __restrict__ z, const float* __restrict__ ls, float* __restrict__ K, int P) {
    __shared__ float zs[64 * 2];
    const int b = blockIdx.x, p = b % P;
    const float* zb = z + (long)b * 128;
    if (threadIdx.x < 128) {
        float v = zb[threadIdx.x] / ls[p * 2 + (threadIdx.x & 1)];
        if (MODE == 2) v *= 0.8493218002880191f;      // sqrt(0.5 * log2 e)
        zs[threadIdx.x] = v;
    }
    __syncthreads();
    float* Kb = K + (long)b * 4096;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = u * 256 + threadIdx.x;
        const int i = q >> 4, jq = q & 15;
        const float a0 = zs[i * 2], a1 = zs[i * 2 + 1];
        float out[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float b0 = zs[(jq * 4 + v) * 2], b1 = zs[(jq * 4 + v) * 2 + 1];
            float d0 = a0 - b0, d1 = a1 - b1;
            float s = fmaf(d1, d1, d0 * d0);
            float k;
            if (MODE == 0) {
                const float x = -0.5f * s;
                const float L2E = 1.4426950408889634f, L2E_LO = 1.9259629911266175e-8f;
                const float hi = x * L2E;
                const float lo = fmaf(x, L2E, -hi) + x * L2E_LO;
                const float e = __builtin_amdgcn_exp2f(hi);
                k = fmaf(e, lo * 0.6931471805599453f, e);
            } else if (MODE == 1) {
                k = __builtin_amdgcn_exp2f(s * -0.7213475204444817f);
            } else if (MODE == 2) {
                k = __builtin_amdgcn_exp2f(-s);
            } else {
                k = s;
            }
            out[v] = k;
        }
        *reinterpret_cast<float4*>(Kb + i * 64 + jq * 4) = make_float4(out[0], out[1], out[2], out[3]);
    }
}


// progressively add the general kernel's features to the specialised kernel
// FEAT bit0: stage z1 and z2 separately (256 loads + divisions); bit1: runtime block decomposition + b / z_div divisions;
// bit2: runtime tile shift in the body; bit3: 64-bit runtime row pitch
template <int FEAT>
__global__ void __launch_bounds__(256) gram_feat(const float* __restrict__ z1, int z1_div, const float* __restrict__ z2, int z2_div,
                                                 const float* __restrict__ ls, float* __restrict__ K, int P, int n, int m, int f,
                                                 int tjq_shift, int tiles_i, int tiles_j) {
    extern __shared__ float sm[];
    float* z1s = sm;
    float* z2s = sm + 128;
    int b = blockIdx.x, i0 = 0, j00 = 0;
    if (FEAT & 2) {
        const int tj = blockIdx.x % tiles_j;
        const int rest = blockIdx.x / tiles_j;
        const int ti = rest % tiles_i;
        b = rest / tiles_i;
        i0 = ti * (1024 >> tjq_shift); j00 = tj * (4 << tjq_shift);
    }
    const int p = b % P;
    const float* z1b = (FEAT & 2) ? z1 + (long)(b / z1_div) * n * f : z1 + (long)b * 128;
    const float* z2b = (FEAT & 2) ? z2 + (long)(b / z2_div) * m * f : z2 + (long)b * 128;
    const float* lp = ls + p * 2;
    if (FEAT & 1) {
        for (int e = threadIdx.x; e < 128; e += 256) z1s[e] = z1b[i0 * 2 + e] / lp[e & 1];
        for (int e = threadIdx.x; e < 128; e += 256) z2s[e] = z2b[j00 * 2 + e] / lp[e & 1];
    } else {
        if (threadIdx.x < 128) { float v = z1b[threadIdx.x] / lp[threadIdx.x & 1]; z1s[threadIdx.x] = v; z2s[threadIdx.x] = v; }
    }
    __syncthreads();
    const int sh = (FEAT & 4) ? tjq_shift : 4;
    const int mm = (FEAT & 8) ? m : 64;
    float* Kb = K + (long)b * ((FEAT & 8) ? (long)n * m : 4096);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int q = u * 256 + threadIdx.x;
        const int il = q >> sh, jq = q & ((1 << sh) - 1);
        const int i = i0 + il, j0 = j00 + jq * 4;
        const float a0 = z1s[il * 2], a1 = z1s[il * 2 + 1];
        float out[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float b0 = z2s[(jq * 4 + v) * 2], b1 = z2s[(jq * 4 + v) * 2 + 1];
            float d0 = a0 - b0, d1 = a1 - b1;
            float s = fmaf(d1, d1, d0 * d0);
            const float x = -0.5f * s;
            const float L2E = 1.4426950408889634f, L2E_LO = 1.9259629911266175e-8f;
            const float hi = x * L2E;
            const float lo = fmaf(x, L2E, -hi) + x * L2E_LO;
            const float e = __builtin_amdgcn_exp2f(hi);
            out[v] = fmaf(e, lo * 0.6931471805599453f, e);
        }
        *reinterpret_cast<float4*>(Kb + (long)i * mm + j0) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

template <int FEAT>
void runf(const float* z, const float* ls, float* K, int B, int P) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gram_feat<FEAT>, dim3(B), dim3(256), 1024, 0, z, 1, z, 1, ls, K, P, 64, 64, 2, 4, 1, 1);
    (void)hipEventRecord(a);
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(gram_feat<FEAT>, dim3(B), dim3(256), 1024, 0, z, 1, z, 1, ls, K, P, 64, 64, 2, 4, 1, 1);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double us = ms / 30 * 1e3, bytes = (double)B * (512 + 16384);
    printf("feat %2d                      %.1f us  %.1f GB/s\n", FEAT, us, bytes / us * 1e-3);
}

template <int MODE>
void run(const float* z, const float* ls, float* K, int B, int P, const char* name) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(gram64<MODE>, dim3(B), dim3(256), 0, 0, z, ls, K, P);
    (void)hipEventRecord(a);
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(gram64<MODE>, dim3(B), dim3(256), 0, 0, z, ls, K, P);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double us = ms / 30 * 1e3, bytes = (double)B * (512 + 16384);
    printf("%-28s %.1f us  %.1f GB/s\n", name, us, bytes / us * 1e-3);
}

void run_product(const float* z, const float* ls, float* K, int B, int P) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (int i = 0; i < 5; ++i) pacoh_gram_rbf_ard(z, 1, z, 1, ls, nullptr, nullptr, 0, K, B, P, 64, 64, 2, PACOH_F32, nullptr);
    (void)hipEventRecord(a);
    for (int i = 0; i < 30; ++i) pacoh_gram_rbf_ard(z, 1, z, 1, ls, nullptr, nullptr, 0, K, B, P, 64, 64, 2, PACOH_F32, nullptr);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double us = ms / 30 * 1e3, bytes = (double)B * (512 + 16384);
    printf("product kernel via C ABI     %.1f us  %.1f GB/s\n", us, bytes / us * 1e-3);
}

int main() {
    const int B = 20480, P = 20;
    std::vector<float> hz((size_t)B * 128), hl(P * 2);
    for (size_t i = 0; i < hz.size(); ++i) hz[i] = (float)((i * 2654435761u) % 2000) / 500.f - 2.f;
    for (int i = 0; i < P * 2; ++i) hl[i] = 0.5f + 0.05f * i;
    float *z, *ls, *K;
    (void)hipMalloc(&z, hz.size() * 4); (void)hipMalloc(&ls, hl.size() * 4); (void)hipMalloc(&K, (size_t)B * 4096 * 4);
    (void)hipMemcpy(z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(ls, hl.data(), hl.size() * 4, hipMemcpyHostToDevice);
    run_product(z, ls, K, B, P);

    run<0>(z, ls, K, B, P, "corrected exp (product)");
    run<1>(z, ls, K, B, P, "plain exp2");
    run<2>(z, ls, K, B, P, "prescaled + exp2");
    run<3>(z, ls, K, B, P, "no exp");
    runf<0>(z, ls, K, B, P); runf<1>(z, ls, K, B, P); runf<2>(z, ls, K, B, P); runf<3>(z, ls, K, B, P);
    runf<4>(z, ls, K, B, P); runf<8>(z, ls, K, B, P); runf<7>(z, ls, K, B, P); runf<15>(z, ls, K, B, P);
    return 0;
}
