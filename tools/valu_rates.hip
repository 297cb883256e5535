// Issue cost of the vector instructions the PACOH kernels are made of, on gfx950: shader cycles per wave64 instruction when W
// waves per SIMD each run a stream of 16 independent instructions of one kind (HIP events, 2.4 GHz assumed).
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rates.hip -o tools/valu_rates && tools/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define LOOP16(STMT) for (int it = 0; it < iters; ++it) { _Pragma("unroll") for (int k = 0; k < 16; ++k) { STMT; } }

template <int OP>
__global__ void __launch_bounds__(64) k_rate(float* out, int iters, float seed) {
    float x[16];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 y[16];
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 z[4] = {{seed, 0, 0, 0}, {0, seed, 0, 0}, {0, 0, seed, 0}, {0, 0, 0, seed}};
    const float a = 1.0f + seed * 1e-3f * (threadIdx.x & 7), b = seed * 1e-4f;
    const f2 a2 = {a, a}, b2 = {b, b};
    const int idx = ((threadIdx.x + 16) & 63) * 4;
#pragma unroll
    for (int q = 0; q < 16; ++q) { x[q] = seed * q; y[q] = f2{seed * q, seed}; }
    if (OP == 0) LOOP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b)))
    if (OP == 1) LOOP16(asm volatile("v_exp_f32 %0, %0" : "+v"(x[k])))
    if (OP == 2) LOOP16(asm volatile("v_rcp_f32 %0, %0" : "+v"(x[k])))
    if (OP == 3) LOOP16(asm volatile("v_rsq_f32 %0, %0" : "+v"(x[k])))
    if (OP == 4) LOOP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[k]) : "v"(a2), "v"(b2)))
    if (OP == 5) LOOP16(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[k]) : "v"(a2)))
    if (OP == 6) LOOP16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a)))
    if (OP == 7) LOOP16(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[k])))
    if (OP == 8) LOOP16(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(x[k]) : "v"(idx)))
    if (OP == 9) LOOP16(asm volatile("v_log_f32 %0, %0" : "+v"(x[k])))
    if (OP == 10) LOOP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a)))
    if (OP == 11) LOOP16(asm volatile("v_sqrt_f32 %0, %0" : "+v"(x[k])))
    if (OP == 12) LOOP16(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(x[k]) : "v"(a) : "s10", "s11"))
    if (OP == 13) LOOP16(asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(x[k]), "v"(a) : "vcc"))
    if (OP == 14) LOOP16(asm volatile("v_cmp_lt_f32_e64 s[10:11], %0, %1" : : "v"(x[k]), "v"(a) : "s10", "s11"))
    if (OP == 15) LOOP16(asm volatile("v_readlane_b32 s10, %0, 5" : : "v"(x[k]) : "s10"))
    if (OP == 16) LOOP16(asm volatile("v_mov_b32 %0, %1" : "=v"(x[k]) : "v"(a)))
    if (OP == 17) LOOP16(asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[k]) : "v"(a)))
    if (OP == 18) LOOP16(asm volatile("ds_swizzle_b32 %0, %0 offset:0x401F\n s_waitcnt lgkmcnt(0)" : "+v"(x[k])))
    if (OP == 19) LOOP16(asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[k]), "+v"(x[(k + 1) & 15])))
    if (OP == 20) LOOP16(asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x[k]) : "v"(a)); if (k == 7) asm volatile("v_cmp_lt_f32_e32 vcc, %0, %1" : : "v"(x[0]), "v"(a) : "vcc"))
    if (OP == 21) LOOP16(asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[k]) : "v"(a), "v"(b)))
    if (OP == 22) LOOP16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[k]) : "v"(a)))
    if (OP == 23) LOOP16(asm volatile("v_fma_f32 %0, %0, %1, s10" : "+v"(x[k]) : "v"(a) : "s10"))
    if (OP == 24) LOOP16(asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(z[k & 3]) : "v"(a), "v"(b)); asm volatile("v_exp_f32 %0, %0" : "+v"(x[k])))
    float s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += x[q] + y[q][0] + y[q][1] + z[q & 3][q >> 2];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int OP>
static void run(const char* name) {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("%-22s", name);
    for (int w = 1; w <= 4; ++w) {
        const int blocks = cus * 4 * w, iters = 4096;
        float* out;
        hipMalloc(&out, sizeof(float) * blocks * 64);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(64), 0, 0, out, iters, 0.37f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_rate<OP>), dim3(blocks), dim3(64), 0, 0, out, iters, 0.37f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        printf("  %d waves: %5.1f", w, ms / 5 * 1e-3 * 2.4e9 / ((double)iters * 16 * w));
        hipFree(out);
    }
    printf("   cycles per instruction and SIMD\n");
}

int main() {
    run<0>("v_fma_f32"); run<6>("v_mul_f32"); run<10>("v_cndmask_b32"); run<7>("v_add_f32 dpp");
    run<4>("v_pk_fma_f32"); run<5>("v_pk_mul_f32");
    run<1>("v_exp_f32"); run<2>("v_rcp_f32"); run<3>("v_rsq_f32"); run<9>("v_log_f32"); run<11>("v_sqrt_f32");
    run<8>("ds_bpermute_b32+wait"); run<18>("ds_swizzle_b32+wait"); run<19>("v_permlane32_swap");
    run<12>("v_cndmask e64 sgpr"); run<20>("v_cndmask vcc (+cmp/16)"); run<13>("v_cmp_lt_f32 vcc"); run<14>("v_cmp_lt_f32 sgpr");
    run<15>("v_readlane_b32"); run<16>("v_mov_b32"); run<17>("v_xor_b32"); run<21>("v_fmac_f32"); run<22>("v_add_f32"); run<23>("v_fma_f32 sgpr src");
    run<24>("mfma + v_exp pair");
    return 0;
}
