// gp_reg_kernel: the register-resident fused task-GP kernel as its own launch -- one 64-lane wavefront (= one workgroup) per
// (task, particle) problem.  The arithmetic is gp_reg_body (gp_reg_body.h, where the algorithm is described); this file holds the
// launch: LDS scratch, register budget per shape, dispatch on the block count.
// Reference lines replaced: random_gp.py:54-89, GPR_meta_mll.py:104-117 (ExactMarginalLogLikelihood + autograd through gpytorch).
#include "gp_reg_body.h"
#include <stdlib.h>

namespace pacoh {

// waves per SIMD the register allocation aims at: 4 (the n <= 64, f <= 2 backward kernel needs 122 registers and 7.7 KB of LDS);
// with f <= 4 the n = 64 kernels need 168 - 186 registers (3 resp. 2 waves).  n > 64: TWO (round 5) -- the live set of the
// algorithm is 44 blocks = 176 registers + temporaries, it was the 28 KB of parked W blocks per problem that held the kernel at one
// wave per SIMD, and the compiler, knowing that, spread over 396 registers.  With the blocks consumed where they are produced
// (gp_reg_body.h) the n = 128, f <= 2 kernel takes 244 registers and 4.8 KB: 0.982 -> 0.682 ms per 20 480 problems
// (profiles/r05_gp_two_waves.txt).  f <= 4 at n = 128 would spill 200 registers at that budget and stays at one wave.
#ifdef PACOH_GPR_W5      // A/B: five waves per SIMD for the n = 64, f <= 2 backward kernel (96 registers: 30 of them spilled)
#define GPR_W64 5
#else
#define GPR_W64 4
#endif
#define GPR_MINW(NB, FP, BWD) ((NB) > 4 ? ((FP) == 2 || (NB) == 6 ? 2 : 1) : ((NB) == 4 && (FP) == 4 ? ((BWD) ? 2 : 3) : ((NB) == 4 && (BWD) ? GPR_W64 : 4)))
template <int NB, int FP, bool BWD, bool HAS_OS = true>
__global__ void __launch_bounds__(64, GPR_MINW(NB, FP, BWD)) gp_reg_kernel(GpMfmaArgs a) {
    constexpr int NP = 16 * NB;
    constexpr int NU = NB * (NB + 1) / 2;
    __shared__ __attribute__((aligned(16))) float zf[NP * FP];      // features / lengthscale
    __shared__ __attribute__((aligned(16))) float rv[NP];           // residual
    __shared__ __attribute__((aligned(16))) float av[NP];           // alpha
    // factor16(): every lane row's registers -> all lane rows; the SAME memory is the 16x16 block transpose scratch (tsc, 320 floats:
    // V_K = -L_KK^-T and the transposes of the gradient loop) -- one wave, in-order LDS, and no transpose is in flight across a
    // factor16() call.  (As two arrays the round-6 factor16 slots made a problem 10.75 KB: 14 instead of 16 problems per CU, +11 %.)
    __shared__ __attribute__((aligned(16))) float fsc[gpreg::GPR_SCR];
    float* tsc = fsc;
    __shared__ __attribute__((aligned(16))) float dzc[BWD ? NP * FP : 1];   // d_z before the chain-rule factors
    // W = K^-1, strictly upper block triangle, each block as the 64 lanes' accumulator registers (one 16-byte slot per lane): parked
    // here between the matrix-core phase that produces it and the gradient loop that consumes it, so that the two phases do not
    // have to share the register file (1 KB per block; with it a problem holds 7.7 KB of LDS = 20 problems per CU)
    __shared__ __attribute__((aligned(16))) float Wl[BWD && NB > 1 && NB <= 4 ? (NU - NB) * 256 : 4];   // (n > 64: no parking, see gp_reg_body.h)
    gpreg::gp_reg_body<NB, FP, BWD, HAS_OS>(a, gpreg::KernelCtx{}, zf, rv, av, fsc, tsc, dzc, Wl);
}

// the posterior predictive on the same body (PRED): mu / var at m test points per problem, n <= 128, f <= 4; V = L^-1 K_xs from the
// registers, nothing parked in LDS
template <int NB, int FP>
__global__ void __launch_bounds__(64, NB > 4 ? (FP == 2 || NB == 6 ? 2 : 1) : (NB >= 3 && FP == 4 ? 3 : 4))
gp_reg_predict_kernel(GpMfmaArgs a, GpPredArgs pa) {
    constexpr int NP = 16 * NB;
    __shared__ __attribute__((aligned(16))) float zf[NP * FP];
    __shared__ __attribute__((aligned(16))) float rv[NP];
    __shared__ __attribute__((aligned(16))) float av[NP];
    __shared__ __attribute__((aligned(16))) float fsc[gpreg::GPR_SCR];
    float* tsc = fsc;                                               // (shared with the transpose scratch: gp_reg_kernel)
    __shared__ __attribute__((aligned(16))) float dzc[4];
    __shared__ __attribute__((aligned(16))) float Wl[4];
    gpreg::gp_reg_body<NB, FP, true, true, gpreg::KernelCtx, true>(a, gpreg::KernelCtx{}, zf, rv, av, fsc, tsc, dzc, Wl, &pa);
}

template <int NB>
static int launch_reg_predict(const GpMfmaArgs& a, const GpPredArgs& pa, int FP, hipStream_t s) {
    if (FP == 2) hipLaunchKernelGGL((gp_reg_predict_kernel<NB, 2>), dim3((unsigned)a.B), dim3(64), 0, s, a, pa);
    else hipLaunchKernelGGL((gp_reg_predict_kernel<NB, 4>), dim3((unsigned)a.B), dim3(64), 0, s, a, pa);
    return launch_status();
}

// returns 1 if this path does not apply (n > 128, f > 4, or PACOH_GP_REG=0 / PACOH_GP_REG_PREDICT=0)
int gp_reg_predict_try(const GpMfmaArgs& a, const GpPredArgs& pa, hipStream_t s) {
    if (!g_sw.gp_reg || !g_sw.gp_reg_predict || a.n > 128 || a.f > 4 || a.n < 1 || pa.m < 1) return 1;
    const int NB = (a.n + 15) / 16;
    const int FP = a.f <= 2 ? 2 : 4;
    switch (NB) {
        case 1: return launch_reg_predict<1>(a, pa, FP, s);
        case 2: return launch_reg_predict<2>(a, pa, FP, s);
        case 3: return launch_reg_predict<3>(a, pa, FP, s);
        case 4: return launch_reg_predict<4>(a, pa, FP, s);
        case 5: case 6: return launch_reg_predict<6>(a, pa, FP, s);
        default: return launch_reg_predict<8>(a, pa, FP, s);
    }
}

template <int NB, bool BWD>
static int launch_reg(const GpMfmaArgs& a, int FP, hipStream_t s) {
    const unsigned pad = g_sw.lds_pad_gp > 0 ? (unsigned)g_sw.lds_pad_gp : 0u;
    if (FP == 2) {
        if constexpr (BWD && (NB == 4 || NB == 8)) {
            if (!a.os && !a.d_os) {
                hipLaunchKernelGGL((gp_reg_kernel<NB, 2, true, false>), dim3((unsigned)a.B), dim3(64), pad, s, a);
                return launch_status();
            }
        }
        hipLaunchKernelGGL((gp_reg_kernel<NB, 2, BWD>), dim3((unsigned)a.B), dim3(64), pad, s, a);
    } else hipLaunchKernelGGL((gp_reg_kernel<NB, 4, BWD>), dim3((unsigned)a.B), dim3(64), pad, s, a);
    return launch_status();
}

// returns 1 if this path does not apply (n > 128, f > 4, or PACOH_GP_REG=0)
int gp_reg_try(const GpMfmaArgs& a, bool bwd, hipStream_t s) {
    if (!g_sw.gp_reg) return 1;
    // (n > 64: one wave still holds the whole matrix -- 244 registers at n = 128, two waves per SIMD -- and beats the LDS-resident
    //  kernel, which runs one wave per SIMD there and moves every block through LDS)
    const int max_n = g_sw.gp_reg_max_n;
    if (a.n > max_n || a.n > 128 || a.f > 4 || a.n < 1) return 1;
    const int NB = (a.n + 15) / 16;
    const int FP = a.f <= 2 ? 2 : 4;
#define PACOH_GPR_NB(nb) case nb: return bwd ? launch_reg<nb, true>(a, FP, s) : launch_reg<nb, false>(a, FP, s);
    switch (NB) { PACOH_GPR_NB(1) PACOH_GPR_NB(2) PACOH_GPR_NB(3) PACOH_GPR_NB(4) case 5: PACOH_GPR_NB(6) default: PACOH_GPR_NB(8) }
#undef PACOH_GPR_NB
}

}  // namespace pacoh
