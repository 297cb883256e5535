// micro-benchmark for VERDICT r5 #7 ("cfg #2 in one launch per iteration, only with a hierarchical ticket"): is a slab reduction INSIDE the
// task launch -- last arriver per XCD over its L2-local slabs, then last arriver of the eight over the partials, then the AdamW-like
// update -- faster than the launch boundary plus the separate reduction launch the product uses (map_task_kernel +
// fused_reduce_slab_kernel<5>)?  The task kernel's body is emulated: 256 workgroups of 512 threads, each busy for BODY_US microseconds
// (cfg #2: 13.4 by rocprofv3) with a per-workgroup skew of up to SKEW_US, then writes its slab of D = 1 153 floats (one 2 x 32 network).
//   variant A  two launches per iteration:  body + slab  |  reduce (32 lanes per element over the 256 slabs) + update
//   variant B  one launch per iteration:    body + slab + fence + ticket per XCD group; the group's last arriver sums its 32 slabs into a
//              partial, fence + global ticket; the last of the eight sums the partials in fixed order and updates
//   variant C  as B with the payload written through (agent-scope atomic stores, read back by atomic loads) and one acq_rel atomic per
//              hand-off instead of __threadfence() + atomicAdd
// Both replayed from a hipGraph of ITERS iterations; results compared (the two sum in different orders: tolerance).
//   hipcc --offload-arch=gfx950 -O3 -o tools/one_launch_probe tools/one_launch_probe.hip && tools/one_launch_probe [body_us] [skew_us]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int WGS = 256, NT = 512, D = 1153, GROUPS = 8;

struct Args {
    float* slab;        // [WGS][D]
    float* partial;     // [GROUPS][D]
    float* theta; float* m; float* v; float* grad;
    unsigned* ticket;   // [GROUPS + 1], zero between launches
    long body_ticks, skew_ticks;      // wall_clock64 ticks (100 MHz)
};

__device__ __forceinline__ void busy(long ticks) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(1);
}
__device__ __forceinline__ float slab_value(int wg, int w, int it) { return 1e-3f * (float)((wg * 131 + w * 7 + it) % 257) - 0.1f; }
__device__ __forceinline__ void update(const Args& a, int w, float g) {       // AdamW-like: a few loads, a few flops, three stores
    const float m = 0.9f * a.m[w] + 0.1f * g, v = 0.999f * a.v[w] + 0.001f * g * g;
    a.m[w] = m; a.v[w] = v; a.grad[w] = g;
    a.theta[w] = a.theta[w] * 0.9999f - 1e-3f * m / (sqrtf(v) + 1e-8f);
}

template <int ONE>      // 0: body + slab only; 1: tickets behind __threadfence(); 2: payload as agent-scope atomic (write-through) stores and
                        // loads, ONE acq_rel atomic per hand-off (what map_wide_kernel's two workgroups do)
__global__ void __launch_bounds__(NT) body_kernel(Args a, int it) {
    const int t = threadIdx.x, wg = blockIdx.x;
    busy(a.body_ticks + (a.skew_ticks * ((wg * 37) & 255)) / 256);
    float* sl = a.slab + (long)wg * D;
    if (ONE == 2) { for (int w = t; w < D; w += NT) __hip_atomic_store(sl + w, slab_value(wg, w, it), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else for (int w = t; w < D; w += NT) sl[w] = slab_value(wg, w, it);
    if (!ONE) return;
    // ---- stage 1: the last arriver of this workgroup's group (blockIdx & 7: the XCD the dispatcher deals it to) reduces the group's slabs
    __shared__ int role;
    const int g = wg & (GROUPS - 1);
    __syncthreads();
    if (t == 0) {
        if (ONE == 1) { __threadfence(); role = atomicAdd(a.ticket + g, 1u) == (unsigned)(WGS / GROUPS - 1); }
        else role = __hip_atomic_fetch_add(a.ticket + g, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(WGS / GROUPS - 1);
    }
    __syncthreads();
    if (!role) return;
    if (ONE == 1) __threadfence();
    for (int w = t; w < D; w += NT) {
        float v[WGS / GROUPS];
#pragma unroll
        for (int c = 0; c < WGS / GROUPS; ++c) {
            const float* q = a.slab + (long)(g + GROUPS * c) * D + w;
            v[c] = ONE == 2 ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
        }
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < WGS / GROUPS; ++c) s += v[c];
        if (ONE == 2) __hip_atomic_store(a.partial + g * D + w, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else a.partial[g * D + w] = s;
    }
    __syncthreads();
    if (t == 0) {
        if (ONE == 1) { __threadfence(); role = atomicAdd(a.ticket + GROUPS, 1u) == (unsigned)(GROUPS - 1); }
        else role = __hip_atomic_fetch_add(a.ticket + GROUPS, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(GROUPS - 1);
    }
    __syncthreads();
    if (!role) return;
    // ---- stage 2: the last of the eight sums the partials in group order and updates
    if (ONE == 1) __threadfence();
    for (int w = t; w < D; w += NT) {
        float v[GROUPS];
#pragma unroll
        for (int c = 0; c < GROUPS; ++c) v[c] = ONE == 2 ? __hip_atomic_load(a.partial + c * D + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.partial[c * D + w];
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < GROUPS; ++c) s += v[c];
        update(a, w, s);
    }
    if (t <= GROUPS) a.ticket[t] = 0;                      // (ready for the next launch)
}

// the product's reduction launch: 32 lanes per output element over the 256 slabs (fused_reduce_slab_kernel<5>), then the update
__global__ void __launch_bounds__(256) reduce_kernel(Args a) {
    const long idx = ((long)blockIdx.x * 256 + threadIdx.x) >> 5;
    const int part = threadIdx.x & 31;
    float s = 0.0f;
    if (idx < D) for (int c = part; c < WGS; c += 32) s += a.slab[(long)c * D + idx];
#pragma unroll
    for (int msk = 1; msk < 32; msk <<= 1) s += __shfl_xor(s, msk, 64);
    if (idx < D && part == 0) update(a, (int)idx, s);
}

static float run_graph(int one, Args a, int iters, hipStream_t s) {
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int it = 0; it < iters; ++it) {
        if (one == 1) hipLaunchKernelGGL(body_kernel<1>, dim3(WGS), dim3(NT), 0, s, a, it);
        else if (one == 2) hipLaunchKernelGGL(body_kernel<2>, dim3(WGS), dim3(NT), 0, s, a, it);
        else {
            hipLaunchKernelGGL(body_kernel<0>, dim3(WGS), dim3(NT), 0, s, a, it);
            hipLaunchKernelGGL(reduce_kernel, dim3((D * 32 + 255) / 256), dim3(256), 0, s, a);
        }
    }
    CK(hipStreamEndCapture(s, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(exec, s)); CK(hipStreamSynchronize(s));                   // warm-up
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(exec, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
    return best / iters * 1e3f;                            // microseconds per iteration
}

int main(int argc, char** argv) {
    const double body_us = argc > 1 ? atof(argv[1]) : 13.4, skew_us = argc > 2 ? atof(argv[2]) : 1.0;
    const int iters = 200;
    hipStream_t s; CK(hipStreamCreate(&s));
    Args a;
    CK(hipMalloc(&a.slab, sizeof(float) * WGS * D)); CK(hipMalloc(&a.partial, sizeof(float) * GROUPS * D));
    CK(hipMalloc(&a.theta, sizeof(float) * D)); CK(hipMalloc(&a.m, sizeof(float) * D)); CK(hipMalloc(&a.v, sizeof(float) * D));
    CK(hipMalloc(&a.grad, sizeof(float) * D)); CK(hipMalloc(&a.ticket, sizeof(unsigned) * 16));
    a.body_ticks = (long)(body_us * 100.0); a.skew_ticks = (long)(skew_us * 100.0);
    std::vector<float> res[3];
    float us[3];
    for (int one = 0; one < 3; ++one) {
        CK(hipMemset(a.theta, 0, sizeof(float) * D)); CK(hipMemset(a.m, 0, sizeof(float) * D)); CK(hipMemset(a.v, 0, sizeof(float) * D));
        CK(hipMemset(a.ticket, 0, sizeof(unsigned) * 16));
        us[one] = run_graph(one, a, iters, s);
        res[one].resize(D);
        CK(hipMemcpy(res[one].data(), a.grad, sizeof(float) * D, hipMemcpyDeviceToHost));
    }
    double err = 0, err2 = 0, nrm = 0;
    for (int w = 0; w < D; ++w) {
        err += (double)(res[0][w] - res[1][w]) * (res[0][w] - res[1][w]); err2 += (double)(res[0][w] - res[2][w]) * (res[0][w] - res[2][w]);
        nrm += (double)res[0][w] * res[0][w];
    }
    // the bare body (no slab consumer): what either variant adds to it
    Args b = a; float us_body;
    {
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int it = 0; it < iters; ++it) hipLaunchKernelGGL(body_kernel<0>, dim3(WGS), dim3(NT), 0, s, b, it);
        CK(hipStreamEndCapture(s, &graph)); CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipGraphLaunch(exec, s)); CK(hipStreamSynchronize(s));
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(exec, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        us_body = best / iters * 1e3f;
    }
    printf("body %.1f us + skew <= %.1f us, 256 workgroups x 512 threads, slab %d floats, %d iterations per graph, best of 5 replays\n", body_us, skew_us, D, iters);
    printf("  body launches alone                         %7.2f us per iteration\n", us_body);
    printf("  A: body | reduction launch (32 lanes/elem)  %7.2f us per iteration  (+%.2f)\n", us[0], us[0] - us_body);
    printf("  B: one launch, per-XCD then global ticket   %7.2f us per iteration  (+%.2f)   [__threadfence() + atomicAdd]\n", us[1], us[1] - us_body);
    printf("  C: the same, write-through payload          %7.2f us per iteration  (+%.2f)   [agent-scope atomic stores / loads, one acq_rel atomic per hand-off]\n", us[2], us[2] - us_body);
    printf("  results: relative difference of the summed gradient  B %.2e  C %.2e (different summation orders)\n", std::sqrt(err / (nrm + 1e-300)), std::sqrt(err2 / (nrm + 1e-300)));
    return 0;
}
