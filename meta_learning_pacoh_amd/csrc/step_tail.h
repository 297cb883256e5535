// Work of a graph-captured SVGD step that does not need a launch of its own (round 3): device functions run by EXTRA workgroups
// of launches the step has anyway.
//   * the particles' pairwise squared distances (+ the snapshot the in-place update reads) depend on the particles only, which do
//     not change before the step's update: extra workgroups of the MLP forward (mlp_fused.hip), or pacoh_svgd_dist_advance;
//   * the NEXT step's operands -- its task batch gathered from the resident task table, its step scalars -- depend on nothing
//     the current step computes: extra workgroups of the current step's update launch (misc.hip, pacoh_svgd_update_next).
// Together with the hyper-parameter transforms, which the update's own threads apply to the elements they have just written, this
// is everything pacoh_step_begin does for an SVGD step: the step is five launches instead of six.
// Step counter protocol: c = *counter is the row of the feed the step in flight works on.  The prologue of a chunk (host:
// pacoh_step_begin on row 0, then c := -1) leaves row 0 gathered and its scalars in sc2[0]; the forward's tail does c += 1 (nothing
// else in that launch reads c); the update reads its scalars from sc2[c & 1] while its tail writes row c + 1's into sc2[(c + 1) & 1]
// and gathers row c + 1's tasks into the batch buffers, which no other workgroup of the update reads.
#pragma once
#include "common.h"

namespace pacoh {

// squared distance of the particle pair (i, j) = (pair / P, pair % P), j <= i, by one 256-thread workgroup, direct differences;
// the diagonal pairs copy their particle into the snapshot.  The caller separates two calls by a barrier (red is reused).
template <typename T>
__device__ __forceinline__ void svgd_dist_block(const T* __restrict__ X, T* __restrict__ d2, int P, int D, T* __restrict__ snap, int pair) {
    __shared__ T red[4];
    const int i = pair / P, j = pair - i * P;
    if (j > i) return;
    const T* xi = X + (long)i * D;
    if (snap && i == j) {                       // the diagonal pairs copy their particle: the in-place update reads the snapshot
        for (int d = threadIdx.x; d < D; d += 256) snap[(long)i * D + d] = xi[d];
        if (threadIdx.x == 0) d2[i * P + i] = T(0);
        return;
    }
    const T* xj = X + (long)j * D;
    T acc = 0;
    for (int d = threadIdx.x; d < D; d += 256) { T df = xi[d] - xj[d]; acc = fma(df, df, acc); }
    acc = subwave_sum<T>(acc, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { const T tot = (red[0] + red[1]) + (red[2] + red[3]); d2[i * P + j] = tot; d2[j * P + i] = tot; }
}

// the forward's tail: lin = index of this workgroup among the nlin tail workgroups
template <typename T>
struct SvgdDistTail {
    const T* X; T* d2; T* snap; int P, D; long* counter;
};
template <typename T>
__device__ __forceinline__ void svgd_dist_tail(const SvgdDistTail<T>& t, int lin, int nlin) {
    if (lin == 0 && threadIdx.x == 0 && t.counter) *t.counter += 1;
    for (int pair = lin; pair < t.P * t.P; pair += nlin) {
        svgd_dist_block<T>(t.X, t.d2, t.P, t.D, t.snap, pair);
        __syncthreads();
    }
}

// the update's tail (and what its main workgroups need to find their scalars and to publish the transformed hyper-parameters)
template <typename T>
struct StepNextArgs {
    const long* counter;                             // nullptr: no pipelining (the kernel behaves as before)
    T* sc2; int n_sc;                                // two rows of step scalars, used alternately
    const long* idx_all; int tb; const T* sc_all;    // the chunk's draws [rows, tb] and scalars [rows, n_sc]
    const T* x; const T* y; const int32_t* n_valid;  // resident task table
    T* ox; T* oy; int32_t* onv; int nx, ny;          // the batch buffers of the next step
    int off_ls, f, off_os, off_noise, tie; T noise_floor; T* ls; T* os; T* noise;      // hyper-parameters of the UPDATED particles
};

template <typename T>
__device__ __forceinline__ void step_next_tail(const StepNextArgs<T>& a, int lin, int nlin) {
    const long c = *a.counter, row = c + 1;
    for (int blk = lin; blk <= a.tb; blk += nlin) {
        if (blk < a.tb) {
            const long t = a.idx_all[row * a.tb + blk];
            const T* sx = a.x + t * a.nx;
            const T* sy = a.y + t * a.ny;
            T* dx = a.ox + (long)blk * a.nx;
            T* dy = a.oy + (long)blk * a.ny;
            for (int q = threadIdx.x; q < a.nx; q += 256) dx[q] = sx[q];
            for (int q = threadIdx.x; q < a.ny; q += 256) dy[q] = sy[q];
            if (threadIdx.x == 0 && a.n_valid) a.onv[blk] = a.n_valid[t];
        } else {
            T* dst = a.sc2 + (row & 1) * a.n_sc;
            for (int q = threadIdx.x; q < a.n_sc; q += 256) dst[q] = a.sc_all[row * a.n_sc + q];
        }
    }
}

}  // namespace pacoh
