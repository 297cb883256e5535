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
// Workgroups larger than 256 threads (the task-fused step kernel, map_task.hip: 512) leave the work to their first 256 threads, so
// that the sum's order -- and with it the median bandwidth -- is the same bits in whichever launch it rides.
template <typename T>
__device__ __forceinline__ void svgd_dist_block(const T* __restrict__ X, T* __restrict__ d2, int P, int D, T* __restrict__ snap, int pair) {
    __shared__ T red[4];
    const int i = pair / P, j = pair - i * P;
    if (j > i) return;
    const T* xi = X + (long)i * D;
    const bool worker = threadIdx.x < 256;
    if (snap && i == j) {                       // the diagonal pairs copy their particle: the in-place update reads the snapshot
        if (worker) {                           // (eight entries per trip, loaded before any is stored: one entry per trip is one memory round trip each)
            for (int d0 = threadIdx.x; d0 < D; d0 += 8 * 256) {
                T v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int d = d0 + 256 * u; v[u] = xi[d < D ? d : D - 1]; }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int d = d0 + 256 * u; if (d < D) snap[(long)i * D + d] = v[u]; }
            }
        }
        if (threadIdx.x == 0) d2[i * P + i] = T(0);
        return;
    }
    const T* xj = X + (long)j * D;
    T acc = 0;
    if (worker) {
        // (eight entries of both particles per trip, requested together, accumulated in the same order: at the launchers' D = 6 566 the
        //  one-entry loop was 26 memory round trips in a row -- as long as the likelihood workgroups this block rides beside)
        for (int d0 = threadIdx.x; d0 < D; d0 += 8 * 256) {
            T vi[8], vj[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int d = d0 + 256 * u, dc = d < D ? d : D - 1; vi[u] = xi[dc]; vj[u] = xj[dc]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (d0 + 256 * u < D) { const T df = vi[u] - vj[u]; acc = fma(df, df, acc); }
        }
    }
    acc = subwave_sum<T>(acc, 64);
    if (worker && (threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { const T tot = (red[0] + red[1]) + (red[2] + red[3]); d2[i * P + j] = tot; d2[j * P + i] = tot; }
}

// Median of the full PxP squared-distance matrix (numpy.median semantics) from its P(P-1)/2 distinct off-diagonal entries:
// the sorted full matrix is P zeros followed by every pair value twice, so entry m of it is 0 for m < P and u[(m-P)/2]
// otherwise (u = sorted pair values).  u is sorted by ONE wavefront entirely in registers: VPL values per lane, bitonic
// network with lane exchanges by shuffle and register exchanges for strides >= 64 -- no LDS, no barriers (the 512-element LDS
// bitonic sort this replaces spent 45 barrier rounds = 16.5 us on it at P = 20; this takes ~1.5 us).
template <typename T, int VPL>
__device__ __forceinline__ T wave_median_full_matrix(const T* __restrict__ d2, int P, int lane) {
    constexpr int NV = 64 * VPL;
    const int npairs = P * (P - 1) / 2;
    T v[VPL];
#pragma unroll
    for (int q = 0; q < VPL; ++q) {
        const int e = q * 64 + lane;
        T val = T(INFINITY);
        if (e < npairs) {
            // pair index e -> (i, j), i < j, row-major over the strict upper triangle
            int i = (int)((T(2 * P - 1) - t_sqrt<T>(T((2 * P - 1) * (2 * P - 1) - 8 * e))) * T(0.5));
            while (i > 0 && i * (2 * P - i - 1) / 2 > e) --i;
            while ((i + 1) * (2 * P - i - 2) / 2 <= e) ++i;
            const int j = i + 1 + (e - i * (2 * P - i - 1) / 2);
            val = d2[i * P + j];
        }
        v[q] = val;
    }
#pragma unroll
    for (int k = 2; k <= NV; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {                           // partner in another register of the same lane
                const int dq = j >> 6;
#pragma unroll
                for (int q = 0; q < VPL; ++q) {
                    if ((q & dq) == 0) {
                        const bool up = (((q * 64) & k) == 0);           // e & k depends only on q here (k > j >= 64)
                        const T a = v[q], c = v[q | dq];
                        const bool sw = (a > c) == up;
                        v[q] = sw ? c : a; v[q | dq] = sw ? a : c;
                    }
                }
            } else {                                 // partner lane = lane ^ j
#pragma unroll
                for (int q = 0; q < VPL; ++q) {
                    const int e = q * 64 + lane;
                    const T other = shfl_xor_t<T>(v[q], j);
                    const bool up = (e & k) == 0;
                    const bool lower = (lane & j) == 0;
                    const T mn = v[q] < other ? v[q] : other, mx = v[q] < other ? other : v[q];
                    v[q] = (lower == up) ? mn : mx;
                }
            }
        }
    }
    const int N = P * P;
    T mids[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int m = h == 0 ? (N - 1) / 2 : N / 2;
        T val = T(0);
        if (m >= P) {
            const int kidx = (m - P) >> 1;
            const int src_lane = kidx & 63, src_q = kidx >> 6;
            T pick = T(0);
#pragma unroll
            for (int q = 0; q < VPL; ++q) pick = (q == src_q) ? v[q] : pick;
            val = __shfl(pick, src_lane, 64);
        }
        mids[h] = val;
    }
    return (mids[0] + mids[1]) * T(0.5);
}


// The median-heuristic bandwidth of an SVGD step (svgd.py:45-51) ahead of its update: the distances are complete once the forward
// launch has retired, the update only needs the scalar.  One wavefront (the first of the calling workgroup); P <= 64.
template <typename T>
__device__ __forceinline__ T svgd_median_bandwidth(const T* __restrict__ d2, int P, int lane) {
    const int npairs = P * (P - 1) / 2;
    T med;
    if (npairs <= 256) med = wave_median_full_matrix<T, 4>(d2, P, lane);
    else if (npairs <= 512) med = wave_median_full_matrix<T, 8>(d2, P, lane);
    else if (npairs <= 1024) med = wave_median_full_matrix<T, 16>(d2, P, lane);
    else med = wave_median_full_matrix<T, 32>(d2, P, lane);
    return t_sqrt<T>(med / (T(2) * t_log<T>(T(P + 1))));
}
template <typename T>
__device__ __forceinline__ void svgd_bandwidth_block(const T* __restrict__ d2, int P, T* __restrict__ bw_out) {
    if (threadIdx.x < 64) {
        const T bw = svgd_median_bandwidth<T>(d2, P, (int)threadIdx.x);
        if (threadIdx.x == 0) *bw_out = bw;
    }
}

// element index of the bandwidth slot in the workspace of pacoh_svgd_update_dev_workspace_bytes: distances | snapshot | median pair | bw
__host__ __device__ inline long svgd_bw_slot(int P, int D) { return (long)P * P + (long)P * D + 2; }

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
    const T* bw_pre;                                 // the step's bandwidth, computed ahead (svgd_bandwidth_block) | nullptr
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
