// All-pairs Euclidean distance statistics for the K sweep of p2 (p2_clustering_optK.py:334-351 gap-statistic
// inertia = pairwise_distances per cluster; internal_eval.py:112-147 silhouette / Dunn): per point i and
// cluster k,  S[i][k] = sum_{j in cluster k} ||x_i - x_j||,  plus the nearest point of every cluster
// and the farthest same-cluster point of every point -- without materialising any n x n matrix (upstream builds n_c x n_c
// float64 matrices: 22 GB for one 53 k-point cluster).
//
// Points arrive SORTED by cluster (seg[k] .. seg[k+1] = rows of cluster k), so a column tile belongs to one
// cluster and its row sums go to one S column: no atomics, fixed summation order (deterministic).
//
// Roofline: this is VALU-bound, not HBM-bound.  Distances are accumulated as sum_d (x_id - x_jd)^2 in f32
// (the ||x||^2 + ||y||^2 - 2 x.y form cancels catastrophically for near points, which is why scikit-learn
// upcasts to f64 for it), 2 packed VALU ops per 2 (pair, d) elements: N^2 D / 64 wave instructions, ~40 ms
// for N = 75 000, D = 256; the 77 MB of latents are read N / 64 times from L2 / Infinity Cache.
//
// Tile: 64 x 64 pairs per workgroup step, 256 threads = 16 x 16, each a 4 x 4 block with rows / columns
// interleaved by 16 (row ty + 16 r, column tx + 16 c) so that the 16 lanes of an LDS access pass read 16
// distinct, conflict-free 16-B bank groups (row stride 36 floats).
#include "dic_common.h"

namespace dic {

constexpr int PT = 64;        // tile edge (points)
constexpr int PDC = 32;       // feature chunk staged through LDS
constexpr int PLD = PDC + 4;  // LDS row stride in floats (144 B: 16 consecutive rows start in distinct bank groups)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float row16_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1));
    v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 8));
    return v;
}
__device__ __forceinline__ float row16_min(float v) {
    v = fminf(v, __shfl_xor(v, 1));
    v = fminf(v, __shfl_xor(v, 2));
    v = fminf(v, __shfl_xor(v, 4));
    v = fminf(v, __shfl_xor(v, 8));
    return v;
}

// INTRA: only the pairs inside a cluster are wanted (the gap statistic's inertia of a reference set, p2:334-351: sum_c n_c^2 pairs instead
// of N^2): a row tile visits the column tiles of the clusters its own rows belong to, and S holds ONE value per point (its own cluster's sum).
template <bool INTRA>
__global__ __launch_bounds__(256) void pairdist_kernel(const float* __restrict__ X, const int* __restrict__ seg, int N, int D, int K,
                                                       float* __restrict__ S, float* __restrict__ Dmin, float* __restrict__ own_max) {
    __shared__ float xi[PT * PLD], xj[PT * PLD];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int i0 = blockIdx.x * PT;
    // staging role: thread loads 16-B pieces (row lr + 32 h, floats lc4 .. lc4+3) of both tiles
    const int lr = tid >> 3, lc4 = (tid & 7) * 4;

    int own[4];            // cluster of each of this thread's rows (rows >= N: -1)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = i0 + ty + 16 * r;
        int lab = -1;
        if (i < N) {
            lab = 0;
            while (lab + 1 < K && seg[lab + 1] <= i) ++lab;        // K <= 64: a short scan
        }
        own[r] = lab;
    }
    float rmax[4] = {0.f, 0.f, 0.f, 0.f};

    int k_lo = 0, k_hi = K - 1;
    if (INTRA) {           // rows are sorted by cluster: the tile's rows span the clusters of its first and of its last valid row (uniform over the workgroup)
        const int first = i0, last = min(i0 + PT, N) - 1;
        while (k_lo + 1 < K && seg[k_lo + 1] <= first) ++k_lo;
        k_hi = k_lo;
        while (k_hi + 1 < K && seg[k_hi + 1] <= last) ++k_hi;
    }
    for (int k = k_lo; k <= k_hi; ++k) {
        const int jbeg = seg[k], jend = seg[k + 1];
        float rsum[4] = {0.f, 0.f, 0.f, 0.f}, rmin[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
        for (int j0 = jbeg; j0 < jend; j0 += PT) {
            v2f acc[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r][c] = (v2f){0.f, 0.f};
            for (int d0 = 0; d0 < D; d0 += PDC) {
                __syncthreads();                                  // previous chunk fully consumed
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int row = lr + 32 * h;
                    const int gi = min(i0 + row, N - 1), gj = max(0, min(j0 + row, N - 1));   // clamped duplicates are masked below
                    v4f a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
                    if (d0 + lc4 < D) {                            // D % 4 == 0: a 16-B piece is inside or outside
                        a = *reinterpret_cast<const v4f*>(X + (size_t)gi * D + d0 + lc4);
                        b = *reinterpret_cast<const v4f*>(X + (size_t)gj * D + d0 + lc4);
                    }
                    *reinterpret_cast<v4f*>(&xi[row * PLD + lc4]) = a;
                    *reinterpret_cast<v4f*>(&xj[row * PLD + lc4]) = -b;      // negated: the difference becomes a packed add
                }
                __syncthreads();
#pragma unroll
                for (int dd = 0; dd < PDC; dd += 4) {
                    v4f a[4], b[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const v4f*>(&xi[(ty + 16 * r) * PLD + dd]);
#pragma unroll
                    for (int c = 0; c < 4; ++c) b[c] = *reinterpret_cast<const v4f*>(&xj[(tx + 16 * c) * PLD + dd]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const v2f lo = a[r].xy + b[c].xy, hi = a[r].zw + b[c].zw;      // b holds -x_j
                            acc[r][c] = __builtin_elementwise_fma(lo, lo, acc[r][c]);
                            acc[r][c] = __builtin_elementwise_fma(hi, hi, acc[r][c]);
                        }
                }
            }
            // distances of this 4 x 4 block; columns past the end of the cluster contribute nothing
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool col_live = j0 + tx + 16 * c < jend;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float dist = sqrtf(acc[r][c].x + acc[r][c].y);
                    if (col_live) {
                        rsum[r] += dist;
                        rmin[r] = fminf(rmin[r], dist);
                        if (own[r] == k) rmax[r] = fmaxf(rmax[r], dist);
                    }
                }
            }
        }
        // end of cluster k: fold the 16 column threads of each row, one store per row
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float s = row16_sum(rsum[r]), mn = row16_min(rmin[r]);
            const int i = i0 + ty + 16 * r;
            if (INTRA) {
                if (tx == 0 && i < N && own[r] == k) S[i] = s;
            } else if (tx == 0 && i < N) {
                S[(size_t)i * K + k] = s;
                if (Dmin) Dmin[(size_t)i * K + k] = mn;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float mx = row16_max(rmax[r]);
        const int i = i0 + ty + 16 * r;
        if (tx == 0 && i < N && own_max) own_max[i] = mx;
    }
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_cluster_pairdist(const float* X, const int32_t* seg, int N, int D, int K, float* S, float* Dmin, float* own_max,
                         dic_stream_t stream) {
    DIC_REQUIRE(N > 0 && D > 0 && K > 0, DIC_ERR_INVALID_ARG, "cluster_pairdist: non-positive size");
    DIC_REQUIRE(D % 4 == 0, DIC_ERR_UNSUPPORTED, "cluster_pairdist: D=%d must be a multiple of 4", D);
    DIC_REQUIRE(K <= 64, DIC_ERR_UNSUPPORTED, "cluster_pairdist: K=%d > 64", K);
    DIC_REQUIRE(X && seg && S, DIC_ERR_INVALID_ARG, "cluster_pairdist: NULL pointer");
    const int grid = (N + PT - 1) / PT;
    hipLaunchKernelGGL(pairdist_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, X, seg, N, D, K, S, Dmin, own_max);
    return check_launch("cluster_pairdist");
}

int dic_cluster_intra_sums(const float* X, const int32_t* seg, int N, int D, int K, float* S_own, dic_stream_t stream) {
    DIC_REQUIRE(N > 0 && D > 0 && K > 0, DIC_ERR_INVALID_ARG, "cluster_intra_sums: non-positive size");
    DIC_REQUIRE(D % 4 == 0, DIC_ERR_UNSUPPORTED, "cluster_intra_sums: D=%d must be a multiple of 4", D);
    DIC_REQUIRE(K <= 64, DIC_ERR_UNSUPPORTED, "cluster_intra_sums: K=%d > 64", K);
    DIC_REQUIRE(X && seg && S_own, DIC_ERR_INVALID_ARG, "cluster_intra_sums: NULL pointer");
    const int grid = (N + PT - 1) / PT;
    hipLaunchKernelGGL(pairdist_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, X, seg, N, D, K, S_own, (float*)nullptr, (float*)nullptr);
    return check_launch("cluster_intra_sums");
}

}  // extern "C"
