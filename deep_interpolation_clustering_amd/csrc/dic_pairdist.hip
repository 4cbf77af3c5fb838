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

// Per-cluster f64 sums of the rows of an f64 matrix: out[k][c] = sum over rows i with lab[i] == k of v[i][c] -- the centroids of the Calinski-Harabasz / Davies-
// Bouldin scores and the per-cluster distance sums of the gap statistic (internal_eval.py:112-147, p2_clustering_optK.py:334-351).  Until round 6 a one-hot f64
// GEMM through rocBLAS (the last library GEMM of the K sweep).  grid (row chunks, 64-column blocks): thread (row phase rq, column c) owns private f64 sums for every
// cluster in LDS ([4][K][64]: no atomics, fixed order), the four phases are added, one partial row block per workgroup; a second kernel adds the chunks in order.
constexpr int SEG_CHUNKS = 64;
__global__ __launch_bounds__(256) void segment_sum_kernel(const double* v, long ldv, const long long* lab, int N, int D, int K, double* part) {
    extern __shared__ double seg_acc[];                       // [4][K][64]
    const int tid = threadIdx.x, c = tid & 63, rq = tid >> 6;
    const int col = blockIdx.y * 64 + c;
    for (int i = tid; i < 4 * K * 64; i += 256) seg_acc[i] = 0.0;
    __syncthreads();
    const int per = (N + gridDim.x - 1) / gridDim.x;
    const int r0 = blockIdx.x * per, r1 = min(N, r0 + per);
    double* mine = seg_acc + (size_t)rq * K * 64 + c;
    if (col < D)
        for (int r = r0 + rq; r < r1; r += 4) {
            const int k = (int)lab[r];
            if (k >= 0 && k < K) mine[k * 64] += v[(size_t)r * ldv + col];
        }
    __syncthreads();
    for (int i = tid; i < K * 64; i += 256) {
        const int k = i >> 6, cc = i & 63;
        if (blockIdx.y * 64 + cc < D)
            part[((size_t)blockIdx.x * K + k) * D + blockIdx.y * 64 + cc] = (seg_acc[i] + seg_acc[K * 64 + i]) + (seg_acc[2 * K * 64 + i] + seg_acc[3 * K * 64 + i]);
    }
}
__global__ __launch_bounds__(256) void segment_sum_finalize(const double* part, int nch, int n, double* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int ch = 0; ch < nch; ++ch) s += part[(size_t)ch * n + i];
    out[i] = s;
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

size_t dic_segment_sum_workspace(int D, int K) { return (D > 0 && K > 0) ? (size_t)SEG_CHUNKS * K * D * sizeof(double) : 0; }

int dic_segment_sum_f64(const double* values, long ldv, const int64_t* labels, int N, int D, int K, double* out, void* workspace, size_t workspace_bytes,
                        dic_stream_t stream) {
    DIC_REQUIRE(N > 0 && D > 0 && K > 0 && ldv >= D, DIC_ERR_INVALID_ARG, "segment_sum_f64: N=%d D=%d K=%d ldv=%ld", N, D, K, ldv);
    DIC_REQUIRE(K <= 64, DIC_ERR_UNSUPPORTED, "segment_sum_f64: K=%d > 64", K);
    DIC_REQUIRE(values && labels && out && workspace, DIC_ERR_INVALID_ARG, "segment_sum_f64: NULL pointer");
    DIC_REQUIRE(workspace_bytes >= dic_segment_sum_workspace(D, K), DIC_ERR_WORKSPACE, "segment_sum_f64: workspace %zu < %zu", workspace_bytes,
                dic_segment_sum_workspace(D, K));
    const size_t lds = (size_t)4 * K * 64 * sizeof(double);
    static size_t lds_set = 0;
    if (lds > lds_set) {
        hipError_t e = hipFuncSetAttribute((const void*)segment_sum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "segment_sum_f64: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        lds_set = lds;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(segment_sum_kernel, dim3(SEG_CHUNKS, (D + 63) / 64), dim3(256), lds, st, values, ldv, (const long long*)labels, N, D, K, (double*)workspace);
    hipLaunchKernelGGL(segment_sum_finalize, dim3((K * D + 255) / 256), dim3(256), 0, st, (const double*)workspace, SEG_CHUNKS, K * D, out);
    return check_launch("segment_sum_f64");
}

}  // extern "C"
