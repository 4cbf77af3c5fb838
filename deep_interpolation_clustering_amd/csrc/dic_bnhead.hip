// Tail of CompressFC (rbf.py:111-125: BatchNorm1d(128) -> ReLU -> [Dropout p=0] -> Linear(128, C)) over all
// N = B*R decoder rows, fused so that the 128-wide hidden activation is never written:
//
//   stats   z -> column sums / sums of squares                                  (1 read of z)
//   fwd     v = b + W relu(gamma (z - mean) rstd + beta)                        (1 read of z, N*C floats out)
//   bwd A   recompute h; column sums of da, da*xhat (BatchNorm backward), dW, db  (1 read of z)
//   bwd B   dz = gamma rstd (da - sum_da/n - xhat sum_dax/n)                    (1 read of z, 1 write of dz)
//
// against BatchNorm fwd/bwd + ReLU fwd/bwd + the separate head kernels (10 passes over (N,128) tensors, 1.8 ms
// of the 11.8 ms step at B = 32768).  All four are HBM streams.  Layout: 16 lanes per row (8 columns = one 16-B
// load each), 4 rows per wave instruction, so every wave access is 1 KB contiguous.  Batch moments are a separate
// first pass because they couple all rows (and, when the batch is sharded, all ranks: the host all-reduces the
// sums between the passes).
#include "dic_bnhead.h"

namespace dic {

// 8 consecutive columns of a row of z as floats: one 16-B load of bf16 (the bf16 step) or two of f32 (the f32 step: the `_f32` entry points)
typedef float bnf32x4 __attribute__((ext_vector_type(4)));
struct Row8 { float v[8]; };
__device__ __forceinline__ Row8 load_row8(const __bf16* p) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(p);
    Row8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r.v[e] = (float)x[e];
    return r;
}
__device__ __forceinline__ Row8 load_row8(const float* p) {
    const bnf32x4 a = *reinterpret_cast<const bnf32x4*>(p), b = *reinterpret_cast<const bnf32x4*>(p + 4);
    Row8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r.v[e] = a[e]; r.v[4 + e] = b[e]; }
    return r;
}
__device__ __forceinline__ void store_row8(__bf16* p, const float (&o)[8]) {
    bf16x8 x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (__bf16)o[e];
    *reinterpret_cast<bf16x8*>(p) = x;
}
__device__ __forceinline__ void store_row8(float* p, const float (&o)[8]) {
    *reinterpret_cast<bnf32x4*>(p) = bnf32x4{o[0], o[1], o[2], o[3]};
    *reinterpret_cast<bnf32x4*>(p + 4) = bnf32x4{o[4], o[5], o[6], o[7]};
}

// partials[blk][2][BK]: sum z, sum z^2
template <typename ZT>
__global__ __launch_bounds__(256) void bn_colstats_kernel(const ZT* z, long N, float* partials) {
    __shared__ float red[4][2][BK];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane >> 4, kc = lane & 15;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[e] = s2[e] = 0.f;
    const long stride = (long)gridDim.x * 16;
    for (long row = ((long)blockIdx.x * 4 + wave) * 4 + slot; row < N; row += stride) {
        const Row8 x = load_row8(z + row * BK + kc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xe = x.v[e];
            s1[e] += xe;
            s2[e] = fmaf(xe, xe, s2[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        s1[e] += __shfl_xor(s1[e], 16);
        s1[e] += __shfl_xor(s1[e], 32);
        s2[e] += __shfl_xor(s2[e], 16);
        s2[e] += __shfl_xor(s2[e], 32);
    }
    if (slot == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[wave][0][kc * 8 + e] = s1[e];
            red[wave][1][kc * 8 + e] = s2[e];
        }
    }
    __syncthreads();
    const int i = threadIdx.x;          // 256 = 2 * BK outputs
    partials[(size_t)blockIdx.x * 2 * BK + i] = red[0][i >> 7][i & 127] + red[1][i >> 7][i & 127] + red[2][i >> 7][i & 127] + red[3][i >> 7][i & 127];
}

__global__ __launch_bounds__(256) void bn_colstats_finalize(const float* partials, int nblk, double nrows, double* sums) {
    __shared__ double red[256];
    const double s = reduce_partials_32x8(partials, nblk, 2 * BK, blockIdx.x * 32, red);
    if (threadIdx.x < 32) sums[blockIdx.x * 32 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) sums[2 * BK] = nrows;        // the row count rides along in the (all-reduced) record
}

// sums [sum z | sum z^2 | rows] (f64, summed over ranks) -> batch moments, and nn.BatchNorm1d's running statistics in the same
// launch (replaces ~15 element-wise torch launches on 128-float vectors per head and step)
__global__ __launch_bounds__(BK) void bn_moments_kernel(const double* sums, float eps, float momentum, float* running_mean,
                                                       float* running_var, long long* num_batches_tracked, float* mean, float* rstd,
                                                       float* count) {
    const int i = threadIdx.x;
    const double n = sums[2 * BK];
    const double m = sums[i] / n;
    const double var = fmax(sums[BK + i] / n - m * m, 0.0);
    mean[i] = (float)m;
    rstd[i] = rsqrtf((float)var + eps);
    if (i == 0) count[0] = (float)n;
    if (running_mean) {
        long long nbt = 0;
        if (num_batches_tracked) nbt = num_batches_tracked[0] + 1;
        const float mom = momentum >= 0.f ? momentum : 1.0f / (float)(nbt > 0 ? nbt : 1);      // momentum=None: cumulative average
        const float unbiased = (float)var * (float)(n / fmax(n - 1.0, 1.0));
        running_mean[i] = (1.0f - mom) * running_mean[i] + mom * (float)m;
        running_var[i] = (1.0f - mom) * running_var[i] + mom * unbiased;
        __syncthreads();
        if (i == 0 && num_batches_tracked) num_batches_tracked[0] = nbt;
    }
}

template <int C, typename ZT>
__global__ __launch_bounds__(256) void bnhead_fwd_kernel(const ZT* z, const float* mean, const float* rstd, const float* gamma,
                                                         const float* beta, const float* W, const float* b, long N, float floor_, float drop_p,
                                                         const unsigned long long* rng, float* v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane >> 4, kc = lane & 15;
    const ColParams p = load_cols(mean, rstd, gamma, beta, kc);
    const Drop drop = load_drop(drop_p, rng);
    float w[C][8];
#pragma unroll
    for (int j = 0; j < C; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) w[j][e] = W[j * BK + kc * 8 + e];
    float bias = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j)
        if (kc == j) bias = b[j];
    const long stride = (long)gridDim.x * 16;
    // sum over the 16 lanes of a row on the vector ALU (DPP row rotations + quad permutes): every lane ends up with the total
    auto row_sum = [](float a) {
        a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x128, 0xF, 0xF, false));     // row_ror:8
        a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x124, 0xF, 0xF, false));     // row_ror:4
        a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
        a += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, a), 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
        return a;
    };
    long row0 = ((long)blockIdx.x * 4 + wave) * 4;
    Row8 xn = {};
    if (row0 + slot < N) xn = load_row8(z + (row0 + slot) * BK + kc * 8);
    for (; row0 < N; row0 += stride) {       // wave-uniform trip count; the next row is in flight while this one is reduced
        const long row = row0 + slot;
        const bool live = row < N;
        const Row8 x = xn;
        if (row + stride < N) xn = load_row8(z + (row + stride) * BK + kc * 8);
        float acc[C], keep[8];
        drop_factors(drop, row, kc, keep);
#pragma unroll
        for (int j = 0; j < C; ++j) acc[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float h = fmaxf(fmaf(x.v[e], p.scale[e], p.shift[e]), floor_) * keep[e];      // floor_ = 0 (ReLU) or -inf (none)
#pragma unroll
            for (int j = 0; j < C; ++j) acc[j] = fmaf(h, w[j][e], acc[j]);
        }
        float out = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float a = row_sum(acc[j]);
            if (kc == j) out = a;
        }
        if (live && kc < C) v[row * C + kc] = out + bias;
    }
}

// recompute the post-ReLU activation h, x-hat and da = (dv W) 1[h > 0] for this lane's 8 columns of one row
template <int C>
__device__ __forceinline__ void recompute_row(const Row8& x, const ColParams& p, const float (&w)[C][8], const float (&g)[C],
                                              float floor_, const float (&keep)[8], float (&h)[8], float (&xhat)[8], float (&da)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float xe = x.v[e];
        const float act = fmaxf(fmaf(xe, p.scale[e], p.shift[e]), floor_);
        h[e] = act * keep[e];                      // what the Linear saw (dropout applied): multiplies dv in dW
        xhat[e] = (xe - p.mean[e]) * p.rstd[e];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) s = fmaf(g[j], w[j][e], s);
        da[e] = act > floor_ ? s * keep[e] : 0.f;
    }
}

// partials[blk][(2 + C) * BK + C]: sum da | sum da*xhat | dW[C][BK] | db[C]
template <int C, typename ZT>
__global__ __launch_bounds__(256) void bnhead_bwd_reduce_kernel(const ZT* z, const float* mean, const float* rstd, const float* gamma,
                                                                const float* beta, const float* W, const float* dv, long N, float floor_, float drop_p,
                                                                const unsigned long long* rng, float* partials) {
    constexpr int NOUT = (2 + C) * BK + C;
    __shared__ float red[4][NOUT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane >> 4, kc = lane & 15;
    const ColParams p = load_cols(mean, rstd, gamma, beta, kc);
    const Drop drop = load_drop(drop_p, rng);
    float w[C][8];
#pragma unroll
    for (int j = 0; j < C; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) w[j][e] = W[j * BK + kc * 8 + e];
    float sda[8], sdx[8], accw[C][8], accb[C];
#pragma unroll
    for (int e = 0; e < 8; ++e) sda[e] = sdx[e] = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        accb[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) accw[j][e] = 0.f;
    }
    const long stride = (long)gridDim.x * 16;
    long row = ((long)blockIdx.x * 4 + wave) * 4 + slot;
    Row8 xn = {};
    float gn[C] = {};
    if (row < N) {
        xn = load_row8(z + row * BK + kc * 8);
#pragma unroll
        for (int j = 0; j < C; ++j) gn[j] = dv[row * C + j];
    }
    for (; row < N; row += stride) {
        const Row8 x = xn;
        float g[C];
#pragma unroll
        for (int j = 0; j < C; ++j) g[j] = gn[j];
        if (row + stride < N) {              // next row in flight while this one is reduced
            xn = load_row8(z + (row + stride) * BK + kc * 8);
#pragma unroll
            for (int j = 0; j < C; ++j) gn[j] = dv[(row + stride) * C + j];
        }
        float h[8], xhat[8], da[8];
        float keep[8];
        drop_factors(drop, row, kc, keep);
        recompute_row<C>(x, p, w, g, floor_, keep, h, xhat, da);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sda[e] += da[e];
            sdx[e] = fmaf(da[e], xhat[e], sdx[e]);
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            accb[j] += g[j];
#pragma unroll
            for (int e = 0; e < 8; ++e) accw[j][e] = fmaf(g[j], h[e], accw[j][e]);
        }
    }
    // fold the 4 row slots of the wave (lanes l, l^16, l^32, l^48 hold the same columns)
    auto fold4 = [](float a) {
        a += __shfl_xor(a, 16);
        a += __shfl_xor(a, 32);
        return a;
    };
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sda[e] = fold4(sda[e]);
        sdx[e] = fold4(sdx[e]);
    }
#pragma unroll
    for (int j = 0; j < C; ++j) {
        accb[j] = fold4(accb[j]);
#pragma unroll
        for (int e = 0; e < 8; ++e) accw[j][e] = fold4(accw[j][e]);
    }
    if (slot == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            red[wave][kc * 8 + e] = sda[e];
            red[wave][BK + kc * 8 + e] = sdx[e];
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave][(2 + j) * BK + kc * 8 + e] = accw[j][e];
            if (kc == 0) red[wave][(2 + C) * BK + j] = accb[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NOUT; i += 256)
        partials[(size_t)blockIdx.x * NOUT + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
}

__global__ __launch_bounds__(256) void bnhead_bwd_finalize(const float* partials, int nblk, int nout, float* sums) {
    __shared__ double red[256];
    const double s = reduce_partials_32x8(partials, nblk, nout, blockIdx.x * 32, red);
    const int i = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x < 32 && i < nout) sums[i] = (float)s;
}

template <int C, typename ZT>
__global__ __launch_bounds__(256) void bnhead_bwd_input_kernel(const ZT* z, const float* mean, const float* rstd, const float* gamma,
                                                               const float* beta, const float* W, const float* dv, const float* sum_da,
                                                               const float* sum_dax, float inv_n, const float* count, long N, float floor_, float drop_p,
                                                               const unsigned long long* rng, ZT* dz) {
    if (count) inv_n = 1.0f / count[0];                      // global row count of the batch moments, on the device
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane >> 4, kc = lane & 15;
    const ColParams p = load_cols(mean, rstd, gamma, beta, kc);
    const Drop drop = load_drop(drop_p, rng);
    float w[C][8], c1[8], c2[8];
#pragma unroll
    for (int j = 0; j < C; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) w[j][e] = W[j * BK + kc * 8 + e];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        c1[e] = sum_da[kc * 8 + e] * inv_n;
        c2[e] = sum_dax[kc * 8 + e] * inv_n;
    }
    const long stride = (long)gridDim.x * 16;
    for (long row = ((long)blockIdx.x * 4 + wave) * 4 + slot; row < N; row += stride) {
        const Row8 x = load_row8(z + row * BK + kc * 8);
        float g[C];
#pragma unroll
        for (int j = 0; j < C; ++j) g[j] = dv[row * C + j];
        float h[8], xhat[8], da[8];
        float keep[8];
        drop_factors(drop, row, kc, keep);
        recompute_row<C>(x, p, w, g, floor_, keep, h, xhat, da);
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = p.scale[e] * (da[e] - c1[e] - xhat[e] * c2[e]);
        store_row8(dz + row * BK + kc * 8, o);
    }
}

static int bnhead_blocks(long N) { return (int)max(1L, min((N + 127) / 128, (long)8 * kNumCU)); }
// the backward reduction holds ~200 VGPRs (2 workgroups per CU) and leaves (2 + C) * 128 partial sums per workgroup
static int bnhead_reduce_blocks(long N) { return (int)max(1L, min((N + 127) / 128, (long)2 * kNumCU)); }

#define DIC_BNHEAD_DISPATCH_C(CV, ...)                           \
    switch (CV) {                                                \
        case 1: { constexpr int C = 1; __VA_ARGS__; } break;     \
        case 2: { constexpr int C = 2; __VA_ARGS__; } break;     \
        case 3: { constexpr int C = 3; __VA_ARGS__; } break;     \
        case 4: { constexpr int C = 4; __VA_ARGS__; } break;     \
        case 5: { constexpr int C = 5; __VA_ARGS__; } break;     \
        case 6: { constexpr int C = 6; __VA_ARGS__; } break;     \
        case 7: { constexpr int C = 7; __VA_ARGS__; } break;     \
        case 8: { constexpr int C = 8; __VA_ARGS__; } break;     \
        case 12: { constexpr int C = 12; __VA_ARGS__; } break;   \
        default: set_error("bnhead: out_features=%d not compiled (1..8, 12)", CV); return DIC_ERR_UNSUPPORTED; \
    }

}  // namespace dic

using namespace dic;

extern "C" {

size_t dic_bn_colstats_workspace(int64_t N, int K) {
    if (N <= 0 || K != BK) return 0;
    return (size_t)bnhead_blocks(N) * 2 * BK * sizeof(float);
}

}  // extern "C"

template <typename ZT>
static int bn_colstats_t(const void* z, int64_t N, int K, double* sums, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(N > 0, DIC_ERR_INVALID_ARG, "bn_colstats: non-positive size");
    DIC_REQUIRE(K == BK, DIC_ERR_UNSUPPORTED, "bn_colstats: width %d (compiled for %d)", K, BK);
    DIC_REQUIRE(z && sums && workspace, DIC_ERR_INVALID_ARG, "bn_colstats: NULL pointer");
    const int nblk = bnhead_blocks(N);
    DIC_REQUIRE(workspace_bytes >= (size_t)nblk * 2 * BK * sizeof(float), DIC_ERR_WORKSPACE, "bn_colstats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_colstats_kernel<ZT>, dim3(nblk), dim3(256), 0, st, (const ZT*)z, (long)N, (float*)workspace);
    hipLaunchKernelGGL(bn_colstats_finalize, dim3(2 * BK / 32), dim3(256), 0, st, (const float*)workspace, nblk, (double)N, sums);
    return check_launch("bn_colstats");
}

extern "C" {
int dic_bn_colstats(const void* z, int64_t N, int K, double* sums, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    return bn_colstats_t<__bf16>(z, N, K, sums, workspace, workspace_bytes, stream);
}
int dic_bn_colstats_f32(const float* z, int64_t N, int K, double* sums, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    return bn_colstats_t<float>(z, N, K, sums, workspace, workspace_bytes, stream);
}

int dic_bn_moments(const double* sums, int K, float eps, float momentum, float* running_mean, float* running_var,
                   int64_t* num_batches_tracked, float* mean, float* rstd, float* count, dic_stream_t stream) {
    DIC_REQUIRE(K == BK, DIC_ERR_UNSUPPORTED, "bn_moments: width %d (compiled for %d)", K, BK);
    DIC_REQUIRE(sums && mean && rstd && count, DIC_ERR_INVALID_ARG, "bn_moments: NULL pointer");
    DIC_REQUIRE((running_mean == nullptr) == (running_var == nullptr), DIC_ERR_INVALID_ARG, "bn_moments: running_mean and running_var go together");
    hipLaunchKernelGGL(bn_moments_kernel, dim3(1), dim3(BK), 0, (hipStream_t)stream, sums, eps, momentum, running_mean, running_var,
                       (long long*)num_batches_tracked, mean, rstd, count);
    return check_launch("bn_moments");
}

}  // extern "C"

template <typename ZT>
static int bnhead_fwd_t(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                        const float* b, int64_t N, int K, int C, int relu, float drop_p, const uint64_t* rng, float* v, dic_stream_t stream) {
    DIC_REQUIRE(N > 0 && C > 0, DIC_ERR_INVALID_ARG, "bnhead_fwd: non-positive size");
    DIC_REQUIRE(K == BK, DIC_ERR_UNSUPPORTED, "bnhead_fwd: in_features %d (compiled for %d)", K, BK);
    DIC_REQUIRE(z && mean && rstd && gamma && beta && W && b && v, DIC_ERR_INVALID_ARG, "bnhead_fwd: NULL pointer");
    DIC_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (drop_p == 0.f || rng), DIC_ERR_INVALID_ARG, "bnhead_fwd: dropout p=%g needs 0 <= p < 1 and an rng state", (double)drop_p);
    const int grid = bnhead_blocks(N);
    DIC_BNHEAD_DISPATCH_C(C, hipLaunchKernelGGL((bnhead_fwd_kernel<C, ZT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const ZT*)z, mean,
                                                rstd, gamma, beta, W, b, (long)N, relu ? 0.f : -INFINITY, drop_p, (const unsigned long long*)rng, v));
    return check_launch("bnhead_fwd");
}

extern "C" {
int dic_bnhead_fwd(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                   const float* b, int64_t N, int K, int C, int relu, float drop_p, const uint64_t* rng, float* v, dic_stream_t stream) {
    return bnhead_fwd_t<__bf16>(z, mean, rstd, gamma, beta, W, b, N, K, C, relu, drop_p, rng, v, stream);
}
int dic_bnhead_fwd_f32(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                       const float* b, int64_t N, int K, int C, int relu, float drop_p, const uint64_t* rng, float* v, dic_stream_t stream) {
    return bnhead_fwd_t<float>(z, mean, rstd, gamma, beta, W, b, N, K, C, relu, drop_p, rng, v, stream);
}

size_t dic_bnhead_bwd_workspace(int64_t N, int K, int C) {
    if (N <= 0 || C <= 0 || K != BK) return 0;
    return (size_t)bnhead_reduce_blocks(N) * ((2 + C) * BK + C) * sizeof(float);
}

}  // extern "C"

template <typename ZT>
static int bnhead_bwd_reduce_t(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                               const float* dv, int64_t N, int K, int C, int relu, float drop_p, const uint64_t* rng, float* sums,
                               void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(N > 0 && C > 0, DIC_ERR_INVALID_ARG, "bnhead_bwd_reduce: non-positive size");
    DIC_REQUIRE(K == BK, DIC_ERR_UNSUPPORTED, "bnhead_bwd_reduce: in_features %d (compiled for %d)", K, BK);
    DIC_REQUIRE(z && mean && rstd && gamma && beta && W && dv && sums && workspace, DIC_ERR_INVALID_ARG, "bnhead_bwd_reduce: NULL pointer");
    const int nblk = bnhead_reduce_blocks(N), nout = (2 + C) * BK + C;
    DIC_REQUIRE(workspace_bytes >= (size_t)nblk * nout * sizeof(float), DIC_ERR_WORKSPACE, "bnhead_bwd_reduce: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    DIC_BNHEAD_DISPATCH_C(C, hipLaunchKernelGGL((bnhead_bwd_reduce_kernel<C, ZT>), dim3(nblk), dim3(256), 0, st, (const ZT*)z, mean, rstd, gamma,
                                                beta, W, dv, (long)N, relu ? 0.f : -INFINITY, drop_p, (const unsigned long long*)rng, (float*)workspace));
    hipLaunchKernelGGL(bnhead_bwd_finalize, dim3((nout + 31) / 32), dim3(256), 0, st, (const float*)workspace, nblk, nout, sums);
    return check_launch("bnhead_bwd_reduce");
}

extern "C" {
int dic_bnhead_bwd_reduce(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                          const float* dv, int64_t N, int K, int C, int relu, float drop_p, const uint64_t* rng, float* sums,
                          void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    return bnhead_bwd_reduce_t<__bf16>(z, mean, rstd, gamma, beta, W, dv, N, K, C, relu, drop_p, rng, sums, workspace, workspace_bytes, stream);
}
int dic_bnhead_bwd_reduce_f32(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                              const float* dv, int64_t N, int K, int C, int relu, float drop_p, const uint64_t* rng, float* sums,
                              void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    return bnhead_bwd_reduce_t<float>(z, mean, rstd, gamma, beta, W, dv, N, K, C, relu, drop_p, rng, sums, workspace, workspace_bytes, stream);
}

}  // extern "C"

template <typename ZT>
static int bnhead_bwd_input_t(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                              const float* dv, const float* sum_da, const float* sum_dax, double inv_n, const float* count, int64_t N, int K, int C, int relu,
                              float drop_p, const uint64_t* rng, void* dz, dic_stream_t stream) {
    DIC_REQUIRE(N > 0 && C > 0, DIC_ERR_INVALID_ARG, "bnhead_bwd_input: non-positive size");
    DIC_REQUIRE(K == BK, DIC_ERR_UNSUPPORTED, "bnhead_bwd_input: in_features %d (compiled for %d)", K, BK);
    DIC_REQUIRE(z && mean && rstd && gamma && beta && W && dv && sum_da && sum_dax && dz, DIC_ERR_INVALID_ARG, "bnhead_bwd_input: NULL pointer");
    const int grid = bnhead_blocks(N);
    DIC_BNHEAD_DISPATCH_C(C, hipLaunchKernelGGL((bnhead_bwd_input_kernel<C, ZT>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const ZT*)z, mean,
                                                rstd, gamma, beta, W, dv, sum_da, sum_dax, (float)inv_n, count, (long)N, relu ? 0.f : -INFINITY, drop_p,
                                                (const unsigned long long*)rng, (ZT*)dz));
    return check_launch("bnhead_bwd_input");
}

extern "C" {
int dic_bnhead_bwd_input(const void* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                         const float* dv, const float* sum_da, const float* sum_dax, double inv_n, const float* count, int64_t N, int K, int C, int relu,
                         float drop_p, const uint64_t* rng, void* dz, dic_stream_t stream) {
    return bnhead_bwd_input_t<__bf16>(z, mean, rstd, gamma, beta, W, dv, sum_da, sum_dax, inv_n, count, N, K, C, relu, drop_p, rng, dz, stream);
}
int dic_bnhead_bwd_input_f32(const float* z, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* W,
                             const float* dv, const float* sum_da, const float* sum_dax, double inv_n, const float* count, int64_t N, int K, int C, int relu,
                             float drop_p, const uint64_t* rng, float* dz, dic_stream_t stream) {
    return bnhead_bwd_input_t<float>(z, mean, rstd, gamma, beta, W, dv, sum_da, sum_dax, inv_n, count, N, K, C, relu, drop_p, rng, dz, stream);
}

}  // extern "C"
