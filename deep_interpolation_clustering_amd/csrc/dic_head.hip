// Last layer of CompressFC (rbf.py:111-125: Linear(128, C) applied to all B*R decoder rows through
// TimeDistributed) as streaming kernels.
//
// Why: with C = 6 outputs this is a 393 216 x 128 x 6 "GEMM" at the bench shape; rocprofv3 shows the
// library picking a 256x16 macro-tile for it (0.95 ms forward, ~0.4 ms for the two skinny backward GEMMs)
// although the operation is a 100 MB read.  One thread per row needs no cross-lane reduction; the C x 128
// weight is uniform across the wave (scalar loads).  bf16 activations in/out, f32 weights and accumulation.
#include "dic_common.h"

namespace dic {

constexpr int HK = 128;       // in_features (CompressFC hidden width, compiled in)


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// v[row][j] = b[j] + sum_k h[row][k] W[j][k]
template <int C>
__global__ __launch_bounds__(256) void head_fwd_kernel(const __bf16* h, const float* W, const float* b, long N, float* v) {
    for (long row = (long)blockIdx.x * blockDim.x + threadIdx.x; row < N; row += (long)gridDim.x * blockDim.x) {
        float acc[C];
#pragma unroll
        for (int j = 0; j < C; ++j) acc[j] = b[j];
        const bf16x8* hp = reinterpret_cast<const bf16x8*>(h + row * HK);
#pragma unroll 4
        for (int kk = 0; kk < HK / 8; ++kk) {
            const bf16x8 x = hp[kk];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xe = (float)x[e];
#pragma unroll
                for (int j = 0; j < C; ++j) acc[j] = fmaf(xe, W[j * HK + kk * 8 + e], acc[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < C; ++j) v[row * C + j] = acc[j];
    }
}

// dh[row][k] = sum_j dv[row][j] W[j][k]
template <int C>
__global__ __launch_bounds__(256) void head_bwd_input_kernel(const float* dv, const float* W, long N, __bf16* dh) {
    for (long row = (long)blockIdx.x * blockDim.x + threadIdx.x; row < N; row += (long)gridDim.x * blockDim.x) {
        float g[C];
#pragma unroll
        for (int j = 0; j < C; ++j) g[j] = dv[row * C + j];
        bf16x8* op = reinterpret_cast<bf16x8*>(dh + row * HK);
#pragma unroll 4
        for (int kk = 0; kk < HK / 8; ++kk) {
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < C; ++j) s = fmaf(g[j], W[j * HK + kk * 8 + e], s);
                o[e] = (__bf16)s;
            }
            op[kk] = o;
        }
    }
}

// dW[j][k] = sum_rows dv[row][j] h[row][k],  db[j] = sum_rows dv[row][j]: per-workgroup partials [C][HK + 1]
// 16 lanes per row (8 columns each, one 16-B load), 4 rows per wave instruction
template <int C>
__global__ __launch_bounds__(256) void head_bwd_weight_kernel(const __bf16* h, const float* dv, long N, float* partials) {
    __shared__ float red[4][C][HK + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, slot = lane >> 4, kc = lane & 15;
    float acc[C][8], accb[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        accb[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
    }
    const long stride = (long)gridDim.x * 16;
    for (long row = ((long)blockIdx.x * 4 + wave) * 4 + slot; row < N; row += stride) {
        const bf16x8 x = *reinterpret_cast<const bf16x8*>(h + row * HK + kc * 8);
        float g[C];
#pragma unroll
        for (int j = 0; j < C; ++j) g[j] = dv[row * C + j];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            accb[j] += g[j];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[j][e] = fmaf(g[j], (float)x[e], acc[j][e]);
        }
    }
    // fold the 4 row slots of the wave (lanes l, l^16, l^32, l^48 hold the same columns)
#pragma unroll
    for (int j = 0; j < C; ++j) {
        accb[j] += __shfl_xor(accb[j], 16);
        accb[j] += __shfl_xor(accb[j], 32);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[j][e] += __shfl_xor(acc[j][e], 16);
            acc[j][e] += __shfl_xor(acc[j][e], 32);
        }
    }
    if (slot == 0) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
#pragma unroll
            for (int e = 0; e < 8; ++e) red[wave][j][kc * 8 + e] = acc[j][e];
            if (kc == 0) red[wave][j][HK] = accb[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * (HK + 1); i += 256) {
        const int j = i / (HK + 1), k = i % (HK + 1);
        partials[(size_t)blockIdx.x * C * (HK + 1) + i] = red[0][j][k] + red[1][j][k] + red[2][j][k] + red[3][j][k];
    }
}

__global__ __launch_bounds__(256) void head_bwd_finalize(const float* partials, int nblk, int C, float* dW, float* db) {
    __shared__ double red[256];
    const int n = C * (HK + 1);
    const double s = reduce_partials_32x8(partials, nblk, n, blockIdx.x * 32, red);
    const int i = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x >= 32 || i >= n) return;
    const int j = i / (HK + 1), k = i % (HK + 1);
    if (k < HK) dW[j * HK + k] = (float)s;
    else db[j] = (float)s;
}

static int head_blocks(long N, long per_block) { return (int)max(1L, min((N + per_block - 1) / per_block, (long)8 * kNumCU)); }

#define DIC_DISPATCH_C(CV, ...)                                  \
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
        case 16: { constexpr int C = 16; __VA_ARGS__; } break;   \
        default: set_error("head: out_features=%d not compiled (1..8, 12, 16)", CV); return DIC_ERR_UNSUPPORTED; \
    }

}  // namespace dic

using namespace dic;

extern "C" {

int dic_head_fwd(const void* h, const float* W, const float* b, int64_t N, int K, int C, float* v, dic_stream_t stream) {
    DIC_REQUIRE(N > 0 && C > 0, DIC_ERR_INVALID_ARG, "head_fwd: non-positive size");
    DIC_REQUIRE(K == HK, DIC_ERR_UNSUPPORTED, "head_fwd: in_features %d (compiled for %d)", K, HK);
    DIC_REQUIRE(h && W && b && v, DIC_ERR_INVALID_ARG, "head_fwd: NULL pointer");
    const int grid = head_blocks(N, 256);
    DIC_DISPATCH_C(C, hipLaunchKernelGGL(head_fwd_kernel<C>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const __bf16*)h, W, b, (long)N, v));
    return check_launch("head_fwd");
}

size_t dic_head_bwd_workspace(int64_t N, int K, int C) {
    if (N <= 0 || C <= 0 || K != HK) return 0;
    return (size_t)head_blocks(N, 16 * 8) * C * (HK + 1) * sizeof(float);
}

int dic_head_bwd(const void* h, const float* W, const float* dv, int64_t N, int K, int C, void* dh, float* dW, float* db,
                 void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(N > 0 && C > 0, DIC_ERR_INVALID_ARG, "head_bwd: non-positive size");
    DIC_REQUIRE(K == HK, DIC_ERR_UNSUPPORTED, "head_bwd: in_features %d (compiled for %d)", K, HK);
    DIC_REQUIRE(h && W && dv && dh && dW && db && workspace, DIC_ERR_INVALID_ARG, "head_bwd: NULL pointer");
    const int nblk = head_blocks(N, 16 * 8);
    DIC_REQUIRE(workspace_bytes >= (size_t)nblk * C * (HK + 1) * sizeof(float), DIC_ERR_WORKSPACE, "head_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int grid = head_blocks(N, 256);
    DIC_DISPATCH_C(C, {
        hipLaunchKernelGGL(head_bwd_input_kernel<C>, dim3(grid), dim3(256), 0, st, dv, W, (long)N, (__bf16*)dh);
        hipLaunchKernelGGL(head_bwd_weight_kernel<C>, dim3(nblk), dim3(256), 0, st, (const __bf16*)h, dv, (long)N, (float*)workspace);
    });
    const int n = C * (HK + 1);
    hipLaunchKernelGGL(head_bwd_finalize, dim3((n + 31) / 32), dim3(256), 0, st, (const float*)workspace, nblk, C, dW, db);
    return check_launch("head_bwd");
}

}  // extern "C"
