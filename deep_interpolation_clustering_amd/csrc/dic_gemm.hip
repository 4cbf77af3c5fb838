// Row-streaming MFMA products for the dense layers of the step in the shapes no specialised kernel covers -- and for the f32 step, whose
// arithmetic is the reference's (clustering_interp.py:14-41: nn.LSTM / nn.Linear in f32, dataloader.py:204):
//
//   gemm_nt   Y[m][n] = sum_k act(A[m][k]) W[n][k] (+ bias[n])      M = (time step, encounter) rows in the hundreds of thousands, N, K <= 1024:
//             the LSTM input projections gx = X.W_ih^T + b, CompressFC's Linear(256,128), and -- with W handed over transposed -- the input
//             gradients dX = dG.W_ih, dX = dZ.W1
//   gemm_tn   D[n][k] (+)= sum_m A[m][n] X[m][k]                      the weight gradients dW = dG^T.X: the reduction runs over the rows, the slow
//             index of both operands -> both MFMA operands leave LDS transposed (ds_read_b64_tr_b16); split over row chunks, f32 partials,
//             fixed-order f64 second stage (deterministic, like every cross-workgroup reduction of this library)
//
// Operand types:  bf16 -> one v_mfma_f32_32x32x16_bf16 per product (the bf16 step's small-batch / odd-shape fallbacks: no library GEMM);
//                 f32  -> THREE: x = hi + lo with hi = bf16(x), lo = bf16(x - hi) on the way into LDS, x.w ~ hi.hi + lo.hi + hi.lo in
//                         f32 accumulators ("bf16x3").  The dropped lo.lo term and the rounding of lo are O(2^-17) per product: the joint step's
//                         losses stay within 1e-6 of the f32 reference (tests/test_gpu_traj.py, oracle emulation in DESIGN.md) at 5x the
//                         exact-f32 MFMA rate (v_mfma_f32_32x32x2_f32: 157 TF/s peak against 2.5 PF/s / 3).
// Tiling: 256 threads = 4 waves in 2 x 2, wave tile 64 x 64 (or 64 x 32) = 32x32 blocks; one LDS tile image, the next tile's global loads
// in flight in registers while the matrix cores work on the current one; 16-B fragment reads at an 80-B row pitch (conflict-free).  Workgroup
// ids are remapped so that the tiles that share an A row block (gemm_nt) / a row chunk (gemm_tn) sit on one XCD and meet in its L2.
#include "dic_common.h"

namespace dic {

typedef __bf16 gbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gbf16x4 __attribute__((ext_vector_type(4)));
typedef short gs16x4 __attribute__((ext_vector_type(4)));
typedef short gs16x8 __attribute__((ext_vector_type(8)));
typedef float gf32x16 __attribute__((ext_vector_type(16)));
typedef float gf32x4 __attribute__((ext_vector_type(4)));

constexpr int GBM = 128;           // rows of A per workgroup (gemm_nt) / output rows n per workgroup (gemm_tn)
constexpr int GBK = 32;            // reduction elements per LDS tile = 2 MFMA k-steps
constexpr int GP = GBK + 8;        // bf16 elements per LDS row of a k-contiguous tile (80 B: conflict-free ds_read_b128)

// 16 consecutive reduction elements of one tile row, as they come out of global memory
template <typename T> struct Chunk16;
template <> struct Chunk16<float> { gf32x4 v[4]; };
template <> struct Chunk16<__bf16> { gbf16x8 v[2]; };

template <typename T> __device__ __forceinline__ void chunk_zero(Chunk16<T>& c);
template <> __device__ __forceinline__ void chunk_zero<float>(Chunk16<float>& c) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c.v[i] = gf32x4{0.f, 0.f, 0.f, 0.f};
}
template <> __device__ __forceinline__ void chunk_zero<__bf16>(Chunk16<__bf16>& c) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) c.v[i][j] = (__bf16)0.f;
}
// elements [k0, k0 + 16) of a row whose valid length is `klen` (klen % 4 == 0 for f32, % 8 == 0 for bf16: vector pieces are all-in or all-out)
__device__ __forceinline__ void chunk_load(Chunk16<float>& c, const float* row, int k0, int klen) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c.v[i] = (k0 + 4 * i < klen) ? *reinterpret_cast<const gf32x4*>(row + k0 + 4 * i) : gf32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void chunk_load(Chunk16<__bf16>& c, const __bf16* row, int k0, int klen) {
    gbf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) c.v[i] = (k0 + 8 * i < klen) ? *reinterpret_cast<const gbf16x8*>(row + k0 + 8 * i) : z;
}
// -> LDS: hi (and lo) images, 16 elements at `dst`
__device__ __forceinline__ void chunk_store(const Chunk16<float>& c, __bf16* hi, __bf16* lo, bool relu) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        gbf16x8 vh, vl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = c.v[2 * h + (j >> 2)][j & 3];
            if (relu) x = fmaxf(x, 0.f);
            const __bf16 xh = (__bf16)x;
            vh[j] = xh;
            vl[j] = (__bf16)(x - (float)xh);
        }
        *reinterpret_cast<gbf16x8*>(hi + 8 * h) = vh;
        *reinterpret_cast<gbf16x8*>(lo + 8 * h) = vl;
    }
}
__device__ __forceinline__ void chunk_store(const Chunk16<__bf16>& c, __bf16* hi, __bf16*, bool relu) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        gbf16x8 v = c.v[h];
        if (relu) {
            const gs16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
            v = __builtin_bit_cast(gbf16x8, __builtin_elementwise_max(__builtin_bit_cast(gs16x8, v), zero));      // bf16 relu on the int16 images
        }
        *reinterpret_cast<gbf16x8*>(hi + 8 * h) = v;
    }
}

// the first 8 elements only (gemm_tn's 64-column X tiles: 8 columns per thread)
__device__ __forceinline__ void chunk_store8(const Chunk16<float>& c, __bf16* hi, __bf16* lo, bool relu) {
    gbf16x8 vh, vl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = c.v[j >> 2][j & 3];
        if (relu) x = fmaxf(x, 0.f);
        const __bf16 xh = (__bf16)x;
        vh[j] = xh;
        vl[j] = (__bf16)(x - (float)xh);
    }
    *reinterpret_cast<gbf16x8*>(hi) = vh;
    *reinterpret_cast<gbf16x8*>(lo) = vl;
}
__device__ __forceinline__ void chunk_store8(const Chunk16<__bf16>& c, __bf16* hi, __bf16*, bool relu) {
    gbf16x8 v = c.v[0];
    if (relu) {
        const gs16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        v = __builtin_bit_cast(gbf16x8, __builtin_elementwise_max(__builtin_bit_cast(gs16x8, v), zero));
    }
    *reinterpret_cast<gbf16x8*>(hi) = v;
}

template <typename T> struct OutCvt;
template <> struct OutCvt<float> { __device__ static float of(float v) { return v; } };
template <> struct OutCvt<__bf16> { __device__ static __bf16 of(float v) { return (__bf16)v; } };

// ------------------------------------------------------------------------------------------------------------------------ gemm_nt
struct GemmNtArgs {
    const void* A; long lda;       // (M, K) rows, element stride lda
    const void* W; long ldw;       // (N, K) rows
    const float* bias;             // (N) or NULL
    void* Y; long ldy;             // (M, N)
    long M; int N, K;
    int relu_a;                    // the product runs on max(A, 0)
    int tiles_n;
    long a_plane = 0;              // AP kernels: A is a pair of bf16 planes -- hi at A, lo a_plane elements behind it (lda in bf16 elements): x = hi + lo
};

// AP ("A in split planes", round 6): the A operand arrives as two bf16 planes hi / lo with A = hi + lo (what the x3 recurrence kernels write: the gate
// gradients) -- its pieces go to the two LDS images as they are, no conversion in the loop; W (small, L2-resident) is still split on the way in.
template <typename TIN, typename TOUT, int BN, bool AP = false>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmNtArgs a) {
    static_assert(!AP || sizeof(TIN) == 4, "split planes stand for an f32 operand");
    constexpr bool SPLIT = sizeof(TIN) == 4;
    constexpr int NB = BN / 64;                              // 32-column blocks per wave
    constexpr int IMG = SPLIT ? 2 : 1;
    __shared__ __align__(16) __bf16 sa[IMG][GBM * GP];
    __shared__ __align__(16) __bf16 sw[IMG][BN * GP];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 1, wn = w >> 1;
    const int lt = xcd_remap(blockIdx.x, gridDim.x);        // consecutive logical tiles (the n-tiles of one row block) share an XCD
    const long m0 = (long)(lt / a.tiles_n) * GBM;
    const int n0 = (lt % a.tiles_n) * BN;
    const TIN* A = reinterpret_cast<const TIN*>(a.A);
    const TIN* W = reinterpret_cast<const TIN*>(a.W);
    const int lrow = tid >> 1, lk = 16 * (tid & 1);         // this thread's tile row / first reduction element of its 16
    const TIN* arow = A + (size_t)min(m0 + lrow, a.M - 1) * a.lda;
    const __bf16* aph = reinterpret_cast<const __bf16*>(a.A) + (size_t)min(m0 + lrow, a.M - 1) * a.lda;      // (AP)
    const __bf16* apl = aph + a.a_plane;
    const bool wload = lrow < BN;
    const bool wvalid = wload && n0 + lrow < a.N;
    const TIN* wrow = W + (size_t)min(n0 + lrow, a.N - 1) * a.ldw;

    gf32x16 acc[2][NB];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[mb][nb][k] = 0.f;

    Chunk16<TIN> ca, cw;
    Chunk16<__bf16> cah, cal;
    const int nkt = (a.K + GBK - 1) / GBK;
    if constexpr (AP) { chunk_load(cah, aph, lk, a.K); chunk_load(cal, apl, lk, a.K); }
    else chunk_load(ca, arow, lk, a.K);
    if (wvalid) chunk_load(cw, wrow, lk, a.K); else chunk_zero(cw);
    for (int kt = 0; kt < nkt; ++kt) {
        if constexpr (AP) {
            chunk_store(cah, &sa[0][lrow * GP + lk], nullptr, false);
            chunk_store(cal, &sa[1][lrow * GP + lk], nullptr, false);
        } else {
            chunk_store(ca, &sa[0][lrow * GP + lk], &sa[IMG - 1][lrow * GP + lk], a.relu_a != 0);
        }
        if (wload) chunk_store(cw, &sw[0][lrow * GP + lk], &sw[IMG - 1][lrow * GP + lk], false);
        __syncthreads();
        if (kt + 1 < nkt) {                                  // in flight while the matrix cores work on this tile
            if constexpr (AP) { chunk_load(cah, aph, (kt + 1) * GBK + lk, a.K); chunk_load(cal, apl, (kt + 1) * GBK + lk, a.K); }
            else chunk_load(ca, arow, (kt + 1) * GBK + lk, a.K);
            if (wvalid) chunk_load(cw, wrow, (kt + 1) * GBK + lk, a.K);
        }
#pragma unroll
        for (int ks = 0; ks < GBK / 16; ++ks) {
            gbf16x8 ah[2], al[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int o = (64 * wm + 32 * mb + r) * GP + ks * 16 + 8 * hh;
                ah[mb] = *reinterpret_cast<const gbf16x8*>(&sa[0][o]);
                if (SPLIT) al[mb] = *reinterpret_cast<const gbf16x8*>(&sa[IMG - 1][o]);
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int o = ((BN / 2) * wn + 32 * nb + r) * GP + ks * 16 + 8 * hh;
                const gbf16x8 bh = *reinterpret_cast<const gbf16x8*>(&sw[0][o]);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bh, acc[mb][nb], 0, 0, 0);
                if (SPLIT) {
                    const gbf16x8 bl = *reinterpret_cast<const gbf16x8*>(&sw[IMG - 1][o]);
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mb], bh, acc[mb][nb], 0, 0, 0);
                        acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bl, acc[mb][nb], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    // C/D layout of a 32x32 block: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    TOUT* Y = reinterpret_cast<TOUT*>(a.Y);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = n0 + (BN / 2) * wn + 32 * nb + r;
        if (n >= a.N) continue;
        const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long m = m0 + 64 * wm + 32 * mb + (k & 3) + 8 * (k >> 2) + 4 * hh;
                if (m < a.M) Y[(size_t)m * a.ldy + n] = OutCvt<TOUT>::of(acc[mb][nb][k] + bv);
            }
    }
}

// The same 128 x 128 tile on EIGHT waves (2 x 4, wave tile 64 x 32): half the conversions, LDS stores and MFMAs per wave and twice the waves per CU for the
// same LDS -- the A/Bs of section 5 say this kernel lives on occupancy.  A thread brings in 8 elements of an A row and 8 of a W row per tile.
template <typename T> struct Chunk8;
template <> struct Chunk8<float> { gf32x4 v[2]; };
template <> struct Chunk8<__bf16> { gbf16x8 v; };
__device__ __forceinline__ void chunk8_load(Chunk8<float>& c, const float* row, int k0, int klen, bool valid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) c.v[i] = (valid && k0 + 4 * i < klen) ? *reinterpret_cast<const gf32x4*>(row + k0 + 4 * i) : gf32x4{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void chunk8_load(Chunk8<__bf16>& c, const __bf16* row, int k0, int klen, bool valid) {
    gbf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.f;
    c.v = (valid && k0 < klen) ? *reinterpret_cast<const gbf16x8*>(row + k0) : z;
}
__device__ __forceinline__ void chunk8_store(const Chunk8<float>& c, __bf16* hi, __bf16* lo, bool relu) {
    gbf16x8 vh, vl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x = c.v[j >> 2][j & 3];
        if (relu) x = fmaxf(x, 0.f);
        const __bf16 xh = (__bf16)x;
        vh[j] = xh;
        vl[j] = (__bf16)(x - (float)xh);
    }
    *reinterpret_cast<gbf16x8*>(hi) = vh;
    *reinterpret_cast<gbf16x8*>(lo) = vl;
}
__device__ __forceinline__ void chunk8_store(const Chunk8<__bf16>& c, __bf16* hi, __bf16*, bool relu) {
    gbf16x8 v = c.v;
    if (relu) {
        const gs16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        v = __builtin_bit_cast(gbf16x8, __builtin_elementwise_max(__builtin_bit_cast(gs16x8, v), zero));
    }
    *reinterpret_cast<gbf16x8*>(hi) = v;
}

template <typename TIN, typename TOUT, bool AP = false>
__global__ __launch_bounds__(512) void gemm_nt8_kernel(GemmNtArgs a) {
    static_assert(!AP || sizeof(TIN) == 4, "split planes stand for an f32 operand");
    constexpr bool SPLIT = sizeof(TIN) == 4;
    constexpr int IMG = SPLIT ? 2 : 1, BN = 128;
    __shared__ __align__(16) __bf16 sa[IMG][GBM * GP];
    __shared__ __align__(16) __bf16 sw[IMG][BN * GP];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 1, wn = w >> 1;      // wave tile: rows 64 wm.., columns 32 wn..
    const int lt = xcd_remap(blockIdx.x, gridDim.x);
    const long m0 = (long)(lt / a.tiles_n) * GBM;
    const int n0 = (lt % a.tiles_n) * BN;
    const TIN* A = reinterpret_cast<const TIN*>(a.A);
    const TIN* W = reinterpret_cast<const TIN*>(a.W);
    const int lrow = tid >> 2, lk = 8 * (tid & 3);
    const TIN* arow = A + (size_t)min(m0 + lrow, a.M - 1) * a.lda;
    const __bf16* aph = reinterpret_cast<const __bf16*>(a.A) + (size_t)min(m0 + lrow, a.M - 1) * a.lda;      // (AP)
    const __bf16* apl = aph + a.a_plane;
    const bool wvalid = n0 + lrow < a.N;
    const TIN* wrow = W + (size_t)min(n0 + lrow, a.N - 1) * a.ldw;

    gf32x16 acc[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[mb][k] = 0.f;
    Chunk8<TIN> ca, cw;
    Chunk8<__bf16> cah, cal;
    const int nkt = (a.K + GBK - 1) / GBK;
    if constexpr (AP) { chunk8_load(cah, aph, lk, a.K, true); chunk8_load(cal, apl, lk, a.K, true); }
    else chunk8_load(ca, arow, lk, a.K, true);
    chunk8_load(cw, wrow, lk, a.K, wvalid);
    for (int kt = 0; kt < nkt; ++kt) {
        if constexpr (AP) {
            chunk8_store(cah, &sa[0][lrow * GP + lk], nullptr, false);
            chunk8_store(cal, &sa[1][lrow * GP + lk], nullptr, false);
        } else {
            chunk8_store(ca, &sa[0][lrow * GP + lk], &sa[IMG - 1][lrow * GP + lk], a.relu_a != 0);
        }
        chunk8_store(cw, &sw[0][lrow * GP + lk], &sw[IMG - 1][lrow * GP + lk], false);
        __syncthreads();
        if (kt + 1 < nkt) {
            if constexpr (AP) { chunk8_load(cah, aph, (kt + 1) * GBK + lk, a.K, true); chunk8_load(cal, apl, (kt + 1) * GBK + lk, a.K, true); }
            else chunk8_load(ca, arow, (kt + 1) * GBK + lk, a.K, true);
            chunk8_load(cw, wrow, (kt + 1) * GBK + lk, a.K, wvalid);
        }
#pragma unroll
        for (int ks = 0; ks < GBK / 16; ++ks) {
            const int ob = (32 * wn + r) * GP + ks * 16 + 8 * hh;
            const gbf16x8 bh = *reinterpret_cast<const gbf16x8*>(&sw[0][ob]);
            gbf16x8 bl;
            if (SPLIT) bl = *reinterpret_cast<const gbf16x8*>(&sw[IMG - 1][ob]);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int o = (64 * wm + 32 * mb + r) * GP + ks * 16 + 8 * hh;
                const gbf16x8 ah = *reinterpret_cast<const gbf16x8*>(&sa[0][o]);
                acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mb], 0, 0, 0);
                if (SPLIT) {
                    const gbf16x8 al = *reinterpret_cast<const gbf16x8*>(&sa[IMG - 1][o]);
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mb], 0, 0, 0);
                    acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mb], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    TOUT* Y = reinterpret_cast<TOUT*>(a.Y);
    const int n = n0 + 32 * wn + r;
    if (n < a.N) {
        const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long m = m0 + 64 * wm + 32 * mb + (k & 3) + 8 * (k >> 2) + 4 * hh;
                if (m < a.M) Y[(size_t)m * a.ldy + n] = OutCvt<TOUT>::of(acc[mb][k] + bv);
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------------ gemm_tn
// Tile images hold 32 rows (reduction index m) x 128 / 64 columns, row pitch 256 + 64 / 128 + 64 B: the 4 rows x 64 B a half-wave takes per
// transposed read (ds_read_b64_tr_b16) fall on disjoint 16-bank groups (pitch = 64 B mod 256 B).
constexpr int TROWS = 32;
struct GemmTnArgs {
    const void* A; long lda;       // (M, N) rows
    const void* X; long ldx;       // (M, K) rows
    const void* X2; long ldx2;     // optional second right-hand operand (M, K2) rows: its product leaves in columns [K, K + K2) of the partials --
    int K2;                        //   dW_ih = dG^T.x and dW_hh = dG^T.h_prev from ONE pass over the gate gradients
    float* part;                   // (S, N, K + K2) partial sums
    long M; int N, K;
    long rows_per_chunk;           // multiple of TROWS
    int tiles_n, tiles_k, tiles_k1;      // tiles_k = tiles_k1 (of X) + those of X2
    int relu_x;                          // the product runs on max(X, 0) (X only, not X2): the decoder LSTM's input is relu(encoder output), rectified on load
    long a_plane = 0;                    // AP kernels: A is a pair of bf16 planes (hi at A, lo a_plane elements behind it; lda in bf16 elements)
};

__device__ __forceinline__ gs16x4 glds_tr16(const __bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) gs16x4*)(p));
}

template <typename TIN, int BKO, bool AP = false>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTnArgs a) {
    static_assert(!AP || sizeof(TIN) == 4, "split planes stand for an f32 operand");
    constexpr bool SPLIT = sizeof(TIN) == 4;
    constexpr int IMG = SPLIT ? 2 : 1;
    constexpr int PA = 128 + 32, PX = BKO + 32;              // row pitches in bf16 elements (320 B / 320 or 192 B)
    constexpr int NBK = BKO / 64;                            // 32-column blocks of X per wave
    __shared__ __align__(16) __bf16 sa[IMG][TROWS * PA];
    __shared__ __align__(16) __bf16 sx[IMG][TROWS * PX];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wn = w & 1, wk = w >> 1;
    const int lt = xcd_remap(blockIdx.x, gridDim.x);        // consecutive logical ids = the output tiles of one row chunk: one XCD
    const int ntile = a.tiles_n * a.tiles_k;
    const int s = lt / ntile, ot = lt % ntile;
    const int kt = ot % a.tiles_k;
    const bool second = kt >= a.tiles_k1;                   // this tile multiplies the second right-hand operand
    const int n0 = (ot / a.tiles_k) * 128, k0 = (second ? kt - a.tiles_k1 : kt) * BKO;
    const long mbeg = (long)s * a.rows_per_chunk, mend = min(a.M, mbeg + a.rows_per_chunk);
    const TIN* A = reinterpret_cast<const TIN*>(a.A);
    const TIN* X = reinterpret_cast<const TIN*>(second ? a.X2 : a.X);
    const long ldx = second ? a.ldx2 : a.ldx;
    const int KX = second ? a.K2 : a.K, KT = a.K + a.K2;
    // loads: row = tid >> 3 of the 32; A: 16 columns [16 seg, +16); X: BKO / 8 columns
    const int lrow = tid >> 3, seg = tid & 7;
    constexpr int XC = BKO / 8;                              // 16 or 8 columns of X per thread
    const int nvalid_a = a.N - n0, kvalid = KX - k0;         // valid columns of this tile (vector pieces all-in or all-out)

    gf32x16 acc[2][NBK];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int kb = 0; kb < NBK; ++kb)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[nb][kb][k] = 0.f;

    Chunk16<TIN> ca, cx;
    Chunk16<__bf16> cah, cal;
    const __bf16* APH = reinterpret_cast<const __bf16*>(a.A);
    auto load = [&](long m) {
        const long row = m + lrow;
        if (row < mend) {
            if constexpr (AP) {
                chunk_load(cah, APH + (size_t)row * a.lda + n0, 16 * seg, nvalid_a);
                chunk_load(cal, APH + a.a_plane + (size_t)row * a.lda + n0, 16 * seg, nvalid_a);
            } else {
                chunk_load(ca, A + (size_t)row * a.lda + n0, 16 * seg, nvalid_a);
            }
            if (XC == 16) chunk_load(cx, X + (size_t)row * ldx + k0, 16 * seg, kvalid);
            else chunk_load(cx, X + (size_t)row * ldx + k0, 8 * seg, min(kvalid, 8 * seg + 8));      // 8 columns: the second half stays zero
        } else {
            if constexpr (AP) { chunk_zero(cah); chunk_zero(cal); }
            else chunk_zero(ca);
            chunk_zero(cx);
        }
    };
    // transposed-read addressing: within each 16-lane group lane 4q+p supplies row q, columns 4p..4p+3 of a 4-row x 16-column block and
    // lane i receives column i of those rows; the 32x32x16 operand of lane l is column (l & 31), rows 8 (l >> 5) + 0..7 of the k-step
    const int kq = (lane & 15) >> 2, kp = lane & 3, cb = (lane >> 4) & 1;
    const int rowoff = 8 * hh + kq;
    auto frag = [&](const __bf16* img, int pitch, int col0, int ks) {
        const __bf16* p = img + (ks * 16 + rowoff) * pitch + col0 + 16 * cb + 4 * kp;
        const gs16x4 lo = glds_tr16(p), hi = glds_tr16(p + 4 * pitch);
        gs16x8 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = lo[j]; f[4 + j] = hi[j]; }
        return __builtin_bit_cast(gbf16x8, f);
    };
    if (mbeg < mend) load(mbeg);
    for (long m = mbeg; m < mend; m += TROWS) {
        if constexpr (AP) {
            chunk_store(cah, &sa[0][lrow * PA + 16 * seg], nullptr, false);
            chunk_store(cal, &sa[1][lrow * PA + 16 * seg], nullptr, false);
        } else {
            chunk_store(ca, &sa[0][lrow * PA + 16 * seg], &sa[IMG - 1][lrow * PA + 16 * seg], false);
        }
        const bool rx = a.relu_x != 0 && !second;
        if (XC == 16) {
            chunk_store(cx, &sx[0][lrow * PX + 16 * seg], &sx[IMG - 1][lrow * PX + 16 * seg], rx);
        } else {                                             // 8 columns per thread: the first half of the chunk
            chunk_store8(cx, &sx[0][lrow * PX + 8 * seg], &sx[IMG - 1][lrow * PX + 8 * seg], rx);
        }
        __syncthreads();
        if (m + TROWS < mend) load(m + TROWS);
#pragma unroll
        for (int ks = 0; ks < TROWS / 16; ++ks) {
            gbf16x8 ah[2], al[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                ah[nb] = frag(sa[0], PA, 64 * wn + 32 * nb, ks);
                if (SPLIT) al[nb] = frag(sa[IMG - 1], PA, 64 * wn + 32 * nb, ks);
            }
#pragma unroll
            for (int kb = 0; kb < NBK; ++kb) {
                const gbf16x8 bh = frag(sx[0], PX, (BKO / 2) * wk + 32 * kb, ks);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[nb], bh, acc[nb][kb], 0, 0, 0);
                if (SPLIT) {
                    const gbf16x8 bl = frag(sx[IMG - 1], PX, (BKO / 2) * wk + 32 * kb, ks);
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[nb], bh, acc[nb][kb], 0, 0, 0);
                        acc[nb][kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[nb], bl, acc[nb][kb], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
    }
    // D block: rows = output row n (the A operand's columns), columns = output column k: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 hh
    float* o = a.part + (size_t)s * a.N * KT + (second ? a.K : 0);
#pragma unroll
    for (int kb = 0; kb < NBK; ++kb) {
        const int k = k0 + (BKO / 2) * wk + 32 * kb + (lane & 31);
        if (k >= KX) continue;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + 64 * wn + 32 * nb + (e & 3) + 8 * (e >> 2) + 4 * hh;
                if (n < a.N) o[(size_t)n * KT + k] = acc[nb][kb][e];
            }
    }
}

// D[n][k] (+)= sum over the S chunks, fixed order, f64; k < kcols only (the padding columns of a packed operand are dropped); columns
// [K1, K) of the partials belong to the second product and go to D2
__global__ __launch_bounds__(256) void gemm_tn_finalize(const float* part, int S, int N, int K, int K1, float* D, long ldd, int kcols, float* D2, long ldd2,
                                                        float beta) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * K) return;
    const int n = i / K, k = i - n * K;
    if (k < K1 && k >= kcols) return;
    double ch[4] = {0.0, 0.0, 0.0, 0.0};
    int s = 0;
    for (; s + 4 <= S; s += 4) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = part[(size_t)(s + j) * N * K + i];
#pragma unroll
        for (int j = 0; j < 4; ++j) ch[j] += (double)v[j];
    }
    for (; s < S; ++s) ch[0] += (double)part[(size_t)s * N * K + i];
    const float r = (float)((ch[0] + ch[1]) + (ch[2] + ch[3]));
    float* dst = k < K1 ? D + (size_t)n * ldd + k : D2 + (size_t)n * ldd2 + (k - K1);
    *dst = beta != 0.f ? fmaf(beta, *dst, r) : r;
}

// ------------------------------------------------------------------------------------------------------------------------ x3_row_proj
// The 256-input projections of the x3 f32 step over all (time step, encounter) rows -- the decoder LSTM's gx = act(x).W_ih^T + b (Nout = 1024) and
// CompressFC's Linear(256, 128) -- in dic_rowproj.hip's form: the weights never move.  Wave w keeps W[n][0..255] of its 32 output columns as hi / lo
// bf16 fragments (split ONCE at start-up: 128 registers), the workgroup streams 32-row tiles of f32 x (global -> registers one tile ahead -> split ->
// two bf16 images in LDS, 560-B pitch: conflict-free straight 16-B reads), 48 MFMAs per tile and wave (hi.hi + lo.hi + hi.lo), f32 outputs straight from
// the accumulators as 128-B row segments.  Against gemm_nt on the same shapes: no weight traffic and no weight conversion inside the loop, a tile's x
// split once per 256 output columns instead of once per 128, one barrier per tile.  grid (row chunks, Nout / (32 NW)): the stripes of a chunk share an XCD.
// (Round 6, measured and not kept: the product issued TRANSPOSED -- A = the resident W fragments, B = the x rows -- so that a lane holds four consecutive columns
// of its row and a tile leaves as four 16-B stores per wave instead of sixteen 4-B ones: 1 592-1 628 against 1 281 us for the decoder's 1024 columns (a store
// instruction then writes 32-B pieces of 32 different rows -- four partial writes per 128-B line -- where the straight product writes whole 128-B row segments),
// 252 against 296 us for CompressFC's 128.  An asm 16-B store also needs wait states before the next tile's accumulator initialisation: the hardware reads a wide
// store's data registers late, and only for its own stores does the compiler keep writes off them -- without them columns 0, 1 of rows 12-15 / 28-31 came out stale.)
// (Also measured, same round: a full tile's outputs through an LDS staging tile, leaving one iteration later as whole 1-KiB rows of 16-B stores -- 1 282 against
// 1 291 us; the next tile's x split piece by piece behind the second half of the MFMA k-steps instead of in a phase of its own -- 1 276 against 1 285 us.  Neither the
// store form nor the conversion phase is what the 3.3 us per tile wait for.)
constexpr int XK = 256, XT = 32;
constexpr int XPITCH = XK * 2 + 48;                  // bytes per LDS row of a bf16 image
constexpr int XIMG = XT * XPITCH;                    // 17 920 B
struct X3ProjArgs {
    const float* x; const float* w; const float* bias; float* out;
    long N; int Nout; int relu_in;
};

template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void x3_row_proj_kernel(X3ProjArgs a) {
    constexpr int NT = NW * 64, NCOLS = 32 * NW, PPT = XT * 32 / NT;       // PPT: 8-float pieces of x per thread and tile
    extern __shared__ __align__(16) unsigned char xsm[];                    // [2 slots][hi image | lo image]
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long N = a.N;
    const int ntiles = (int)((N + XT - 1) / XT), nch = gridDim.x;
    const int n0 = blockIdx.y * NCOLS, ncol = n0 + 32 * w + (lane & 31);

    gbf16x8 whi[XK / 16], wlo[XK / 16];              // B operand: W[ncol][16 ks + 8 hh .. + 7] as hi + lo
    {
        const float* wr = a.w + (size_t)min(ncol, a.Nout - 1) * XK;
#pragma unroll
        for (int ks = 0; ks < XK / 16; ++ks) {
            const gf32x4 p0 = *reinterpret_cast<const gf32x4*>(wr + 16 * ks + 8 * hh), p1 = *reinterpret_cast<const gf32x4*>(wr + 16 * ks + 8 * hh + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = j < 4 ? p0[j & 3] : p1[j & 3];
                const __bf16 h = (__bf16)v;
                whi[ks][j] = h;
                wlo[ks][j] = (__bf16)(v - (float)h);
            }
        }
    }
    const float bn = (a.bias && ncol < a.Nout) ? a.bias[ncol] : 0.f;

    // x pieces: piece p = tid + NT j of a tile: row p / 32, floats 8 (p % 32) .. + 7
    gf32x4 px[PPT][2];
    // hand-issued loads (asm: the compiler neither counts nor waits for them): the wait in front of `land` is OURS and counted -- the tile's output stores
    // issued behind these loads stay in flight across it (the compiler's own wait for a plain load cannot know how many of the predicated stores went out and
    // drains the queue: every tile then waits for the acknowledgement of the previous tile's stores)
    auto request = [&](int tile) {
        const long r0 = (long)tile * XT;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int p = tid + NT * j;
            const float* src = a.x + (size_t)min(r0 + (p >> 5), N - 1) * XK + (p & 31) * 8;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(px[j][0]) : "v"(src) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=&v"(px[j][1]) : "v"(src) : "memory");
        }
    };
    auto arrived = [&](bool counted) {      // the 2 PPT loads of the last `request` have landed; `counted`: exactly 16 stores were issued behind them
        if (counted) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < PPT; ++j) { asm volatile("" : "+v"(px[j][0])); asm volatile("" : "+v"(px[j][1])); }
    };
    auto land = [&](int slot) {
        unsigned char* base = xsm + slot * 2 * XIMG;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int p = tid + NT * j;
            gbf16x8 vh, vl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = px[j][e >> 2][e & 3];
                if (a.relu_in) v = fmaxf(v, 0.f);
                const __bf16 h = (__bf16)v;
                vh[e] = h;
                vl[e] = (__bf16)(v - (float)h);
            }
            unsigned char* dst = base + (p >> 5) * XPITCH + (p & 31) * 16;
            *reinterpret_cast<gbf16x8*>(dst) = vh;
            *reinterpret_cast<gbf16x8*>(dst + XIMG) = vl;
        }
    };
    const int a_off = (lane & 31) * XPITCH + hh * 16;          // A operand: row (lane & 31), 16-B piece 2 ks + hh

    int tile = blockIdx.x;
    if (tile < ntiles) { request(tile); arrived(false); land(0); }
    if (tile + nch < ntiles) request(tile + nch);
    __syncthreads();
    int slot = 0;
    bool stored16 = false;                  // the previous iteration issued all of its 16 stores (a full tile, all columns inside) behind the loads in flight
    for (; tile < ntiles; tile += nch) {
        const unsigned char* base = xsm + slot * 2 * XIMG;
        gf32x16 acc;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = bn;
#pragma unroll
        for (int ks = 0; ks < XK / 16; ++ks) {
            const gbf16x8 ah = *reinterpret_cast<const gbf16x8*>(base + a_off + ks * 32);
            const gbf16x8 al = *reinterpret_cast<const gbf16x8*>(base + XIMG + a_off + ks * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, whi[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, whi[ks], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, wlo[ks], acc, 0, 0, 0);
        }
        if (tile + nch < ntiles) { arrived(stored16); land(slot ^ 1); }      // (the other slot was read in the previous iteration: everybody is past its barrier)
        if (tile + 2 * nch < ntiles) request(tile + 2 * nch);
        // C/D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        const long r0 = (long)tile * XT;
        const bool full = r0 + XT <= N && n0 + NCOLS <= a.Nout;          // (workgroup-uniform)
        if (full) {                                                       // 16 unconditional stores: what the counted wait of the next iteration counts
            float* o = a.out + (size_t)(r0 + 4 * hh) * a.Nout + ncol;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                float* dst = o + (size_t)((k & 3) + 8 * (k >> 2)) * a.Nout;
                asm volatile("global_store_dword %0, %1, off" :: "v"(dst), "v"(acc[k]) : "memory");
            }
        } else if (ncol < a.Nout) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long m = r0 + (k & 3) + 8 * (k >> 2) + 4 * hh;
                if (m < N) a.out[(size_t)m * a.Nout + ncol] = acc[k];
            }
        }
        stored16 = full;
        // LDS-only barrier (round 6): __syncthreads() carries s_waitcnt vmcnt(0), i.e. every tile waited for the acknowledgement of its own 16 output stores
        // per wave (and for the next tile's loads) before the next tile's MFMAs could start -- SQ_WAIT_INST_ANY 0.50, matrix cores 41 % busy.  What the
        // barrier orders is the LDS slot hand-over only; the loads are waited for where `land` consumes them (a counted wait: the stores behind them fly on)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        slot ^= 1;
    }
}

static int tn_chunks(long M, int N, int K, int K2, long* rows_per_chunk, int* bko) {
    *bko = (K > 64 || K2 > 64) ? 128 : 64;
    const int tiles = ((N + 127) / 128) * ((K + *bko - 1) / *bko + (K2 + *bko - 1) / *bko);
    const long row_tiles = (M + TROWS - 1) / TROWS;
    long S = max(1L, min(row_tiles, (long)(4 * kNumCU + tiles - 1) / tiles));        // ~4 workgroups per CU
    S = min(S, 256L);
    long rpc = (row_tiles + S - 1) / S * TROWS;
    *rows_per_chunk = rpc;
    return (int)((M + rpc - 1) / rpc);
}

}  // namespace dic

using namespace dic;

extern "C" {

static int gemm_nt_impl(int in_dtype, int out_dtype, const void* A, long a_plane, long lda, const void* W, long ldw, const float* bias, long M, int N, int K,
                        void* Y, long ldy, int relu_a, dic_stream_t stream) {
    DIC_REQUIRE(in_dtype == DIC_DTYPE_F32 || in_dtype == DIC_DTYPE_BF16, DIC_ERR_INVALID_ARG, "gemm_nt: input dtype %d", in_dtype);
    const bool ap = a_plane != 0;
    DIC_REQUIRE(!ap || (in_dtype == DIC_DTYPE_F32 && out_dtype == DIC_DTYPE_F32 && !relu_a && K % 8 == 0 && lda % 8 == 0), DIC_ERR_UNSUPPORTED,
                "gemm_nt: a split-plane A operand goes with f32 W / Y, no rectification, K and lda multiples of 8 (K=%d lda=%ld)", K, lda);
    DIC_REQUIRE(out_dtype == DIC_DTYPE_F32 || out_dtype == DIC_DTYPE_BF16, DIC_ERR_INVALID_ARG, "gemm_nt: output dtype %d", out_dtype);
    DIC_REQUIRE(A && W && Y, DIC_ERR_INVALID_ARG, "gemm_nt: NULL pointer");
    DIC_REQUIRE(M > 0 && N > 0 && K > 0 && lda >= K && ldw >= K && ldy >= N, DIC_ERR_INVALID_ARG, "gemm_nt: M=%ld N=%d K=%d lda=%ld ldw=%ld ldy=%ld", M, N, K, lda, ldw, ldy);
    const int vec = in_dtype == DIC_DTYPE_F32 ? 4 : 8;       // 16-B vector loads
    DIC_REQUIRE(K % vec == 0 && lda % vec == 0 && ldw % vec == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)W & 15) == 0, DIC_ERR_UNSUPPORTED,
                "gemm_nt: K, lda, ldw must be multiples of %d elements and the operands 16-B aligned (K=%d lda=%ld ldw=%ld): pad the rows", vec, K, lda, ldw);
    const int bn = N > 64 ? 128 : 64;
    GemmNtArgs a{A, lda, W, ldw, bias, Y, ldy, M, N, K, relu_a, (N + bn - 1) / bn};
    a.a_plane = a_plane;
    const long tiles = ((M + GBM - 1) / GBM) * a.tiles_n;
    DIC_REQUIRE(tiles < (1L << 31), DIC_ERR_UNSUPPORTED, "gemm_nt: %ld tiles", tiles);
    const dim3 grid((unsigned)tiles), blk(256);
    hipStream_t st = (hipStream_t)stream;
    const bool f32in = in_dtype == DIC_DTYPE_F32, f32out = out_dtype == DIC_DTYPE_F32;
#define DIC_NT(TI, TO, BN) hipLaunchKernelGGL((gemm_nt_kernel<TI, TO, BN>), grid, blk, 0, st, a)
    static const bool eight = [] { const char* e = getenv("DIC_GEMM_NT8"); return !(e && e[0] == '0'); }();      // (A/B switch: 0 = four waves per 128 x 128 tile)
    if (ap) {
        if (bn == 128) hipLaunchKernelGGL((gemm_nt8_kernel<float, float, true>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((gemm_nt_kernel<float, float, 64, true>), grid, blk, 0, st, a);
    } else if (bn == 128 && eight) {
#define DIC_NT8(TI, TO) hipLaunchKernelGGL((gemm_nt8_kernel<TI, TO>), grid, dim3(512), 0, st, a)
        if (f32in) { if (f32out) DIC_NT8(float, float); else DIC_NT8(float, __bf16); }
        else { if (f32out) DIC_NT8(__bf16, float); else DIC_NT8(__bf16, __bf16); }
#undef DIC_NT8
    } else if (bn == 128) {
        if (f32in) { if (f32out) DIC_NT(float, float, 128); else DIC_NT(float, __bf16, 128); }
        else { if (f32out) DIC_NT(__bf16, float, 128); else DIC_NT(__bf16, __bf16, 128); }
    } else {
        if (f32in) { if (f32out) DIC_NT(float, float, 64); else DIC_NT(float, __bf16, 64); }
        else { if (f32out) DIC_NT(__bf16, float, 64); else DIC_NT(__bf16, __bf16, 64); }
    }
#undef DIC_NT
    return check_launch("gemm_nt");
}

int dic_gemm_nt(int in_dtype, int out_dtype, const void* A, long lda, const void* W, long ldw, const float* bias, long M, int N, int K,
                void* Y, long ldy, int relu_a, dic_stream_t stream) {
    return gemm_nt_impl(in_dtype, out_dtype, A, 0, lda, W, ldw, bias, M, N, K, Y, ldy, relu_a, stream);
}

int dic_gemm_nt_planes(const void* A_hi, long a_plane, long lda, const float* W, long ldw, const float* bias, long M, int N, int K, float* Y, long ldy,
                       dic_stream_t stream) {
    DIC_REQUIRE(a_plane > 0, DIC_ERR_INVALID_ARG, "gemm_nt_planes: plane stride %ld", a_plane);
    return gemm_nt_impl(DIC_DTYPE_F32, DIC_DTYPE_F32, A_hi, a_plane, lda, W, ldw, bias, M, N, K, Y, ldy, 0, stream);
}

size_t dic_gemm_tn_workspace(long M, int N, int K, int K2) {
    if (M <= 0 || N <= 0 || K <= 0 || K2 < 0) return 0;
    long rpc; int bko;
    const int S = tn_chunks(M, N, K, K2, &rpc, &bko);
    return (size_t)S * N * (K + K2) * sizeof(float);
}

static int gemm_tn_impl(int in_dtype, const void* A, long a_plane, long lda, const void* X, long ldx, long M, int N, int K, float* D, long ldd, int kcols,
                        const void* X2, long ldx2, int K2, float* D2, long ldd2, int accumulate, int relu_x, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(in_dtype == DIC_DTYPE_F32 || in_dtype == DIC_DTYPE_BF16, DIC_ERR_INVALID_ARG, "gemm_tn: input dtype %d", in_dtype);
    const bool ap = a_plane != 0;
    DIC_REQUIRE(!ap || (in_dtype == DIC_DTYPE_F32 && N % 8 == 0 && lda % 8 == 0), DIC_ERR_UNSUPPORTED,
                "gemm_tn: a split-plane A operand goes with f32 X, N and lda multiples of 8 (N=%d lda=%ld)", N, lda);
    DIC_REQUIRE(A && X && D && workspace, DIC_ERR_INVALID_ARG, "gemm_tn: NULL pointer");
    DIC_REQUIRE(M > 0 && N > 0 && K > 0 && lda >= N && ldx >= K && kcols > 0 && kcols <= K && ldd >= kcols, DIC_ERR_INVALID_ARG,
                "gemm_tn: M=%ld N=%d K=%d lda=%ld ldx=%ld ldd=%ld kcols=%d", M, N, K, lda, ldx, ldd, kcols);
    if (!X2) K2 = 0;
    DIC_REQUIRE(!X2 || (K2 > 0 && D2 && ldx2 >= K2 && ldd2 >= K2), DIC_ERR_INVALID_ARG, "gemm_tn: second operand K2=%d ldx2=%ld ldd2=%ld", K2, ldx2, ldd2);
    const int vec = in_dtype == DIC_DTYPE_F32 ? 4 : 8;
    DIC_REQUIRE(N % vec == 0 && K % vec == 0 && lda % vec == 0 && ldx % vec == 0 && ((uintptr_t)A & 15) == 0 && ((uintptr_t)X & 15) == 0, DIC_ERR_UNSUPPORTED,
                "gemm_tn: N, K, lda, ldx must be multiples of %d elements and the operands 16-B aligned (N=%d K=%d lda=%ld ldx=%ld): pad the rows", vec, N, K, lda, ldx);
    DIC_REQUIRE(!X2 || (K2 % vec == 0 && ldx2 % vec == 0 && ((uintptr_t)X2 & 15) == 0), DIC_ERR_UNSUPPORTED,
                "gemm_tn: K2, ldx2 must be multiples of %d elements and X2 16-B aligned (K2=%d ldx2=%ld)", vec, K2, ldx2);
    long rpc; int bko;
    const int S = tn_chunks(M, N, K, K2, &rpc, &bko);
    const int KT = K + K2;
    DIC_REQUIRE(workspace_bytes >= (size_t)S * N * KT * sizeof(float), DIC_ERR_WORKSPACE, "gemm_tn: workspace %zu < %zu", workspace_bytes,
                (size_t)S * N * KT * sizeof(float));
    const int tk1 = (K + bko - 1) / bko, tk2 = (K2 + bko - 1) / bko;
    GemmTnArgs a{A, lda, X, ldx, X2, ldx2, K2, (float*)workspace, M, N, K, rpc, (N + 127) / 128, tk1 + tk2, tk1, relu_x};
    a.a_plane = a_plane;
    const dim3 grid((unsigned)(S * a.tiles_n * a.tiles_k)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (ap) {
        if (bko == 128) hipLaunchKernelGGL((gemm_tn_kernel<float, 128, true>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((gemm_tn_kernel<float, 64, true>), grid, blk, 0, st, a);
    } else if (in_dtype == DIC_DTYPE_F32) {
        if (bko == 128) hipLaunchKernelGGL((gemm_tn_kernel<float, 128>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((gemm_tn_kernel<float, 64>), grid, blk, 0, st, a);
    } else {
        if (bko == 128) hipLaunchKernelGGL((gemm_tn_kernel<__bf16, 128>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((gemm_tn_kernel<__bf16, 64>), grid, blk, 0, st, a);
    }
    hipLaunchKernelGGL(gemm_tn_finalize, dim3((N * KT + 255) / 256), dim3(256), 0, st, (const float*)workspace, S, N, KT, K, D, ldd, kcols, D2, ldd2,
                       accumulate ? 1.0f : 0.0f);
    return check_launch("gemm_tn");
}

int dic_gemm_tn(int in_dtype, const void* A, long lda, const void* X, long ldx, long M, int N, int K, float* D, long ldd, int kcols,
                const void* X2, long ldx2, int K2, float* D2, long ldd2, int accumulate, int relu_x, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    return gemm_tn_impl(in_dtype, A, 0, lda, X, ldx, M, N, K, D, ldd, kcols, X2, ldx2, K2, D2, ldd2, accumulate, relu_x, workspace, workspace_bytes, stream);
}

int dic_gemm_tn_planes(const void* A_hi, long a_plane, long lda, const float* X, long ldx, long M, int N, int K, float* D, long ldd, int kcols,
                       const float* X2, long ldx2, int K2, float* D2, long ldd2, int accumulate, int relu_x, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(a_plane > 0, DIC_ERR_INVALID_ARG, "gemm_tn_planes: plane stride %ld", a_plane);
    return gemm_tn_impl(DIC_DTYPE_F32, A_hi, a_plane, lda, X, ldx, M, N, K, D, ldd, kcols, X2, ldx2, K2, D2, ldd2, accumulate, relu_x, workspace, workspace_bytes, stream);
}

int dic_x3_row_proj(const float* x, const float* w, const float* bias, int64_t N, int in_features, int out_features, float* out, int relu_input,
                    dic_stream_t stream) {
    DIC_REQUIRE(N > 0, DIC_ERR_INVALID_ARG, "x3_row_proj: non-positive row count");
    DIC_REQUIRE(in_features == XK && out_features > 0 && (out_features % 256 == 0 || out_features == 128), DIC_ERR_UNSUPPORTED,
                "x3_row_proj: (%d -> %d) (compiled for 256 -> 128 or a multiple of 256)", in_features, out_features);
    DIC_REQUIRE(x && w && out, DIC_ERR_INVALID_ARG, "x3_row_proj: NULL pointer");
    DIC_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0, DIC_ERR_UNSUPPORTED, "x3_row_proj: x and w must be 16-B aligned");
    X3ProjArgs a{x, w, bias, out, (long)N, out_features, relu_input != 0};
    const int lds = 2 * 2 * XIMG;
    const int ntiles = (int)((N + XT - 1) / XT);
    hipStream_t st = (hipStream_t)stream;
    if (out_features == 128) {
        static bool attr4 = false;
        if (!attr4) {
            hipError_t e = hipFuncSetAttribute((const void*)x3_row_proj_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "x3_row_proj: cannot reserve %d B of LDS: %s", lds, hipGetErrorString(e));
            attr4 = true;
        }
        const int nch = max(1, min(ntiles, 2 * kNumCU));
        hipLaunchKernelGGL(x3_row_proj_kernel<4>, dim3(nch, 1), dim3(256), lds, st, a);
    } else {
        static bool attr8 = false;
        if (!attr8) {
            hipError_t e = hipFuncSetAttribute((const void*)x3_row_proj_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "x3_row_proj: cannot reserve %d B of LDS: %s", lds, hipGetErrorString(e));
            attr8 = true;
        }
        const int stripes = out_features / 256;
        int nch = max(1, min(ntiles, kNumCU / stripes * 1));
        nch = nch >= 8 ? nch / 8 * 8 : nch;                          // a multiple of 8: the stripes of a row chunk share an XCD
        hipLaunchKernelGGL(x3_row_proj_kernel<8>, dim3(nch, stripes), dim3(512), lds, st, a);
    }
    return check_launch("x3_row_proj");
}

}  // extern "C"
