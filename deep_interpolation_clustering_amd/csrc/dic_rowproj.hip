// Row-wise projections with a 256-wide input applied to every (time step, encounter) row:
//     out[row][n] = bias[n] + sum_k x[row][k] W[n][k]          x (N,256) bf16, W (Nout,256) bf16, out (N,Nout) bf16
// i.e. (a) the decoder LSTM's input projection gx = relu(enc_out) . W_ih^T + (b_ih + b_hh) (clustering_interp.py:47-59 through nn.LSTM:
// N = 786 432 rows, Nout = 1024, 412 GFLOP, 1.6 GB of output at B = 32 768).  The library GEMM for this shape has K = 256 only --
// four k-iterations per 256x256 macro-tile, so its prologue / epilogue never overlap -- and ran at 0.58-0.60 ms (3.4 TB/s of its
// own traffic).  Here the weights never move: wave w of a workgroup keeps W[n][0..255] for its 32 output columns in 64 registers
// (the B operand of all 16 MFMA k-steps), the workgroup streams 32-row tiles of x through LDS (global -> registers -> LDS, one
// tile ahead, two images, one barrier per tile), and the 32 x 256 output block leaves through an LDS staging tile as whole
// 512-B row segments.  grid (chunks, Nout / 256): the workgroups of a row chunk sit on one XCD (chunk count a multiple of 8), so
// x is fetched from HBM about once and served to the other column stripes by that XCD's L2.
// (b) CompressFC's first layer Linear(256, 128) over the same rows (rbf.py:111-125): 4 waves per workgroup, one 128-column stripe,
// and -- the layer feeds a training-mode BatchNorm1d -- the per-column sums of z and z^2 (of the bf16 values actually stored) ride
// along in registers: the separate column-statistics pass over the 201 MB z tensor disappears.
#include "dic_common.h"

namespace dic {

constexpr int PK = 256;                              // input width (K)
constexpr int PT = 32;                               // rows per tile
constexpr int PX_PITCH = PK * 2 + 48;                // 560 B: the 32 rows of a straight 16-B read fall on disjoint bank groups
constexpr int P_TILE = PT * PX_PITCH;
__host__ __device__ constexpr int p_stage_pitch(int nw) { return nw * 64 + 16; }          // staging rows of the (32 nw)-column output block
__host__ __device__ constexpr int p_lds(int nw) { return 2 * P_TILE + 2 * PT * p_stage_pitch(nw); }    // 69 632 B (8 waves) / 53 248 B (4 waves)

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));

struct RowProjArgs {
    const __bf16* x;       // (N, 256)
    const __bf16* w;       // (Nout, 256)
    const __bf16* bias;    // (Nout) or NULL
    __bf16* out;           // (N, Nout)
    long N;
    int Nout;
    float* stat_part;      // STATS: (gridDim.x, 2, 32 NW) per-workgroup column sums of out and out^2
    int nbt;               // NATIVE: 32-row tiles per time step (batch / 32)
    int relu_in;           // the product is taken of relu(x): x is the raw output of the encoder LSTM (clustering_interp.py:38-41 applies
                           // F.relu between the two), rectified on its way into LDS -- no rectified copy of the encoder output in HBM
};

// relu of packed bf16 pairs as ONE integer instruction per register: a negative bf16 is a negative int16, so max(x, 0) on the 16-bit
// halves (v_pk_max_i16) zeroes exactly the negative halves (and -0.0); `floor` = 0 applies it, 0x8000 per half (INT16_MIN) is the identity
typedef short ps16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_relu(unsigned x, unsigned floor) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(ps16x2, x), __builtin_bit_cast(ps16x2, floor)));
}
__device__ __forceinline__ uint4 pk_relu4(uint4 v, unsigned floor) {
    return make_uint4(pk_relu(v.x, floor), pk_relu(v.y, floor), pk_relu(v.z, floor), pk_relu(v.w, floor));
}

// NATIVE (the decoder's gx, Nout = 1024 = 2 directions x 4 gates x 128 units, read back only by the recurrence kernel): the product is
// issued TRANSPOSED -- D[output column][row] = W . x^T, the same fragments with the operands swapped -- so lane (row r, hh) ends up
// with the 16 pre-activations the recurrence kernel's lane (r, hh) of wave (column / 32) % 4 wants in its accumulators, and writes
// them as two 16-B pieces of the lane-native gx form (gxn_off in dic_lstm.hip: ((((((t nbt + bt) 2 + dir) 4 + w) 4 + g) 2 + qp) 2 + hh)
// 32 + r) 8): 1 KiB per wave instruction, no staging tile, one barrier per tile.

template <int NW, bool STATS, bool NATIVE>
__global__ __launch_bounds__(NW * 64, 2) void row_proj_kernel(RowProjArgs a) {
    static_assert(!NATIVE || (NW == 8 && !STATS), "lane-native output: 256-column stripes of the 1024 gate columns");
    constexpr int NT = NW * 64, PS_PITCH = p_stage_pitch(NW), P_STAGE = PT * PS_PITCH, NCOLS = 32 * NW, PPT = PT * 32 / NT;   // PPT: x pieces per thread and tile
    extern __shared__ __align__(16) unsigned char psm[];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long N = a.N;
    const int ntiles = (int)((N + PT - 1) / PT), nch = gridDim.x;
    const int n0 = blockIdx.y * NCOLS, ncol = n0 + 32 * w + (lane & 31);
    unsigned char* stage = psm + 2 * P_TILE;

    pbf16x8 wreg[PK / 16];                 // B operand: W[ncol][16 ks + 8 hh .. + 7]
#pragma unroll
    for (int ks = 0; ks < PK / 16; ++ks) wreg[ks] = *reinterpret_cast<const pbf16x8*>(a.w + (size_t)ncol * PK + 16 * ks + 8 * hh);
    const float bn = a.bias ? (float)a.bias[ncol] : 0.f;
    float bk[16];                          // NATIVE: the bias of the 16 output columns this lane accumulates
    if constexpr (NATIVE) {
#pragma unroll
        for (int k = 0; k < 16; ++k) bk[k] = a.bias ? (float)a.bias[n0 + 32 * w + (k & 3) + 8 * (k >> 2) + 4 * hh] : 0.f;
    }

    const int xrow = tid >> 5, xpc = tid & 31;               // rows xrow + (NT / 32) j; 32 pieces of 16 B per row
    static_assert(PPT == 2 || PPT == 4, "x pieces per thread and tile");
    uint4 px0, px1, px2 = {}, px3 = {};      // (named registers: an indexed array of these ends up in scratch memory)
    auto piece = [&](long r0, int j) {       // (clamped, always valid addresses; rows past the end are never stored nor counted)
        return *reinterpret_cast<const uint4*>(a.x + (size_t)min(r0 + xrow + (NT / 32) * j, N - 1) * PK + xpc * 8);
    };
#define DIC_RP_REQUEST(TILE)                                                         \
    do {                                                                             \
        const long r0_ = (long)(TILE) * PT;                                          \
        px0 = piece(r0_, 0); px1 = piece(r0_, 1);                                    \
        if constexpr (PPT == 4) { px2 = piece(r0_, 2); px3 = piece(r0_, 3); }        \
    } while (0)
#define DIC_RP_LAND(SLOT)                                                                                              \
    do {                                                                                                               \
        unsigned char* base_ = psm + (SLOT) * P_TILE + xrow * PX_PITCH + xpc * 16;                                     \
        *reinterpret_cast<uint4*>(base_) = pk_relu4(px0, xfloor);                                                      \
        *reinterpret_cast<uint4*>(base_ + (NT / 32) * PX_PITCH) = pk_relu4(px1, xfloor);                               \
        if constexpr (PPT == 4) {                                                                                      \
            *reinterpret_cast<uint4*>(base_ + 2 * (NT / 32) * PX_PITCH) = pk_relu4(px2, xfloor);                       \
            *reinterpret_cast<uint4*>(base_ + 3 * (NT / 32) * PX_PITCH) = pk_relu4(px3, xfloor);                       \
        }                                                                                                              \
    } while (0)
    const unsigned xfloor = a.relu_in ? 0u : 0x80008000u;
    const int a_off = (lane & 31) * PX_PITCH + hh * 16;       // A operand: row (lane & 31), 16-B piece 2 ks + hh

    int tile = blockIdx.x;
    if (tile < ntiles) { DIC_RP_REQUEST(tile); DIC_RP_LAND(0); }
    if (tile + nch < ntiles) DIC_RP_REQUEST(tile + nch);
    __syncthreads();
    int slot = 0;
    double s1 = 0.0, s2 = 0.0;             // STATS: this lane's column, its 16 rows of every tile
    for (; tile < ntiles; tile += nch) {
        const unsigned char* base = psm + slot * P_TILE;
        pf32x16 acc;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = NATIVE ? bk[k] : bn;
#pragma unroll
        for (int ks = 0; ks < PK / 16; ++ks) {
            const pbf16x8 af = *reinterpret_cast<const pbf16x8*>(base + a_off + ks * 32);
            if constexpr (NATIVE) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[ks], af, acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wreg[ks], acc, 0, 0, 0);
        }
        if (tile + nch < ntiles) DIC_RP_LAND(slot ^ 1);
        if (tile + 2 * nch < ntiles) DIC_RP_REQUEST(tile + 2 * nch);
        if constexpr (NATIVE) {
            // tile = 32 rows of one time step (the batch is a multiple of 32): t = tile / nbt, bt = tile % nbt; stripe y = direction y / 2,
            // gates 2 (y % 2) + w / 4; recurrence wave w % 4.  C/D layout: output column (reg & 3) + 8 (reg >> 2) + 4 hh, row = lane & 31
            const size_t blk = ((size_t)tile * 2 + (blockIdx.y >> 1)) * 4 + (w & 3);
            const size_t o = ((blk * 4 + 2 * (blockIdx.y & 1) + (w >> 2)) * 2 * 2 + hh) * 32 + (lane & 31);
            pbf16x8 lo, hi;
#pragma unroll
            for (int e = 0; e < 8; ++e) { lo[e] = (__bf16)acc[e]; hi[e] = (__bf16)acc[8 + e]; }
#ifndef DIC_RP_PLAIN_STORES   // streaming (nontemporal) stores for gx, which only another kernel reads back: they do not push the x tiles the four
            // stripes of a chunk share out of the XCD's L2 (same-box A/B: 477 -> 454 us)
            typedef int pi32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(__builtin_bit_cast(pi32x4, lo), reinterpret_cast<pi32x4*>(a.out + o * 8));
            __builtin_nontemporal_store(__builtin_bit_cast(pi32x4, hi), reinterpret_cast<pi32x4*>(a.out + (o + 2 * 32) * 8));
#else
            *reinterpret_cast<pbf16x8*>(a.out + o * 8) = lo;
            *reinterpret_cast<pbf16x8*>(a.out + (o + 2 * 32) * 8) = hi;
#endif
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // (LDS-only, round 6: __syncthreads() also drains the vector-memory queue -- the loads just requested, the stores just issued)
            asm volatile("" ::: "memory");
            slot ^= 1;
            continue;
        }
        // C/D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        unsigned char* sb = stage + slot * P_STAGE;
        const long r0 = (long)tile * PT;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = (k & 3) + 8 * (k >> 2) + 4 * hh;
            const __bf16 ob = (__bf16)acc[k];
            *reinterpret_cast<__bf16*>(sb + m * PS_PITCH + (32 * w + (lane & 31)) * 2) = ob;
            if (STATS && r0 + m < N) {
                const float of = (float)ob;
                t1 += of;
                t2 = fmaf(of, of, t2);
            }
        }
        if (STATS) { s1 += (double)t1; s2 += (double)t2; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // (LDS-only, round 6: see above)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < PT * (NCOLS / 8) / NT; ++k) {
            const int p = tid + NT * k, row = p / (NCOLS / 8), pc = p % (NCOLS / 8);
            if (r0 + row < N)
                *reinterpret_cast<uint4*>(a.out + (size_t)(r0 + row) * a.Nout + n0 + pc * 8) = *reinterpret_cast<const uint4*>(sb + row * PS_PITCH + pc * 16);
        }
        slot ^= 1;
    }
#undef DIC_RP_REQUEST
#undef DIC_RP_LAND
    if (STATS) {        // the two row halves of a column meet, then one partial row per workgroup (columns are wave-private)
        s1 += __shfl_xor(s1, 32);
        s2 += __shfl_xor(s2, 32);
        if (hh == 0) {
            a.stat_part[(size_t)blockIdx.x * 2 * NCOLS + 32 * w + (lane & 31)] = (float)s1;
            a.stat_part[(size_t)blockIdx.x * 2 * NCOLS + NCOLS + 32 * w + (lane & 31)] = (float)s2;
        }
    }
}

// [sum z | sum z^2 | rows] in f64 (what dic_bn_colstats delivers): fixed-order reduction of the per-workgroup sums
__global__ __launch_bounds__(256) void row_proj_stats_finalize(const float* partials, int nblk, int n, double nrows, double* sums) {
    __shared__ double red[256];
    const double s = reduce_partials_32x8(partials, nblk, n, blockIdx.x * 32, red);
    if (threadIdx.x < 32 && blockIdx.x * 32 + (int)threadIdx.x < n) sums[blockIdx.x * 32 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) sums[n] = nrows;
}

static int row_proj_chunks(long N, int stripes, int wg_per_cu) {
    const int ntiles = (int)((N + PT - 1) / PT);
    int nch = max(1, min(ntiles, wg_per_cu * kNumCU / stripes));
    return nch >= 8 ? nch / 8 * 8 : nch;                          // a multiple of 8: the stripes of a row chunk share an XCD (and its L2)
}

template <int NW, bool STATS, bool NATIVE>
static int row_proj_launch(const RowProjArgs& a, int nch, int stripes, hipStream_t st) {
    constexpr int lds = NATIVE ? 2 * P_TILE : p_lds(NW);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)row_proj_kernel<NW, STATS, NATIVE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "row_proj: cannot reserve %d B of LDS: %s", lds, hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((row_proj_kernel<NW, STATS, NATIVE>), dim3(nch, stripes), dim3(NW * 64), lds, st, a);
    return DIC_OK;
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_row_proj(const void* x, const void* w, const void* bias, int64_t N, int in_features, int out_features, void* out, int lane_native_batch,
                 int relu_input, dic_stream_t stream) {
    DIC_REQUIRE(N > 0, DIC_ERR_INVALID_ARG, "row_proj: non-positive row count");
    DIC_REQUIRE(in_features == PK && out_features > 0 && out_features % 256 == 0, DIC_ERR_UNSUPPORTED,
                "row_proj: (%d -> %d) (compiled for 256 inputs and a multiple of 256 outputs)", in_features, out_features);
    DIC_REQUIRE(x && w && out, DIC_ERR_INVALID_ARG, "row_proj: NULL pointer");
    const int stripes = out_features / 256;
    DIC_REQUIRE(lane_native_batch == 0 || (out_features == 1024 && lane_native_batch > 0 && lane_native_batch % 64 == 0 && N % lane_native_batch == 0),
                DIC_ERR_INVALID_ARG, "row_proj: lane-native output needs 1024 output columns and rows = steps x a batch that is a multiple of 64 (batch %d, %lld rows)",
                lane_native_batch, (long long)N);
    RowProjArgs a{(const __bf16*)x, (const __bf16*)w, (const __bf16*)bias, (__bf16*)out, (long)N, out_features, nullptr, lane_native_batch / 32, relu_input != 0};
    int rc = lane_native_batch ? row_proj_launch<8, false, true>(a, row_proj_chunks(N, stripes, 2), stripes, (hipStream_t)stream)
                               : row_proj_launch<8, false, false>(a, row_proj_chunks(N, stripes, 2), stripes, (hipStream_t)stream);
    return rc ? rc : check_launch("row_proj");
}

size_t dic_row_proj_stats_workspace(int64_t N, int out_features) {
    if (N <= 0 || out_features != 128) return 0;
    return (size_t)row_proj_chunks(N, 1, 3) * 2 * 128 * sizeof(float);
}

int dic_row_proj_stats(const void* x, const void* w, const void* bias, int64_t N, int in_features, int out_features, void* out, double* sums,
                       void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(N > 0, DIC_ERR_INVALID_ARG, "row_proj_stats: non-positive row count");
    DIC_REQUIRE(in_features == PK && out_features == 128, DIC_ERR_UNSUPPORTED, "row_proj_stats: (%d -> %d) (compiled for 256 -> 128)", in_features,
                out_features);
    DIC_REQUIRE(x && w && out && sums && workspace, DIC_ERR_INVALID_ARG, "row_proj_stats: NULL pointer");
    const int nch = row_proj_chunks(N, 1, 3);
    DIC_REQUIRE(workspace_bytes >= (size_t)nch * 2 * 128 * sizeof(float), DIC_ERR_WORKSPACE, "row_proj_stats: workspace too small");
    RowProjArgs a{(const __bf16*)x, (const __bf16*)w, (const __bf16*)bias, (__bf16*)out, (long)N, out_features, (float*)workspace, 0, 0};
    int rc = row_proj_launch<4, true, false>(a, nch, 1, (hipStream_t)stream);
    if (rc) return rc;
    hipLaunchKernelGGL(row_proj_stats_finalize, dim3(2 * 128 / 32), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nch, 2 * 128, (double)N, sums);
    return check_launch("row_proj_stats");
}

}  // extern "C"
