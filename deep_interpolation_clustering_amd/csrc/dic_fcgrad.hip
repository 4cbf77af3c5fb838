// Backward of the first CompressFC layer, Linear(256, 128) applied to every (time step, encounter) row of the decoder output
// (rbf.py:111-125 `nn.Linear(in_dim, 128)`, TimeDistributed utils.py:202-224), from ONE pass over the rows:
//     dX[row][i] = sum_o dZ[row][o] W[o][i]            (N x 256, bf16: the decoder LSTM's output gradient)
//     dW[o][i]   = sum_row dZ[row][o] X[row][i]        (128 x 256, f32)
// The two library GEMMs this replaces each read dZ (N x 128 bf16, 201 MB at B = 32768) and ran at ~4 TB/s of their own traffic;
// here dZ and X are read once, dX is written once: 1.0 GB instead of 1.2 GB, and no split-K chunk products to sum.
//
// 512 threads = 8 waves, one workgroup per CU, 32-row tiles.  Wave w owns the input columns i in [32 w, 32 w + 32):
//   dX: one 32x32 block, K = the 128 outputs in 8 MFMA k-steps; A = dZ rows straight from the LDS tile (16-B row pieces),
//       B = W[o][i] for its 32 columns, resident in 32 registers for the whole kernel;
//   dW: four 32x32 blocks (all 128 outputs x its 32 columns), K = the tile's 32 rows in 2 k-steps; both operands come out of
//       LDS transposed (ds_read_b64_tr_b16): the reduction index (row) is the slow index of dZ and X in memory.
// Tiles go global -> registers -> LDS (one 16-B piece of dZ and two of X per thread and tile, requested one tile ahead), two LDS
// images so that a single barrier per tile suffices; the dX block leaves through an LDS staging tile as whole 512-B rows.
// Row pitches of 256 + 80 / 512 + 80 B (= 20 dwords mod 64) serve both kinds of read without bank conflicts: the straight 16-B reads
// (the 16 rows of a ds_read_b128 lane group land on 16 different 4-bank slots: slot = 5 row mod 16) and the transposed reads, whose
// 32-lane half takes 4 rows x 64 B -- with rows FOUR apart (4 x 20 dwords = 16 mod 64: four disjoint 16-bank groups), which is
// free to choose because the tile row is the reduction index of dW: dZ and X fragments only have to agree on the order.  (Round 2's
// 256 + 48 / 512 + 48 with rows 0..3 of a transposed read side by side overlapped by 4 banks per row: SQ_LDS_BANK_CONFLICT 23 %.)
// The dX block reaches the staging tile as packed bf16 pairs (one DPP swap per pair of accumulator rows) instead of 2-byte stores.
// Deterministic: per-workgroup dW partials, fixed-order f64 second stage.
#include "dic_bnhead.h"

namespace dic {

constexpr int FO = 128, FI = 256;                    // out / in features of the layer
constexpr int FT = 32;                               // rows per tile
constexpr int FZ_PITCH = FO * 2 + 80;                // 336 B
constexpr int FX_PITCH = FI * 2 + 80;                // 592 B
constexpr int FS_PITCH = FI * 2 + 64;                // 576 B: dX staging rows (rows m, m + 1 written by one store: 16 banks apart)
constexpr int F_TILE = FT * (FZ_PITCH + FX_PITCH);   // 29 696 B per tile image
constexpr int F_STAGE = FT * FS_PITCH;               // 18 432 B
constexpr int F_LDS = 2 * F_TILE + 2 * F_STAGE;      // 96 256 B
constexpr int F_W2_MAX = 12;                         // FUSED: 128-float rows kept behind the staging tiles: the head weight W2 (C rows) + 6 rows of column constants
constexpr int F_LDS_FUSED = F_LDS + F_W2_MAX * FO * 4;

typedef __bf16 fbf16x8 __attribute__((ext_vector_type(8)));
typedef short fs16x4 __attribute__((ext_vector_type(4)));
typedef short fs16x8 __attribute__((ext_vector_type(8)));
typedef float ff32x16 __attribute__((ext_vector_type(16)));

struct FcBwdArgs {
    const __bf16* dz;      // (N, 128)
    const __bf16* x;       // (N, 256)
    const __bf16* w;       // (128, 256)
    __bf16* dx;            // (N, 256) or NULL
    float* partials;       // (gridDim.x, 128, 256)
    long N;
    // FUSED: dZ is not read but formed on the way into LDS from the BatchNorm -> ReLU -> Dropout -> Linear(128, C) tail behind this layer
    // (what dic_bnhead_bwd_input would have written, same arithmetic in the same order): z = this layer's own output, dv = the gradient
    // of the tail's output, the tail's parameters, and the column sums of its first backward pass (dic_bnhead_bwd_reduce)
    const __bf16* z;       // (N, 128)
    const float* dv;       // (N, C)
    const float *mean, *rstd, *gamma, *beta, *w2;      // (128) x 4, (C, 128)
    const float *sum_da, *sum_dax, *count;             // (128), (128), (1): global row count of the batch moments
    float floor_, drop_p;
    const unsigned long long* rng;
};

__device__ __forceinline__ fs16x4 f_lds_tr16(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fs16x4*)(p));
}

template <bool FUSED, int C>
__global__ __launch_bounds__(512) void fc_bwd_kernel(FcBwdArgs a) {
    extern __shared__ __align__(16) unsigned char fsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long N = a.N;
    const int ntiles = (int)((N + FT - 1) / FT), nwg = gridDim.x;
    unsigned char* stage = fsm + 2 * F_TILE;

    // B operand of dX: W[o = 16 ks + 8 kg + j][i = 32 w + (lane & 31)], j = 0..7 -- resident
    fbf16x8 wreg[FO / 16];
    {
        const int i = 32 * w + (lane & 31), kg = lane >> 5;
#pragma unroll
        for (int ks = 0; ks < FO / 16; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) wreg[ks][j] = a.w[(size_t)(16 * ks + 8 * kg + j) * FI + i];
    }
    ff32x16 dw[4];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int k = 0; k < 16; ++k) dw[mb][k] = 0.f;

    // global -> register pieces of a tile (rows past the end read as zeros: they add nothing to dW and their dX is not stored)
    const int zrow = tid >> 4, zpc = tid & 15;               // dZ: 32 rows x 16 pieces
    const int xrow = tid >> 5, xpc = tid & 31;               // X: rows xrow, xrow + 16, 32 pieces each
    uint4 pz, px0, px1;
    // FUSED: this thread's 8 columns (zpc) of the tail: BatchNorm constants, the two mean terms of its backward; W2 in LDS
    // h = max(scale z + shift, floor),  dz = scale (da - c1 - xhat c2),  xhat = (z - mean) rstd,  c1 = sum_da / n,  c2 = sum_dax / n:
    // the arithmetic of dic_bnhead_bwd_input, operation for operation (bit-identical dZ).  The six constants per column live in LDS
    // behind W2 (the kernel is at its register limit).
    static_assert(C + 6 <= F_W2_MAX, "LDS rows behind the staging tiles: W2, scale, shift, mean, rstd, c1, c2");
    Drop drop;
    float pg[C];
    long prow = 0;
    float* w2s = reinterpret_cast<float*>(fsm + F_LDS);
    if constexpr (FUSED) {
        const ColParams cp = load_cols(a.mean, a.rstd, a.gamma, a.beta, zpc);
        drop = load_drop(a.drop_p, a.rng);
        const float inv_n = 1.0f / a.count[0];
        if (zrow == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = zpc * 8 + e;
                w2s[(C + 0) * FO + k] = cp.scale[e];
                w2s[(C + 1) * FO + k] = cp.shift[e];
                w2s[(C + 2) * FO + k] = cp.mean[e];
                w2s[(C + 3) * FO + k] = cp.rstd[e];
                w2s[(C + 4) * FO + k] = a.sum_da[k] * inv_n;
                w2s[(C + 5) * FO + k] = a.sum_dax[k] * inv_n;
            }
        }
        for (int i = tid; i < C * FO; i += 512) w2s[i] = a.w2[i];
    }
    auto request = [&](int tile) {           // (loads from clamped, always valid rows; the rows past the end are zeroed by a select)
        const long r0 = (long)tile * FT;
        const long rz = min(r0 + zrow, N - 1), ra = min(r0 + xrow, N - 1), rb = min(r0 + xrow + 16, N - 1);
        const uint4 vz = *reinterpret_cast<const uint4*>((FUSED ? a.z : a.dz) + (size_t)rz * FO + zpc * 8);
        if constexpr (FUSED) {
            prow = r0 + zrow;
#pragma unroll
            for (int j = 0; j < C; ++j) pg[j] = a.dv[(size_t)rz * C + j];
        }
        const uint4 va = *reinterpret_cast<const uint4*>(a.x + (size_t)ra * FI + xpc * 8);
        const uint4 vb = *reinterpret_cast<const uint4*>(a.x + (size_t)rb * FI + xpc * 8);
        const bool kz = r0 + zrow < N, ka = r0 + xrow < N, kb = r0 + xrow + 16 < N;
        pz = make_uint4(kz ? vz.x : 0u, kz ? vz.y : 0u, kz ? vz.z : 0u, kz ? vz.w : 0u);
        px0 = make_uint4(ka ? va.x : 0u, ka ? va.y : 0u, ka ? va.z : 0u, ka ? va.w : 0u);
        px1 = make_uint4(kb ? vb.x : 0u, kb ? vb.y : 0u, kb ? vb.z : 0u, kb ? vb.w : 0u);
    };
    auto land = [&](int slot) {
        unsigned char* base = fsm + slot * F_TILE;
        if constexpr (FUSED) {
            // da = (dv W2) keep 1[h > floor]   (dic_bnhead.hip: recompute_row + bwd B, the mean terms folded into ca + cb z)
            const fbf16x8 x = __builtin_bit_cast(fbf16x8, pz);
            float keep[8];
            drop_factors(drop, prow, zpc, keep);
            fbf16x8 o;
#pragma unroll
            for (int half = 0; half < 2; ++half) {        // four columns at a time: all the LDS operands of eight at once overflow the register file
                const float* col = w2s + zpc * 8 + 4 * half;
                float4 sacc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < C; ++j) {      // (W2 and the column constants come from LDS: held in registers they spill -- 96 values per thread)
                    const float4 wj = *reinterpret_cast<const float4*>(col + j * FO);
                    sacc.x = fmaf(pg[j], wj.x, sacc.x); sacc.y = fmaf(pg[j], wj.y, sacc.y);
                    sacc.z = fmaf(pg[j], wj.z, sacc.z); sacc.w = fmaf(pg[j], wj.w, sacc.w);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float4 k0 = *reinterpret_cast<const float4*>(col + C * FO), k1 = *reinterpret_cast<const float4*>(col + (C + 1) * FO);
                const float4 k2 = *reinterpret_cast<const float4*>(col + (C + 2) * FO), k3 = *reinterpret_cast<const float4*>(col + (C + 3) * FO);
                const float4 k4 = *reinterpret_cast<const float4*>(col + (C + 4) * FO), k5 = *reinterpret_cast<const float4*>(col + (C + 5) * FO);
                const float sv[4] = {sacc.x, sacc.y, sacc.z, sacc.w}, scale[4] = {k0.x, k0.y, k0.z, k0.w}, shift[4] = {k1.x, k1.y, k1.z, k1.w};
                const float mean[4] = {k2.x, k2.y, k2.z, k2.w}, rstd[4] = {k3.x, k3.y, k3.z, k3.w}, c1[4] = {k4.x, k4.y, k4.z, k4.w}, c2[4] = {k5.x, k5.y, k5.z, k5.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = 4 * half + i;
                    const float xe = (float)x[e];
                    const float act = fmaxf(fmaf(xe, scale[i], shift[i]), a.floor_);
                    const float xhat = (xe - mean[i]) * rstd[i];
                    const float da = act > a.floor_ ? sv[i] * keep[e] : 0.f;
                    o[e] = (__bf16)(scale[i] * (da - c1[i] - xhat * c2[i]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const uint4 v = __builtin_bit_cast(uint4, o);
            const bool kz = prow < N;
            pz = make_uint4(kz ? v.x : 0u, kz ? v.y : 0u, kz ? v.z : 0u, kz ? v.w : 0u);
        }
        *reinterpret_cast<uint4*>(base + zrow * FZ_PITCH + zpc * 16) = pz;
        *reinterpret_cast<uint4*>(base + FT * FZ_PITCH + xrow * FX_PITCH + xpc * 16) = px0;
        *reinterpret_cast<uint4*>(base + FT * FZ_PITCH + (xrow + 16) * FX_PITCH + xpc * 16) = px1;
    };

    // transposed-read addressing (see lstm_dw_kernel): lane 4q+p of a 16-lane group supplies "row" q, columns 4p..4p+3; lane l of the
    // 32x32x16 operand takes column (l & 31) and 8 reduction terms.  Which tile rows those are is free (the same for dZ and X): the q-th
    // row of a read is tile row 16 ks + 4 q + 2 (l >> 5) (+ 1 for the second read), see the note on the pitches
    const int kq = (lane & 15) >> 2, kp = lane & 3, cb = (lane >> 4) & 1, hh = lane >> 5;
    const int rowoff = 4 * kq + 2 * hh;
    auto frag = [&](const unsigned char* p, int pitch) {
        const fs16x4 lo = f_lds_tr16(p), hi = f_lds_tr16(p + pitch);
        fs16x8 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = lo[j]; f[4 + j] = hi[j]; }
        return __builtin_bit_cast(fbf16x8, f);
    };
    const int za_off = (lane & 31) * FZ_PITCH + hh * 16;                               // dX's A: row (lane & 31), 16-B piece 2 ks + hh
    const int zt_off = rowoff * FZ_PITCH + (16 * cb + 4 * kp) * 2;                     // dW's A: + 64 mb (column block), + 16 ks rows
    const int xt_off = FT * FZ_PITCH + rowoff * FX_PITCH + (32 * w + 16 * cb + 4 * kp) * 2;

    int tile = blockIdx.x;
    if constexpr (FUSED) __syncthreads();                    // W2 is in LDS
    if (tile < ntiles) { request(tile); land(0); }
    if (tile + nwg < ntiles) request(tile + nwg);
    __syncthreads();
    int slot = 0;
    auto store_rows = [&](int st_slot, long r0) {             // the dX block of a tile: staging -> global, whole 512-B rows
        if (!a.dx || r0 < 0) return;
        const unsigned char* sb = stage + st_slot * F_STAGE;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = tid + 512 * k, row = p >> 5, pc = p & 31;
            if (r0 + row < N)
                *reinterpret_cast<uint4*>(a.dx + (size_t)(r0 + row) * FI + pc * 8) = *reinterpret_cast<const uint4*>(sb + row * FS_PITCH + pc * 16);
        }
    };
    for (; tile < ntiles; tile += nwg) {
        const unsigned char* base = fsm + slot * F_TILE;
        // ---- dX block of this wave
        ff32x16 acc;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
#pragma unroll
        for (int ks = 0; ks < FO / 16; ++ks) {
            const fbf16x8 af = *reinterpret_cast<const fbf16x8*>(base + za_off + ks * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wreg[ks], acc, 0, 0, 0);
        }
        // ---- dW blocks
#pragma unroll
        for (int ks = 0; ks < FT / 16; ++ks) {
            const fbf16x8 bf = frag(base + xt_off + ks * 16 * FX_PITCH, FX_PITCH);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const fbf16x8 af = frag(base + zt_off + ks * 16 * FZ_PITCH + mb * 64, FZ_PITCH);
                dw[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, dw[mb], 0, 0, 0);
            }
        }
        // ---- next tile's pieces -> the other image, the one after that requested; this tile's dX block -> staging
        if (tile + nwg < ntiles) land(slot ^ 1);
        if (tile + 2 * nwg < ntiles) request(tile + 2 * nwg);
        if (a.dx) {
            // accumulators 2p, 2p + 1 of a lane are rows m, m + 1 of its column: the even lane of a pair takes both columns of row m, the odd
            // lane those of row m + 1 (one quad_perm [1,0,3,2] swap), each stores ONE packed dword
            unsigned char* sb = stage + slot * F_STAGE;
            const int odd = lane & 1;
            unsigned char* sdst = sb + odd * FS_PITCH + (32 * w + ((lane & 31) & ~1)) * 2;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k0 = 2 * kk, m = (k0 & 3) + 8 * (k0 >> 2) + 4 * hh;
                // (both candidates are formed and ONE select picks: a select between acc[k0] and acc[k0 + 1] themselves becomes a dynamic
                // vector index, i.e. a 16-way select chain)
                const float a0 = acc[k0], a1 = acc[k0 + 1];       // (locals: __builtin_bit_cast of a vector ELEMENT lvalue reads element 0)
                const float n0 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a0), 0xB1, 0xF, 0xF, true));
                const float n1 = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a1), 0xB1, 0xF, 0xF, true));
                typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
                const bf16x2_t pe = {(__bf16)a0, (__bf16)n0}, po = {(__bf16)n1, (__bf16)a1};
                const unsigned pk = odd ? __builtin_bit_cast(unsigned, po) : __builtin_bit_cast(unsigned, pe);
                *reinterpret_cast<unsigned*>(sdst + m * FS_PITCH) = pk;
            }
        }
        // LDS-only barrier (round 6): __syncthreads() carries s_waitcnt vmcnt(0) -- every 32-row tile then waited out the full latency of the loads `request`
        // had just issued (they are consumed a whole tile later, in `land`, where the compiler's own wait stands) and of the previous tile's row stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        store_rows(slot, (long)tile * FT);
        slot ^= 1;
    }
    // C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* o = a.partials + (size_t)blockIdx.x * FO * FI;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = 32 * mb + (k & 3) + 8 * (k >> 2) + 4 * hh;
            o[(size_t)m * FI + 32 * w + (lane & 31)] = dw[mb][k];
        }
}

__global__ __launch_bounds__(256) void fc_bwd_finalize(const float* partials, int nwg, float* dw) {
    __shared__ double red[256];
    const double s = reduce_partials_32x8(partials, nwg, FO * FI, blockIdx.x * 32, red);
    const int i = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x < 32 && i < FO * FI) dw[i] = (float)s;
}

template <bool FUSED, int C>
static int fc_bwd_launch(const FcBwdArgs& a, int nwg, float* dw, hipStream_t st) {
    constexpr int lds = FUSED ? F_LDS_FUSED : F_LDS;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)fc_bwd_kernel<FUSED, C>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "fc_bwd: cannot reserve %d B of LDS: %s", lds, hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((fc_bwd_kernel<FUSED, C>), dim3(nwg), dim3(512), lds, st, a);
    hipLaunchKernelGGL(fc_bwd_finalize, dim3(FO * FI / 32), dim3(256), 0, st, (const float*)a.partials, nwg, dw);
    return check_launch("fc_bwd");
}

static int fc_bwd_blocks(long N) { return (int)max(1L, min((N + FT - 1) / FT, (long)kNumCU)); }

}  // namespace dic

using namespace dic;

extern "C" {

size_t dic_fc_bwd_workspace(int64_t N, int in_features, int out_features) {
    if (N <= 0 || in_features != FI || out_features != FO) return 0;
    return (size_t)fc_bwd_blocks(N) * FO * FI * sizeof(float);
}

int dic_fc_bwd(const void* dz, const void* x, const void* w, int64_t N, int in_features, int out_features, void* dx, float* dw,
               void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(N > 0, DIC_ERR_INVALID_ARG, "fc_bwd: non-positive row count");
    DIC_REQUIRE(in_features == FI && out_features == FO, DIC_ERR_UNSUPPORTED, "fc_bwd: Linear(%d, %d) (compiled for Linear(%d, %d))", in_features,
                out_features, FI, FO);
    DIC_REQUIRE(dz && x && w && dw && workspace, DIC_ERR_INVALID_ARG, "fc_bwd: NULL pointer");
    const int nwg = fc_bwd_blocks(N);
    DIC_REQUIRE(workspace_bytes >= (size_t)nwg * FO * FI * sizeof(float), DIC_ERR_WORKSPACE, "fc_bwd: workspace %zu B too small", workspace_bytes);
    FcBwdArgs a{};
    a.dz = (const __bf16*)dz; a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.dx = (__bf16*)dx; a.partials = (float*)workspace; a.N = (long)N;
    return fc_bwd_launch<false, 1>(a, nwg, dw, (hipStream_t)stream);
}

int dic_fc_bwd_bnhead(const void* z, const float* dv, const float* mean, const float* rstd, const float* gamma, const float* beta, const float* w2,
                      const float* sum_da, const float* sum_dax, const float* count, int C, int relu, float drop_p, const uint64_t* rng,
                      const void* x, const void* w, int64_t N, int in_features, int out_features, void* dx, float* dw,
                      void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(N > 0, DIC_ERR_INVALID_ARG, "fc_bwd_bnhead: non-positive row count");
    DIC_REQUIRE(in_features == FI && out_features == FO, DIC_ERR_UNSUPPORTED, "fc_bwd_bnhead: Linear(%d, %d) (compiled for Linear(%d, %d))", in_features,
                out_features, FI, FO);
    DIC_REQUIRE(C == 6, DIC_ERR_UNSUPPORTED, "fc_bwd_bnhead: head width %d (compiled for the reference's 6 channels; use dic_bnhead_bwd_input + dic_fc_bwd)", C);
    DIC_REQUIRE(z && dv && mean && rstd && gamma && beta && w2 && sum_da && sum_dax && count && x && w && dw && workspace, DIC_ERR_INVALID_ARG,
                "fc_bwd_bnhead: NULL pointer");
    const int nwg = fc_bwd_blocks(N);
    DIC_REQUIRE(workspace_bytes >= (size_t)nwg * FO * FI * sizeof(float), DIC_ERR_WORKSPACE, "fc_bwd_bnhead: workspace %zu B too small", workspace_bytes);
    FcBwdArgs a{};
    a.x = (const __bf16*)x; a.w = (const __bf16*)w; a.dx = (__bf16*)dx; a.partials = (float*)workspace; a.N = (long)N;
    a.z = (const __bf16*)z; a.dv = dv; a.mean = mean; a.rstd = rstd; a.gamma = gamma; a.beta = beta; a.w2 = w2;
    a.sum_da = sum_da; a.sum_dax = sum_dax; a.count = count; a.floor_ = relu ? 0.f : -INFINITY; a.drop_p = drop_p; a.rng = (const unsigned long long*)rng;
    return fc_bwd_launch<true, 6>(a, nwg, dw, (hipStream_t)stream);
}

}  // extern "C"
