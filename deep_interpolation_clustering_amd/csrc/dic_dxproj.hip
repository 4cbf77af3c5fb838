// The decoder LSTM's input gradient  dX = dG . W_ih  over every (time step, encounter) row (clustering_interp.py:29-41 through nn.LSTM's
// backward):
//     dx[row][n] = sum_k dg[row][k] w[k][n]        dg (N, 1024) bf16 = [direction][gate][unit] gate gradients, w (1024, 256) bf16, dx (N, 256) bf16
// N = 786 432 rows at B = 32 768: 412 GFLOP over 1.6 GB of dG.  ONE kernel forms it: dic_lstm_dx_tile, 256 x 256 macro-tiles with both
// operands streamed through LDS-DMA rings (below).  Round 3's resident-weight design (W_ih in 256 registers per wave, dG streamed: every CU
// took in its row share of dG twice, 571 us against 447 us for the library GEMM) and the library-GEMM branch were removed in round 6 -- one
// way to form each product (docs/history.md round 3 / 5 for the measurements).
#include "dic_common.h"

namespace dic {

constexpr int XK = 1024;                      // K = 2 directions x 4 gates x 128 units
constexpr int XN = 256;                       // decoder input width
typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef float xf32x16 __attribute__((ext_vector_type(16)));

// LDS-DMA as asm (not counted by the compiler: see dic_lstmgrad.hip): 64 lanes x 16 B -> 1 KiB at lds_dst
__device__ __forceinline__ void xdma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// 256 x 256 macro-tiles, nothing resident (round 5).  dX (N, 256) = dG (N, 1024) . W_ih (1024, 256) as an "NT" product of two
// row-major operands with k contiguous: dG rows and the rows of W_ih^T (256, 1024).  One persistent workgroup of 8 waves per CU walks its
// 256-row tiles; a tile's k range streams through LDS in 64-deep slabs (128 B = one cache line per row): dG slabs (32 KB, from HBM) through a
// 3-slot ring, W^T slabs (32 KB, L2-resident: the whole matrix is 512 KB) through a 2-slot ring -- all 160 KB of the CU's LDS, filled by LDS-DMA
// only (asm: the compiler does not count it), one raw barrier and one counted vmcnt wait per slab, the ring running on ACROSS tile boundaries.
// Every CU takes in its row share of dG ONCE (6.3 MB at B = 32 768) plus 6.3 MB of W^T from L2.
// Wave (wm, wn) = (w & 3, w >> 2) owns 64 rows x 128 columns: the product is issued transposed, D^T = W^T . dG^T (A fragment = 32 rows of W^T,
// B fragment = 32 rows of dG), so lane (m = lane & 31, hh) ends up with output row m and FOUR consecutive columns per accumulator quad; one
// v_permlane32_swap per quad pair makes that eight -- 16-B stores straight from the registers, no staging tile (there is no LDS left for one).
// LDS image: rows of 128 B back to back, the eight 16-B pieces of row r stored at position p ^ ((r >> 1) & 7) -- the permutation is applied
// through the DMA's per-lane SOURCE address (a DMA instruction fills 1 KiB = 8 rows in lane order), and it makes every 16-lane group of a
// ds_read_b128 fragment read (16 rows, same logical piece) cover all 64 banks once.
constexpr int TM = 256, TN = 256, TK = 64;
constexpr int T_ROWB = TK * 2;                        // bytes of a row inside a slab
constexpr int T_ASLOT = TM * T_ROWB;                  // 32 KB: a slab of dG rows
constexpr int T_BSLOT = TN * T_ROWB;                  // 32 KB: a slab of W^T rows
#ifndef DIC_DXT_NA
#define DIC_DXT_NA 3
#define DIC_DXT_NB 2
#endif
constexpr int T_NA = DIC_DXT_NA, T_NB = DIC_DXT_NB;       // (ring depths: -D overrides for A/B builds, scripts/dx_experiments.sh)
constexpr int T_LDS = T_NA * T_ASLOT + T_NB * T_BSLOT;   // 163 840 B
constexpr int T_SLABS = XK / TK;                      // 16 slabs per tile
static_assert(TN == XN && T_LDS <= 160 * 1024 && T_NB >= 2 && T_NB <= T_NA, "dx_tile: tile / LDS budget / ring depths");

struct DxTileArgs {
    const __bf16* dg;      // (N, 1024)
    const __bf16* wt;      // (256, 1024) = W_ih^T
    __bf16* dx;            // (N, 256)
    long N;
};

__global__ __launch_bounds__(512, 1) void dx_tile_kernel(DxTileArgs a) {
    extern __shared__ __align__(16) unsigned char tsm[];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, l31 = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 3, wn = w >> 2;
    const long N = a.N;
    const int ntiles = (int)((N + TM - 1) / TM), nch = gridDim.x;
    const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / nch + 1 : 0;
    const int S = my_tiles * T_SLABS;
    if (S == 0) return;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)tsm);
    const unsigned ldsA = lds0, ldsB = lds0 + T_NA * T_ASLOT;
    // DMA: wave w fills the 1-KiB pieces c = w, w + 8, w + 16, w + 24 of a slab (rows 8c .. 8c + 7); lane L -> row 8c + (L >> 3), physical piece L & 7,
    // i.e. logical piece (L & 7) ^ ((row >> 1) & 7) with (row >> 1) & 7 = (4 (w & 1) + (L >> 4)) & 7 for every piece of this wave (c = w mod 8)
    const unsigned v_dma = (unsigned)(lane >> 3) * (XK * 2) + (unsigned)(((lane & 7) ^ ((4 * (w & 1) + (lane >> 4)) & 7)) * 16);
    auto tile_row0 = [&](int i) { return min((long)((int)blockIdx.x + i * nch) * TM, N - TM); };
    auto issue_a = [&](int s) {
        const int i = s / T_SLABS, ks = s % T_SLABS;
        const __bf16* src = a.dg + (size_t)tile_row0(i) * XK + ks * TK;
        const unsigned dst = ldsA + (s % T_NA) * T_ASLOT;
#pragma unroll
        for (int j = 0; j < TM / 64; ++j) {
            const int c = w + 8 * j;
            xdma16(src + (size_t)(8 * c) * XK, v_dma, dst + c * 1024);
        }
    };
    auto issue_b = [&](int s) {
        const int ks = s % T_SLABS;
        const __bf16* src = a.wt + ks * TK;
        const unsigned dst = ldsB + (s % T_NB) * T_BSLOT;
#pragma unroll
        for (int j = 0; j < TN / 64; ++j) {
            const int c = w + 8 * j;
            xdma16(src + (size_t)(8 * c) * XK, v_dma, dst + c * 1024);
        }
    };
    // fragment reads: row (32 block + l31), logical piece 2 kk + hh -> physical piece ^ sw
    const int sw = (l31 >> 1) & 7;
    int poff[TK / 16];
#pragma unroll
    for (int kk = 0; kk < TK / 16; ++kk) poff[kk] = ((2 * kk + hh) ^ sw) * 16;
    const int a_row = (128 * wn + l31) * T_ROWB;          // W^T rows of this wave's 4 column blocks (+ 32 nb rows)
    const int b_row = (64 * wm + l31) * T_ROWB;           // dG rows of this wave's 2 row blocks (+ 32 mb rows)

    // prologue = what iterations -(T_NA - 1) .. -1 of the steady state would have issued, in its order (W^T slab first, then the dG slab)
#pragma unroll
    for (int it = 1 - T_NA; it < 0; ++it) {
        if (it + T_NB - 1 >= 0 && it + T_NB - 1 < S) issue_b(it + T_NB - 1);
        if (it + T_NA - 1 < S) issue_a(it + T_NA - 1);
    }
    xf32x16 acc[4][2];
    // a finished tile's accumulators -> bf16 -> global: called one iteration LATE (behind the next tile's first barrier and DMA issue), so that the 16
    // stores are the youngest entries of the in-order memory queue and nothing waits for them for the next two slabs
    auto store_tile = [&](int tile_i) {
            // D^T layout: lane (m = l31, hh), register k -> column 32 nb + (k & 3) + 8 (k >> 2) + 4 hh.  Quads g = k >> 2: (0, 1) and (2, 3) are traded
            // between the lane halves so that each lane holds eight consecutive columns: [g even own | partner's] for hh = 0, [partner's | g odd own] for hh = 1
            const long r0 = tile_row0(tile_i);
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                __bf16* orow = a.dx + (size_t)(r0 + 64 * wm + 32 * mb + l31) * XN + 128 * wn + 8 * hh;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        typedef __bf16 obf16x2 __attribute__((ext_vector_type(2)));
                        unsigned q[2][2];
#pragma unroll
                        for (int g = 0; g < 2; ++g)
#pragma unroll
                            for (int h2 = 0; h2 < 2; ++h2) {
                                const float v0 = acc[nb][mb][4 * (2 * gp + g) + 2 * h2], v1 = acc[nb][mb][4 * (2 * gp + g) + 2 * h2 + 1];
                                obf16x2 pr = {(__bf16)v0, (__bf16)v1};
                                q[g][h2] = __builtin_bit_cast(unsigned, pr);
                            }
                        // permlane32_swap(x, y): x of lanes 32..63 <-> y of lanes 0..31
                        const auto s0 = __builtin_amdgcn_permlane32_swap(q[0][0], q[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(q[0][1], q[1][1], false, false);
                        uint4 v;
                        v.x = s0[0]; v.y = s1[0]; v.z = s0[1]; v.w = s1[1];
                        *reinterpret_cast<uint4*>(orow + 32 * nb + 16 * gp) = v;
                    }
            }
    };
    for (int s = 0; s < S; ++s) {
        const int ks = s % T_SLABS;
        // this wave's pieces of slab s (dG: issued two iterations ago, W^T: one) have landed once only what was issued AFTER them is still in
        // flight: the 4 dG pieces of slab s + 1 -- and, on the first slab of a later tile, the previous tile's 16 output stores behind them
        // Counted wait.  An iteration issues 4 DMA instructions for W^T slab j + NB - 1, then 4 for dG slab j + NA - 1.  W^T slab s went out first thing in
        // iteration s - NB + 1; behind it came that iteration's dG slab and the whole iterations s - NB + 2 .. s - 1: 4 + 8 (NB - 2) instructions may
        // still be in flight when it has landed (dG slab s is older when NA > NB; with NA == NB it is that iteration's own dG slab and only the
        // 8 (NB - 2) count).  A tile's 16 output stores are issued in the FIRST iteration of the next tile, behind that iteration's DMA instructions: they
        // sit behind the slabs waited for in the NB - 1 iterations after it.  Towards the end of the stream fewer instructions were issued than the count
        // assumes: drain.
        {
            constexpr int inflight = (T_NA > T_NB ? 4 : 0) + 8 * (T_NB - 2);
            const int tail = S - 1 - s;
            if (tail >= T_NA - 1) {
                if (ks >= 1 && ks <= T_NB - 1 && s >= T_SLABS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(inflight + 16) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(inflight) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // everybody's pieces of slab s are in; everybody is done reading slab s - 1 (its two slots are free)
        const unsigned char* A = tsm + T_NA * T_ASLOT + (s % T_NB) * T_BSLOT + a_row;
        const unsigned char* Bm = tsm + (s % T_NA) * T_ASLOT + b_row;
        // fragments of k-step kk + 1 are requested before the MFMAs of k-step kk are issued (two register sets): a ds_read_b128 takes ~100+ cycles to
        // come back, and both waves of a SIMD leave the barrier together -- with one register set every group of MFMAs waited for the reads just issued
        xbf16x8 af[2][4], bf[2][2];
        auto load_frags = [&](int kk, int set) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) bf[set][mb] = *reinterpret_cast<const xbf16x8*>(Bm + mb * 32 * T_ROWB + poff[kk]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) af[set][nb] = *reinterpret_cast<const xbf16x8*>(A + nb * 32 * T_ROWB + poff[kk]);
        };
        load_frags(0, 0);
        __builtin_amdgcn_sched_barrier(0);          // (the first fragment reads go out BEFORE the DMA instructions: their latency hides behind the DMA issue)
#ifndef DIC_DXT_EXP_NOB
        if (s + T_NB - 1 < S) issue_b(s + T_NB - 1);
#endif
#ifndef DIC_DXT_EXP_NOA
        if (s + T_NA - 1 < S) issue_a(s + T_NA - 1);
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) {
#ifndef DIC_DXT_EXP_NOSTORE
            if (s > 0) store_tile(s / T_SLABS - 1);
#endif
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[nb][mb][k] = 0.f;
        }
#pragma unroll
        for (int kk = 0; kk < TK / 16; ++kk) {
            if (kk + 1 < TK / 16) load_frags(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);          // (left alone the scheduler folds the two register sets back into one and re-serialises reads and MFMAs)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
#ifdef DIC_DXT_EXP_NOMMA
                    if (nb + mb == 0) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][nb], bf[kk & 1][mb], acc[0][0], 0, 0, 0);
#else
                    acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][nb], bf[kk & 1][mb], acc[nb][mb], 0, 0, 0);
#endif
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#ifndef DIC_DXT_EXP_NOSTORE
    store_tile(my_tiles - 1);
#else
    if (a.N < 0) store_tile(my_tiles - 1);
#endif
}

static int dx_tile_chunks(long N) {
    const int ntiles = (int)((N + TM - 1) / TM);
    return max(1, min(ntiles, kNumCU));
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_lstm_dx_tile(const void* dg, const void* w_ih_t, int64_t N, int gate_columns, int in_features, void* dx, dic_stream_t stream) {
    DIC_REQUIRE(N >= TM, DIC_ERR_INVALID_ARG, "lstm_dx_tile: %lld rows (needs at least %d)", (long long)N, TM);
    DIC_REQUIRE(gate_columns == XK && in_features == XN, DIC_ERR_UNSUPPORTED, "lstm_dx_tile: (%d gate columns -> %d inputs) (compiled for 1024 -> 256)",
                gate_columns, in_features);
    DIC_REQUIRE(dg && w_ih_t && dx, DIC_ERR_INVALID_ARG, "lstm_dx_tile: NULL pointer");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)dx_tile_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_dx_tile: cannot reserve %d B of LDS: %s", T_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    DxTileArgs a{(const __bf16*)dg, (const __bf16*)w_ih_t, (__bf16*)dx, (long)N};
    hipLaunchKernelGGL(dx_tile_kernel, dim3(dx_tile_chunks(N)), dim3(512), T_LDS, (hipStream_t)stream, a);
    return check_launch("lstm_dx_tile");
}

}  // extern "C"
