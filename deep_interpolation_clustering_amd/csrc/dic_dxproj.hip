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

// ------------------------------------------------------------------------------------------------------------------------------------------
// The same product for the f32 step on split products (DIC_DTYPE_F32X3; round 6): dX (N, 256) f32 = dG . W_ih with dG and W_ih^T as PAIRS OF bf16 PLANES
// (x = hi + lo; the recurrence backward writes dG that way, W_ih^T is split by the caller) and every product hi.hi + lo.hi + hi.lo.  Same machine as
// dx_tile_kernel -- one persistent 8-wave workgroup per CU, 256 x 256 macro-tiles, all-LDS-DMA rings with one raw barrier and one counted wait per slab,
// the transposed product, stores straight from the accumulators -- with a slab = BOTH planes of a 32-deep k range (2 x 16 KB: the same 32 KB per slab and
// ring slot, the same four DMA instructions per wave, operand and slab), 48 MFMAs per wave and slab instead of 32 (three terms x two k-steps x eight blocks),
// and f32 output (a lane's accumulator quad is four consecutive columns: one 16-B store).  Until then: dic_gemm_nt on the f32 operand (128 x 128 tiles, W
// converted in the loop, two barriers per 32-deep tile): 1.62 ms at B = 32 768, matrix cores 41 % busy.
// LDS image of a plane: 64-B rows back to back, 16-B piece p of row r at p ^ ((r >> 2) & 3) (through the DMA's source address): the 16 rows of every
// ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: MI355X_MICROARCH.md) then cover all 64 banks once.
constexpr int PK = 32;                                // k per slab
constexpr int P_ROWB = PK * 2;                        // 64 B
constexpr int P_PLANE = TM * P_ROWB;                  // 16 KB: one plane of a slab (dG rows or W^T rows: TM == TN)
constexpr int P_SLOT = 2 * P_PLANE;                   // 32 KB: hi | lo
constexpr int P_SLABS = XK / PK;                      // 32 slabs per tile
constexpr int P_LDS = (T_NA + T_NB) * P_SLOT;         // 163 840 B
static_assert(P_LDS <= 160 * 1024 && TM == TN, "dx_tile_x3: LDS budget");

struct DxTileX3Args {
    const __bf16* dg; long dg_plane;      // (N, 1024) hi plane; lo plane dg_plane elements behind it
    const __bf16* wt; long wt_plane;      // (256, 1024) = W_ih^T hi plane; lo plane wt_plane elements behind it
    float* dx;                            // (N, 256)
    long N;
};

__global__ __launch_bounds__(512, 1) void dx_tile_x3_kernel(DxTileX3Args a) {
    extern __shared__ __align__(16) unsigned char tsm[];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, l31 = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 3, wn = w >> 2;
    const long N = a.N;
    const int ntiles = (int)((N + TM - 1) / TM), nch = gridDim.x;
    const int my_tiles = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / nch + 1 : 0;
    const int S = my_tiles * P_SLABS;
    if (S == 0) return;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)tsm);
    const unsigned ldsA = lds0, ldsB = lds0 + T_NA * P_SLOT;
    // DMA: a plane of a slab = 16 instructions of 16 rows; wave w issues c = w, w + 8, w + 16, w + 24 of the 32 (plane c >> 4, row group c & 15); lane L -> row
    // 16 rg + (L >> 2), physical piece L & 3 = logical piece (L & 3) ^ ((row >> 2) & 3), and (row >> 2) & 3 = (L >> 4) & 3 for every row group
    const unsigned v_dma = (unsigned)(lane >> 2) * (XK * 2) + (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
    auto tile_row0 = [&](int i) { return min((long)((int)blockIdx.x + i * nch) * TM, N - TM); };
    auto issue_a = [&](int s) {
        const int i = s / P_SLABS, ks = s % P_SLABS;
        const __bf16* src = a.dg + (size_t)tile_row0(i) * XK + ks * PK;
        const unsigned dst = ldsA + (s % T_NA) * P_SLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = w + 8 * j, pl = c >> 4, rg = c & 15;
            xdma16(src + (size_t)pl * a.dg_plane + (size_t)(16 * rg) * XK, v_dma, dst + pl * P_PLANE + rg * 1024);
        }
    };
    auto issue_b = [&](int s) {
        const int ks = s % P_SLABS;
        const __bf16* src = a.wt + ks * PK;
        const unsigned dst = ldsB + (s % T_NB) * P_SLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = w + 8 * j, pl = c >> 4, rg = c & 15;
            xdma16(src + (size_t)pl * a.wt_plane + (size_t)(16 * rg) * XK, v_dma, dst + pl * P_PLANE + rg * 1024);
        }
    };
    // fragment reads: row (32 block + l31), logical piece 2 kk + hh -> physical piece ^ ((l31 >> 2) & 3)
    const int sw = (l31 >> 2) & 3;
    int poff[PK / 16];
#pragma unroll
    for (int kk = 0; kk < PK / 16; ++kk) poff[kk] = ((2 * kk + hh) ^ sw) * 16;
    const int a_row = (128 * wn + l31) * P_ROWB;          // W^T rows of this wave's 4 column blocks (+ 32 nb rows)
    const int b_row = (64 * wm + l31) * P_ROWB;           // dG rows of this wave's 2 row blocks (+ 32 mb rows)

#pragma unroll
    for (int it = 1 - T_NA; it < 0; ++it) {
        if (it + T_NB - 1 >= 0 && it + T_NB - 1 < S) issue_b(it + T_NB - 1);
        if (it + T_NA - 1 < S) issue_a(it + T_NA - 1);
    }
    xf32x16 acc[4][2];
    // a finished tile's accumulators -> global, one iteration LATE (see dx_tile_kernel): D^T layout, lane (m = l31, hh), register k -> column
    // 32 nb + (k & 3) + 8 (k >> 2) + 4 hh: quad g = k >> 2 is four consecutive f32 columns
    auto store_tile = [&](int tile_i) {
        const long r0 = tile_row0(tile_i);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            float* orow = a.dx + (size_t)(r0 + 64 * wm + 32 * mb + l31) * XN + 128 * wn + 4 * hh;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    typedef float of32x4 __attribute__((ext_vector_type(4)));
                    const of32x4 v = {acc[nb][mb][4 * g], acc[nb][mb][4 * g + 1], acc[nb][mb][4 * g + 2], acc[nb][mb][4 * g + 3]};
                    *reinterpret_cast<of32x4*>(orow + 32 * nb + 8 * g) = v;
                }
        }
    };
    for (int s = 0; s < S; ++s) {
        const int ks = s % P_SLABS;
        // counted wait: as dx_tile_kernel (4 + 4 DMA instructions per wave and iteration), with 32 output stores per wave behind the first DMAs of a later tile
        {
            constexpr int inflight = (T_NA > T_NB ? 4 : 0) + 8 * (T_NB - 2);
            const int tail = S - 1 - s;
            if (tail >= T_NA - 1) {
                if (ks >= 1 && ks <= T_NB - 1 && s >= P_SLABS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(inflight + 32) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(inflight) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // everybody's pieces of slab s are in; everybody is done reading slab s - 1 (its two slots are free)
        const unsigned char* A = tsm + T_NA * P_SLOT + (s % T_NB) * P_SLOT + a_row;
        const unsigned char* Bm = tsm + (s % T_NA) * P_SLOT + b_row;
        // registers: the hi fragments of a k-step in two sets (the next k-step's are requested under the current one's second term), the lo fragments in one
        // (requested under the first term, which does not use them)
        xbf16x8 ah[2][4], bh[2][2], al[4], bl[2];
        auto load_hi = [&](int kk, int set) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) bh[set][mb] = *reinterpret_cast<const xbf16x8*>(Bm + mb * 32 * P_ROWB + poff[kk]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) ah[set][nb] = *reinterpret_cast<const xbf16x8*>(A + nb * 32 * P_ROWB + poff[kk]);
        };
        auto load_lo = [&](int kk) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) bl[mb] = *reinterpret_cast<const xbf16x8*>(Bm + P_PLANE + mb * 32 * P_ROWB + poff[kk]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) al[nb] = *reinterpret_cast<const xbf16x8*>(A + P_PLANE + nb * 32 * P_ROWB + poff[kk]);
        };
        load_hi(0, 0);
        __builtin_amdgcn_sched_barrier(0);          // (the first fragment reads go out BEFORE the DMA instructions)
        if (s + T_NB - 1 < S) issue_b(s + T_NB - 1);
        if (s + T_NA - 1 < S) issue_a(s + T_NA - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) {
            if (s > 0) store_tile(s / P_SLABS - 1);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[nb][mb][k] = 0.f;
        }
#pragma unroll
        for (int kk = 0; kk < PK / 16; ++kk) {
            load_lo(kk);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk & 1][nb], bh[kk & 1][mb], acc[nb][mb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < PK / 16) load_hi(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[nb], bh[kk & 1][mb], acc[nb][mb], 0, 0, 0);
                    acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk & 1][nb], bl[mb], acc[nb][mb], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    store_tile(my_tiles - 1);
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

int dic_lstm_dx_tile_x3(const void* dg_hi, long dg_plane, const void* w_ih_t_hi, long wt_plane, int64_t N, int gate_columns, int in_features, float* dx,
                        dic_stream_t stream) {
    DIC_REQUIRE(N >= TM, DIC_ERR_INVALID_ARG, "lstm_dx_tile_x3: %lld rows (needs at least %d)", (long long)N, TM);
    DIC_REQUIRE(gate_columns == XK && in_features == XN, DIC_ERR_UNSUPPORTED, "lstm_dx_tile_x3: (%d gate columns -> %d inputs) (compiled for 1024 -> 256)",
                gate_columns, in_features);
    DIC_REQUIRE(dg_hi && w_ih_t_hi && dx && dg_plane > 0 && wt_plane > 0, DIC_ERR_INVALID_ARG, "lstm_dx_tile_x3: NULL pointer / plane stride");
    DIC_REQUIRE(((uintptr_t)dg_hi & 15) == 0 && ((uintptr_t)w_ih_t_hi & 15) == 0 && ((uintptr_t)dx & 15) == 0 && dg_plane % 8 == 0 && wt_plane % 8 == 0,
                DIC_ERR_UNSUPPORTED, "lstm_dx_tile_x3: operands must be 16-B aligned");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)dx_tile_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_dx_tile_x3: cannot reserve %d B of LDS: %s", P_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    DxTileX3Args a{(const __bf16*)dg_hi, dg_plane, (const __bf16*)w_ih_t_hi, wt_plane, dx, (long)N};
    hipLaunchKernelGGL(dx_tile_x3_kernel, dim3(dx_tile_chunks(N)), dim3(512), P_LDS, (hipStream_t)stream, a);
    return check_launch("lstm_dx_tile_x3");
}

}  // extern "C"
