// The decoder LSTM's input gradient  dX = dG . W_ih  over every (time step, encounter) row (clustering_interp.py:29-41 through nn.LSTM's
// backward):
//     dx[row][n] = sum_k dg[row][k] w[k][n]        dg (N, 1024) bf16 = [direction][gate][unit] gate gradients, w (1024, 256) bf16, dx (N, 256) bf16
// N = 786 432 rows at B = 32 768: 412 GFLOP over 1.6 GB of dG -- the last library GEMM of the timed step in round 2 (hipBLASLt
// MT256x256x64 stream-K: 0.445 ms, 4.4 TB/s of its own traffic).  Same idea as dic_rowproj.hip, with the roles of the long and the short
// dimension swapped: the WEIGHTS never move.  One workgroup of 4 waves per CU, one wave per SIMD; wave w keeps w[0..1023][n] for its 32
// output columns in 256 registers (the B operands of all 64 MFMA k-steps), the workgroup streams 32-row tiles of dG (64 KB each) into
// a two-slot LDS ring by LDS-DMA (asm, counted vmcnt: the next tile stays in flight across the barriers), every wave reads the whole
// tile as A fragments (ds_read_b128, rows at a 2064-B pitch: conflict-free), and the 32 x 128 output block leaves through an LDS
// staging tile as 16-B pieces.  grid (chunks, 2 column stripes); the chunk count is a multiple of 8, so both stripes of a row chunk
// sit on one XCD and the second one finds the dG tiles in that XCD's L2: HBM sees dG about once.
//
// MEASURED (round 3, B = 32 768, same box): 571 us against 447 us for the library GEMM -- this kernel LOSES and is off by default
// (DIC_DX_KERNEL=1 turns it on; tests/test_gpu_lstm.py keeps it correct).  Why: with the weights resident, W_ih (512 KB) is the whole
// register file of a CU, so a workgroup can hold only half of the output columns and EVERY CU has to take in the full 1.6 GB / 128 row
// share of dG (12.5 MB per CU; the library's 256 x 256 macro-tiles take in 6.3 MB per CU and re-read the small W from L2 instead).  One CU
// ingests ~25-30 GB/s through LDS-DMA (MI355X_MICROARCH.md, ldsdma-fill), i.e. >= 0.42 ms for this design whatever the ring depth
// (two whole-tile slots: 661 us; four half-tile slots, 96 KB in flight: 571 us).  Swapping which operand stays resident does not pay when
// the streamed operand is 3 000 times larger than the resident one AND the resident one fills the register file.
#include "dic_common.h"

namespace dic {

constexpr int XK = 1024;                      // K = 2 directions x 4 gates x 128 units
constexpr int XN = 256;                       // decoder input width
constexpr int XT = 32;                        // rows per tile
constexpr int XW = 4;                         // waves per workgroup
constexpr int XKH = XK / 2;                   // a ring slot holds HALF the k range of a 32-row tile (one direction's 512 gate columns)
constexpr int XA_PITCH = XKH * 2 + 16;        // 1040 B = 260 dwords = 4 (mod 64): the 16 rows of a ds_read_b128 lane group cover all 64 banks
constexpr int XA_SLOT = XT * XA_PITCH;        // 33 280 B
constexpr int XNS = 4;                        // ring slots: one being read, three (96 KB per CU) in flight
constexpr int XS_PITCH = XW * 64 + 16;        // staging rows of the 32 x 128 output block (272 B)
constexpr int XS_TILE = XT * XS_PITCH;        // 8 704 B
constexpr int X_LDS = XNS * XA_SLOT + XS_TILE;  // 141 824 B: one workgroup per CU

typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef float xf32x16 __attribute__((ext_vector_type(16)));

struct DxProjArgs {
    const __bf16* dg;      // (N, 1024)
    const __bf16* w;       // (1024, 256)
    __bf16* dx;            // (N, 256)
    long N;
};

// LDS-DMA as asm (not counted by the compiler: see dic_lstmgrad.hip): 64 lanes x 16 B -> 1 KiB at lds_dst
__device__ __forceinline__ void xdma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(XW * 64, 1) void dx_proj_kernel(DxProjArgs a) {
    extern __shared__ __align__(16) unsigned char xsm[];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long N = a.N;
    const int ntiles = (int)((N + XT - 1) / XT), nch = gridDim.x;
    const int n0 = blockIdx.y * (32 * XW), ncol = n0 + 32 * w + (lane & 31);

    // B operands: lane (n = lane & 31, hh) of k-step ks holds w[16 ks + 8 hh + j][ncol], j = 0..7 (strided 2-B loads, once per kernel;
    // the 512-KB weight matrix is L2-resident)
    xbf16x8 wreg[XK / 16];
    const __bf16* wp = a.w + (size_t)(8 * hh) * XN + ncol;
#pragma unroll
    for (int ks = 0; ks < XK / 16; ++ks) {
        xbf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = wp[j * XN];
        wreg[ks] = f;
        wp += 16 * XN;
        // (one running pointer, laundered: left alone the compiler materialises all 512 load addresses up front and spills them; and a
        // fence every four k-steps bounds the landing registers in flight)
        asm volatile("" : "+v"(wp));
        if ((ks & 3) == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)xsm);
    const unsigned v_lane = lane * 16;
    unsigned char* stage = xsm + XNS * XA_SLOT;
    // The row tiles of this workgroup are cut into HALF-tiles h = 2 i + (k half): rows [r0, r0 + 32) of tile blockIdx.x + i nch, gate
    // columns [512 (h & 1), + 512).  r0 = min(32 tile, N - 32): the last tile of a ragged row count is shifted back (its first rows are
    // written twice, with the same values).  Half-tile h lives in ring slot h % 4.
    const int my_tiles = blockIdx.x < ntiles ? (ntiles - 1 - blockIdx.x) / nch + 1 : 0, nh = 2 * my_tiles;
    auto request = [&](int h) {                   // 32 half rows of 1 KiB, 8 per wave
        const long r0 = min((long)(blockIdx.x + (h >> 1) * nch) * XT, N - XT);
        const unsigned base = lds0 + (h & (XNS - 1)) * XA_SLOT;
        const __bf16* src = a.dg + (size_t)r0 * XK + (h & 1) * XKH;
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = p * XW + w;
            xdma16(src + (size_t)row * XK, v_lane, base + row * XA_PITCH);
        }
    };
    const int a_off = (lane & 31) * XA_PITCH + hh * 16;       // A operand: row (lane & 31), 16-B piece 2 ks + hh
    auto copy_out = [&](int i) {                   // tile i of this workgroup: staged output block -> global, 512 pieces of 16 B, two per thread
        const long r0 = min((long)(blockIdx.x + i * nch) * XT, N - XT);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = tid + 256 * k, row = p >> 4, pc = p & 15;
            const uint4 v = *reinterpret_cast<const uint4*>(stage + row * XS_PITCH + pc * 16);
            *reinterpret_cast<uint4*>(a.dx + (size_t)(r0 + row) * XN + n0 + pc * 8) = v;
        }
    };

    for (int h = 0; h < min(nh, XNS - 1); ++h) request(h);
    xf32x16 acc;
    for (int i = 0; i < my_tiles; ++i) {
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int h = 2 * i + kh;
            // this wave's DMA of half-tile h has landed once at most the operations issued after it are outstanding: the 8 + 8 DMA
            // instructions of the next two half-tiles (the two output stores a tile issues in between only make the wait stricter)
            if (h + 2 < nh) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (h + 1 < nh) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();          // ... and every other wave's; all waves are done with half-tile h - 1 (its slot is free) and,
                                                   // at kh == 0, have staged the previous tile's output columns
            if (h + XNS - 1 < nh) request(h + XNS - 1);
            if (kh == 0 && i > 0) copy_out(i - 1);
            const unsigned char* base = xsm + (h & (XNS - 1)) * XA_SLOT;
            if (kh == 0) {
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[k] = 0.f;
            }
#pragma unroll
            for (int ks = 0; ks < XKH / 16; ++ks) {
                const xbf16x8 af = *reinterpret_cast<const xbf16x8*>(base + a_off + ks * 32);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wreg[kh * (XKH / 16) + ks], acc, 0, 0, 0);
                // (a scheduling fence every eight k-steps: the compiler otherwise hoists all fragment reads and spills)
                if ((ks & 7) == 7) asm volatile("" ::: "memory");
            }
            if (kh == 1) {      // C/D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int m = (k & 3) + 8 * (k >> 2) + 4 * hh;
                    *reinterpret_cast<__bf16*>(stage + m * XS_PITCH + (32 * w + (lane & 31)) * 2) = (__bf16)acc[k];
                }
            }
        }
    }
    if (my_tiles > 0) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        copy_out(my_tiles - 1);
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// Round 5: the OTHER design -- 256 x 256 macro-tiles, nothing resident.  dX (N, 256) = dG (N, 1024) . W_ih (1024, 256) as an "NT" product of two
// row-major operands with k contiguous: dG rows and the rows of W_ih^T (256, 1024).  One persistent workgroup of 8 waves per CU walks its
// 256-row tiles; a tile's k range streams through LDS in 64-deep slabs (128 B = one cache line per row): dG slabs (32 KB, from HBM) through a
// 3-slot ring, W^T slabs (32 KB, L2-resident: the whole matrix is 512 KB) through a 2-slot ring -- all 160 KB of the CU's LDS, filled by LDS-DMA
// only (asm: the compiler does not count it), one raw barrier and one counted vmcnt wait per slab, the ring running on ACROSS tile boundaries.
// Every CU takes in its row share of dG ONCE (6.3 MB at B = 32 768) plus 6.3 MB of W^T from L2; round 3's resident-weight kernel above took in
// dG twice (12.5 MB per CU) and lost to the library on exactly that.
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

static int dx_proj_chunks(long N) {
    const int ntiles = (int)((N + XT - 1) / XT);
    int nch = max(1, min(ntiles, kNumCU / 2));
    return nch >= 8 ? nch / 8 * 8 : nch;
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_lstm_dx_wide(const void* dg, const void* w_ih, int64_t N, int gate_columns, int in_features, void* dx, dic_stream_t stream) {
    DIC_REQUIRE(N >= XT, DIC_ERR_INVALID_ARG, "lstm_dx_wide: %lld rows (needs at least %d)", (long long)N, XT);
    DIC_REQUIRE(gate_columns == XK && in_features == XN, DIC_ERR_UNSUPPORTED, "lstm_dx_wide: (%d gate columns -> %d inputs) (compiled for 1024 -> 256)",
                gate_columns, in_features);
    DIC_REQUIRE(dg && w_ih && dx, DIC_ERR_INVALID_ARG, "lstm_dx_wide: NULL pointer");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)dx_proj_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_dx_wide: cannot reserve %d B of LDS: %s", X_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    DxProjArgs a{(const __bf16*)dg, (const __bf16*)w_ih, (__bf16*)dx, (long)N};
    hipLaunchKernelGGL(dx_proj_kernel, dim3(dx_proj_chunks(N), XN / (32 * XW)), dim3(XW * 64), X_LDS, (hipStream_t)stream, a);
    return check_launch("lstm_dx_wide");
}

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
