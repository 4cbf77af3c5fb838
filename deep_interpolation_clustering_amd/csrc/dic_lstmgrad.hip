// Parameter-side kernels of the bidirectional LSTMs (clustering_interp.py:14-41: EncoderRNN / DecoderRNN, nn.LSTM parameters
// weight_ih_l0[_reverse], weight_hh_l0[_reverse], bias_ih_l0[_reverse], bias_hh_l0[_reverse]; gate order i,f,g,o):
//
//   lstm_pack_kernel     the eight f32 parameters -> the bf16 operands the recurrence kernels / library GEMMs take
//                        (W_ih padded to the MFMA k-step with the bias as a constant-one input column, W_hh, W_hh^T, bias):
//                        ONE launch instead of ~20 stack / add / cast / pad / transpose launches per LSTM and step.
//   lstm_dw_kernel       encoder weight gradients: dW_hh = sum_t dG_t^T h_{t-1} (h_{t+1} for the reverse direction) and
//                        dW_ih = sum_t dG_t^T x_t from ONE pass over dG (rocprofv3, round 1: three library GEMMs, each re-reading
//                        the 1.6 GB dG at its own HBM floor).  MFMA v_mfma_f32_32x32x16_bf16 with BOTH operands transposed on the
//                        way out of LDS (ds_read_b64_tr_b16): the reduction runs over the (t, b) rows, which is the slow index of
//                        dG, of h and of x in memory.
//   lstm_dw_finalize     fixed-order f64 reduction of the per-workgroup partial sums, written (or accumulated) straight into the
//                        gradients of the nn.LSTM parameters -- no (2,4H,.) staging tensors, no autograd AccumulateGrad adds.
//   lstm_unpack_grads    the same scatter for gradients that came out of library GEMMs (decoder).
#include "dic_common.h"

namespace dic {

constexpr int GH = 128;            // hidden size
constexpr int G4 = 4 * GH;         // gate rows per direction
constexpr int XW = 32;             // packed encoder input width (MFMA k-step multiple; column I carries the constant one)
constexpr int TR = 32;             // (t, b) rows per tile = 2 MFMA k-steps
constexpr int NBUF = 3;            // LDS ring: one tile being read, two in flight
// LDS image of a tile.  A transposed read (ds_read_b64_tr_b16) takes 4 rows x 64 contiguous bytes per 32-lane half:
//   dG rows (1 KiB, one LDS-DMA instruction each) sit at a pitch of 1024 + 64 B -> the 4 rows fall on disjoint 16-bank groups;
//   h rows (256 B) arrive four to a DMA instruction, contiguous, so the spreading is an XOR on the 16-B piece index instead
//   (piece c of row r is stored at position c ^ ((r & 3) << 2): applied to the SOURCE address of the DMA and to the reads);
//   x rows (64 B) are conflict-free as they lie.
#ifndef DIC_DW_DGPAD
#define DIC_DW_DGPAD 64
#endif
constexpr int DG_PITCH = 2 * G4 + DIC_DW_DGPAD;     // 1088
constexpr int H_PITCH = 2 * GH;           // 256
constexpr int X_PITCH = 2 * XW;           // 64
constexpr int LDS_DG = TR * DG_PITCH, LDS_H = TR * H_PITCH, LDS_X = TR * X_PITCH;
constexpr int SLOT = LDS_DG + LDS_H + LDS_X;        // 45 056 B
// the input gradient dX = dG . W_ih rides along (the library GEMM for it re-read the whole 1.6 GB dG): W_ih^T of this direction
// sits in LDS ([19 input columns][512 gate rows], the 18 features + nothing else is ever read downstream), 16x16x32 MFMAs
constexpr int WT_COLS = 19, WT_PITCH = 2 * G4 + 16;  // 1040 B rows: 16 lanes x 16 B reads spread over the banks
constexpr int LDS_WT = WT_COLS * WT_PITCH;           // 19 760 B
constexpr int LDS_RED = 2 * 4 * 1024;                // two parities x four 1-KiB partial tiles (k-halves are summed across wave pairs)
constexpr int DW_LDS = NBUF * SLOT + LDS_WT + LDS_RED;   // 163 120 B of the CU's 163 840: one workgroup per CU

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct LstmParams {     // device pointers of one bidirectional nn.LSTM layer (f32), index = direction
    const float* w_ih[2]; const float* w_hh[2]; const float* b_ih[2]; const float* b_hh[2];
};
struct LstmGrads {
    float* w_ih[2]; float* w_hh[2]; float* b_ih[2]; float* b_hh[2];
};

// ------------------------------------------------------------------------------------------------ pack
// wih (2*4H, Ip) bf16: columns [0,I) = W_ih, column I = b_ih + b_hh when `bias_col` (narrow inputs: the projection runs inside
// the recurrence kernel with a constant-one input column), other padding 0;  whh (2,4H,H) bf16;  whh_t (2,H,4H) bf16;
// bias (2*4H) bf16 = b_ih + b_hh (the addmm operand of the library projection).
// wih_t (Ip, 2*4H) bf16 (optional) = wih transposed: the k-contiguous operand of dic_lstm_dx_tile (decoder: Ip = I = 256).
template <typename T>
__global__ __launch_bounds__(256) void lstm_pack_kernel(LstmParams p, int I, int Ip, int bias_col, T* wih, T* whh, T* whh_t, T* bias, T* wih_t) {
    const int n_ih = 2 * G4 * Ip, n_hh = 2 * G4 * GH;
    const int total = n_ih + 2 * n_hh + 2 * G4 + (wih_t ? n_ih : 0);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        if (i >= n_ih + 2 * n_hh + 2 * G4) {
            // wih_t[c][row] = wih[row][c]: consecutive threads walk the gate rows (coalesced writes; the f32 source rows stay in L2)
            const int j = i - (n_ih + 2 * n_hh + 2 * G4), c = j / (2 * G4), row = j - c * (2 * G4), d = row / G4, g = row - d * G4;
            float v = 0.f;
            if (c < I) v = p.w_ih[d][(size_t)g * I + c];
            else if (c == I && bias_col) v = p.b_ih[d][g] + p.b_hh[d][g];
            wih_t[j] = (T)v;
            continue;
        }
        if (i < n_ih) {
            const int row = i / Ip, c = i - row * Ip, d = row / G4, g = row - d * G4;
            float v = 0.f;
            if (c < I) v = p.w_ih[d][(size_t)g * I + c];
            else if (c == I && bias_col) v = p.b_ih[d][g] + p.b_hh[d][g];
            wih[i] = (T)v;
        } else if (i < n_ih + n_hh) {
            const int j = i - n_ih, d = j / (G4 * GH), k = j - d * (G4 * GH);
            whh[j] = (T)p.w_hh[d][k];
        } else if (i < n_ih + 2 * n_hh) {
            // whh_t[d][u][n] = whh[d][n][u]: consecutive threads walk n (coalesced writes; the 256-KB source stays in L2)
            const int j = i - n_ih - n_hh, d = j / (G4 * GH), k = j - d * (G4 * GH), u = k / G4, n = k - u * G4;
            if (whh_t) whh_t[j] = (T)p.w_hh[d][(size_t)n * GH + u];
        } else {
            const int j = i - n_ih - 2 * n_hh, d = j / G4, g = j - d * G4;
            if (bias) bias[j] = (T)(p.b_ih[d][g] + p.b_hh[d][g]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ fused dW (encoder)
struct DwArgs {
    const __bf16* dg;      // (R*B, 2, 4H) gate gradients
    const __bf16* hext;    // ((R+2)*B, 2H)  the layer's outputs with one extra time slot at either end: slot 0 holds h0 of the forward
                           //                direction in [:H], slot R+1 h0 of the reverse direction in [H:] (zeros without an h0)
    const __bf16* x;       // (R*B, XW)      packed inputs
    float* partials;       // (gridDim.x, 2, 4H, NW) per-workgroup sums
    const __bf16* wih;     // (2*4H, XW) packed input weights, or NULL: no input gradient wanted
    __bf16* dxp;           // (2, R*B, XW) per-direction input gradients dG[d] . W_ih[d] (columns >= 19 are not written)
    int R, B;
};
typedef float f32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}
// LDS-DMA written as asm so that hipcc does NOT count it: with the builtin, the compiler's LDS-DMA tracking puts an
// s_waitcnt vmcnt(0) in front of the first ds_read after any outstanding DMA (it cannot tell the ring slots apart), which
// serialises load and compute.  Uncounted, the loop's own counted vmcnt + raw s_barrier are the only waits; the loop issues
// no other vector-memory instruction, so nothing of the compiler's bookkeeping is disturbed.  M0 = LDS destination base
// (lane i lands at base + 16 i / 4 i); saved and restored around the statement (cdna_hip_programming.md 5.7).
__device__ __forceinline__ void dma16(const void* sbase, unsigned voff, unsigned lds_dst) {      // 64 lanes x 16 B -> 1 KiB at lds_dst
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma4(const void* sbase, unsigned voff, unsigned lds_dst) {       // 64 lanes x 4 B -> 256 B at lds_dst
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// grid (nch, 2 directions), 512 threads = 8 waves (two per SIMD), one workgroup per CU.  Wave w owns gate rows [64w, 64w+64) of
// its direction: 2 m-blocks x 5 n-blocks of 32x32 f32 accumulators (160 registers; four waves x 320 spilled accumulators).  The R*B rows are cut into tiles of 32 without
// regard to the (t, b) structure: the recurrent input of row i is row i (forward) / i + 2B (reverse) of the extended output
// buffer.  Every byte comes in by LDS-DMA into a ring of three tile images; per tile ONE raw s_barrier and a COUNTED
// s_waitcnt vmcnt, so two tiles (90 KB per CU) stay in flight across the barrier while the matrix cores work on the third
// (a __syncthreads() here drains the DMA queue: measured 0.55 ms instead of the 0.4 ms HBM floor).
// XWT = packed input width: 32 (3C < 32, the reference's six vitals: 5 n-blocks, the input gradient dX rides along) or 64 (3C < 64,
// BASELINE configs[3]'s twelve channels: 6 n-blocks = 192 accumulator registers like lstm_dw_wide's waves, x rows of 128 B with their 64-B
// halves XOR-swizzled by (row >> 1) & 1 through the DMA source addresses so that the 4 rows of a transposed read stay on disjoint
// bank groups; no fused dX -- its W_ih^T image does not fit next to the three tile slots -- dic_gemm_nt forms it)
template <int XWT>
__global__ __launch_bounds__(512) void lstm_dw_kernel(DwArgs a) {
    static_assert(XWT == 32 || XWT == 64, "packed input rows are 32 or 64 wide");
    constexpr int NBX = XWT / 32, NBN = 4 + NBX;          // n-blocks of x / in total
    constexpr int NWT = GH + XWT;                          // output columns [h_prev | x]
    constexpr int XP = 2 * XWT, L_X = TR * XP;             // x row pitch / tile bytes
    constexpr int SLOTT = LDS_DG + LDS_H + L_X;
    extern __shared__ __align__(16) unsigned char dwsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y;
    const long nrows = (long)a.R * a.B;
    const int ntiles = (int)((nrows + TR - 1) / TR), nch = gridDim.x;
    const __bf16* hsrc = a.hext + (dir ? (size_t)2 * a.B * 2 * GH + GH : 0);        // row i of this view = h_prev of row i

    f32x16 acc[2][NBN];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBN; ++nb)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[mb][nb][k] = 0.f;

    // a tile = rows [r0, r0 + 32), r0 = min(32 tile, nrows - 32): the last tile of a ragged row count is shifted back so that it ends
    // with the data; the rows it shares with its predecessor get their h / x rows zeroed after landing
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)dwsm);
    const unsigned v_dg = lane * 16;                                                     // per-lane byte offsets of the three streams
    const unsigned v_h = (lane >> 4) * (2 * GH * 2) + (((lane & 15) ^ ((lane >> 4) << 2)) * 16);
    // x: 64-B rows, four per 4-B-per-lane instruction as they lie; 128-B rows, two per instruction, half j of row r stored at j ^ ((r >> 1) & 1)
    const unsigned v_x = lane * 4;
    const unsigned v_x64[2] = {(unsigned)((lane >> 5) * 128 + (((lane >> 4) & 1) * 64) + (lane & 15) * 4),
                               (unsigned)((lane >> 5) * 128 + ((((lane >> 4) & 1) ^ 1) * 64) + (lane & 15) * 4)};
    auto request = [&](int tile, int slot) {
        const long r0 = min((long)tile * TR, nrows - TR);
        const unsigned base = lds0 + slot * SLOTT;
#pragma unroll
        for (int p = 0; p < 4; ++p) {              // dG: one row (1 KiB of this direction) per instruction
            const int row = p * 8 + w;
            dma16(a.dg + ((size_t)(r0 + row) * 2 + dir) * G4, v_dg, base + row * DG_PITCH);
        }
        // h: four 256-B rows per instruction, 16-B pieces XOR-swizzled through the source address
        dma16(hsrc + (size_t)(r0 + w * 4) * 2 * GH, v_h, base + LDS_DG + w * 4 * H_PITCH);
        // x: four 64-B rows per 4-B-per-lane instruction (every wave issues the same number of instructions: one counted wait fits all)
        if constexpr (XWT == 32) {
            dma4(a.x + (size_t)r0 * XWT + w * 128, v_x, base + LDS_DG + LDS_H + w * 256);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)            // rows 4 w + 2 i + {0, 1}: (row >> 1) & 1 == i
                dma4(a.x + (size_t)(r0 + 4 * w + 2 * i) * XWT, v_x64[i], base + LDS_DG + LDS_H + (4 * w + 2 * i) * XP);
        }
    };

    // transposed-read addressing (ds_read_b64_tr_b16): within each 16-lane group, lane 4q+p supplies row q, columns 4p..4p+3 of a
    // 4-row x 16-column block and lane i receives column i of those 4 rows.  For the 32x32x16 operand lane l needs column
    // (l & 31) and rows 8*(l >> 5) + 0..7 of the k-step: group (l >> 4) & 1 takes columns 16.., two reads take rows +0..3, +4..7.
    const int kq = (lane & 15) >> 2, kp = lane & 3, cb = (lane >> 4) & 1, hh = lane >> 5;
    const int rowoff = 8 * hh + kq;                                    // (rowoff & 3) == kq for both reads (+4 rows)
    const int pa_off = rowoff * DG_PITCH + (64 * w + 16 * cb + 4 * kp) * 2;
    int px_off[NBX];
#pragma unroll
    for (int nbx = 0; nbx < NBX; ++nbx)
        px_off[nbx] = LDS_DG + LDS_H + rowoff * XP + (XWT == 64 ? ((nbx ^ ((rowoff >> 1) & 1)) * 64) : 0) + (16 * cb + 4 * kp) * 2;
    int ph_off[4];                                                      // 16-B piece (4 nb + 2 cb + (kp >> 1)) ^ (kq << 2), 8-B half kp & 1
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) ph_off[nb] = LDS_DG + rowoff * H_PITCH + (((4 * nb + 2 * cb + (kp >> 1)) ^ (kq << 2)) * 16) + (kp & 1) * 8;
    auto frag = [&](const unsigned char* p, int pitch) {
        const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 4 * pitch);
        s16x8 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = lo[j]; f[4 + j] = hi[j]; }
        return __builtin_bit_cast(bf16x8, f);
    };

    // ---- dX: this direction's W_ih^T -> LDS once; per tile, wave w multiplies quadrant (rows 16 (q >> 1).., input columns 16 (q & 1)..),
    // q = w & 3, over half of the 512 gate columns (w >> 2); the two halves meet through LDS one tile later (no extra barrier)
    unsigned char* wt = dwsm + NBUF * SLOTT;
    float* red = reinterpret_cast<float*>(wt + LDS_WT);
    const bool want_dx = XWT == 32 && a.wih != nullptr;
    if (want_dx) {
        for (int i = tid; i < WT_COLS * G4; i += 512) {
            const int xc = i / G4, k = i - xc * G4;
            *reinterpret_cast<__bf16*>(wt + xc * WT_PITCH + k * 2) = a.wih[((size_t)dir * G4 + k) * XWT + xc];
        }
    }
    const int q4 = w & 3, rw = q4 >> 1, xcb = q4 & 1, khalf = w >> 2, li = lane & 15, kq4 = lane >> 4;
    const int dxa_off = (16 * rw + li) * DG_PITCH + (khalf * 256 + 8 * kq4) * 2;
    const int dxb_off = min(16 * xcb + li, WT_COLS - 1) * WT_PITCH + (khalf * 256 + 8 * kq4) * 2;
    long prev_r0 = -1;
    int parity = 0;
    f32x4_t own_prev = {0.f, 0.f, 0.f, 0.f};
    auto finish_dx = [&](int par, long r0p) {          // waves 0-3: add the other k-half (parked in LDS by waves 4-7), round, store the 16x16 tile
        if (want_dx && khalf == 0 && r0p >= 0) {
            const f32x4_t o = *reinterpret_cast<const f32x4_t*>(red + (par * 4 + q4) * 256 + lane * 4);
            __bf16* dst = a.dxp + ((size_t)dir * nrows + r0p + 16 * rw + 4 * kq4) * XWT + 16 * xcb + li;
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[(size_t)e * XWT] = (__bf16)(o[e] + own_prev[e]);
        }
    };

    int tile = blockIdx.x, slot = 0;
    if (tile < ntiles) request(tile, 0);
    if (tile + nch < ntiles) request(tile + nch, 1);
    for (; tile < ntiles; tile += nch) {
        // this wave's DMA of the current tile has landed once at most the next tile's instructions are outstanding
        // (the dX stores of waves 0-3 are ordinary global stores issued after the DMAs they must not be confused with: up to 4 per tile)
        if (tile + nch < ntiles) {                 // 6 / 7 = LDS-DMA instructions per wave and tile: 4 dG + 1 h + 1 / 2 x
            if constexpr (XWT == 32) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();              // ... and every other wave's; all waves are done reading the slot refilled next
        finish_dx(parity ^ 1, prev_r0);            // the previous tile's input gradients (its partials were parked before this barrier)
        if (tile + 2 * nch < ntiles) request(tile + 2 * nch, slot == 0 ? 2 : slot - 1);
        unsigned char* base = dwsm + slot * SLOTT;
        if ((long)tile * TR + TR > nrows) {        // the shifted last tile: its first rows were summed by the previous tile already
            const int dup = (int)((long)tile * TR + TR - nrows);
            for (int i = tid; i < dup * (H_PITCH + XP) / 16; i += 512) {
                const int nh = dup * H_PITCH / 16;
                unsigned char* dst = i < nh ? base + LDS_DG + i * 16 : base + LDS_DG + LDS_H + (i - nh) * 16;
                *reinterpret_cast<uint4*>(dst) = make_uint4(0u, 0u, 0u, 0u);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#pragma unroll 1                               // (unrolled, the compiler hoists every fragment read of a tile and spills)
        for (int ks = 0; ks < TR / 16; ++ks) {
            const unsigned char* kb = base + ks * 16 * DG_PITCH;
            bf16x8 af[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) af[mb] = frag(kb + pa_off + mb * 64, DG_PITCH);
#pragma unroll
            for (int nb = 0; nb < NBN; ++nb) {
                const bf16x8 bfg = nb < 4 ? frag(base + ks * 16 * H_PITCH + ph_off[nb < 4 ? nb : 0], H_PITCH)
                                          : frag(base + ks * 16 * XP + px_off[nb < 4 ? 0 : nb - 4], XP);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bfg, acc[mb][nb], 0, 0, 0);
            }
        }
        if (want_dx) {       // D[row][input column] = sum over this wave's 256 gate columns of dG[row][k] W_ih[k][column]
            f32x4_t dacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(base + dxa_off + ks * 64);
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(wt + dxb_off + ks * 64);
                dacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, dacc, 0, 0, 0);
            }
            if (khalf) *reinterpret_cast<f32x4_t*>(red + (parity * 4 + q4) * 256 + lane * 4) = dacc;
            else own_prev = dacc;
            prev_r0 = min((long)tile * TR, nrows - TR);
            parity ^= 1;
        }
        slot = slot == NBUF - 1 ? 0 : slot + 1;
    }
    if (want_dx) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        finish_dx(parity ^ 1, prev_r0);
    }
    // C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* o = a.partials + ((size_t)blockIdx.x * 2 + dir) * G4 * NWT;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBN; ++nb)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int m = 64 * w + 32 * mb + (k & 3) + 8 * (k >> 2) + 4 * hh;
                o[(size_t)m * NWT + 32 * nb + (lane & 31)] = acc[mb][nb][k];
            }
}

// ------------------------------------------------------------------------------------------------ fused dW (decoder, 256-wide input)
// The same one-pass product for the decoder LSTM, whose input rows are the 256 rectified encoder outputs: per direction
// dW[512 gate rows][128 h | 256 x] = sum over (t, b) rows.  The 384 output columns are 12 n-blocks, so a workgroup takes HALF of a
// direction's gate rows: grid (nch, 4 = direction x half); wave w owns 2 m-blocks (64 gate rows, w & 3) x 6 n-blocks (w >> 2) = 192
// accumulator registers and reads 2 + 6 operand fragments per k-step (1 x 12 would read 13: the kernel is bound by LDS fragment
// reads + MFMA issue, 0.55 ms that way).
// The four workgroups of a chunk walk the same rows at the same time and sit on the same XCD (nch is a multiple of 8 and the
// dispatcher deals workgroup ids round-robin over the 8 XCDs), so the x and h tiles they share come out of that XCD's L2 for three
// of them: HBM sees dG once and x, h about once, instead of the three library GEMMs' 1.6 + 1.6 + 0.8 GB.
// Tile image: dG half rows, x rows (512 B each, two per LDS-DMA instruction) and h rows (256 B, four per instruction), all with the
// 16-B pieces XOR-swizzled by (row & 3) << 2 through the DMA's SOURCE addresses (rows at a 256-B-multiple pitch would otherwise put
// the 4 rows of a transposed read on the same banks).
constexpr int DXW = 256;                    // decoder input width
constexpr int DNW = GH + DXW;               // 384 B-operand columns [h_prev | x]
constexpr int DMH = G4 / 2;                 // gate rows per workgroup
static_assert(DNW / 32 == 12, "12 n-blocks = 2 wave groups x 6");
constexpr int WD_DG = TR * DMH * 2, WD_H = TR * GH * 2, WD_X = TR * DXW * 2;      // 16 KB, 8 KB, 16 KB
constexpr int WD_SLOT = WD_DG + WD_H + WD_X;                                       // 40 960 B
constexpr int WD_LDS = NBUF * WD_SLOT;                                             // 122 880 B
constexpr int WD_OUT = 4 * DMH * DNW;       // outputs per partial: [direction x half][gate row][h cols | x cols]

struct DwWideArgs {
    const __bf16* dg;      // (R*B, 2, 4H)
    const __bf16* hext;    // ((R+2)*B, 2H): see DwArgs
    const __bf16* x;       // (R*B, 256)
    float* partials;       // (gridDim.x, 4, 256, 384)
    int R, B;
    int x_relu;            // x is the raw encoder output: the input the LSTM saw is relu(x) (rectified in the B fragments, one v_pk_max_i16 each)
};

__global__ __launch_bounds__(512) void lstm_dw_wide_kernel(DwWideArgs a) {
    extern __shared__ __align__(16) unsigned char dwsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y >> 1, mh = blockIdx.y & 1;
    const long nrows = (long)a.R * a.B;
    const int ntiles = (int)((nrows + TR - 1) / TR), nch = gridDim.x;
    const __bf16* hsrc = a.hext + (dir ? (size_t)2 * a.B * 2 * GH + GH : 0);        // row i of this view = h_prev of row i
    const __bf16* gsrc = a.dg + (size_t)dir * G4 + mh * DMH;                           // row i: + i * 2 * G4

    const int mg = w & 3, ng = w >> 2;
    f32x16 acc[2][6];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 6; ++nb)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[mb][nb][k] = 0.f;

    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)dwsm);
    // 512-B rows, two per instruction: lane = 32 (row in pair) + stored piece; rows 2 p + (lane >> 5) with p = w, w + 8: (row & 3) is a
    // per-lane constant.  256-B rows, four per instruction: lane = 16 (row in quad) + stored piece, rows 4 w + (lane >> 4).
    const int r2 = lane >> 5, sw2 = ((2 * (w & 1) + r2) & 3) << 2;
    const unsigned v_dg = r2 * (2 * G4 * 2) + (((lane & 31) ^ sw2) * 16);
    const unsigned v_x = r2 * (DXW * 2) + (((lane & 31) ^ sw2) * 16);
    const unsigned v_h = (lane >> 4) * (2 * GH * 2) + (((lane & 15) ^ ((lane >> 4) << 2)) * 16);
    auto request = [&](int tile, int slot) {
        const long r0 = min((long)tile * TR, nrows - TR);
        const unsigned base = lds0 + slot * WD_SLOT;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int row = 2 * (w + 8 * p);
            dma16(gsrc + (size_t)(r0 + row) * 2 * G4, v_dg, base + row * (DMH * 2));
            dma16(a.x + (size_t)(r0 + row) * DXW, v_x, base + WD_DG + WD_H + row * (DXW * 2));
        }
        dma16(hsrc + (size_t)(r0 + w * 4) * 2 * GH, v_h, base + WD_DG + w * 4 * (GH * 2));
    };

    // transposed reads: see lstm_dw_kernel.  Piece (16 B) index of columns 16 cb + 4 kp.. of 32-column block j: 4 j + 2 cb + (kp >> 1),
    // stored at index ^ (kq << 2) ((row & 3) == kq for both reads of a fragment)
    const int kq = (lane & 15) >> 2, kp = lane & 3, cb = (lane >> 4) & 1, hh = lane >> 5;
    const int rowoff = 8 * hh + kq;
    auto piece = [&](int j) { return (((4 * j + 2 * cb + (kp >> 1)) ^ (kq << 2)) * 16) + (kp & 1) * 8; };
    int pa_off[2], pb_off[6], pb_pitch[6];         // n-block j = 6 ng + i of [h (4 blocks) | x (8 blocks)]
    short pb_floor[6];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) pa_off[mb] = rowoff * (DMH * 2) + piece(2 * mg + mb);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int j = 6 * ng + i;
        pb_pitch[i] = j < 4 ? GH * 2 : DXW * 2;
        pb_floor[i] = (j >= 4 && a.x_relu) ? (short)0 : (short)-32768;
        pb_off[i] = j < 4 ? WD_DG + rowoff * (GH * 2) + piece(j) : WD_DG + WD_H + rowoff * (DXW * 2) + piece(j - 4);
    }
    auto frag = [&](const unsigned char* p, int pitch) {
        const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 4 * pitch);
        s16x8 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = lo[j]; f[4 + j] = hi[j]; }
        return __builtin_bit_cast(bf16x8, f);
    };

    int tile = blockIdx.x, slot = 0;
    if (tile < ntiles) request(tile, 0);
    if (tile + nch < ntiles) request(tile + nch, 1);
    for (; tile < ntiles; tile += nch) {
        if (tile + nch < ntiles) asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");       // 5 = LDS-DMA instructions per wave and tile
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (tile + 2 * nch < ntiles) request(tile + 2 * nch, slot == 0 ? 2 : slot - 1);
        unsigned char* base = dwsm + slot * WD_SLOT;
        if ((long)tile * TR + TR > nrows) {        // the shifted last tile: its first rows were summed by the previous tile already
            const int dup = (int)((long)tile * TR + TR - nrows);
            const int nh = dup * GH * 2 / 16, nx = dup * DXW * 2 / 16;
            for (int i = tid; i < nh + nx; i += 512) {
                unsigned char* dst = i < nh ? base + WD_DG + i * 16 : base + WD_DG + WD_H + (i - nh) * 16;
                *reinterpret_cast<uint4*>(dst) = make_uint4(0u, 0u, 0u, 0u);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
#ifdef DIC_DWW_EXP_NOMMA        // experiment (scripts/dww_experiments.sh): the DMA ring and barriers alone
        const int nks = a.R < 0 ? 1 : 0;
#else
        const int nks = TR / 16;
#endif
#pragma unroll 1
        for (int ks = 0; ks < nks; ++ks) {
            bf16x8 af[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) af[mb] = frag(base + ks * 16 * (DMH * 2) + pa_off[mb], DMH * 2);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                bf16x8 bfg = frag(base + ks * 16 * pb_pitch[i] + pb_off[i], pb_pitch[i]);
                {   // relu of the x columns (h columns: identity floor): max on the int16 halves, see pk_relu in dic_rowproj.hip
                    typedef short s16x8v __attribute__((ext_vector_type(8)));
                    const short fl = pb_floor[i];
                    const s16x8v flv = {fl, fl, fl, fl, fl, fl, fl, fl};
                    bfg = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8v, bfg), flv));
                }
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bfg, acc[mb][i], 0, 0, 0);
            }
        }
        slot = slot == NBUF - 1 ? 0 : slot + 1;
    }
    // C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* o = a.partials + ((size_t)blockIdx.x * 4 + blockIdx.y) * DMH * DNW;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int m = 64 * mg + 32 * mb + (k & 3) + 8 * (k >> 2) + 4 * hh;
                o[(size_t)m * DNW + 32 * (6 * ng + i) + (lane & 31)] = acc[mb][i][k];
            }
}

// One thread per FOUR consecutive outputs (16-B loads: a wave reads 1 KiB of each partial row), fixed-order f64 sums over the nch
// partial rows with eight loads in flight.  (Round 2 ran this through reduce_partials_32x8 -- 32 outputs x 8 slices per workgroup, made
// for a handful of outputs over thousands of rows: 100 MB of partials in 128-B pieces took 217 us per step, 0.46 TB/s.)
__global__ __launch_bounds__(256) void lstm_dw_wide_finalize(const float* partials, int nch, LstmGrads g, float beta) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= WD_OUT) return;
    double ch[8][4];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) ch[k][e] = 0.0;
    const float* src = partials + i;
    int b = 0;
    for (; b + 8 <= nch; b += 8) {
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float4*>(src + (size_t)(b + k) * WD_OUT);
#pragma unroll
        for (int k = 0; k < 8; ++k) { ch[k][0] += (double)v[k].x; ch[k][1] += (double)v[k].y; ch[k][2] += (double)v[k].z; ch[k][3] += (double)v[k].w; }
    }
    for (; b < nch; ++b) {
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)b * WD_OUT);
        ch[0][0] += (double)v.x; ch[0][1] += (double)v.y; ch[0][2] += (double)v.z; ch[0][3] += (double)v.w;
    }
    const int part = i / (DMH * DNW), rem = i - part * (DMH * DNW), m = (part & 1) * DMH + rem / DNW, n = rem % DNW, d = part >> 1;
    float* dst = n < GH ? g.w_hh[d] + (size_t)m * GH + n : g.w_ih[d] + (size_t)m * DXW + (n - GH);      // (n .. n+3 stay on one side: GH % 4 == 0)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const double s = ((ch[0][e] + ch[1][e]) + (ch[2][e] + ch[3][e])) + ((ch[4][e] + ch[5][e]) + (ch[6][e] + ch[7][e]));
        dst[e] = beta != 0.f ? fmaf(beta, dst[e], (float)s) : (float)s;
    }
}
static_assert(WD_OUT % 4 == 0 && DNW % 4 == 0 && GH % 4 == 0, "four outputs per thread");

static int dw_wide_chunks(int R, int B) {
    const int ntiles = (int)(((long)R * B + TR - 1) / TR);
    const int n = max(1, min(ntiles, kNumCU / 4));      // x 4 (direction, half) = one workgroup per CU
    return n >= 8 ? n / 8 * 8 : n;                      // a multiple of 8: the four workgroups of a chunk land on one XCD
}

// out_i = sum over workgroups of partial_i (fixed order, f64), scattered into the nn.LSTM gradients: h columns -> weight_hh,
// x columns [0, I) -> weight_ih (column I of the packed input is the constant one: its sum is the bias gradient, which the
// recurrence backward already delivers in f32).  beta = 0 overwrites, 1 accumulates.
__global__ __launch_bounds__(256) void lstm_dw_finalize(const float* partials, int nch, int I, int nw, LstmGrads g, float beta) {
    __shared__ double red[256];
    const int n_out = 2 * G4 * nw;
    const double s = reduce_partials_32x8(partials, nch, n_out, blockIdx.x * 32, red);
    const int i = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x >= 32 || i >= n_out) return;
    const int d = i / (G4 * nw), rem = i - d * (G4 * nw), m = rem / nw, n = rem - m * nw;
    float* dst;
    if (n < GH) dst = g.w_hh[d] + (size_t)m * GH + n;
    else if (n - GH < I) dst = g.w_ih[d] + (size_t)m * I + (n - GH);
    else return;
    *dst = beta != 0.f ? fmaf(beta, *dst, (float)s) : (float)s;
}

// Gradients that came out of library GEMMs / the recurrence backward as (2,4H,ldw) / (2,4H,H) / (2,4H) f32 staging tensors ->
// the eight parameter gradients (any of the three sources may be NULL).
__global__ __launch_bounds__(256) void lstm_unpack_grads_kernel(const float* dw_ih, int ldw, const float* dw_hh, const float* dbias,
                                                               int I, LstmGrads g, float beta) {
    const int n_ih = dw_ih ? 2 * G4 * I : 0, n_hh = dw_hh ? 2 * G4 * GH : 0, n_b = dbias ? 2 * G4 : 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n_ih + n_hh + n_b; i += gridDim.x * 256) {
        if (i < n_ih) {
            const int row = i / I, c = i - row * I, d = row / G4, m = row - d * G4;
            float* dst = g.w_ih[d] + (size_t)m * I + c;
            const float v = dw_ih[(size_t)row * ldw + c];
            *dst = beta != 0.f ? fmaf(beta, *dst, v) : v;
        } else if (i < n_ih + n_hh) {
            const int j = i - n_ih, d = j / (G4 * GH), k = j - d * (G4 * GH);
            float* dst = g.w_hh[d] + k;
            *dst = beta != 0.f ? fmaf(beta, *dst, dw_hh[j]) : dw_hh[j];
        } else {
            const int j = i - n_ih - n_hh, d = j / G4, m = j - d * G4;
            const float v = dbias[j];
            float* d1 = g.b_ih[d] + m;
            float* d2 = g.b_hh[d] + m;
            *d1 = beta != 0.f ? fmaf(beta, *d1, v) : v;
            *d2 = beta != 0.f ? fmaf(beta, *d2, v) : v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ x3: weight gradients from split planes (round 6)
// The f32 step on split products (DIC_DTYPE_F32X3): dW[gate row][h_prev | x] = sum over the (t, b) rows of dG^T [h_prev | x] with every product hi.hi + lo.hi +
// hi.lo.  Until round 5 this was dic_gemm_tn on f32 operands: 128 x 128 output tiles, both operands through registers, converted in the loop -- five launches,
// 4.9 ms of a 19 ms step at B = 32 768, each CU taking in its operands three to four times over (per-CU ingest bound).  Here, in the form of lstm_dw_kernel /
// lstm_dw_wide_kernel above:
//   * dG arrives as the two bf16 PLANES the x3 recurrence backward writes (hi | lo, dG = hi + lo) and goes from HBM into LDS by LDS-DMA, untouched, through a
//     3-slot ring of 16-row tiles (one MFMA k-step); every CU takes in its rows of dG ONCE;
//   * h_prev and x are f32 (the layer's outputs / inputs as every other consumer reads them): 1 + 1 (encoder) or 1 + 2 (decoder) 16-B loads per thread and
//     tile, hand-issued (asm: counted in the same vmcnt queue as the DMAs by THIS code, not drained by the compiler's), split into hi / lo on their way into a
//     2-slot pair of LDS images one tile ahead; the decoder's x is rectified there when the LSTM ran on relu(x);
//   * one raw barrier and one counted wait per tile; 30 (encoder: wave = 64 gate rows x 5 column blocks) or 36 (decoder: 64 x 6) MFMAs per wave and tile.
// Same partial-sum layouts as the bf16 kernels: their finalize kernels (fixed-order f64) write the parameter gradients.
constexpr int TX = 16;                                   // rows per tile
template <int XWT, int DMH> struct DwX3 {
    static constexpr int NG = G4 / DMH;                  // column groups of waves: 1 (a workgroup owns all 512 gate rows of a direction) or 2 (half of them)
    static constexpr int MW = DMH / 64;                  // waves along the gate rows
    static constexpr int NBT = 4 + XWT / 32;             // 32-column blocks of [h_prev (4) | x]: 5 (encoder) or 12 (decoder)
    static constexpr int NBN = (NBT + NG - 1) / NG;      // ... per wave: 3 (the second column group of the encoder has two: its third is idle) or 6
    static constexpr int NWT = GH + XWT;
    static constexpr int DGP = DMH == G4 ? 2 * DMH + 64 : 2 * DMH;        // dG image row pitch in bytes: 1088, or 512 with XOR-swizzled 16-B pieces
    static constexpr int L_PL = TX * DGP, L_DG = 2 * L_PL;                 // one plane / both planes of a tile
    static constexpr int HP = 2 * GH + 64;               // 320 B: pitch = 64 B mod 256 B -> the 4 rows of a transposed read fall on disjoint bank groups
    static constexpr int XP = XWT == 32 ? 64 : 2 * XWT + 64;               // 64-B rows are conflict-free as they lie; 576 B
    static constexpr int L_H = TX * HP, L_X = TX * XP;   // one image
    static constexpr int L_HX = 2 * L_H + 2 * L_X;       // hi + lo of both
    static constexpr int L_RED = XWT == 32 ? 2 * 8 * 1024 : 0;             // encoder: two parities x eight waves x one 16 x 16 f32 partial tile of the fused dX
    static constexpr int LDS = 3 * L_DG + 2 * L_HX + L_RED;
    static constexpr int NXL = XWT == 32 ? 1 : 2;        // 16-B loads of x per thread and tile
    static constexpr int NDMA = DMH == G4 ? 4 : 2;       // LDS-DMA instructions per wave and tile
};

struct DwX3Args {
    const __bf16* dg;      // planes: (2, R*B, 2*4H) -- hi at dg, lo dg_plane elements behind it
    long dg_plane;
    const float* hext;     // ((R+2)*B, 2H) f32: see DwArgs
    const float* x;        // (R*B, ldx) f32
    int ldx;               // row length of x in elements (a multiple of 4; <= XWT)
    float* partials;
    int R, B, x_relu;
    const float* wih;      // encoder only, or NULL: (2*4H, ldx) f32 packed input weights -> the input gradient rides along:
    float* dxp;            //   (4, R*B, ldx) f32 partial input gradients, one per (direction, half of its gate rows): their sum is dX = dG . W_ih
};

__device__ __forceinline__ f32x4_t gload16(const void* sbase, unsigned voff) {      // 16 B per lane, hand-issued: see the waits in the kernel
    f32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(v) : "v"(voff), "s"(sbase) : "memory");
    return v;
}
__device__ __forceinline__ void split4_store(unsigned char* hi, unsigned char* lo, f32x4_t v, bool relu, bool zero) {
    bf16x8 dummy; (void)dummy;
    typedef __bf16 b4 __attribute__((ext_vector_type(4)));
    b4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float x = zero ? 0.f : v[j];
        if (relu) x = fmaxf(x, 0.f);
        const __bf16 xh = (__bf16)x;
        h[j] = xh;
        l[j] = (__bf16)(x - (float)xh);
    }
    *reinterpret_cast<b4*>(hi) = h;
    *reinterpret_cast<b4*>(lo) = l;
}

#ifdef DIC_DWX3_EXP_TIMING      // experiment: per-phase cycle stamps of lane 0 of every wave of workgroup (5, 0), tiles 8..39 (scripts/dwx3_timing.py)
__device__ unsigned long long dic_dwx3_stamps[8][32][8];
#define DWX_STAMP(tile, slot)                                                                                         \
    do {                                                                                                              \
        if (blockIdx.x == 5 && blockIdx.y == 0 && (threadIdx.x & 63) == 0 && (tile) >= 8 && (tile) < 40)               \
            dic_dwx3_stamps[threadIdx.x >> 6][(tile) - 8][slot] = __builtin_readcyclecounter();                        \
    } while (0)
#else
#define DWX_STAMP(tile, slot) do {} while (0)
#endif

template <int XWT, int DMH>
__global__ __launch_bounds__(512) void lstm_dwx3_kernel(DwX3Args a) {
    typedef DwX3<XWT, DMH> C;
    extern __shared__ __align__(16) unsigned char dwsm[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = blockIdx.y / C::NG, mh = blockIdx.y % C::NG;
    const int mg = w % C::MW, ng = w / C::MW;
    const long nrows = (long)a.R * a.B;
    const int ntiles = (int)((nrows + TX - 1) / TX), nch = gridDim.x;
    const int mine = (int)blockIdx.x < ntiles ? (ntiles - 1 - (int)blockIdx.x) / nch + 1 : 0;
    if (mine == 0) return;                                               // (uniform: the whole workgroup)
    const float* hsrc = a.hext + (dir ? (size_t)2 * a.B * 2 * GH + GH : 0);                  // row i of this view = h_prev of row i
    const __bf16* gsrc = a.dg + (size_t)dir * G4 + mh * DMH;                                   // row i: + i * 2 * G4 (either plane)

    f32x16 acc[2][C::NBN];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < C::NBN; ++nb)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[mb][nb][k] = 0.f;

    unsigned char* const hx0 = dwsm + 3 * C::L_DG;                                            // the two image slots behind the three dG slots
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)dwsm);
    // tile j of this workgroup (clamped: requests past the end re-read the last tile into a slot nobody computes on -- every wave issues the same number
    // of memory instructions per iteration, which is what the counted waits count)
    auto tile_r0 = [&](int j) { return min((long)((int)blockIdx.x + min(j, mine - 1) * nch) * TX, nrows - TX); };
    // ---- dG planes by LDS-DMA.  1-KiB rows (DMH = 512): one row per instruction, rows w and w + 8 of either plane.  512-B rows: two per instruction (row
    // pair w), 16-B pieces XOR-swizzled by (row & 3) << 2 through the source address (as lstm_dw_wide_kernel)
    const int r2 = lane >> 5, sw2 = ((2 * (w & 1) + r2) & 3) << 2;
    const unsigned v_dg = DMH == G4 ? (unsigned)lane * 16 : (unsigned)(r2 * (2 * G4 * 2) + (((lane & 31) ^ sw2) * 16));
    auto request_dg = [&](int j) {
        const long r0 = tile_r0(j);
        const unsigned base = lds0 + (j % 3) * C::L_DG;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const __bf16* src = gsrc + (size_t)p * a.dg_plane;
            if constexpr (DMH == G4) {
#pragma unroll
                for (int k = 0; k < 2; ++k) dma16(src + (size_t)(r0 + w + 8 * k) * 2 * G4, v_dg, base + p * C::L_PL + (w + 8 * k) * C::DGP);
            } else {
                dma16(src + (size_t)(r0 + 2 * w) * 2 * G4, v_dg, base + p * C::L_PL + 2 * w * C::DGP);
            }
        }
    };
    // ---- h_prev / x through registers: thread -> (row, 16-B piece)
    const int hrow = tid >> 5, hc4 = tid & 31;
    const unsigned v_h = (unsigned)(hrow * (2 * GH * 4) + hc4 * 16);
    const int xpc = a.ldx >> 2;                                          // 16-B pieces per x row
    int xrow[C::NXL], xc4[C::NXL];
    unsigned v_x[C::NXL];
    bool xon[C::NXL];
#pragma unroll
    for (int i = 0; i < C::NXL; ++i) {
        const int p = tid + 512 * i;
        xon[i] = p < TX * xpc;
        const int pc = xon[i] ? p : 0;
        xrow[i] = pc / xpc; xc4[i] = pc - xrow[i] * xpc;
        v_x[i] = (unsigned)((xrow[i] * a.ldx + 4 * xc4[i]) * 4);
    }
    // (the LDS positions of a thread's pieces: row * pitch + 8 piece, and row < dup <=> position < dup * pitch -- the rows and pieces themselves need not stay)
    const int h_pos = hrow * C::HP + hc4 * 8;
    int x_pos[C::NXL];
#pragma unroll
    for (int i = 0; i < C::NXL; ++i) x_pos[i] = xrow[i] * C::XP + xc4[i] * 8;
    f32x4_t hreg, xreg[C::NXL];
    auto request_hx = [&](int j) {
        const long r0 = tile_r0(j);
        hreg = gload16(hsrc + (size_t)r0 * 2 * GH, v_h);
#pragma unroll
        for (int i = 0; i < C::NXL; ++i) xreg[i] = gload16(a.x + (size_t)r0 * a.ldx, v_x[i]);
    };
    // split -> the images of slot j & 1, in 1 + NXL pieces (h, then the x registers) so that the main loop can put one piece behind each of its first MFMA groups;
    // rows the shifted last tile shares with its predecessor: zeros
    constexpr int NPIECE = 1 + C::NXL;
    static_assert(NPIECE <= C::NBN, "dwx3: one MFMA group per piece");
    auto store_piece = [&](int j, int pc) {
        const long t0 = (long)((int)blockIdx.x + min(j, mine - 1) * nch) * TX;
        const int dup = (int)max(0L, t0 + TX - nrows);
        unsigned char* base = hx0 + (j & 1) * C::L_HX;
        if (C::NBN >= 6 && dup == 0) {           // (uniform; every tile but a shifted last one: no zero selects -- the wide form only: the branch cost the narrow one 4 %)
            if (pc == 0) split4_store(base + h_pos, base + C::L_H + h_pos, hreg, false, false);
            else if (xon[pc - 1]) split4_store(base + 2 * C::L_H + x_pos[pc - 1], base + 2 * C::L_H + C::L_X + x_pos[pc - 1], xreg[pc - 1], a.x_relu != 0, false);
        } else if (pc == 0) {
            split4_store(base + h_pos, base + C::L_H + h_pos, hreg, false, h_pos < dup * C::HP);
        } else {
            const int i = pc - 1;
            if (xon[i]) split4_store(base + 2 * C::L_H + x_pos[i], base + 2 * C::L_H + C::L_X + x_pos[i], xreg[i], a.x_relu != 0, x_pos[i] < dup * C::XP);
        }
    };
    auto store_hx = [&](int j) {
#pragma unroll
        for (int pc = 0; pc < NPIECE; ++pc) store_piece(j, pc);
    };

    // ---- transposed-read addressing (ds_read_b64_tr_b16; see lstm_dw_kernel)
    const int kq = (lane & 15) >> 2, kp = lane & 3, cb = (lane >> 4) & 1, hh = lane >> 5;
    const int rowoff = 8 * hh + kq;
    auto piece = [&](int j) { return (((4 * j + 2 * cb + (kp >> 1)) ^ (kq << 2)) * 16) + (kp & 1) * 8; };
    int pa_off[2], pb_pitch[C::NBN], pb_lo[C::NBN];
    int pb_sel[C::NBN], pb_soff[C::NBN];                                 // (scalar) column block i of this wave: in the h image or the x image, and its byte offset there
    const int pb_h = rowoff * C::HP + (16 * cb + 4 * kp) * 2, pb_x = 2 * C::L_H + rowoff * C::XP + (16 * cb + 4 * kp) * 2;      // (per lane: two registers, not one per block)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
        pa_off[mb] = DMH == G4 ? rowoff * C::DGP + (64 * mg + 32 * mb + 16 * cb + 4 * kp) * 2 : rowoff * C::DGP + piece(2 * mg + mb);
#pragma unroll
    for (int i = 0; i < C::NBN; ++i) {
        const int jn = min(C::NBN * ng + i, C::NBT - 1);                 // column block of [h (4 blocks) | x] (an idle slot repeats the last one: never stored)
        pb_pitch[i] = jn < 4 ? C::HP : C::XP;
        pb_lo[i] = jn < 4 ? C::L_H : C::L_X;
        pb_sel[i] = jn < 4;
        pb_soff[i] = 64 * (jn < 4 ? jn : jn - 4);
    }
    auto pb_off = [&](int i) { return (pb_sel[i] ? pb_h : pb_x) + pb_soff[i]; };
    auto frag = [&](const unsigned char* p, int pitch) {
        const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 4 * pitch);
        s16x8 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = lo[j]; f[4 + j] = hi[j]; }
        return __builtin_bit_cast(bf16x8, f);
    };

    // ---- encoder: the input gradient dX[row][input column] = sum over the gate columns of dG[row][k] W_ih[k][column] from the SAME dG tiles (the separate
    // product re-read all of dG: 0.78 ms at B = 32 768).  16x16x32 MFMAs: wave (mg, ng) multiplies the tile's 16 rows by gate columns [64 mg, 64 mg + 64) of this
    // workgroup's half direction for the 16 input columns of block ng -- W_ih^T hi / lo for those, split once at start-up: 16 registers -- and parks its 16 x 16
    // partial tile in LDS; one barrier later (the next tile's) thread (row, column) adds the four partials of its column block in a fixed order and stores the
    // f32 result into THIS workgroup's slice of dx_parts (direction x half: four partial tensors, summed by the caller).
    constexpr bool DXOK = XWT == 32 && C::NG == 2;
    const bool want_dx = DXOK && a.wih != nullptr;
    float* red = reinterpret_cast<float*>(dwsm + 3 * C::L_DG + 2 * C::L_HX);
    bf16x8 wdh[2], wdl[2];                                               // [k-step]
    const int li = lane & 15, kg4 = lane >> 4;
    if (want_dx) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int col = 16 * ng + li, k = mh * DMH + 64 * mg + 32 * ks + 8 * kg4 + jj;
                const float v = col < a.ldx ? a.wih[((size_t)dir * G4 + k) * a.ldx + col] : 0.f;
                const __bf16 vh = (__bf16)v;
                wdh[ks][jj] = vh;
                wdl[ks][jj] = (__bf16)(v - (float)vh);
            }
    }
    // A fragment: row li of the tile, gate columns 64 mg + 32 ks + 8 kg4 .. + 7 of the (swizzled) half-direction image: 16-B piece 8 mg + 4 ks + kg4, stored at
    // piece ^ ((row & 3) << 2)
    int dxa_off[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) dxa_off[ks] = li * C::DGP + (((8 * mg + 4 * ks + kg4) ^ ((li & 3) << 2)) * 16);
    auto finish_dx = [&](int jprev) {                                    // thread (row = tid >> 5, column = tid & 31) of tile jprev: its partials were parked before the barrier
        if (!want_dx || jprev < 0) return;
        const int row = tid >> 5, col = tid & 31;
        const float* src = red + (jprev & 1) * (8 * 256) + (col >> 4) * (4 * 256) + ((col & 15) + 16 * (row >> 2)) * 4 + (row & 3);
        const float sum = (src[0] + src[256]) + (src[512] + src[768]);   // the four gate-column slices (mg) of column group col >> 4
        if (col < a.ldx) a.dxp[((size_t)blockIdx.y * nrows + tile_r0(jprev) + row) * a.ldx + col] = sum;
    };

    // ---- prologue: the x images' padding columns are zero for good; tile 0 in, tile 1 requested
    if constexpr (XWT == 32) {
        for (int i = tid; i < 2 * C::L_HX / 16; i += 512) reinterpret_cast<uint4*>(hx0)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }
    request_hx(0);
    request_dg(0);
    asm volatile("s_waitcnt vmcnt(%0)" :: "i"(C::NDMA) : "memory");     // h / x of tile 0 have landed (its DMAs may still fly)
    asm volatile("" : "+v"(hreg));
#pragma unroll
    for (int i = 0; i < C::NXL; ++i) asm volatile("" : "+v"(xreg[i]));
    store_hx(0);
    request_hx(1);
    request_dg(1);
    for (int j = 0; j < mine; ++j) {
        DWX_STAMP(j, 0);
        // queue, oldest first: [DMA(j)] h / x (j + 1), DMA(j + 1): everything up to the h / x loads has landed once only the last tile's DMAs are outstanding
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "i"(C::NDMA) : "memory");
        DWX_STAMP(j, 1);
        asm volatile("" : "+v"(hreg));
#pragma unroll
        for (int i = 0; i < C::NXL; ++i) asm volatile("" : "+v"(xreg[i]));
        __builtin_amdgcn_s_barrier();              // every wave's DMA of tile j has landed and its image writes are visible; all are done with tile j - 1
        DWX_STAMP(j, 2);
        if constexpr (DXOK) finish_dx(j - 1);
        // the narrow form (three MFMA groups per tile and wave, the partial dX product behind them) splits and requests up front: spreading its two pieces over
        // the groups delayed its requests by most of the tile's matrix work -- 1 361 against 1 292 us
        constexpr bool SPREAD = C::NBN >= 6;
        if constexpr (!SPREAD) {
            store_hx(j + 1);                       // -> the image slot tile j - 1 was read from
            request_hx(j + 2);
            request_dg(j + 2);                     // -> the dG slot tile j - 1 was read from
        }
        DWX_STAMP(j, 3);
        DWX_STAMP(j, 4);
        const unsigned char* dgb = dwsm + (j % 3) * C::L_DG;
        const unsigned char* hxb = hx0 + (j & 1) * C::L_HX;
        bf16x8 ah[2], al[2];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            ah[mb] = frag(dgb + pa_off[mb], C::DGP);
            al[mb] = frag(dgb + C::L_PL + pa_off[mb], C::DGP);
        }
        if constexpr (!SPREAD) {
            // (the narrow form keeps the plain loop: three column blocks do not repay the ordering fences of the pipelined one below -- 1 305 against 1 270 us)
#pragma unroll
            for (int i = 0; i < C::NBN; ++i) {
                const bf16x8 bh = frag(hxb + pb_off(i), pb_pitch[i]);
                const bf16x8 bl = frag(hxb + pb_off(i) + pb_lo[i], pb_pitch[i]);
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bh, acc[mb][i], 0, 0, 0);
                    acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mb], bh, acc[mb][i], 0, 0, 0);
                    acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bl, acc[mb][i], 0, 0, 0);
                }
            }
        } else {
            // the [h | x] fragments of column block i + 1 are requested UNDER the MFMAs of block i, into the registers those MFMAs have just read (there is no room
            // for a second set at 255 registers): hi (i + 1) behind the four products on hi (i), lo (i + 1) behind the two on lo (i).  As first written -- both
            // fragments read, waited for, then six MFMAs -- every block exposed an LDS round trip: the matrix cores were 52-56 % busy (rocprofv3).
            bf16x8 bh = frag(hxb + pb_off(0), pb_pitch[0]);
            bf16x8 bl = frag(hxb + pb_off(0) + pb_lo[0], pb_pitch[0]);
    #pragma unroll
            for (int i = 0; i < C::NBN; ++i) {
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int mb = 0; mb < 2; ++mb) acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bh, acc[mb][i], 0, 0, 0);
    #pragma unroll
                for (int mb = 0; mb < 2; ++mb) acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mb], bh, acc[mb][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i + 1 < C::NBN) bh = frag(hxb + pb_off(i + 1), pb_pitch[i + 1]);
                // the NEXT tile's h / x registers -> split -> the other image slot (the one tile j - 1 was read from), one piece per MFMA group: vector and LDS-write
                // work in the shadow of the four MFMAs just issued (cycle stamps, scripts/dwx3_timing.py: done in one go behind the barrier it cost the wave that
                // issues second 1 170 of its 4 430 cycles per tile; this way the launch takes 2 053 instead of 2 209 us on the same box -- the two waves of a SIMD still
                // convert at the same time, so most of that vector work stays exposed); the requests for tile j + 2 reuse those registers and follow the last piece
                // (Measured and not kept: the two waves of a SIMD converting at OPPOSITE ends of the loop, so that one's vector work meets the other's MFMAs -- the
                // late wave then requests tile j + 2 five MFMA groups later and waits 700 cycles for it at the top of the next tile: 2 297 against 2 053 us.)
                if constexpr (SPREAD) {
                    if (i < NPIECE) store_piece(j + 1, i);
                    if (i == NPIECE - 1) {
                        request_hx(j + 2);
                        request_dg(j + 2);             // -> the dG slot tile j - 1 was read from
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                for (int mb = 0; mb < 2; ++mb) acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bl, acc[mb][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i + 1 < C::NBN) bl = frag(hxb + pb_off(i + 1) + pb_lo[i + 1], pb_pitch[i + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        DWX_STAMP(j, 5);
        if constexpr (DXOK) {
            if (want_dx) {
                f32x4_t d0 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 avh = *reinterpret_cast<const bf16x8*>(dgb + dxa_off[ks]);
                    const bf16x8 avl = *reinterpret_cast<const bf16x8*>(dgb + C::L_PL + dxa_off[ks]);
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(avh, wdh[ks], d0, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(avl, wdh[ks], d0, 0, 0, 0);
                    d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(avh, wdl[ks], d0, 0, 0, 0);
                }
                *reinterpret_cast<f32x4_t*>(red + (j & 1) * (8 * 256) + (ng * 4 + mg) * 256 + lane * 4) = d0;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // (the clamped requests past the end)
    if constexpr (DXOK) {
        if (want_dx) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            finish_dx(mine - 1);
        }
    }
    // C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* o = a.partials + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * DMH * C::NWT;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int i = 0; i < C::NBN; ++i) {
            if (C::NBN * ng + i >= C::NBT) continue;                      // (the encoder's idle slot)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int m = 64 * mg + 32 * mb + (k & 3) + 8 * (k >> 2) + 4 * hh;
                o[(size_t)m * C::NWT + 32 * (C::NBN * ng + i) + (lane & 31)] = acc[mb][i][k];
            }
        }
}

static int dw_chunks(int R, int B) {
    const int ntiles = (int)(((long)R * B + TR - 1) / TR);
    return max(1, min(ntiles, kNumCU / 2));        // x 2 directions = one workgroup per CU
}
static int dwx3_chunks(int R, int B, bool wide) {
    const int ntiles = (int)(((long)R * B + TX - 1) / TX);
    (void)wide;
    const int n = max(1, min(ntiles, kNumCU / 4));      // x 4 (direction, half) = one workgroup per CU
    return n >= 8 ? n / 8 * 8 : n;                      // a multiple of 8: the four workgroups of a chunk land on one XCD (they share the h / x rows through its L2)
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_lstm_pack(int dtype, const float* const* params, int H, int I, int Ip, int bias_col, void* wih, void* whh, void* whh_t, void* bias,
                  void* wih_t, dic_stream_t stream) {
    DIC_REQUIRE(dtype == DIC_DTYPE_F32 || dtype == DIC_DTYPE_BF16, DIC_ERR_INVALID_ARG, "lstm_pack: dtype %d", dtype);
    DIC_REQUIRE(H == GH, DIC_ERR_UNSUPPORTED, "lstm_pack: hidden size %d (compiled for %d)", H, GH);
    DIC_REQUIRE(params && wih && whh, DIC_ERR_INVALID_ARG, "lstm_pack: NULL pointer");
    DIC_REQUIRE(I > 0 && Ip >= I + (bias_col ? 1 : 0) && Ip <= 1024, DIC_ERR_INVALID_ARG, "lstm_pack: I=%d Ip=%d bias_col=%d", I, Ip, bias_col);
    LstmParams p;
    for (int d = 0; d < 2; ++d) {
        p.w_ih[d] = params[4 * d + 0]; p.w_hh[d] = params[4 * d + 1]; p.b_ih[d] = params[4 * d + 2]; p.b_hh[d] = params[4 * d + 3];
        DIC_REQUIRE(p.w_ih[d] && p.w_hh[d] && p.b_ih[d] && p.b_hh[d], DIC_ERR_INVALID_ARG, "lstm_pack: NULL parameter (direction %d)", d);
    }
    const int total = 2 * G4 * Ip + 4 * G4 * GH + 2 * G4 + (wih_t ? 2 * G4 * Ip : 0);
    const dim3 grid(min((total + 255) / 256, 2 * kNumCU));
    if (dtype == DIC_DTYPE_BF16)
        hipLaunchKernelGGL(lstm_pack_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, p, I, Ip, bias_col, (__bf16*)wih, (__bf16*)whh,
                           (__bf16*)whh_t, (__bf16*)bias, (__bf16*)wih_t);
    else
        hipLaunchKernelGGL(lstm_pack_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, p, I, Ip, bias_col, (float*)wih, (float*)whh,
                           (float*)whh_t, (float*)bias, (float*)wih_t);
    return check_launch("lstm_pack");
}

static int grads_from(float* const* grads, LstmGrads* g, bool need_w, bool need_b, const char* who) {
    DIC_REQUIRE(grads, DIC_ERR_INVALID_ARG, "%s: grads is NULL", who);
    for (int d = 0; d < 2; ++d) {
        g->w_ih[d] = grads[4 * d + 0]; g->w_hh[d] = grads[4 * d + 1]; g->b_ih[d] = grads[4 * d + 2]; g->b_hh[d] = grads[4 * d + 3];
        DIC_REQUIRE(!need_w || (g->w_ih[d] && g->w_hh[d]), DIC_ERR_INVALID_ARG, "%s: NULL weight gradient (direction %d)", who, d);
        DIC_REQUIRE(!need_b || (g->b_ih[d] && g->b_hh[d]), DIC_ERR_INVALID_ARG, "%s: NULL bias gradient (direction %d)", who, d);
    }
    return DIC_OK;
}

size_t dic_lstm_dw_workspace(int R, int B) {        // (sized for the 64-wide packed rows: 2 x 4H x (H + 64) sums per workgroup)
    if (R <= 0 || B <= 0) return 0;
    return (size_t)dw_chunks(R, B) * 2 * G4 * (GH + 64) * sizeof(float);
}

int dic_lstm_dw(const void* dgx, const void* out_ext, const void* x, const void* wih, void* dx_parts, int R, int B, int H, int I, int Ip,
                float* const* grads, int accumulate, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(R > 0 && B > 0, DIC_ERR_INVALID_ARG, "lstm_dw: non-positive size");
    DIC_REQUIRE(H == GH, DIC_ERR_UNSUPPORTED, "lstm_dw: hidden size %d (compiled for %d)", H, GH);
    DIC_REQUIRE((Ip == XW || Ip == 64) && I > 0 && I <= Ip, DIC_ERR_UNSUPPORTED, "lstm_dw: packed input width %d / %d (compiled for %d and 64)", I, Ip, XW);
    DIC_REQUIRE(dgx && out_ext && x && workspace, DIC_ERR_INVALID_ARG, "lstm_dw: NULL pointer");
    DIC_REQUIRE((long)R * B >= TR, DIC_ERR_UNSUPPORTED, "lstm_dw: R*B = %ld rows < one %d-row tile (use the GEMM path)", (long)R * B, TR);
    LstmGrads g;
    int rc = grads_from(grads, &g, true, false, "lstm_dw");
    if (rc) return rc;
    const int nch = dw_chunks(R, B);
    const int wide = Ip == 64, nw = GH + Ip, n_out = 2 * G4 * nw;
    DIC_REQUIRE(workspace_bytes >= (size_t)nch * n_out * sizeof(float), DIC_ERR_WORKSPACE, "lstm_dw: workspace %zu < %zu", workspace_bytes,
                (size_t)nch * n_out * sizeof(float));
    const int lds = wide ? NBUF * (LDS_DG + LDS_H + TR * 2 * 64) : DW_LDS;
    static bool attr_set[2] = {false, false};
    if (!attr_set[wide]) {
        hipError_t e = hipFuncSetAttribute(wide ? (const void*)lstm_dw_kernel<64> : (const void*)lstm_dw_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_dw: cannot reserve %d B of LDS: %s", lds, hipGetErrorString(e));
        attr_set[wide] = true;
    }
    hipStream_t st = (hipStream_t)stream;
    DIC_REQUIRE((wih == nullptr) == (dx_parts == nullptr), DIC_ERR_INVALID_ARG, "lstm_dw: wih and dx_parts go together");
    DIC_REQUIRE(!wih || (!wide && I <= WT_COLS), DIC_ERR_UNSUPPORTED, "lstm_dw: the fused input gradient covers %d input columns of 32-wide rows, got %d of %d", WT_COLS, I, Ip);
    DwArgs a{(const __bf16*)dgx, (const __bf16*)out_ext, (const __bf16*)x, (float*)workspace, (const __bf16*)wih, (__bf16*)dx_parts, R, B};
    if (wide) hipLaunchKernelGGL(lstm_dw_kernel<64>, dim3(nch, 2), dim3(512), lds, st, a);
    else hipLaunchKernelGGL(lstm_dw_kernel<32>, dim3(nch, 2), dim3(512), lds, st, a);
    hipLaunchKernelGGL(lstm_dw_finalize, dim3((n_out + 31) / 32), dim3(256), 0, st, (const float*)workspace, nch, I, nw, g,
                       accumulate ? 1.0f : 0.0f);
    return check_launch("lstm_dw");
}

size_t dic_lstm_dw_wide_workspace(int R, int B) {
    if (R <= 0 || B <= 0) return 0;
    return (size_t)dw_wide_chunks(R, B) * WD_OUT * sizeof(float);
}

int dic_lstm_dw_wide(const void* dgx, const void* out_ext, const void* x, int x_relu, int R, int B, int H, int I, float* const* grads, int accumulate,
                     void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(R > 0 && B > 0, DIC_ERR_INVALID_ARG, "lstm_dw_wide: non-positive size");
    DIC_REQUIRE(H == GH, DIC_ERR_UNSUPPORTED, "lstm_dw_wide: hidden size %d (compiled for %d)", H, GH);
    DIC_REQUIRE(I == DXW, DIC_ERR_UNSUPPORTED, "lstm_dw_wide: input width %d (compiled for %d)", I, DXW);
    DIC_REQUIRE(dgx && out_ext && x && workspace, DIC_ERR_INVALID_ARG, "lstm_dw_wide: NULL pointer");
    DIC_REQUIRE((long)R * B >= TR, DIC_ERR_UNSUPPORTED, "lstm_dw_wide: R*B = %ld rows < one %d-row tile (use the GEMM path)", (long)R * B, TR);
    LstmGrads g;
    int rc = grads_from(grads, &g, true, false, "lstm_dw_wide");
    if (rc) return rc;
    const int nch = dw_wide_chunks(R, B);
    DIC_REQUIRE(workspace_bytes >= (size_t)nch * WD_OUT * sizeof(float), DIC_ERR_WORKSPACE, "lstm_dw_wide: workspace %zu < %zu", workspace_bytes,
                (size_t)nch * WD_OUT * sizeof(float));
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)lstm_dw_wide_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WD_LDS);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_dw_wide: cannot reserve %d B of LDS: %s", WD_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    DwWideArgs a{(const __bf16*)dgx, (const __bf16*)out_ext, (const __bf16*)x, (float*)workspace, R, B, x_relu != 0};
    hipLaunchKernelGGL(lstm_dw_wide_kernel, dim3(nch, 4), dim3(512), WD_LDS, st, a);
    hipLaunchKernelGGL(lstm_dw_wide_finalize, dim3((WD_OUT / 4 + 255) / 256), dim3(256), 0, st, (const float*)workspace, nch, g, accumulate ? 1.0f : 0.0f);
    return check_launch("lstm_dw_wide");
}

size_t dic_lstm_dw_x3_workspace(int R, int B, int I) {
    if (R <= 0 || B <= 0 || I <= 0) return 0;
    const bool wide = I == DXW;
    return (size_t)dwx3_chunks(R, B, wide) * (wide ? (size_t)WD_OUT : (size_t)2 * G4 * (GH + XW)) * sizeof(float);
}

int dic_lstm_dw_x3(const void* dg_hi, long dg_plane, const float* out_ext, const float* x, int ldx, int x_relu, const float* wih, float* dx_parts, int R, int B, int H,
                   int I, float* const* grads, int accumulate, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(R > 0 && B > 0, DIC_ERR_INVALID_ARG, "lstm_dw_x3: non-positive size");
    DIC_REQUIRE(H == GH, DIC_ERR_UNSUPPORTED, "lstm_dw_x3: hidden size %d (compiled for %d)", H, GH);
    const bool wide = I == DXW;
    DIC_REQUIRE(wide ? ldx == DXW : (I > 0 && I <= ldx && ldx <= XW && ldx % 4 == 0), DIC_ERR_UNSUPPORTED,
                "lstm_dw_x3: input width %d in rows of %d (compiled for %d, or up to %d in rows that are a multiple of 4)", I, ldx, DXW, XW);
    DIC_REQUIRE(dg_hi && dg_plane > 0 && out_ext && x && workspace, DIC_ERR_INVALID_ARG, "lstm_dw_x3: NULL pointer / plane stride");
    DIC_REQUIRE(((uintptr_t)dg_hi & 15) == 0 && dg_plane % 8 == 0 && ((uintptr_t)out_ext & 15) == 0 && ((uintptr_t)x & 15) == 0, DIC_ERR_UNSUPPORTED,
                "lstm_dw_x3: operands must be 16-B aligned");
    DIC_REQUIRE((long)R * B >= TX, DIC_ERR_UNSUPPORTED, "lstm_dw_x3: R*B = %ld rows < one %d-row tile (use the GEMM path)", (long)R * B, TX);
    DIC_REQUIRE((wih == nullptr) == (dx_parts == nullptr) && (!wih || !wide), DIC_ERR_INVALID_ARG, "lstm_dw_x3: wih and dx_parts go together (narrow input only)");
    LstmGrads g;
    int rc = grads_from(grads, &g, true, false, "lstm_dw_x3");
    if (rc) return rc;
    const int nch = dwx3_chunks(R, B, wide);
    DIC_REQUIRE(workspace_bytes >= dic_lstm_dw_x3_workspace(R, B, I), DIC_ERR_WORKSPACE, "lstm_dw_x3: workspace %zu < %zu", workspace_bytes,
                dic_lstm_dw_x3_workspace(R, B, I));
    hipStream_t st = (hipStream_t)stream;
    DwX3Args a{(const __bf16*)dg_hi, dg_plane, out_ext, x, ldx, (float*)workspace, R, B, x_relu != 0, wih, dx_parts};
    static bool attr_set[2] = {false, false};
    const int lds = wide ? DwX3<DXW, G4 / 2>::LDS : DwX3<XW, G4 / 2>::LDS;
    if (!attr_set[wide]) {
        hipError_t e = hipFuncSetAttribute(wide ? (const void*)lstm_dwx3_kernel<DXW, G4 / 2> : (const void*)lstm_dwx3_kernel<XW, G4 / 2>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_dw_x3: cannot reserve %d B of LDS: %s", lds, hipGetErrorString(e));
        attr_set[wide] = true;
    }
    if (wide) {
        hipLaunchKernelGGL((lstm_dwx3_kernel<DXW, G4 / 2>), dim3(nch, 4), dim3(512), lds, st, a);
        hipLaunchKernelGGL(lstm_dw_wide_finalize, dim3((WD_OUT / 4 + 255) / 256), dim3(256), 0, st, (const float*)workspace, nch, g, accumulate ? 1.0f : 0.0f);
    } else {
        const int nw = GH + XW, n_out = 2 * G4 * nw;
        hipLaunchKernelGGL((lstm_dwx3_kernel<XW, G4 / 2>), dim3(nch, 4), dim3(512), lds, st, a);
        hipLaunchKernelGGL(lstm_dw_finalize, dim3((n_out + 31) / 32), dim3(256), 0, st, (const float*)workspace, nch, I, nw, g, accumulate ? 1.0f : 0.0f);
    }
    return check_launch("lstm_dw_x3");
}

int dic_lstm_unpack_grads(const float* dw_ih, int ldw, const float* dw_hh, const float* dbias, int H, int I, float* const* grads,
                          int accumulate, dic_stream_t stream) {
    DIC_REQUIRE(H == GH, DIC_ERR_UNSUPPORTED, "lstm_unpack_grads: hidden size %d (compiled for %d)", H, GH);
    DIC_REQUIRE(I > 0 && (!dw_ih || ldw >= I), DIC_ERR_INVALID_ARG, "lstm_unpack_grads: I=%d ldw=%d", I, ldw);
    LstmGrads g;
    int rc = grads_from(grads, &g, dw_ih || dw_hh, dbias != nullptr, "lstm_unpack_grads");
    if (rc) return rc;
    const int total = (dw_ih ? 2 * G4 * I : 0) + (dw_hh ? 2 * G4 * GH : 0) + (dbias ? 2 * G4 : 0);
    if (total == 0) return DIC_OK;
    hipLaunchKernelGGL(lstm_unpack_grads_kernel, dim3(min((total + 255) / 256, 2 * kNumCU)), dim3(256), 0, (hipStream_t)stream, dw_ih, ldw,
                       dw_hh, dbias, I, g, accumulate ? 1.0f : 0.0f);
    return check_launch("lstm_unpack_grads");
}

#ifdef DIC_DWX3_EXP_TIMING
int dic_dwx3_debug_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dic_dwx3_stamps), sizeof(unsigned long long) * 8 * 32 * 8);
}
#endif

}  // extern "C"
