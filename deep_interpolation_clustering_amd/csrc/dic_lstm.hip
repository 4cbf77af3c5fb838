// Persistent recurrence kernels of the bidirectional LSTM (hidden size 128) that sits between the
// interpolation and de-interpolation kernels (clustering_interp.py:14-41: EncoderRNN / DecoderRNN,
// torch.nn.LSTM semantics, gate order i,f,g,o).
//
// Why: rocprofv3 of the joint step (profiles/r1_step_*_kernel_stats.csv) shows MIOpen's LSTM -- ~100 small
// GEMM + point-wise launches per direction-step, a 2.5 ms generic bias-gradient reduce, sub-tensor copies --
// taking ~90 % of the step; the north-star's "MFMA only if the dense layers prove the bottleneck" clause.
//
// Split of the work: the time-parallel GEMMs (input projection X.W_ih^T for all steps, dX, dW_ih, dW_hh,
// bias gradient) stay large hipBLASLt GEMMs issued from Python; these kernels do the SEQUENTIAL part only:
//   forward :  G_t = gx_t + h_{t-1}.W_hh^T ; i,f,o = sigmoid, g = tanh ; c_t = f c_{t-1} + i g ; h_t = o tanh(c_t)
//   backward:  gate gradients from (dh_t, dc_t), dh_{t-1} = dG_t.W_hh, dc_{t-1} = dc_t f_t
// One 256-thread workgroup owns 32*DIC_LSTM_NB batch rows of one direction for the whole sequence (h, c, dh, dc never
// leave the chip).  The MFMA is issued TRANSPOSED -- D[gate col][batch] = W_hh[gate col][k] . h^T[k][batch]
// (v_mfma_f32_32x32x16_bf16) -- so that wave w's A operand is the slice of W_hh for hidden units
// [32w,32w+32): 32 fragments = 128 VGPRs loaded ONCE and kept for all R steps (weights never re-read), the
// B operand is the 64x128 bf16 h tile in LDS (272-B padded rows: conflict-free ds_read_b128), and the four
// gate accumulators of a hidden unit land in the same lane/register, so the gate math is register-local.
// bf16 operands, f32 accumulation, f32 cell state.
#include "dic_common.h"

namespace dic {

constexpr int LH = 128;            // hidden size (compiled in)
#ifndef DIC_LSTM_NB
#define DIC_LSTM_NB 2              // 32-wide batch tiles per workgroup (1 -> 2 workgroups/CU measured slower: 0.77/1.33 ms vs
#endif                             // 0.74/1.06 ms fwd/bwd at B=16384)
constexpr int LNB = DIC_LSTM_NB;
constexpr int LBM = 32 * LNB;      // batch rows per workgroup
constexpr int HSTR = LH + 8;       // bf16 elements per LDS row of the h tile (272 B)
constexpr int GSTR = 4 * LH + 8;   // bf16 elements per LDS row of the dG tile (1040 B)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int V> struct IC { static constexpr int value = V; };      // compile-time index passed through a generic lambda

// (DIC_NT_LOAD / DIC_NT_STORE, dic_common.h: same-box A/B of lstm_bwd8 at B = 32 768 with its saved-state loads and 16-B dG row stores
// nontemporal: 794 -> 763 us; its c_prev and dL/dout loads too: another -1.5 %.  The ENCODER forward (98 % writes) gains 3 % from nontemporal
// saved-state stores (538 -> 522 us, three alternating same-box runs); the decoder forward does not (772 vs 777 us) and keeps plain stores.
// Tried and dropped elsewhere -- the forward kernels' `out` stores, 16 B of each of 32 rows per instruction, i.e. partial lines: TWICE as
// long; the decoder forward's gx LDS-DMA: +6 %; row_proj's x loads (each tile is read by four column stripes): +15 %; fc_bwd, the
// weight-gradient kernels' DMA: nothing.)


#ifdef DIC_LSTM_EXP_NOMATH      // experiment: gate non-linearities replaced by one FMA each (timing only, wrong results)
__device__ __forceinline__ float sigmoid_fast(float x) { return fmaf(x, 0.01f, 0.5f); }
__device__ __forceinline__ float tanh_fast(float x) { return x * 0.01f; }
#else
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp2(-kLog2e * x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + fast_exp2(2.0f * kLog2e * x)); }
#endif

// "Lane-native" layout of the tensors only these two kernels exchange (saved gates, cell states): element
// (t, batch b, dir, gate g, unit u) lives at
//     ((((((t*NBT + b/32)*2 + dir)*4 + u/32)*G + g)*4 + (u%32)/8)*2 + (u%8)/4)*32 + b%32)*4 + u%4
// so the 8-byte (bf16x4) / 16-byte (f32x4) access of every lane of a wave instruction is contiguous with its
// neighbours': one 512-B / 1-KiB transaction instead of 32 scattered 16-B pieces (the MFMA accumulator
// layout puts batch rows on lanes, hidden units on registers).  B is padded to a multiple of 32.
__device__ __forceinline__ size_t native_off(int t, int nbt, int bt, int dir, int w, int G, int g, int q, int hh, int r) {
    size_t o = (size_t)t * nbt + bt;
    o = (o * 2 + dir) * 4 + w;
    o = (o * G + g) * 4 + q;
    o = (o * 2 + hh) * 32 + r;
    return o * 4;
}

// (h0, c0, hn, cn and their gradients) element offset of (direction, batch row): nn.LSTM's (2,B,H) layout, or batch-major (B,2,H)
// -- the latter makes hn.view(B, 2H) the concatenated latent [h_fwd | h_rev] and feeds the next LSTM with no copy
__device__ __forceinline__ size_t state_off(int bm, int dir, int b, int B) { return (bm ? (size_t)b * 2 + dir : (size_t)dir * B + b) * LH; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a fence that also drains the vector-memory queue
// (s_waitcnt vmcnt(0)): every saved-state / output store and every prefetch load of the step would have to retire before the
// barrier releases.  With this form they stay in flight across it (measured on the 32-row kernels of dic_lstm32.hip).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

constexpr int LXK = 32;            // input width of the fused-projection variant (two MFMA k-steps)
constexpr int XSTR = LXK + 8;      // bf16 elements per LDS row of the x tile (80 B: conflict-free ds_read_b128)

#ifdef DIC_LSTM_EXP_TIMING      // experiment: per-phase cycle stamps of workgroup (0,0), wave 0 (scripts/lstm_experiments.sh)
__device__ unsigned long long dic_lstm_stamps[2][32][8];
#define DIC_STAMP(kern, step, slot)                                                          \
    do {                                                                                     \
        if (blockIdx.x == 7 && blockIdx.y == 0 && threadIdx.x == 0 && (step) < 32)            \
            dic_lstm_stamps[kern][step][slot] = __builtin_readcyclecounter();                \
    } while (0)
#else
#define DIC_STAMP(kern, step, slot) do {} while (0)
#endif

struct LstmFwdArgs {
    const __bf16* gx;      // (R,B,2,4,H) input projection + both biases      [lstm_fwd_kernel<FWD_GX>; FWD_GXN: the lane-native form, gxn_off]
    const __bf16* x;       // (R,B,32) inputs and  wih (2,4H,32) input weights [lstm_fwd_kernel<FWD_PROJ>: projection in-kernel]
    const __bf16* wih;
    const __bf16* whh;     // (2,4H,H)
    const float* h0; const float* c0;     // (2,B,H) or NULL (zeros)
    __bf16* out;           // (R,B,2H): forward direction in [:H], reverse in [H:]
    __bf16* out_relu;      // optional second output relu(out), same layout (what the decoder reads of the encoder: clustering_interp.py:38-41)
    float* hn; float* cn;  // (2,B,H)
    __bf16* gates;         // lane-native (R,Bpad,2,4,H) post-activation i,f,g,o, or NULL (inference)
    __bf16* cs;            // lane-native (R,Bpad,2,H) cell states rounded to bf16 (the recurrence itself carries c in f32), or NULL
    int R, B;
    int bm;                // state tensors batch-major (B,2,H) instead of (2,B,H)
    int boundary;          // out is time slots 1..R of an (R+2,B,2H) buffer: also write h0 (bf16; zeros without one) into slot 0 [:, :H]
                           // (forward direction) / slot R+1 [:, H:] (reverse) -- every step's recurrent input for the weight-gradient kernels
};

// Lane-native form of gx (written by dic_row_proj in its lane-native mode, read only here): element (t, 32-row tile bt, dir,
// wave w, gate g, half qp of the lane's 16 units, hh, batch row r, e) with hidden unit u = 32 w + 16 qp + 4 hh + 8 (e / 4) + e % 4,
// i.e. accumulator register 8 qp + e of lane (r, hh) -- every wave-wide 16-B access is one contiguous 1-KiB piece and both
// kernels touch it straight from / into their MFMA accumulator registers.  B must be a multiple of 64.
__host__ __device__ __forceinline__ size_t gxn_off(int t, int nbt, int bt, int dir, int w, int g, int qp, int hh, int r) {
    size_t o = ((size_t)t * nbt + bt) * 2 + dir;
    o = ((o * 4 + w) * 4 + g) * 2 + qp;
    return ((o * 2 + hh) * 32 + r) * 8;
}

enum { FWD_GX = 0, FWD_PROJ = 1, FWD_GXN = 2 };
// FWD_GX:   G_t starts from a precomputed row-major gx_t (library GEMM / row-major dic_row_proj: any batch size).
// FWD_PROJ: G_t = x_t.W_ih^T + h_{t-1}.W_hh^T with W_ih (4H x 32) resident in registers next to W_hh, for narrow
//           inputs (encoder: 3C = 18 channels): the (R,B,8H) gx tensor -- a 1.6 GB write and a 1.6 GB read at
//           B = 32768 for 50 MB of actual input -- never exists.  The caller folds the bias in as a constant-one
//           input column.
// FWD_GXN:  gx_t in the lane-native form: every wave brings in only what its own lanes consume (16 LDS-DMA pieces of 1 KiB per
//           step, a wave-private 16-KiB region), so the staged tile needs no workgroup barrier of its own -- counted vmcnt waits
//           per 32-row half instead -- and its reads are contiguous; permutation-matrix MFMAs drop it into the accumulators.
template <int MODE>
__global__ __launch_bounds__(256, LNB == 1 ? 2 : 1) void lstm_fwd_kernel(LstmFwdArgs a) {
    constexpr bool PROJ = MODE == FWD_PROJ, GXN = MODE == FWD_GXN;
    extern __shared__ __align__(16) __bf16 fsm[];
    __bf16 (*hbuf)[LBM * HSTR] = reinterpret_cast<__bf16 (*)[LBM * HSTR]>(fsm);      // [2][LBM*HSTR]
    __bf16* gst = fsm + 2 * LBM * HSTR;                    // PROJ ? [2][LBM][XSTR] x tiles : [LBM][GSTR] staged gx tile
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int dir = blockIdx.y, b0 = blockIdx.x * LBM, B = a.B, R = a.R;
    const int nbt = gridDim.x * LNB;                       // 32-row batch tiles in the padded batch

    // A operand: this wave's W_hh rows (4 gates x 32 hidden units) for all 8 k-steps, resident in VGPRs
    bf16x8 wf[4][8];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            wf[g][ks] = *reinterpret_cast<const bf16x8*>(a.whh + ((size_t)(dir * 4 + g) * LH + 32 * w + r) * LH + ks * 16 + 8 * hh);
    // A-operand slices of the 32x32 identity for the two 16-wide k-steps: lane (row r, k-group hh) holds I[r][ks*16 + 8hh + j]
    bf16x8 eye[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) eye[ks][j] = (__bf16)((r == ks * 16 + 8 * hh + j) ? 1.0f : 0.0f);
    // FWD_GXN: a lane's 16-B piece (qp) of the lane-native gx is the B-operand column of batch row r with k = 8 hh + e <-> hidden
    // unit 16 qp + 8 (e / 4) + 4 hh + e % 4 of this wave: A = that permutation matrix (row m holds a one at the k of unit m)
    bf16x8 perm[2];
#pragma unroll
    for (int qp = 0; qp < 2; ++qp)
#pragma unroll
        for (int e = 0; e < 8; ++e) perm[qp][e] = (__bf16)((r == 16 * qp + 8 * (e >> 2) + 4 * hh + (e & 3)) ? 1.0f : 0.0f);
    bf16x8 wx[4][LXK / 16];
    if constexpr (PROJ) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int ks = 0; ks < LXK / 16; ++ks)
                wx[g][ks] = *reinterpret_cast<const bf16x8*>(a.wih + ((size_t)(dir * 4 + g) * LH + 32 * w + r) * LXK + ks * 16 + 8 * hh);
    }

    // cell state and h_0: lane owns batch row (nb*32 + r) and hidden units 32w + 8q + 4hh + {0..3}, q = 0..3
    float c[LNB][16];
#pragma unroll
    for (int nb = 0; nb < LNB; ++nb) {
        const int b = b0 + nb * 32 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = 32 * w + 8 * q + 4 * hh;
            f32x4 hv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
            if (b < B) {
                if (a.h0) hv = *reinterpret_cast<const f32x4*>(a.h0 + state_off(a.bm, dir, b, B) + u);
                if (a.c0) cv = *reinterpret_cast<const f32x4*>(a.c0 + state_off(a.bm, dir, b, B) + u);
            }
            bf16x4 hb;
#pragma unroll
            for (int j = 0; j < 4; ++j) { hb[j] = (__bf16)hv[j]; c[nb][4 * q + j] = cv[j]; }
            *reinterpret_cast<bf16x4*>(&hbuf[0][(nb * 32 + r) * HSTR + u]) = hb;
            if (a.boundary && b < B) {
                __bf16* slot = dir ? a.out + (size_t)R * B * 2 * LH : a.out - (size_t)B * 2 * LH;
                *reinterpret_cast<bf16x4*>(slot + (size_t)b * 2 * LH + dir * LH + u) = hb;
            }
        }
    }
    // gx tile of a step -> LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, asynchronous): one instruction moves
    // one whole 1-KiB row (row-major, as the GEMM wrote it) to a padded LDS row; the accumulator layout (batch
    // rows on lanes) is read back from LDS.  The tile of step s+1 is requested right after step s has consumed
    // its own, so it lands during the MFMA / gate-math / store phases (__syncthreads waits for it).
    auto request_gx = [&](int step) {
        const int t = dir ? R - 1 - step : step;
#pragma unroll
        for (int k = 0; k < LBM / 4; ++k) {
            const int rowl = k * 4 + w;
            const int b = min(b0 + rowl, B - 1);
            const __bf16* src = a.gx + (((size_t)t * B + b) * 2 + dir) * 4 * LH + lane * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(gst + rowl * GSTR), 16, 0, 0);
        }
    };
    // PROJ: the 64 x 32 x tile of a step is 4 KB -- one 16-B piece per thread, prefetched one step ahead through a
    // register and written to the other LDS buffer before the step's closing barrier
    const int xrow = tid >> 2, xpc = tid & 3;
    auto load_x = [&](int step) {
        const int t = dir ? R - 1 - step : step;
        return *reinterpret_cast<const bf16x8*>(a.x + ((size_t)t * B + min(b0 + xrow, B - 1)) * LXK + xpc * 8);
    };
    bf16x8 xnext = {};
    // FWD_GXN: this wave's pieces of one 32-row half of a step -> its private LDS region [nb][gate][qp][lane * 16 B]
    __bf16* const gwave = gst + w * (LNB * 4 * 2 * 512);
    auto request_gxn = [&](int nb, int step) {
        const int t = dir ? R - 1 - step : step;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                const __bf16* src = a.gx + gxn_off(t, nbt, blockIdx.x * LNB + nb, dir, w, g, qp, hh, r);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(gwave + ((nb * 4 + g) * 2 + qp) * 512), 16, 0, 0);
            }
    };
    // the pieces of a half have landed once at most the operations issued after them remain in flight: the other half's 8 pieces and
    // one step's saved-state / output stores (6 per unit group x 8 groups; B is a multiple of 64 here: no row is masked)
    // (half 1 of the last step: no pieces of a next step were requested behind it)
    auto gxn_landed = [&](bool other_half_behind) {
        if (a.gates && !a.out_relu) {
            if (other_half_behind) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    if constexpr (PROJ) {
        *reinterpret_cast<bf16x8*>(gst + xrow * XSTR + xpc * 8) = load_x(0);
    } else if constexpr (GXN) {
        request_gxn(0, 0);
        request_gxn(1, 0);
    } else {
        request_gx(0);
    }
    __syncthreads();
    // The rectified copy of the output (what the decoder reads of the encoder) leaves through the LDS tile of h that the next step
    // multiplies anyway, as whole 256-B row halves per 16 lanes, issued at the start of the following step so that the stores drain
    // under its MFMAs: four 16-B stores per lane and step replace an element-wise pass over (R,B,2H) (read + write, 0.15 ms at
    // B = 32768).  (The raw rows keep going out of the accumulator registers: moving them to this path as well was measured 9 %
    // slower for the whole kernel.)
    auto store_relu_rows = [&](int t_of_rows, int buf) {
        if (!a.out_relu) return;
#pragma unroll
        for (int k = 0; k < LBM * 16 / 256; ++k) {
            const int i = k * 256 + tid, row = i >> 4, pc = i & 15;
            if (b0 + row < B) {
                const uint4 v = *reinterpret_cast<const uint4*>(&hbuf[buf][row * HSTR + pc * 8]);
                // relu on packed bf16 pairs: a half with its sign bit set becomes 0 ((sign >> 15) * 0xffff = that half's mask)
                auto rl = [](unsigned x) { return x & ~(((x & 0x80008000u) >> 15) * 0xffffu); };
                *reinterpret_cast<uint4*>(a.out_relu + ((size_t)t_of_rows * B + b0 + row) * 2 * LH + dir * LH + pc * 8) =
                    make_uint4(rl(v.x), rl(v.y), rl(v.z), rl(v.w));
            }
        }
    };
    for (int step = 0; step < R; ++step) {
        const int t = dir ? R - 1 - step : step;
        const int cur = step & 1;
        f32x16 acc[4][LNB];
        DIC_STAMP(0, step, 0);
        if constexpr (PROJ) {
            if (step + 1 < R) xnext = load_x(step + 1);        // in flight across the MFMA and gate-math phases
            if (step > 0) store_relu_rows(dir ? R - step : step - 1, cur);
        } else if constexpr (GXN) {
            if (step > 0) store_relu_rows(dir ? R - step : step - 1, cur);
        } else {
            // accumulators start from the input projection of this step, brought into the accumulator layout BY the matrix core:
            // acc = I . gx^T with two 32x16 slices of the identity as A operand and 16-B row pieces of the staged tile as B
            // (exact: 1.0 x bf16 in f32).  16 ds_read_b128 + 16 MFMAs per wave instead of 64 ds_read_b64 + 128 conversions.
#pragma unroll
            for (int nb = 0; nb < LNB; ++nb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const __bf16* gp = gst + (nb * 32 + r) * GSTR + g * LH + 32 * w + 8 * hh;
                    f32x16 z;
#pragma unroll
                    for (int k = 0; k < 16; ++k) z[k] = 0.f;
                    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eye[0], *reinterpret_cast<const bf16x8*>(gp), z, 0, 0, 0);
                    acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(eye[1], *reinterpret_cast<const bf16x8*>(gp + 16), z, 0, 0, 0);
                }
            lds_barrier();                                     // every wave has read its part of the staged tile
            if (step + 1 < R) request_gx(step + 1);
            if (step > 0) store_relu_rows(dir ? R - step : step - 1, cur);      // (issued after the DMA: the counted wait below stays conservative)
        }
        DIC_STAMP(0, step, 1);
        const bool last = step == R - 1;
        // Software pipeline over the two 32-row halves: the MFMAs of half 1 are issued between the four gate-math groups of
        // half 0 (an MFMA occupies the vector issue port for 8 of its 32 cycles: the transcendental-heavy gate math runs in its
        // shadow), and the stores of half 0 drain while half 1 computes.
        static_assert(LNB == 2, "the half-step software pipeline is written for two 32-row halves");
        auto x_part = [&](int nb) {
            if constexpr (PROJ) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[g][nb][k] = 0.f;
                // G = W_ih . x_t^T (bias included: constant-one input column)
#pragma unroll
                for (int ks = 0; ks < LXK / 16; ++ks) {
                    const bf16x8 xb = *reinterpret_cast<const bf16x8*>(gst + cur * LBM * XSTR + (nb * 32 + r) * XSTR + ks * 16 + 8 * hh);
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wx[g][ks], xb, acc[g][nb], 0, 0, 0);
                }
            }
        };
        auto gxn_part = [&](auto nbc) {            // accumulators of a half <- its staged pre-activations (exact: 1.0 x bf16 in f32); the region refills
            constexpr int nb = decltype(nbc)::value;
            if constexpr (GXN) {
                if (step > 0) gxn_landed(nb == 0 || step + 1 < R);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const __bf16* gp = gwave + (nb * 4 + g) * 2 * 512 + lane * 8;
                    f32x16 z;
#pragma unroll
                    for (int k = 0; k < 16; ++k) z[k] = 0.f;
                    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(perm[0], *reinterpret_cast<const bf16x8*>(gp), z, 0, 0, 0);
                    acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(perm[1], *reinterpret_cast<const bf16x8*>(gp + 512), z, 0, 0, 0);
                }
                if (step + 1 < R) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads above have returned: the region may be overwritten
                    request_gxn(nb, step + 1);
                }
            }
        };
        auto h_part = [&](int nb, int ks) {        // G += W_hh . h_{t-1}^T, one k-step
            const bf16x8 hb = *reinterpret_cast<const bf16x8*>(&hbuf[cur][(nb * 32 + r) * HSTR + ks * 16 + 8 * hh]);
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[g][ks], hb, acc[g][nb], 0, 0, 0);
        };
        auto gate_math = [&](int nb, int q) {      // register-local: the four gates of a unit sit in the same lane / register
            const int b = b0 + nb * 32 + r;
            const bool ok = GXN || b < B;       // (lane-native gx: the batch tiles by 64 rows)

                const int u = 32 * w + 8 * q + 4 * hh;
                bf16x4 hb, ib, fb, gb, ob;
                f32x4 cv, hv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int k = 4 * q + j;
                    const float ig = sigmoid_fast(acc[0][nb][k]);
                    const float fg = sigmoid_fast(acc[1][nb][k]);
                    const float gg = tanh_fast(acc[2][nb][k]);
                    const float og = sigmoid_fast(acc[3][nb][k]);
                    const float cn = fmaf(fg, c[nb][k], ig * gg);
                    const float hn = og * tanh_fast(cn);
                    c[nb][k] = cn;
                    cv[j] = cn; hv[j] = hn;
                    hb[j] = (__bf16)hn; ib[j] = (__bf16)ig; fb[j] = (__bf16)fg; gb[j] = (__bf16)gg; ob[j] = (__bf16)og;
                }
                *reinterpret_cast<bf16x4*>(&hbuf[cur ^ 1][(nb * 32 + r) * HSTR + u]) = hb;
#ifdef DIC_LSTM_EXP_NOSTORE     // experiment: no saved-state stores
                if (false) {
#else
                if (a.gates) {
#endif
                    const int bt = blockIdx.x * LNB + nb;
                    *reinterpret_cast<bf16x4*>(a.gates + native_off(t, nbt, bt, dir, w, 4, 0, q, hh, r)) = ib;
                    *reinterpret_cast<bf16x4*>(a.gates + native_off(t, nbt, bt, dir, w, 4, 1, q, hh, r)) = fb;
                    *reinterpret_cast<bf16x4*>(a.gates + native_off(t, nbt, bt, dir, w, 4, 2, q, hh, r)) = gb;
                    *reinterpret_cast<bf16x4*>(a.gates + native_off(t, nbt, bt, dir, w, 4, 3, q, hh, r)) = ob;
                    bf16x4 cb;
#pragma unroll
                    for (int j = 0; j < 4; ++j) cb[j] = (__bf16)cv[j];
                    *reinterpret_cast<bf16x4*>(a.cs + native_off(t, nbt, bt, dir, w, 1, 0, q, hh, r)) = cb;
                }
                if (ok) {
                    const size_t row = (size_t)t * B + b;
                    *reinterpret_cast<bf16x4*>(a.out + row * 2 * LH + dir * LH + u) = hb;
                    if (last) {
                        *reinterpret_cast<f32x4*>(a.hn + state_off(a.bm, dir, b, B) + u) = hv;
                        *reinterpret_cast<f32x4*>(a.cn + state_off(a.bm, dir, b, B) + u) = cv;
                    }
                }
                    };
        x_part(0);
        gxn_part(IC<0>{});
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) h_part(0, ks);
        x_part(1);
        gxn_part(IC<1>{});
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            gate_math(0, q);
            h_part(1, 2 * q);
            h_part(1, 2 * q + 1);
        }
        DIC_STAMP(0, step, 2);
#pragma unroll
        for (int q = 0; q < 4; ++q) gate_math(1, q);
        if constexpr (PROJ) {
            if (step + 1 < R) *reinterpret_cast<bf16x8*>(gst + (cur ^ 1) * LBM * XSTR + xrow * XSTR + xpc * 8) = xnext;
        }
        DIC_STAMP(0, step, 3);
        if constexpr (MODE == FWD_GX) {
            // this wave's LDS-DMA pieces of the next gx tile have landed once only operations issued after them remain in flight:
            // the saved-state stores (5 per unit group x 8 groups; the `out` stores may have been branched over) -- a counted wait
            if (a.gates) asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        lds_barrier();
        DIC_STAMP(0, step, 4);
    }
    store_relu_rows(dir ? 0 : R - 1, R & 1);               // the last step's rows (its h went to buffer (R - 1) & 1 ^ 1)
}

// The fused-projection forward with EIGHT waves per workgroup (two per SIMD), 16 hidden units each.  Its per-step chain -- LDS reads,
// MFMAs, gate math, barrier -- is what bounds this variant (scripts/lstm_scale.py: 126 us for a workgroup alone on the chip, 143 us with
// all 256 CUs busy: it moves 40 % fewer bytes per step than the gx variants), and half the MFMAs and half the gate math per wave with a
// second wave on the SIMD to fill the gaps shorten exactly that.  A wave's MFMA A operand stacks two gates of its 16 units in one
// 32-row block ([i | f] and [g | o]): accumulator register k < 8 is the first gate of unit 16 w + (k & 3) + 8 (k >> 2) + 4 hh, register
// 8 + k the second gate of the same unit -- the four gates of a unit still meet in one lane, and the unit <-> (lane, register) map
// is the one of the 4-wave kernel (its wave w / 2, unit group 2 (w & 1) + q), so the saved gates / cell states land where lstm_bwd
// expects them and every number is bit-identical (same MFMA k order).
// XK = packed input width: 32 (3C < 32: the reference's six vitals) or 64 (3C < 64: BASELINE configs[3]'s twelve channels, 36 features + bias).
template <int XK>
__global__ __launch_bounds__(512, 1) void lstm_fwd8_proj_kernel(LstmFwdArgs a) {
    static_assert(LNB == 2, "two 32-row halves per workgroup");
    static_assert(XK == 32 || XK == 64, "packed input rows are 32 or 64 wide");
    constexpr int XS = XK + 8;                // bf16 elements per LDS row of the x tile (80 / 144 B: conflict-free ds_read_b128)
    extern __shared__ __align__(16) __bf16 fsm[];
    __bf16 (*hbuf)[LBM * HSTR] = reinterpret_cast<__bf16 (*)[LBM * HSTR]>(fsm);      // [2][LBM*HSTR]
    __bf16* gst = fsm + 2 * LBM * HSTR;                                               // [2][LBM][XS] x tiles
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);                           // 0..7
    const int dir = blockIdx.y, b0 = blockIdx.x * LBM, B = a.B, R = a.R;
    const int nbt = gridDim.x * LNB;
    const int wq = w >> 1, qb = 2 * (w & 1);              // wave / first unit group of the 4-wave kernel's layout these 16 units belong to

    bf16x8 wf[2][8], wx[2][XK / 16];
    {
        const int grow = (dir * 4 + (r >> 4)) * LH + 16 * w + (r & 15);               // block 0: gates 0 / 1 (rows 0-15 / 16-31); block 1: + 2 gates
#pragma unroll
        for (int bk = 0; bk < 2; ++bk) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                wf[bk][ks] = *reinterpret_cast<const bf16x8*>(a.whh + ((size_t)grow + 2 * bk * LH) * LH + ks * 16 + 8 * hh);
#pragma unroll
            for (int ks = 0; ks < XK / 16; ++ks)
                wx[bk][ks] = *reinterpret_cast<const bf16x8*>(a.wih + ((size_t)grow + 2 * bk * LH) * XK + ks * 16 + 8 * hh);
        }
    }
    // lane owns batch row (nb*32 + r) and hidden units 16 w + 8 q + 4 hh + {0..3}, q = 0..1: element 4 q + j
    float c[LNB][8];
    bf16x4 hkeep[LNB];     // unit group 0's h of each half, until group 1's is there (see gate_math)
#pragma unroll
    for (int nb = 0; nb < LNB; ++nb) {
        const int b = b0 + nb * 32 + r;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int u = 16 * w + 8 * q + 4 * hh;
            f32x4 hv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
            if (b < B) {
                if (a.h0) hv = *reinterpret_cast<const f32x4*>(a.h0 + state_off(a.bm, dir, b, B) + u);
                if (a.c0) cv = *reinterpret_cast<const f32x4*>(a.c0 + state_off(a.bm, dir, b, B) + u);
            }
            bf16x4 hb;
#pragma unroll
            for (int j = 0; j < 4; ++j) { hb[j] = (__bf16)hv[j]; c[nb][4 * q + j] = cv[j]; }
            *reinterpret_cast<bf16x4*>(&hbuf[0][(nb * 32 + r) * HSTR + u]) = hb;
            if (a.boundary && b < B) {
                __bf16* slot = dir ? a.out + (size_t)R * B * 2 * LH : a.out - (size_t)B * 2 * LH;
                *reinterpret_cast<bf16x4*>(slot + (size_t)b * 2 * LH + dir * LH + u) = hb;
            }
        }
    }
    constexpr int XPC = XK / 8;                           // 16-B pieces per row: 64 rows x 4 (first 256 threads) or x 8 (all 512)
    const bool xloader = tid < LBM * XPC;
    const int xrow = (tid & (LBM * XPC - 1)) / XPC, xpc = tid & (XPC - 1);
    auto load_x = [&](int step) {
        const int t = dir ? R - 1 - step : step;
        return *reinterpret_cast<const bf16x8*>(a.x + ((size_t)t * B + min(b0 + xrow, B - 1)) * XK + xpc * 8);
    };
    bf16x8 xnext = {};
    if (xloader) *reinterpret_cast<bf16x8*>(gst + xrow * XS + xpc * 8) = load_x(0);
    __syncthreads();
    auto store_relu_rows = [&](int t_of_rows, int buf) {
        if (!a.out_relu) return;
#pragma unroll
        for (int k = 0; k < LBM * 16 / 512; ++k) {
            const int i = k * 512 + tid, row = i >> 4, pc = ((i & 15) - (row & 1)) & 15;      // odd rows rotated by one 16-B piece (LDS banks, see below)
            if (b0 + row < B) {
                const uint4 v = *reinterpret_cast<const uint4*>(&hbuf[buf][row * HSTR + pc * 8]);
                auto rl = [](unsigned x) { return x & ~(((x & 0x80008000u) >> 15) * 0xffffu); };
                *reinterpret_cast<uint4*>(a.out_relu + ((size_t)t_of_rows * B + b0 + row) * 2 * LH + dir * LH + pc * 8) =
                    make_uint4(rl(v.x), rl(v.y), rl(v.z), rl(v.w));
            }
        }
    };
    // the step's h rows LDS -> global as whole 256-B pieces, one step late (see lstm_fwd8_gxn_kernel; same-box A/B 501 -> 482 us here)
    // (round 4: a ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27}, ... -- eight pieces of an even row and the OTHER eight of the odd row
    //  behind it; at the 272-B pitch piece p of row r sits on slot (r + p) mod 16, so piece 12 of the even row met piece 11 of the odd one: a
    //  doubled slot in every group.  Lane i of an odd row now copies piece (i - 1) & 15 -- the 16 lanes still cover the row, conflict-free.)
    auto store_out_rows = [&](int t_of_rows, int buf) {
#pragma unroll
        for (int k = 0; k < LBM * 16 / 512; ++k) {
            const int i = k * 512 + tid, row = i >> 4, pc = ((i & 15) - (row & 1)) & 15;      // odd rows rotated by one 16-B piece (LDS banks, see below)
            if (b0 + row < B)
                *reinterpret_cast<uint4*>(a.out + ((size_t)t_of_rows * B + b0 + row) * 2 * LH + dir * LH + pc * 8) =
                    *reinterpret_cast<const uint4*>(&hbuf[buf][row * HSTR + pc * 8]);
        }
    };
    for (int step = 0; step < R; ++step) {
        const int t = dir ? R - 1 - step : step;
        const int cur = step & 1;
        f32x16 acc[2][LNB];
        if (xloader && step + 1 < R) xnext = load_x(step + 1);
        if (step > 0) store_relu_rows(dir ? R - step : step - 1, cur);
        if (step > 0) store_out_rows(dir ? R - step : step - 1, cur);
        const bool last = step == R - 1;
        auto x_part = [&](int nb) {                // G = W_ih . x_t^T (bias included: constant-one input column)
#pragma unroll
            for (int bk = 0; bk < 2; ++bk)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[bk][nb][k] = 0.f;
#pragma unroll
            for (int ks = 0; ks < XK / 16; ++ks) {
                const bf16x8 xb = *reinterpret_cast<const bf16x8*>(gst + cur * LBM * XS + (nb * 32 + r) * XS + ks * 16 + 8 * hh);
#pragma unroll
                for (int bk = 0; bk < 2; ++bk) acc[bk][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wx[bk][ks], xb, acc[bk][nb], 0, 0, 0);
            }
        };
        auto h_part = [&](int nb, int ks) {        // G += W_hh . h_{t-1}^T, one k-step
            const bf16x8 hb = *reinterpret_cast<const bf16x8*>(&hbuf[cur][(nb * 32 + r) * HSTR + ks * 16 + 8 * hh]);
#pragma unroll
            for (int bk = 0; bk < 2; ++bk) acc[bk][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[bk][ks], hb, acc[bk][nb], 0, 0, 0);
        };
        auto gate_math = [&](int nb, int q) {
            const int b = b0 + nb * 32 + r;
            const bool ok = b < B;
            const int u = 16 * w + 8 * q + 4 * hh;
            bf16x4 hb, ib, fb, gb, ob;
            f32x4 cv, hv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * q + j;
                const float ig = sigmoid_fast(acc[0][nb][e]);
                const float fg = sigmoid_fast(acc[0][nb][8 + e]);
                const float gg = tanh_fast(acc[1][nb][e]);
                const float og = sigmoid_fast(acc[1][nb][8 + e]);
                const float cn = fmaf(fg, c[nb][e], ig * gg);
                const float hn = og * tanh_fast(cn);
                c[nb][e] = cn;
                cv[j] = cn; hv[j] = hn;
                hb[j] = (__bf16)hn; ib[j] = (__bf16)ig; fb[j] = (__bf16)fg; gb[j] = (__bf16)gg; ob[j] = (__bf16)og;
            }
            // h_t -> the LDS tile as ONE 16-B store per lane and unit-group pair (round 4; was two 8-B stores: ds_write_b64 is served in groups
            // of 16 consecutive lanes with banks mod 32, and at the 272-B pitch rows r and r + 8 share their banks -- 2-way): the lane halves
            // trade group 1 of the lower half for group 0 of the upper one, so that lane (r, hh) holds units 16 w + 8 hh .. + 7 of its row
            if (q == 0) {
                hkeep[nb] = hb;
            } else {
                typedef unsigned hu32x2 __attribute__((ext_vector_type(2)));
                typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));
                hu32x2 A = __builtin_bit_cast(hu32x2, hkeep[nb]), Bv = __builtin_bit_cast(hu32x2, hb);
                const auto s0 = __builtin_amdgcn_permlane32_swap(A[0], Bv[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(A[1], Bv[1], false, false);
                const hu32x4 v16 = {s0[0], s1[0], s0[1], s1[1]};
                *reinterpret_cast<hu32x4*>(&hbuf[cur ^ 1][(nb * 32 + r) * HSTR + 16 * w + 8 * hh]) = v16;
            }
            if (a.gates) {
                const int bt = blockIdx.x * LNB + nb;
#ifndef DIC_LSTM_EXP_RECOMPUTE   // experiment (scripts/lstm_recompute_ab.sh, timing only): the backward recomputes the gates, the forward does not save them
                DIC_NT_STORE(bf16x4, a.gates + native_off(t, nbt, bt, dir, wq, 4, 0, qb + q, hh, r), ib);
                DIC_NT_STORE(bf16x4, a.gates + native_off(t, nbt, bt, dir, wq, 4, 1, qb + q, hh, r), fb);
                DIC_NT_STORE(bf16x4, a.gates + native_off(t, nbt, bt, dir, wq, 4, 2, qb + q, hh, r), gb);
                DIC_NT_STORE(bf16x4, a.gates + native_off(t, nbt, bt, dir, wq, 4, 3, qb + q, hh, r), ob);
#endif
                bf16x4 cb;
#pragma unroll
                for (int j = 0; j < 4; ++j) cb[j] = (__bf16)cv[j];
                DIC_NT_STORE(bf16x4, a.cs + native_off(t, nbt, bt, dir, wq, 1, 0, qb + q, hh, r), cb);
            }
            if (ok) {
                if (last) {
                    *reinterpret_cast<f32x4*>(a.hn + state_off(a.bm, dir, b, B) + u) = hv;
                    *reinterpret_cast<f32x4*>(a.cn + state_off(a.bm, dir, b, B) + u) = cv;
                }
            }
        };
        x_part(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) h_part(0, ks);
        x_part(1);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            gate_math(0, q);
#pragma unroll
            for (int ks = 4 * q; ks < 4 * q + 4; ++ks) h_part(1, ks);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) gate_math(1, q);
        if (xloader && step + 1 < R) *reinterpret_cast<bf16x8*>(gst + (cur ^ 1) * LBM * XS + xrow * XS + xpc * 8) = xnext;
        lds_barrier();
    }
    store_relu_rows(dir ? 0 : R - 1, R & 1);
    store_out_rows(dir ? 0 : R - 1, R & 1);
}


// The same eight-wave layout for the decoder's forward on lane-native gx (FWD_GXN): a piece of that layout (unit half qp of wave w) is
// exactly the 16 units of eight-wave wave 2 w + qp, so every wave stages 4 pieces per 32-row half and step.
__global__ __launch_bounds__(512, 1) void lstm_fwd8_gxn_kernel(LstmFwdArgs a) {
    static_assert(LNB == 2, "two 32-row halves per workgroup");
    extern __shared__ __align__(16) __bf16 fsm[];
    __bf16 (*hbuf)[LBM * HSTR] = reinterpret_cast<__bf16 (*)[LBM * HSTR]>(fsm);      // [2][LBM*HSTR]
    __bf16* gst = fsm + 2 * LBM * HSTR;                                               // [8 waves][LNB][4 gates][512]: the wave's own lane-native gx pieces
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);                           // 0..7
    const int dir = blockIdx.y, b0 = blockIdx.x * LBM, B = a.B, R = a.R;
    const int nbt = gridDim.x * LNB;
    const int wq = w >> 1, qb = 2 * (w & 1);              // wave / first unit group of the 4-wave kernel's layout these 16 units belong to

    bf16x8 wf[2][8];
    {
        const int grow = (dir * 4 + (r >> 4)) * LH + 16 * w + (r & 15);               // block 0: gates 0 / 1 (rows 0-15 / 16-31); block 1: + 2 gates
#pragma unroll
        for (int bk = 0; bk < 2; ++bk) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
                wf[bk][ks] = *reinterpret_cast<const bf16x8*>(a.whh + ((size_t)grow + 2 * bk * LH) * LH + ks * 16 + 8 * hh);
        }
    }
    // lane owns batch row (nb*32 + r) and hidden units 16 w + 8 q + 4 hh + {0..3}, q = 0..1: element 4 q + j
    float c[LNB][8];
    bf16x4 hkeep[LNB];     // unit group 0's h of each half, until group 1's is there (see gate_math)
#pragma unroll
    for (int nb = 0; nb < LNB; ++nb) {
        const int b = b0 + nb * 32 + r;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int u = 16 * w + 8 * q + 4 * hh;
            f32x4 hv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
            if (b < B) {
                if (a.h0) hv = *reinterpret_cast<const f32x4*>(a.h0 + state_off(a.bm, dir, b, B) + u);
                if (a.c0) cv = *reinterpret_cast<const f32x4*>(a.c0 + state_off(a.bm, dir, b, B) + u);
            }
            bf16x4 hb;
#pragma unroll
            for (int j = 0; j < 4; ++j) { hb[j] = (__bf16)hv[j]; c[nb][4 * q + j] = cv[j]; }
            *reinterpret_cast<bf16x4*>(&hbuf[0][(nb * 32 + r) * HSTR + u]) = hb;
            if (a.boundary && b < B) {
                __bf16* slot = dir ? a.out + (size_t)R * B * 2 * LH : a.out - (size_t)B * 2 * LH;
                *reinterpret_cast<bf16x4*>(slot + (size_t)b * 2 * LH + dir * LH + u) = hb;
            }
        }
    }
    // A operands that drop a lane-native piece (B operand: lane (row r, hh) holds the 8 values of units 8 (e / 4) + 4 hh + e % 4 of the
    // wave's 16) into rows 0-15 (first gate of the block) or 16-31 (second gate) of the accumulator block: 0/1 matrices, exact
    bf16x8 perm[2];
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int e = 0; e < 8; ++e) perm[half][e] = (__bf16)((r == 16 * half + 8 * (e >> 2) + 4 * hh + (e & 3)) ? 1.0f : 0.0f);
    __bf16* const gwave = gst + w * (LNB * 4 * 512);
    auto request_gxn = [&](int nb, int step) {          // this wave's 4 pieces (gates) of one 32-row half: unit half w & 1 of the 4-wave kernel's wave w / 2
        const int t = dir ? R - 1 - step : step;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const __bf16* src = a.gx + gxn_off(t, nbt, blockIdx.x * LNB + nb, dir, wq, g, w & 1, hh, r);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(gwave + (nb * 4 + g) * 512), 16, 0, 0);
        }
    };
    // the pieces of a half have landed once at most the operations issued after them remain in flight: the other half's 4 pieces and one
    // step's stores of this wave (5 saved-state stores per unit group x 4 groups + the 2 row stores of `store_out_rows` at the step's top; the
    // batch is a multiple of 64: no row is masked)
    auto gxn_landed = [&](bool other_half_behind) {
        if (a.gates && !a.out_relu) {
            if (other_half_behind) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    request_gxn(0, 0);
    request_gxn(1, 0);
    __syncthreads();
    auto store_relu_rows = [&](int t_of_rows, int buf) {
        if (!a.out_relu) return;
#pragma unroll
        for (int k = 0; k < LBM * 16 / 512; ++k) {
            const int i = k * 512 + tid, row = i >> 4, pc = ((i & 15) - (row & 1)) & 15;      // odd rows rotated by one 16-B piece (LDS banks, see below)
            if (b0 + row < B) {
                const uint4 v = *reinterpret_cast<const uint4*>(&hbuf[buf][row * HSTR + pc * 8]);
                auto rl = [](unsigned x) { return x & ~(((x & 0x80008000u) >> 15) * 0xffffu); };
                *reinterpret_cast<uint4*>(a.out_relu + ((size_t)t_of_rows * B + b0 + row) * 2 * LH + dir * LH + pc * 8) =
                    make_uint4(rl(v.x), rl(v.y), rl(v.z), rl(v.w));
            }
        }
    };
    // the step's h rows LDS -> global as whole 256-B pieces, one step late (round 2 let every lane store its own 8 bytes: 32 rows x 16 B per
    // instruction, 1 024 partial-line writes per workgroup and step instead of 128 line-sized ones; same-box A/B 701.7 -> 688.6 us)
    auto store_out_rows = [&](int t_of_rows, int buf) {
#pragma unroll
        for (int k = 0; k < LBM * 16 / 512; ++k) {
            const int i = k * 512 + tid, row = i >> 4, pc = ((i & 15) - (row & 1)) & 15;      // odd rows rotated by one 16-B piece (LDS banks, see below)
            const uint4 v = *reinterpret_cast<const uint4*>(&hbuf[buf][row * HSTR + pc * 8]);
            *reinterpret_cast<uint4*>(a.out + ((size_t)t_of_rows * B + b0 + row) * 2 * LH + dir * LH + pc * 8) = v;
        }
    };
    for (int step = 0; step < R; ++step) {
        const int t = dir ? R - 1 - step : step;
        const int cur = step & 1;
        f32x16 acc[2][LNB];
        if (step > 0) store_relu_rows(dir ? R - step : step - 1, cur);
        if (step > 0) store_out_rows(dir ? R - step : step - 1, cur);
        const bool last = step == R - 1;
        auto x_part = [&](int nb) {                // accumulators of a half <- its staged pre-activations (exact: 1.0 x bf16 in f32); the region refills
            if (step > 0) gxn_landed(nb == 0 || step + 1 < R);
#pragma unroll
            for (int bk = 0; bk < 2; ++bk) {
                const __bf16* gp = gwave + (nb * 4 + 2 * bk) * 512 + lane * 8;
                f32x16 z;
#pragma unroll
                for (int k = 0; k < 16; ++k) z[k] = 0.f;
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(perm[0], *reinterpret_cast<const bf16x8*>(gp), z, 0, 0, 0);
                acc[bk][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(perm[1], *reinterpret_cast<const bf16x8*>(gp + 512), z, 0, 0, 0);
            }
            if (step + 1 < R) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the reads above have returned: the region may be overwritten
                request_gxn(nb, step + 1);
            }
        };
        auto h_part = [&](int nb, int ks) {        // G += W_hh . h_{t-1}^T, one k-step
            const bf16x8 hb = *reinterpret_cast<const bf16x8*>(&hbuf[cur][(nb * 32 + r) * HSTR + ks * 16 + 8 * hh]);
#pragma unroll
            for (int bk = 0; bk < 2; ++bk) acc[bk][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[bk][ks], hb, acc[bk][nb], 0, 0, 0);
        };
        auto gate_math = [&](int nb, int q) {
            const int b = b0 + nb * 32 + r;
            const bool ok = true;               // (lane-native gx: the batch tiles by 64 rows)
            const int u = 16 * w + 8 * q + 4 * hh;
            bf16x4 hb, ib, fb, gb, ob;
            f32x4 cv, hv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * q + j;
                const float ig = sigmoid_fast(acc[0][nb][e]);
                const float fg = sigmoid_fast(acc[0][nb][8 + e]);
                const float gg = tanh_fast(acc[1][nb][e]);
                const float og = sigmoid_fast(acc[1][nb][8 + e]);
                const float cn = fmaf(fg, c[nb][e], ig * gg);
                const float hn = og * tanh_fast(cn);
                c[nb][e] = cn;
                cv[j] = cn; hv[j] = hn;
                hb[j] = (__bf16)hn; ib[j] = (__bf16)ig; fb[j] = (__bf16)fg; gb[j] = (__bf16)gg; ob[j] = (__bf16)og;
            }
            // h_t -> the LDS tile as ONE 16-B store per lane and unit-group pair (round 4; was two 8-B stores: ds_write_b64 is served in groups
            // of 16 consecutive lanes with banks mod 32, and at the 272-B pitch rows r and r + 8 share their banks -- 2-way): the lane halves
            // trade group 1 of the lower half for group 0 of the upper one, so that lane (r, hh) holds units 16 w + 8 hh .. + 7 of its row
            if (q == 0) {
                hkeep[nb] = hb;
            } else {
                typedef unsigned hu32x2 __attribute__((ext_vector_type(2)));
                typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));
                hu32x2 A = __builtin_bit_cast(hu32x2, hkeep[nb]), Bv = __builtin_bit_cast(hu32x2, hb);
                const auto s0 = __builtin_amdgcn_permlane32_swap(A[0], Bv[0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(A[1], Bv[1], false, false);
                const hu32x4 v16 = {s0[0], s1[0], s0[1], s1[1]};
                *reinterpret_cast<hu32x4*>(&hbuf[cur ^ 1][(nb * 32 + r) * HSTR + 16 * w + 8 * hh]) = v16;
            }
            if (a.gates) {
                const int bt = blockIdx.x * LNB + nb;
#ifndef DIC_LSTM_EXP_RECOMPUTE   // experiment (scripts/lstm_recompute_ab.sh, timing only): the backward recomputes the gates, the forward does not save them
                *reinterpret_cast<bf16x4*>(a.gates + native_off(t, nbt, bt, dir, wq, 4, 0, qb + q, hh, r)) = ib;
                *reinterpret_cast<bf16x4*>(a.gates + native_off(t, nbt, bt, dir, wq, 4, 1, qb + q, hh, r)) = fb;
                *reinterpret_cast<bf16x4*>(a.gates + native_off(t, nbt, bt, dir, wq, 4, 2, qb + q, hh, r)) = gb;
                *reinterpret_cast<bf16x4*>(a.gates + native_off(t, nbt, bt, dir, wq, 4, 3, qb + q, hh, r)) = ob;
#endif
                bf16x4 cb;
#pragma unroll
                for (int j = 0; j < 4; ++j) cb[j] = (__bf16)cv[j];
                *reinterpret_cast<bf16x4*>(a.cs + native_off(t, nbt, bt, dir, wq, 1, 0, qb + q, hh, r)) = cb;
            }
            if (ok) {
                if (last) {
                    *reinterpret_cast<f32x4*>(a.hn + state_off(a.bm, dir, b, B) + u) = hv;
                    *reinterpret_cast<f32x4*>(a.cn + state_off(a.bm, dir, b, B) + u) = cv;
                }
            }
        };
        x_part(0);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) h_part(0, ks);
        x_part(1);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            gate_math(0, q);
#pragma unroll
            for (int ks = 4 * q; ks < 4 * q + 4; ++ks) h_part(1, ks);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) gate_math(1, q);
        lds_barrier();
    }
    store_relu_rows(dir ? 0 : R - 1, R & 1);
    store_out_rows(dir ? 0 : R - 1, R & 1);
}



struct LstmBwdArgs {
    const __bf16* whh_t;   // (2,H,4H): whh_t[d][u][n] = whh[d][n][u]
    const __bf16* gates;   // lane-native, as written by lstm_fwd_kernel
    const __bf16* cs;      // lane-native, bf16
    const float* c0;       // (2,B,H) or NULL
    const __bf16* dout;    // (R,B,2H) or NULL
    const float* dhn; const float* dcn;   // (2,B,H) or NULL
    __bf16* dgx;           // (R,B,2,4,H) pre-activation gate gradients
    float* dh0; float* dc0;               // (2,B,H)
    float* dbias_part;     // (gridDim.x, 2, 4H) per-workgroup sums of dG over its rows and all steps, or NULL
    int R, B;
    int bm;                // state tensors (c0, dhn, dcn, dh0, dc0) batch-major (B,2,H) instead of (2,B,H)
    int relu;              // dout is the gradient of relu(out): it passes where h_t > 0, i.e. (o being a sigmoid) where tanh(c_t) > 0
};

__global__ __launch_bounds__(256, LNB == 1 ? 2 : 1) void lstm_bwd_kernel(LstmBwdArgs a) {
    extern __shared__ __align__(16) __bf16 dgt[];      // [LBM][GSTR], then the staged rows of dL/dout: [LNB][32][LH]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, hh = lane >> 5;
    const int dir = blockIdx.y, b0 = blockIdx.x * LBM, B = a.B, R = a.R;
    __bf16* dob = dgt + LBM * GSTR;
    const int nbt = gridDim.x * LNB;

    // A operand: W_hh^T rows for this wave's 32 hidden units, all 32 k-steps over the 4H gate columns
    bf16x8 wt[32];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks)
        wt[ks] = *reinterpret_cast<const bf16x8*>(a.whh_t + ((size_t)dir * LH + 32 * w + r) * 4 * LH + ks * 16 + 8 * hh);

#ifdef DIC_LSTM_EXP_RECOMPUTE
    f32x16 exp_acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) exp_acc[k] = 0.f;
#endif
    f32x16 dh[LNB];        // recurrent dL/dh arriving at the current step
    float dc[LNB][16];
    float ccar[LNB][16];   // c of the step processed next (= this step's c_prev): each cell state is read once
    float bsum[8];         // bias gradient: running column sums of dG (columns lane*8 .. +7 of this wave's rows)
#pragma unroll
    for (int k = 0; k < 8; ++k) bsum[k] = 0.f;
#pragma unroll
    for (int nb = 0; nb < LNB; ++nb) {
        const int b = b0 + nb * 32 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = 32 * w + 8 * q + 4 * hh;
            f32x4 hv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
            if (b < B) {
                if (a.dhn) hv = *reinterpret_cast<const f32x4*>(a.dhn + state_off(a.bm, dir, b, B) + u);
                if (a.dcn) cv = *reinterpret_cast<const f32x4*>(a.dcn + state_off(a.bm, dir, b, B) + u);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { dh[nb][4 * q + j] = hv[j]; dc[nb][4 * q + j] = cv[j]; }
        }
    }

    // Software pipeline over the two 32-row halves, each an independent recurrence chain, half a step apart:
    //     phase X(s):  MFMA + row stores of half 0, step s   ||   gate-gradient math of half 1, step s
    //     phase Y(s):  MFMA + row stores of half 1, step s   ||   gate-gradient math of half 0, step s+1
    // The VALU-heavy math of one half issues in the shadow of the other half's MFMAs, and the inputs of (half, step+1) are
    // requested right after (half, step) has consumed its registers -- a full step before they are needed, with the same
    // 96 VGPRs the one-deep pipeline used.  One barrier per phase: it publishes the half just written to LDS and retires the
    // reads of the half about to be overwritten.
    static_assert(LNB == 2, "the half-step software pipeline is written for two 32-row halves");
    struct StepIn { bf16x4 ib, fb, gb, ob, cp; };
    StepIn in[LNB][4];
    // (half and unit group are compile-time constants: a run-time index into the register arrays sends them to scratch memory)
    auto load_q = [&](auto nbc, auto qc, int step) {
        constexpr int nb = decltype(nbc)::value, q = decltype(qc)::value;
        const int t = dir ? step : R - 1 - step;           // reverse of the forward visiting order
        const bool first_fwd = step == R - 1;               // this t was the forward pass' first step
        const int tp = dir ? t + 1 : t - 1;                 // forward predecessor
        const int b = min(b0 + nb * 32 + r, B - 1);
        const int bt = blockIdx.x * LNB + nb;
        {
            const int u = 32 * w + 8 * q + 4 * hh;
            StepIn& d = in[nb][q];
            // (uniform base + this lane's 8 bytes: the base is scalar arithmetic, the loads take it as an SGPR pair)
            const unsigned lane_b = (unsigned)lane * 8u;                       // bytes
            const char* gbase = reinterpret_cast<const char*>(a.gates + native_off(t, nbt, bt, dir, w, 4, 0, q, 0, 0));
            constexpr unsigned GATE_STRIDE = 4 * 2 * 32 * 4 * 2;               // bytes between the planes of two gates
            d.ib = *reinterpret_cast<const bf16x4*>(gbase + lane_b);
#ifdef DIC_LSTM_EXP_RECOMPUTE    // experiment (timing only, wrong results): ONE 2-B-per-unit plane stands in for the h_prev row a recomputing backward reads
            d.fb = d.ib; d.gb = d.ib; d.ob = d.ib;
#else
            d.fb = *reinterpret_cast<const bf16x4*>(gbase + (GATE_STRIDE + lane_b));
            d.gb = *reinterpret_cast<const bf16x4*>(gbase + (2 * GATE_STRIDE + lane_b));
            d.ob = *reinterpret_cast<const bf16x4*>(gbase + (3 * GATE_STRIDE + lane_b));
#endif
            bf16x4 cp = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
            if (!first_fwd) cp = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const char*>(a.cs + native_off(tp, nbt, bt, dir, w, 1, 0, q, 0, 0)) + lane_b);
            else if (a.c0) {
                const f32x4 c0v = *reinterpret_cast<const f32x4*>(a.c0 + state_off(a.bm, dir, b, B) + u);
#pragma unroll
                for (int j = 0; j < 4; ++j) cp[j] = (__bf16)c0v[j];
            }
            d.cp = cp;
        }
    };
    // dL/dout of a 32-row half and step: WHOLE 256-B row pieces (16 lanes x 16 B; a load instruction = 4 rows) one step ahead into
    // registers, then into LDS with the 16-B pieces of row r rotated by r, where the gate math picks up its 8 bytes (2-way conflict at
    // worst).  Round 2 let every lane fetch its own 8 bytes straight from its row: 32 rows x 16 B per instruction, eight instructions per
    // wave and step -- 1 024 cache-line requests per workgroup and step for 16 KB, more than all its other loads together (640).
    bf16x8 dstage[LNB][2];
    auto dout_load = [&](auto nbc, int step) {
        constexpr int nb = decltype(nbc)::value;
        const int t = dir ? step : R - 1 - step;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 8 * w + 4 * i + (lane >> 4);
            const int b = min(b0 + nb * 32 + row, B - 1);
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
            if (a.dout) v = *reinterpret_cast<const bf16x8*>(a.dout + ((size_t)t * B + b) * 2 * LH + dir * LH + 8 * (lane & 15));
            dstage[nb][i] = v;
        }
    };
    auto dout_store = [&](auto nbc) {
        constexpr int nb = decltype(nbc)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = 8 * w + 4 * i + (lane >> 4);
            *reinterpret_cast<bf16x8*>(dob + (nb * 32 + row) * LH + (((lane & 15) + row) & 15) * 8) = dstage[nb][i];
        }
    };
    auto load_half = [&](auto nbc, int step) {
        load_q(nbc, IC<0>{}, step); load_q(nbc, IC<1>{}, step); load_q(nbc, IC<2>{}, step); load_q(nbc, IC<3>{}, step);
    };
    // gate gradients of hidden units 32w + 8q + 4hh .. +3 of half nb -> its rows of the LDS dG tile; dc and the c carry advance
    auto math_q = [&](int nb, int q) {
        [[maybe_unused]] const int u = 32 * w + 8 * q + 4 * hh;
        const StepIn& x = in[nb][q];
        const bf16x4 go = *reinterpret_cast<const bf16x4*>(dob + (nb * 32 + r) * LH + (((4 * w + q) + r) & 15) * 8 + 4 * hh);
        bf16x4 di, df, dg, dO;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * q + j;
#ifdef DIC_LSTM_EXP_RECOMPUTE    // ... and the four activations of the recomputed pre-activations (3 sigmoids + 1 tanh = 8 transcendentals per unit)
            const float ig = sigmoid_fast((float)x.ib[j]), fg = sigmoid_fast((float)x.fb[j] + 0.1f), gg = tanh_fast((float)x.gb[j]), og = sigmoid_fast((float)x.ob[j] - 0.1f);
#else
            const float ig = (float)x.ib[j], fg = (float)x.fb[j], gg = (float)x.gb[j], og = (float)x.ob[j];
#endif
            const float tc = tanh_fast(ccar[nb][k]);
            const float dht = dh[nb][k] + ((a.relu && !(tc > 0.f)) ? 0.f : (float)go[j]);
            const float dct = fmaf(dht * og, 1.0f - tc * tc, dc[nb][k]);
            const float vi = dct * gg * ig * (1.0f - ig), vf = dct * (float)x.cp[j] * fg * (1.0f - fg);
            const float vg = dct * ig * (1.0f - gg * gg), vo = dht * tc * og * (1.0f - og);
            di[j] = (__bf16)vi; df[j] = (__bf16)vf; dg[j] = (__bf16)vg; dO[j] = (__bf16)vo;
            dc[nb][k] = dct * fg;
            ccar[nb][k] = (float)x.cp[j];                 // this step's c_prev is the next visited step's c
        }
#ifdef DIC_LSTM_BWD_B64_WRITES       // round 2's form: four 8-B stores per lane, 16 consecutive rows per LDS pass -> rows r and r + 8 share banks
        __bf16* lp = dgt + (nb * 32 + r) * GSTR + u;
        *reinterpret_cast<bf16x4*>(lp) = di;
        *reinterpret_cast<bf16x4*>(lp + LH) = df;
        *reinterpret_cast<bf16x4*>(lp + 2 * LH) = dg;
        *reinterpret_cast<bf16x4*>(lp + 3 * LH) = dO;
#else
        // The two lane halves hold units u..u+3 (hh = 0) and u+4..u+7 (hh = 1) of the same row: four v_permlane32_swap trade them so that
        // the lower half ends up with the whole 8-unit (16-B) pieces of gates i and g, the upper half with those of f and o, and each lane
        // issues TWO 16-B stores.  A ds_write_b128 pass covers 8 consecutive rows: at the 1040-B row pitch they fall on 8 x 4 distinct
        // banks -- no conflict (the 8-B stores were 2-way: SQ_LDS_BANK_CONFLICT 22 % of the kernel's LDS cycles in round 2).
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x2 I = __builtin_bit_cast(u32x2, di), F = __builtin_bit_cast(u32x2, df), G = __builtin_bit_cast(u32x2, dg), O = __builtin_bit_cast(u32x2, dO);
        {   // permlane32_swap(x, y): x of lanes 32..63 <-> y of lanes 0..31
            auto s0 = __builtin_amdgcn_permlane32_swap(I[0], F[0], false, false); I[0] = s0[0]; F[0] = s0[1];
            auto s1 = __builtin_amdgcn_permlane32_swap(I[1], F[1], false, false); I[1] = s1[0]; F[1] = s1[1];
            auto s2 = __builtin_amdgcn_permlane32_swap(G[0], O[0], false, false); G[0] = s2[0]; O[0] = s2[1];
            auto s3 = __builtin_amdgcn_permlane32_swap(G[1], O[1], false, false); G[1] = s3[0]; O[1] = s3[1];
        }
        // lower half: (I, F) = gate i of units u..u+7 (this lane's four, then the partner's), (G, O) = gate g; upper half: gates f and o
        __bf16* lp = dgt + (nb * 32 + r) * GSTR + (32 * w + 8 * q) + hh * LH;
        const u32x4 c0 = {I[0], I[1], F[0], F[1]}, c1 = {G[0], G[1], O[0], O[1]};
        *reinterpret_cast<u32x4*>(lp) = c0;
        *reinterpret_cast<u32x4*>(lp + 2 * LH) = c1;
#endif
    };
    // one of this wave's 8 rows of half nb: dG row LDS -> global (whole 1-KiB row per wave instruction, row-major for the
    // weight-gradient GEMMs) + bias column sums, and four k-steps of dh_{t-1}[u][b] = sum_n W_hh[n][u] dG_t[b][n].
    // The B fragments of a group of four k-steps are requested ONE GROUP AHEAD into a two-slot ring (`fetch_b`), so that the MFMAs never
    // wait on an LDS read issued right in front of them (round 2's form read each fragment immediately before its MFMA: hipcc put an
    // `s_waitcnt lgkmcnt(0)` in front of all 64 MFMAs of a step, an exposed LDS round trip each -- the wave is alone on its SIMD).
    bf16x8 gring[2][4];
    auto fetch_b = [&](int nb, int k) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            gring[k & 1][i] = *reinterpret_cast<const bf16x8*>(dgt + (nb * 32 + r) * GSTR + (4 * k + i) * 16 + 8 * hh);
    };
    auto row_and_mfma = [&](int nb, int k, int t) {
        const int rowl = nb * 32 + k * 4 + w;
        const int b = b0 + rowl;
        bf16x8 v;
        if (b < B) v = *reinterpret_cast<const bf16x8*>(dgt + rowl * GSTR + lane * 8);
        if (k + 1 < 8) fetch_b(nb, k + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ks = 4 * k + i;
            const bf16x8 gbv = gring[k & 1][i];
#ifndef DIC_LSTM_BWD_ASM_MFMA
            dh[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[ks], gbv, dh[nb], 0, 0, 0);
#else
            // (experiment, bit-identical, no gain: 0.827 -> 0.831 ms)  The MFMA as asm with its A operand constrained to the accumulation
            // registers: W_hh^T (128 registers) then LIVES there and is read in place -- through the builtin hipcc parks the weights in
            // AGPRs as spill space and copies four of them back with v_accvgpr_read in front of every MFMA (256 copies per step).  Wait
            // states by hand (cdna_hip_programming.md 5.7 item 2): `s_nop 1` covers a v_accvgpr_write of the zeroed accumulator just
            // before; the accumulate chain itself needs none; the readers of D are held off by the `s_nop 11` after the phase's last MFMA.
            asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(dh[nb]) : "a"(wt[ks]), "v"(gbv));
#endif
#ifdef DIC_LSTM_EXP_RECOMPUTE    // ... and the 80 MFMAs per wave and step of W_hh.h_prev + W_ih.x (5 per 4 of the backward's own; OPTIMISTIC: they reuse the
            // resident W_hh^T fragments and the dG tile as operands, so the 128 extra weight registers and the h / x tiles in LDS cost nothing here)
            exp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[ks], gbv, exp_acc, 0, 0, 0);
            if ((ks & 3) == 3) exp_acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wt[ks ^ 1], gbv, exp_acc, 0, 0, 0);
#endif
        }
        if (b < B) {
            *reinterpret_cast<bf16x8*>(a.dgx + (((size_t)t * B + b) * 2 + dir) * 4 * LH + lane * 8) = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[e] += (float)v[e];
        }
#ifdef DIC_LSTM_BWD_ASM_MFMA
        if (k == 7) asm volatile("s_nop 11" : "+a"(dh[nb]));      // D of the phase's last MFMA -> its first reader: 12 wait states (8-pass XDL)
#endif
    };
    {   // prologue: c of the first visited step, the inputs of both halves, and half 0's first gate gradients
        const int t0 = dir ? 0 : R - 1;
#pragma unroll
        for (int nb = 0; nb < LNB; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x4 ct = *reinterpret_cast<const bf16x4*>(a.cs + native_off(t0, nbt, blockIdx.x * LNB + nb, dir, w, 1, 0, q, hh, r));
#pragma unroll
                for (int j = 0; j < 4; ++j) ccar[nb][4 * q + j] = (float)ct[j];
            }
        dout_load(IC<0>{}, 0);
        dout_load(IC<1>{}, 0);
        load_half(IC<0>{}, 0);
        load_half(IC<1>{}, 0);
        dout_store(IC<0>{});
        dout_store(IC<1>{});
        if (R > 1) { dout_load(IC<0>{}, 1); dout_load(IC<1>{}, 1); }
        lds_barrier();                                     // both halves' first rows of dL/dout are in LDS
#pragma unroll
        for (int q = 0; q < 4; ++q) math_q(0, q);
        if (R > 1) load_half(IC<0>{}, 1);
    }

    for (int step = 0; step < R; ++step) {
        const int t = dir ? step : R - 1 - step;
        DIC_STAMP(1, step, 0);
        lds_barrier();                                     // half 0 of dG_t is complete; nobody reads half 1 of the previous step any more
        DIC_STAMP(1, step, 1);
        // ---- phase X: MFMA + stores of half 0  ||  math of half 1
#pragma unroll
        for (int k = 0; k < 16; ++k) dh[0][k] = 0.f;
        // (each unit group's registers are refilled as soon as its math has consumed them: a full step of lead time, and the
        // requests spread over the step instead of one burst per phase)
        fetch_b(0, 0);
#define DIC_BWD_X(Q)                                                   \
        math_q(1, Q);                                                  \
        if (step + 1 < R) load_q(IC<1>{}, IC<Q>{}, step + 1);          \
        row_and_mfma(0, 2 * Q, t);                                     \
        row_and_mfma(0, 2 * Q + 1, t);
        DIC_BWD_X(0) DIC_BWD_X(1) DIC_BWD_X(2) DIC_BWD_X(3)
#undef DIC_BWD_X
        if (step + 1 < R) dout_store(IC<0>{});            // half 0's dL/dout of step + 1: read by phase Y's math (its last readers left before this phase's barrier)
        if (step + 2 < R) dout_load(IC<0>{}, step + 2);
        DIC_STAMP(1, step, 2);
        lds_barrier();                                     // half 1 of dG_t is complete; the reads of half 0 have retired
        DIC_STAMP(1, step, 3);
        // ---- phase Y: MFMA + stores of half 1  ||  math of half 0 for the next step
#pragma unroll
        for (int k = 0; k < 16; ++k) dh[1][k] = 0.f;
        fetch_b(1, 0);
        if (step + 1 < R) {
#define DIC_BWD_Y(Q)                                                   \
            math_q(0, Q);                                              \
            if (step + 2 < R) load_q(IC<0>{}, IC<Q>{}, step + 2);      \
            row_and_mfma(1, 2 * Q, t);                                 \
            row_and_mfma(1, 2 * Q + 1, t);
            DIC_BWD_Y(0) DIC_BWD_Y(1) DIC_BWD_Y(2) DIC_BWD_Y(3)
#undef DIC_BWD_Y
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) row_and_mfma(1, k, t);
        }
        if (step + 1 < R) dout_store(IC<1>{});            // half 1's dL/dout of step + 1: read by the next phase X
        if (step + 2 < R) dout_load(IC<1>{}, step + 2);
        DIC_STAMP(1, step, 4);
        DIC_STAMP(1, step, 5);
    }
#ifdef DIC_LSTM_EXP_RECOMPUTE
    asm volatile("" :: "v"(exp_acc));                      // (keeps the experiment's products alive)
#endif
    __syncthreads();                                       // the dG tile is free for the bias reduction below
    if (a.dbias_part) {     // add the 4 waves' column sums through LDS (the dG tile is free now): one partial per workgroup
        float* red = reinterpret_cast<float*>(dgt);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[w * 4 * LH + lane * 8 + e] = bsum[e];
        __syncthreads();
        for (int i = tid; i < 4 * LH; i += 256)
            a.dbias_part[((size_t)blockIdx.x * 2 + dir) * 4 * LH + i] = red[i] + red[4 * LH + i] + red[8 * LH + i] + red[12 * LH + i];
    }
#pragma unroll
    for (int nb = 0; nb < LNB; ++nb) {
        const int b = b0 + nb * 32 + r;
        if (b >= B) continue;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = 32 * w + 8 * q + 4 * hh;
            f32x4 hv, cv;
#pragma unroll
            for (int j = 0; j < 4; ++j) { hv[j] = dh[nb][4 * q + j]; cv[j] = dc[nb][4 * q + j]; }
            *reinterpret_cast<f32x4*>(a.dh0 + state_off(a.bm, dir, b, B) + u) = hv;
            *reinterpret_cast<f32x4*>(a.dc0 + state_off(a.bm, dir, b, B) + u) = cv;
        }
    }
}

// ---- the same backward with EIGHT waves per workgroup (round 3): a wave owns 16 hidden units (unit group pair q = 2 (w8 & 1) + {0, 1} of
// unit block w8 >> 1 in the saved-state layout), i.e. half the gate math, half the MFMA time and half the row copies of a four-wave wave,
// and two waves per SIMD fill each other's stalls.  dh[unit][batch] runs on v_mfma_f32_16x16x32_bf16 -- 16 units x 16 batch rows per
// MFMA, two batch blocks per 32-row half -- with the A rows ordered so that D lane (n, g) holds what math lane n + 16 g needs, up to one
// v_permlane16_swap per register between the two batch blocks (the recipe of dic_lstm32.hip's lstm_rec_bwd8_kernel, where it is
// bit-identical to the 32x32x16 form).  Same half-step software pipeline, same LDS images, same results.
typedef float f32x4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void lstm_bwd8_kernel(LstmBwdArgs a) {
    extern __shared__ __align__(16) __bf16 dgt[];      // [LBM][GSTR], then the staged rows of dL/dout: [LNB][32][LH]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5, n16 = lane & 15, g4 = lane >> 4;
    const int w8 = __builtin_amdgcn_readfirstlane(tid >> 6), w = w8 >> 1, qh = w8 & 1;      // w: the 32-unit block of the saved-state layout
    const int dir = blockIdx.y, b0 = blockIdx.x * LBM, B = a.B, R = a.R;
    __bf16* dob = dgt + LBM * GSTR;
    const int nbt = gridDim.x * LNB;
    // LDS image of the dG tile in THIS kernel (round 4): rows at a 1024-B pitch, the 16-B pieces of row b stored at piece ^ (b & 15).
    // Round 3 used the four-wave kernel's 1040-B pitch; PMC: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.41, and the access is fetch_b
    // below -- a ds_read_b128 is served in four 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS), i.e.
    // eight rows n of one k-group g and the OTHER eight rows of k-group g + 1: at a pitch of 65 slots row n sits on slot (n + g) mod 16 and
    // row 12 of group 0 meets row 11 of group 1 -- one doubled slot in every group, 8 LDS cycles per fragment instead of 4 (64 fragments per
    // step and wave: 256 of the 264 conflict cycles; the other 8 were the dL/dout reads, see dout_store).  With the XOR image slot =
    // (piece ^ n) & 15: a bijection of n within the rows of one k-group, and {4..11} is closed under ^ 1, ^ 2, ^ 3 -- conflict-free; the
    // 16-B stores of math_q (eight consecutive rows per group, banks mod 32) and the row copies (lane reads piece lane ^ (row & 15), so it
    // still holds columns 8 lane .. + 7) stay conflict-free too.  Same values, same order: bit-identical.
    constexpr int G8 = 4 * LH;                             // bf16 elements per LDS row of the dG tile (1024 B)
    // per-lane element offsets, computed once (everything else of an address is an immediate): fetch_b's piece for k-step 4 m + j of rows
    // n16 / 16 + n16, and math_q's two pieces of row r
    int fbo[4], mqo[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) fbo[j] = n16 * G8 + (((4 * j + g4) ^ n16) & 15) * 8;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) mqo[qq] = r * G8 + ((4 * w + 2 * qh + qq + 16 * hh) ^ (r & 15)) * 8;
    static_assert(LNB == 2, "the half-step software pipeline is written for two 32-row halves");

    // A operand (16x16x32): lane (m = lane & 15, kg = lane >> 4) holds W_hh^T[unit s(m)][32 ks + 8 kg .. + 7], s(m) = 8 ((m >> 2) & 1) + 4 (m >> 3) + (m & 3)
    bf16x8 wt[16];
    {
        const int unit = 16 * w8 + 8 * ((n16 >> 2) & 1) + 4 * (n16 >> 3) + (n16 & 3);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks)
            wt[ks] = *reinterpret_cast<const bf16x8*>(a.whh_t + ((size_t)dir * LH + unit) * 4 * LH + ks * 32 + 8 * g4);
    }
    float dh[LNB][8];      // element e = 4 qq + j: unit 16 w8 + 8 qq + 4 hh + j of batch row r
    float dc[LNB][8];
    bf16x4 ccar[LNB][2];   // c of the step processed next, as the bf16 it was saved in (packed: the wave has 256 registers)
    float bsum[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bsum[k] = 0.f;
#pragma unroll
    for (int nb = 0; nb < LNB; ++nb) {
        const int b = b0 + nb * 32 + r;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int u = 16 * w8 + 8 * qq + 4 * hh;
            f32x4 hv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
            if (b < B) {
                if (a.dhn) hv = *reinterpret_cast<const f32x4*>(a.dhn + state_off(a.bm, dir, b, B) + u);
                if (a.dcn) cv = *reinterpret_cast<const f32x4*>(a.dcn + state_off(a.bm, dir, b, B) + u);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { dh[nb][4 * qq + j] = hv[j]; dc[nb][4 * qq + j] = cv[j]; }
        }
    }
    struct StepIn { bf16x4 ib, fb, gb, ob, cp; };
    StepIn in[LNB][2];
    auto load_q = [&](auto nbc, auto qc, int step) {
        constexpr int nb = decltype(nbc)::value, qq = decltype(qc)::value;
        const int q = 2 * qh + qq;
        const int t = dir ? step : R - 1 - step;
        const bool first_fwd = step == R - 1;
        const int tp = dir ? t + 1 : t - 1;
        const int b = min(b0 + nb * 32 + r, B - 1);
        const int bt = blockIdx.x * LNB + nb;
        const int u = 16 * w8 + 8 * qq + 4 * hh;
        StepIn& d = in[nb][qq];
        const unsigned lane_b = (unsigned)lane * 8u;
        const char* gbase = reinterpret_cast<const char*>(a.gates + native_off(t, nbt, bt, dir, w, 4, 0, q, 0, 0));
        constexpr unsigned GATE_STRIDE = 4 * 2 * 32 * 4 * 2;
        d.ib = DIC_NT_LOAD(bf16x4, gbase + lane_b);
        d.fb = DIC_NT_LOAD(bf16x4, gbase + (GATE_STRIDE + lane_b));
        d.gb = DIC_NT_LOAD(bf16x4, gbase + (2 * GATE_STRIDE + lane_b));
        d.ob = DIC_NT_LOAD(bf16x4, gbase + (3 * GATE_STRIDE + lane_b));
        bf16x4 cp = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        if (!first_fwd) cp = DIC_NT_LOAD(bf16x4, reinterpret_cast<const char*>(a.cs + native_off(tp, nbt, bt, dir, w, 1, 0, q, 0, 0)) + lane_b);
        else if (a.c0) {
            const f32x4 c0v = *reinterpret_cast<const f32x4*>(a.c0 + state_off(a.bm, dir, b, B) + u);
#pragma unroll
            for (int j = 0; j < 4; ++j) cp[j] = (__bf16)c0v[j];
        }
        d.cp = cp;
    };
    auto load_half = [&](auto nbc, int step) { load_q(nbc, IC<0>{}, step); load_q(nbc, IC<1>{}, step); };
    // dL/dout of a 32-row half and step through LDS (see lstm_bwd_kernel): eight waves x one instruction of 4 rows x 256 B
    bf16x8 dstage[LNB];
    auto dout_load = [&](auto nbc, int step) {
        constexpr int nb = decltype(nbc)::value;
        const int t = dir ? step : R - 1 - step;
        const int row = 4 * w8 + (lane >> 4);
        const int b = min(b0 + nb * 32 + row, B - 1);
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (__bf16)0.f;
        if (a.dout) v = DIC_NT_LOAD(bf16x8, a.dout + ((size_t)t * B + b) * 2 * LH + dir * LH + 8 * (lane & 15));
        dstage[nb] = v;
    };
    auto dout_store = [&](auto nbc) {
        constexpr int nb = decltype(nbc)::value;
        const int row = 4 * w8 + (lane >> 4);
        // rows 16-31 of a half (waves 4-7) keep the two 8-B halves of every piece swapped: math lanes r and r + 16 read the same piece slot
        // (rotation by r mod 16) -- from opposite halves now, so the 32 lanes of a ds_read_b64 group cover all 64 banks (was 2-way)
        bf16x8 v = dstage[nb];
        if (w8 >= 4) v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 0, 1, 2, 3);
        *reinterpret_cast<bf16x8*>(dob + (nb * 32 + row) * LH + (((lane & 15) + row) & 15) * 8) = v;
    };
    auto math_q = [&](int nb, int qq) {
        const int q = 2 * qh + qq;
        const StepIn& x = in[nb][qq];
        const bf16x4 go = *reinterpret_cast<const bf16x4*>(dob + (nb * 32 + r) * LH + (((4 * w + q) + r) & 15) * 8 + 4 * (hh ^ (r >> 4)));
        bf16x4 di, df, dg, dO;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = 4 * qq + j;
            const float ig = (float)x.ib[j], fg = (float)x.fb[j], gg = (float)x.gb[j], og = (float)x.ob[j];
            const float tc = tanh_fast((float)ccar[nb][qq][j]);
            const float dht = dh[nb][k] + ((a.relu && !(tc > 0.f)) ? 0.f : (float)go[j]);
            const float dct = fmaf(dht * og, 1.0f - tc * tc, dc[nb][k]);
            const float vi = dct * gg * ig * (1.0f - ig), vf = dct * (float)x.cp[j] * fg * (1.0f - fg);
            const float vg = dct * ig * (1.0f - gg * gg), vo = dht * tc * og * (1.0f - og);
            di[j] = (__bf16)vi; df[j] = (__bf16)vf; dg[j] = (__bf16)vg; dO[j] = (__bf16)vo;
            dc[nb][k] = dct * fg;
        }
        ccar[nb][qq] = x.cp;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        u32x2 I = __builtin_bit_cast(u32x2, di), F = __builtin_bit_cast(u32x2, df), G = __builtin_bit_cast(u32x2, dg), O = __builtin_bit_cast(u32x2, dO);
        {
            auto s0 = __builtin_amdgcn_permlane32_swap(I[0], F[0], false, false); I[0] = s0[0]; F[0] = s0[1];
            auto s1 = __builtin_amdgcn_permlane32_swap(I[1], F[1], false, false); I[1] = s1[0]; F[1] = s1[1];
            auto s2 = __builtin_amdgcn_permlane32_swap(G[0], O[0], false, false); G[0] = s2[0]; O[0] = s2[1];
            auto s3 = __builtin_amdgcn_permlane32_swap(G[1], O[1], false, false); G[1] = s3[0]; O[1] = s3[1];
        }
        __bf16* lp = dgt + nb * 32 * G8 + mqo[qq];             // logical piece 4 w + q + 16 hh (c1: + 32) of row r, XOR image
        const u32x4 c0 = {I[0], I[1], F[0], F[1]}, c1 = {G[0], G[1], O[0], O[1]};
        *reinterpret_cast<u32x4*>(lp) = c0;
        *reinterpret_cast<u32x4*>(lp + 32 * 8) = c1;
    };
    // the recurrent product of half nb in four groups of (4 k-steps x 2 batch blocks), B fragments one group ahead; one of this wave's four
    // dG rows of the half leaves for global memory with every group
    f32x4v acc0, acc1;
    bf16x8 gring[2];                // the B fragments of ONE k-step (two batch blocks): the MFMAs read their operands at issue, so the next k-step's
                                    // reads may overwrite the registers right behind them; the wave's partner on the SIMD covers the round trip
                                    // (a wave has 256 registers here: a deeper ring spills)
    auto fetch_b = [&](int nb, int ks) {
        const __bf16* fp = dgt + fbo[ks & 3] + nb * 32 * G8 + (ks >> 2) * 128;      // piece (4 ks + g4) ^ n16 = 16 (ks >> 2) + ((4 (ks & 3) + g4) ^ n16)
        gring[0] = *reinterpret_cast<const bf16x8*>(fp);
        gring[1] = *reinterpret_cast<const bf16x8*>(fp + 16 * G8);          // row 16 + n16: the same XOR value
    };
    auto mfma_sub = [&](int nb, int sg) {      // two k-steps
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ks = 2 * sg + i;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[ks], gring[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wt[ks], gring[1], acc1, 0, 0, 0);
            if (ks + 1 < 16) fetch_b(nb, ks + 1);
        }
    };
    auto row_and_mfma = [&](int nb, int k, int t) {
        const int rowl = nb * 32 + k * 8 + w8;
        const int b = b0 + rowl;
        bf16x8 v;
        if (b < B) v = *reinterpret_cast<const bf16x8*>(dgt + rowl * G8 + (lane ^ (rowl & 15)) * 8);      // logical piece `lane` of the row
        mfma_sub(nb, 2 * k);
        mfma_sub(nb, 2 * k + 1);
        if (b < B) {
            DIC_NT_STORE(bf16x8, a.dgx + (((size_t)t * B + b) * 2 + dir) * 4 * LH + lane * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) bsum[e] += (float)v[e];
        }
    };
    auto finish_dh = [&](int nb) {     // even 16-lane rows keep acc0 and take the odd partner's acc0 (batch n), odd rows keep acc1 and take the even partner's (batch 16 + n)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const auto s = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, (float)acc0[e]), __builtin_bit_cast(unsigned, (float)acc1[e]), false, false);
            dh[nb][e] = __builtin_bit_cast(float, (unsigned)s[0]);
            dh[nb][4 + e] = __builtin_bit_cast(float, (unsigned)s[1]);
        }
    };
    {   // prologue
        const int t0 = dir ? 0 : R - 1;
#pragma unroll
        for (int nb = 0; nb < LNB; ++nb)
#pragma unroll
            for (int qq = 0; qq < 2; ++qq) {
                ccar[nb][qq] = *reinterpret_cast<const bf16x4*>(a.cs + native_off(t0, nbt, blockIdx.x * LNB + nb, dir, w, 1, 0, 2 * qh + qq, hh, r));
            }
        dout_load(IC<0>{}, 0);
        dout_load(IC<1>{}, 0);
        load_half(IC<0>{}, 0);
        load_half(IC<1>{}, 0);
        dout_store(IC<0>{});
        dout_store(IC<1>{});
        if (R > 1) { dout_load(IC<0>{}, 1); dout_load(IC<1>{}, 1); }
        lds_barrier();
        math_q(0, 0); math_q(0, 1);
        if (R > 1) load_half(IC<0>{}, 1);
    }
    for (int step = 0; step < R; ++step) {
        const int t = dir ? step : R - 1 - step;
        lds_barrier();                                     // half 0 of dG_t is complete; nobody reads half 1 of the previous step any more
        // ---- phase X: MFMA + stores of half 0  ||  math of half 1
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        fetch_b(0, 0);
#define DIC_BWD8_X(Q)                                                  \
        math_q(1, Q);                                                  \
        if (step + 1 < R) load_q(IC<1>{}, IC<Q>{}, step + 1);          \
        row_and_mfma(0, 2 * Q, t);                                     \
        row_and_mfma(0, 2 * Q + 1, t);
        DIC_BWD8_X(0) DIC_BWD8_X(1)
#undef DIC_BWD8_X
        finish_dh(0);
        if (step + 1 < R) dout_store(IC<0>{});
        if (step + 2 < R) dout_load(IC<0>{}, step + 2);
        lds_barrier();                                     // half 1 of dG_t is complete; the reads of half 0 have retired
        // ---- phase Y: MFMA + stores of half 1  ||  math of half 0 for the next step
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        fetch_b(1, 0);
        if (step + 1 < R) {
#define DIC_BWD8_Y(Q)                                                  \
            math_q(0, Q);                                              \
            if (step + 2 < R) load_q(IC<0>{}, IC<Q>{}, step + 2);      \
            row_and_mfma(1, 2 * Q, t);                                 \
            row_and_mfma(1, 2 * Q + 1, t);
            DIC_BWD8_Y(0) DIC_BWD8_Y(1)
#undef DIC_BWD8_Y
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) row_and_mfma(1, k, t);
        }
        finish_dh(1);
        if (step + 1 < R) dout_store(IC<1>{});
        if (step + 2 < R) dout_load(IC<1>{}, step + 2);
    }
    __syncthreads();                                       // the dG tile is free for the bias reduction below
    if (a.dbias_part) {
        float* red = reinterpret_cast<float*>(dgt);
#pragma unroll
        for (int e = 0; e < 8; ++e) red[w8 * 4 * LH + lane * 8 + e] = bsum[e];
        __syncthreads();
        for (int i = tid; i < 4 * LH; i += 512) {
            float sum = 0.f;
#pragma unroll
            for (int ww = 0; ww < 8; ++ww) sum += red[ww * 4 * LH + i];
            a.dbias_part[((size_t)blockIdx.x * 2 + dir) * 4 * LH + i] = sum;
        }
    }
#pragma unroll
    for (int nb = 0; nb < LNB; ++nb) {
        const int b = b0 + nb * 32 + r;
        if (b >= B) continue;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int u = 16 * w8 + 8 * qq + 4 * hh;
            f32x4 hv, cv;
#pragma unroll
            for (int j = 0; j < 4; ++j) { hv[j] = dh[nb][4 * qq + j]; cv[j] = dc[nb][4 * qq + j]; }
            *reinterpret_cast<f32x4*>(a.dh0 + state_off(a.bm, dir, b, B) + u) = hv;
            *reinterpret_cast<f32x4*>(a.dc0 + state_off(a.bm, dir, b, B) + u) = cv;
        }
    }
}

__global__ __launch_bounds__(256) void lstm_dbias_finalize(const float* partials, int nblk, float* dbias) {
    __shared__ double red[256];
    const int n = 2 * 4 * LH;
    const double s = reduce_partials_32x8(partials, nblk, n, blockIdx.x * 32, red);
    const int i = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x < 32 && i < n) dbias[i] = (float)s;
}

}  // namespace dic

using namespace dic;

extern "C" {

static int lstm_fwd_launch(int mode, const LstmFwdArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(2 * LBM * HSTR + (mode == FWD_PROJ ? 2 * LBM * XSTR : mode == FWD_GX ? LBM * GSTR : LBM * 4 * LH)) * sizeof(__bf16);
    static bool attr_set[3] = {false, false, false};
    const void* fn = mode == FWD_PROJ ? (const void*)lstm_fwd_kernel<FWD_PROJ>
                   : mode == FWD_GXN ? (const void*)lstm_fwd_kernel<FWD_GXN> : (const void*)lstm_fwd_kernel<FWD_GX>;
    if (!attr_set[mode]) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_fwd: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        attr_set[mode] = true;
    }
    const dim3 grid((a.B + LBM - 1) / LBM, 2);
    if (mode == FWD_PROJ) hipLaunchKernelGGL(lstm_fwd_kernel<FWD_PROJ>, grid, dim3(256), lds, st, a);
    else if (mode == FWD_GXN) hipLaunchKernelGGL(lstm_fwd_kernel<FWD_GXN>, grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL(lstm_fwd_kernel<FWD_GX>, grid, dim3(256), lds, st, a);
    return check_launch("lstm_fwd");
}

int dic_lstm_fwd(const void* gx, int gx_lane_native, const void* whh, const float* h0, const float* c0, int R, int B, int H,
                 void* out, void* out_relu, float* hn, float* cn, void* gates, void* cs, int state_batch_major, int write_boundary, dic_stream_t stream) {
    DIC_REQUIRE(R > 0 && B > 0, DIC_ERR_INVALID_ARG, "lstm_fwd: non-positive size");
    DIC_REQUIRE(H == LH, DIC_ERR_UNSUPPORTED, "lstm_fwd: hidden size %d (compiled for %d)", H, LH);
    DIC_REQUIRE(gx && whh && out && hn && cn, DIC_ERR_INVALID_ARG, "lstm_fwd: NULL pointer");
    DIC_REQUIRE((gates == nullptr) == (cs == nullptr), DIC_ERR_INVALID_ARG, "lstm_fwd: gates and cs go together");
    DIC_REQUIRE(!gx_lane_native || B % LBM == 0, DIC_ERR_INVALID_ARG, "lstm_fwd: lane-native gx needs a batch that is a multiple of %d (got %d)", LBM, B);
    LstmFwdArgs a{(const __bf16*)gx, nullptr, nullptr, (const __bf16*)whh, h0, c0, (__bf16*)out, (__bf16*)out_relu, hn, cn, (__bf16*)gates, (__bf16*)cs, R, B, state_batch_major != 0, write_boundary != 0};
    if (gx_lane_native == 2) {          // eight waves per workgroup (see lstm_fwd8_gxn_kernel)
        const size_t lds = (size_t)(2 * LBM * HSTR + LBM * 4 * LH) * sizeof(__bf16);
        static bool attr_set = false;
        if (!attr_set) {
            hipError_t e = hipFuncSetAttribute((const void*)lstm_fwd8_gxn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_fwd: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            attr_set = true;
        }
        hipLaunchKernelGGL(lstm_fwd8_gxn_kernel, dim3(B / LBM, 2), dim3(512), lds, (hipStream_t)stream, a);
        return check_launch("lstm_fwd");
    }
    return lstm_fwd_launch(gx_lane_native ? FWD_GXN : FWD_GX, a, (hipStream_t)stream);
}

int dic_lstm_fwd_proj(const void* x, const void* wih, const void* whh, const float* h0, const float* c0, int R, int B, int H,
                      int I, void* out, void* out_relu, float* hn, float* cn, void* gates, void* cs, int state_batch_major, int write_boundary,
                      int eight_waves, dic_stream_t stream) {
    DIC_REQUIRE(R > 0 && B > 0, DIC_ERR_INVALID_ARG, "lstm_fwd_proj: non-positive size");
    DIC_REQUIRE(H == LH, DIC_ERR_UNSUPPORTED, "lstm_fwd_proj: hidden size %d (compiled for %d)", H, LH);
    DIC_REQUIRE(I == LXK || I == 64, DIC_ERR_UNSUPPORTED, "lstm_fwd_proj: input width %d (compiled for %d and 64: zero-pad narrower inputs)", I, LXK);
    if (I == 64) eight_waves = 1;                  // (the 64-wide rows exist in the eight-wave kernel only)
    DIC_REQUIRE(x && wih && whh && out && hn && cn, DIC_ERR_INVALID_ARG, "lstm_fwd_proj: NULL pointer");
    DIC_REQUIRE((gates == nullptr) == (cs == nullptr), DIC_ERR_INVALID_ARG, "lstm_fwd_proj: gates and cs go together");
    LstmFwdArgs a{nullptr, (const __bf16*)x, (const __bf16*)wih, (const __bf16*)whh, h0, c0, (__bf16*)out, (__bf16*)out_relu, hn, cn, (__bf16*)gates, (__bf16*)cs, R, B, state_batch_major != 0, write_boundary != 0};
    if (eight_waves) {
        const int wide = I == 64;
        const size_t lds = (size_t)(2 * LBM * HSTR + 2 * LBM * (I + 8)) * sizeof(__bf16);
        static bool attr_set[2] = {false, false};
        if (!attr_set[wide]) {
            hipError_t e = hipFuncSetAttribute(wide ? (const void*)lstm_fwd8_proj_kernel<64> : (const void*)lstm_fwd8_proj_kernel<32>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_fwd_proj: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            attr_set[wide] = true;
        }
        if (wide) hipLaunchKernelGGL(lstm_fwd8_proj_kernel<64>, dim3((B + LBM - 1) / LBM, 2), dim3(512), lds, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(lstm_fwd8_proj_kernel<32>, dim3((B + LBM - 1) / LBM, 2), dim3(512), lds, (hipStream_t)stream, a);
        return check_launch("lstm_fwd_proj");
    }
    return lstm_fwd_launch(FWD_PROJ, a, (hipStream_t)stream);
}

#ifdef DIC_LSTM_EXP_TIMING
int dic_lstm_debug_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dic_lstm_stamps), sizeof(unsigned long long) * 2 * 32 * 8);
}
#endif

size_t dic_lstm_bwd_workspace(int B) { return B > 0 ? (size_t)((B + LBM - 1) / LBM) * 2 * 4 * LH * sizeof(float) : 0; }

int dic_lstm_bwd(const void* whh_t, const void* gates, const void* cs, const float* c0, const void* dout,
                 const float* dhn, const float* dcn, int R, int B, int H, void* dgx, float* dh0, float* dc0,
                 float* dbias, void* workspace, size_t workspace_bytes, int state_batch_major, int dout_of_relu, dic_stream_t stream) {
    DIC_REQUIRE(R > 0 && B > 0, DIC_ERR_INVALID_ARG, "lstm_bwd: non-positive size");
    DIC_REQUIRE(H == LH, DIC_ERR_UNSUPPORTED, "lstm_bwd: hidden size %d (compiled for %d)", H, LH);
    DIC_REQUIRE(whh_t && gates && cs && dgx && dh0 && dc0, DIC_ERR_INVALID_ARG, "lstm_bwd: NULL pointer");
    static const size_t lds = (size_t)(LBM * GSTR + LBM * LH) * sizeof(__bf16);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)lstm_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_bwd: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        attr_set = true;
    }
    const int nwg = (B + LBM - 1) / LBM;
    DIC_REQUIRE(!dbias || (workspace && workspace_bytes >= dic_lstm_bwd_workspace(B)), DIC_ERR_WORKSPACE,
                "lstm_bwd: dbias needs %zu B of workspace", dic_lstm_bwd_workspace(B));
    LstmBwdArgs a{(const __bf16*)whh_t, (const __bf16*)gates, (const __bf16*)cs, c0, (const __bf16*)dout, dhn, dcn, (__bf16*)dgx, dh0, dc0,
                  dbias ? (float*)workspace : nullptr, R, B, state_batch_major != 0, dout_of_relu != 0};
    static const bool eight = [] { const char* e = getenv("DIC_BWD_EIGHT_WAVES"); return !(e && e[0] == '0'); }();
    if (eight) {
        static bool attr8_set = false;
        if (!attr8_set) {
            hipError_t e = hipFuncSetAttribute((const void*)lstm_bwd8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_bwd8: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
            attr8_set = true;
        }
        hipLaunchKernelGGL(lstm_bwd8_kernel, dim3(nwg, 2), dim3(512), lds, (hipStream_t)stream, a);
    } else {
        hipLaunchKernelGGL(lstm_bwd_kernel, dim3(nwg, 2), dim3(256), lds, (hipStream_t)stream, a);
    }
    if (dbias)
        hipLaunchKernelGGL(lstm_dbias_finalize, dim3(2 * 4 * LH / 32), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, nwg,
                           dbias);
    return check_launch("lstm_bwd");
}

}  // extern "C"
