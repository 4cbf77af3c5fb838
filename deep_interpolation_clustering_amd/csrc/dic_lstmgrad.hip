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
constexpr int NW = GH + XW;        // B-operand columns of the fused product: [h_prev | x]
constexpr int TR = 64;             // (t, b) rows per tile = 4 MFMA k-steps
// LDS row pitches (bytes).  A transposed read takes 4 rows x 64 contiguous bytes per 32-lane half: a pitch of 64 (mod 256)
// puts the 4 rows on disjoint 16-bank groups.
constexpr int DG_PITCH = 2 * G4 + 64;     // 1088
constexpr int H_PITCH = 2 * GH + 64;      // 320
constexpr int X_PITCH = 2 * XW;           // 64
constexpr int LDS_DG = TR * DG_PITCH, LDS_H = TR * H_PITCH, LDS_X = TR * X_PITCH;
constexpr int DW_LDS = 2 * LDS_DG + LDS_H + LDS_X;  // 163 840 B = the whole LDS of a CU: two dG buffers (LDS-DMA), one h | x buffer
constexpr int DW_OUT = 2 * G4 * NW;                 // outputs per partial: [dir][gate row][h cols | x cols]

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
__global__ __launch_bounds__(256) void lstm_pack_kernel(LstmParams p, int I, int Ip, int bias_col, __bf16* wih, __bf16* whh,
                                                       __bf16* whh_t, __bf16* bias) {
    const int n_ih = 2 * G4 * Ip, n_hh = 2 * G4 * GH;
    const int total = n_ih + 2 * n_hh + 2 * G4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        if (i < n_ih) {
            const int row = i / Ip, c = i - row * Ip, d = row / G4, g = row - d * G4;
            float v = 0.f;
            if (c < I) v = p.w_ih[d][(size_t)g * I + c];
            else if (c == I && bias_col) v = p.b_ih[d][g] + p.b_hh[d][g];
            wih[i] = (__bf16)v;
        } else if (i < n_ih + n_hh) {
            const int j = i - n_ih, d = j / (G4 * GH), k = j - d * (G4 * GH);
            whh[j] = (__bf16)p.w_hh[d][k];
        } else if (i < n_ih + 2 * n_hh) {
            // whh_t[d][u][n] = whh[d][n][u]: consecutive threads walk n (coalesced writes; the 256-KB source stays in L2)
            const int j = i - n_ih - n_hh, d = j / (G4 * GH), k = j - d * (G4 * GH), u = k / G4, n = k - u * G4;
            if (whh_t) whh_t[j] = (__bf16)p.w_hh[d][(size_t)n * GH + u];
        } else {
            const int j = i - n_ih - 2 * n_hh, d = j / G4, g = j - d * G4;
            if (bias) bias[j] = (__bf16)(p.b_ih[d][g] + p.b_hh[d][g]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ fused dW (encoder)
struct DwArgs {
    const __bf16* dg;      // (R,B,2,4H) gate gradients
    const __bf16* out;     // (R,B,2H)   the layer's own outputs h_t (forward | reverse)
    const __bf16* x;       // (R,B,XW)   packed inputs
    const float* h0;       // (2,B,H) initial hidden state or NULL (zeros)
    float* partials;       // (gridDim.x, 2, 4H, NW) per-workgroup sums
    int R, B;
};

__device__ __forceinline__ s16x4 lds_tr16(const unsigned char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
}

// grid (nch, 2 directions), 256 threads = 4 waves, one workgroup per CU.  Wave w owns gate rows [128w, 128w+128) of its
// direction: 4 m-blocks x 5 n-blocks of 32x32 f32 accumulators (320 VGPRs).  Tiles of 64 (t, b) rows stream through LDS: the dG
// rows (1 KiB each) by LDS-DMA into one of two buffers -- no staging registers, the next tile lands while the matrix cores
// work on the current one -- the narrow h / x rows through registers.
__global__ __launch_bounds__(256, 1) void lstm_dw_kernel(DwArgs a) {
    extern __shared__ __align__(16) unsigned char dwsm[];
    unsigned char* sdg0 = dwsm;
    unsigned char* sh = dwsm + 2 * LDS_DG;
    unsigned char* sx = sh + LDS_H;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int dir = blockIdx.y, B = a.B, R = a.R;
    const int nbt = (B + TR - 1) / TR, ntiles = R * nbt, nch = gridDim.x;

    f32x16 acc[4][5];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 5; ++nb)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[mb][nb][k] = 0.f;

    uint4 rh[4], rx;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
    auto request_dg = [&](int idx, int buf) {      // rows past the batch are clamped: their h / x rows are zero, so they add nothing
        const int t = idx / nbt, b0 = (idx - t * nbt) * TR;
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int row = p * 4 + w;
            const int b = min(b0 + row, B - 1);
            const __bf16* src = a.dg + (((size_t)t * B + b) * 2 + dir) * G4 + lane * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(sdg0 + buf * LDS_DG + row * DG_PITCH), 16, 0, 0);
        }
    };
    auto load_hx = [&](int idx) {
        const int t = idx / nbt, b0 = (idx - t * nbt) * TR;
        const int tp = dir ? t + 1 : t - 1;     // the step whose output fed this step's recurrent product
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = p * 16 + (tid >> 4), ch = tid & 15;
            const int b = b0 + row;
            uint4 v = zero4;
            if (b < B) {
                if (tp >= 0 && tp < R) {
                    v = *reinterpret_cast<const uint4*>(a.out + ((size_t)tp * B + b) * 2 * GH + dir * GH + ch * 8);
                } else if (a.h0) {
                    const float* hp = a.h0 + ((size_t)dir * B + b) * GH + ch * 8;
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(hp), hi = *reinterpret_cast<const f32x4*>(hp + 4);
                    bf16x8 hb;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { hb[j] = (__bf16)lo[j]; hb[4 + j] = (__bf16)hi[j]; }
                    v = *reinterpret_cast<const uint4*>(&hb);
                }
            }
            rh[p] = v;
        }
        {
            const int row = tid >> 2, ch = tid & 3;
            const int b = b0 + row;
            rx = b < B ? *reinterpret_cast<const uint4*>(a.x + ((size_t)t * B + b) * XW + ch * 8) : zero4;
        }
    };
    auto store_hx = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<uint4*>(sh + (p * 16 + (tid >> 4)) * H_PITCH + (tid & 15) * 16) = rh[p];
        *reinterpret_cast<uint4*>(sx + (tid >> 2) * X_PITCH + (tid & 3) * 16) = rx;
    };

    // transposed-read addressing (ds_read_b64_tr_b16): within each 16-lane group, lane 4q+p supplies row q, columns 4p..4p+3 of a
    // 4-row x 16-column block and lane i receives column i of those 4 rows.  For the 32x32x16 operand lane l needs column
    // (l & 31) and rows 8*(l >> 5) + 0..7 of the k-step: group (l >> 4) & 1 takes columns 16.., two reads take rows +0..3, +4..7.
    const int kq = (lane & 15) >> 2, kp = lane & 3, cb = (lane >> 4) & 1, hh = lane >> 5;
    const int rowoff = 8 * hh + kq;
    const unsigned char* pa0 = sdg0 + rowoff * DG_PITCH + (128 * w + 16 * cb + 4 * kp) * 2;
    const unsigned char* ph = sh + rowoff * H_PITCH + (16 * cb + 4 * kp) * 2;
    const unsigned char* px = sx + rowoff * X_PITCH + (16 * cb + 4 * kp) * 2;
    auto frag = [&](const unsigned char* p, int pitch) {
        const s16x4 lo = lds_tr16(p), hi = lds_tr16(p + 4 * pitch);
        s16x8 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = lo[j]; f[4 + j] = hi[j]; }
        return __builtin_bit_cast(bf16x8, f);
    };

    int idx = blockIdx.x, cur = 0;
    if (idx < ntiles) { request_dg(idx, 0); load_hx(idx); }
    while (idx < ntiles) {
        __syncthreads();                       // every wave's DMA of this tile has landed (the barrier waits for vmcnt 0) and the
                                               // previous tile's LDS reads have retired
        store_hx();
        const int next = idx + nch;
        if (next < ntiles) { request_dg(next, cur ^ 1); load_hx(next); }
        __syncthreads();                       // h | x rows visible
        const unsigned char* pa = pa0 + cur * LDS_DG;
#pragma unroll 1                               // (unrolled, the compiler hoists all 72 fragment reads of a tile and spills)
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) af[mb] = frag(pa + ks * 16 * DG_PITCH + mb * 64, DG_PITCH);
#pragma unroll
            for (int nb = 0; nb < 5; ++nb) {
                const bf16x8 bfg = nb < 4 ? frag(ph + ks * 16 * H_PITCH + nb * 64, H_PITCH) : frag(px + ks * 16 * X_PITCH, X_PITCH);
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mb], bfg, acc[mb][nb], 0, 0, 0);
            }
        }
        idx = next;
        cur ^= 1;
    }
    // C/D layout of the 32x32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* o = a.partials + ((size_t)blockIdx.x * 2 + dir) * G4 * NW;
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 5; ++nb)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int m = 128 * w + 32 * mb + (k & 3) + 8 * (k >> 2) + 4 * hh;
                o[(size_t)m * NW + 32 * nb + (lane & 31)] = acc[mb][nb][k];
            }
}

// out_i = sum over workgroups of partial_i (fixed order, f64), scattered into the nn.LSTM gradients: h columns -> weight_hh,
// x columns [0, I) -> weight_ih (column I of the packed input is the constant one: its sum is the bias gradient, which the
// recurrence backward already delivers in f32).  beta = 0 overwrites, 1 accumulates.
__global__ __launch_bounds__(256) void lstm_dw_finalize(const float* partials, int nch, int I, LstmGrads g, float beta) {
    __shared__ double red[256];
    const double s = reduce_partials_32x8(partials, nch, DW_OUT, blockIdx.x * 32, red);
    const int i = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x >= 32 || i >= DW_OUT) return;
    const int d = i / (G4 * NW), rem = i - d * (G4 * NW), m = rem / NW, n = rem - m * NW;
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

static int dw_chunks(int R, int B) {
    const int ntiles = R * ((B + TR - 1) / TR);
    return max(1, min(ntiles, kNumCU / 2));        // x 2 directions = one workgroup per CU
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_lstm_pack(const float* const* params, int H, int I, int Ip, int bias_col, void* wih, void* whh, void* whh_t, void* bias,
                  dic_stream_t stream) {
    DIC_REQUIRE(H == GH, DIC_ERR_UNSUPPORTED, "lstm_pack: hidden size %d (compiled for %d)", H, GH);
    DIC_REQUIRE(params && wih && whh, DIC_ERR_INVALID_ARG, "lstm_pack: NULL pointer");
    DIC_REQUIRE(I > 0 && Ip >= I + (bias_col ? 1 : 0) && Ip <= 1024, DIC_ERR_INVALID_ARG, "lstm_pack: I=%d Ip=%d bias_col=%d", I, Ip, bias_col);
    LstmParams p;
    for (int d = 0; d < 2; ++d) {
        p.w_ih[d] = params[4 * d + 0]; p.w_hh[d] = params[4 * d + 1]; p.b_ih[d] = params[4 * d + 2]; p.b_hh[d] = params[4 * d + 3];
        DIC_REQUIRE(p.w_ih[d] && p.w_hh[d] && p.b_ih[d] && p.b_hh[d], DIC_ERR_INVALID_ARG, "lstm_pack: NULL parameter (direction %d)", d);
    }
    const int total = 2 * G4 * Ip + 4 * G4 * GH + 2 * G4;
    hipLaunchKernelGGL(lstm_pack_kernel, dim3(min((total + 255) / 256, 2 * kNumCU)), dim3(256), 0, (hipStream_t)stream, p, I, Ip, bias_col,
                       (__bf16*)wih, (__bf16*)whh, (__bf16*)whh_t, (__bf16*)bias);
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

size_t dic_lstm_dw_workspace(int R, int B) {
    if (R <= 0 || B <= 0) return 0;
    return (size_t)dw_chunks(R, B) * DW_OUT * sizeof(float);
}

int dic_lstm_dw(const void* dgx, const void* out, const void* x, const float* h0, int R, int B, int H, int I, int Ip,
                float* const* grads, int accumulate, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(R > 0 && B > 0, DIC_ERR_INVALID_ARG, "lstm_dw: non-positive size");
    DIC_REQUIRE(H == GH, DIC_ERR_UNSUPPORTED, "lstm_dw: hidden size %d (compiled for %d)", H, GH);
    DIC_REQUIRE(Ip == XW && I > 0 && I <= XW, DIC_ERR_UNSUPPORTED, "lstm_dw: packed input width %d / %d (compiled for %d)", I, Ip, XW);
    DIC_REQUIRE(dgx && out && x && workspace, DIC_ERR_INVALID_ARG, "lstm_dw: NULL pointer");
    LstmGrads g;
    int rc = grads_from(grads, &g, true, false, "lstm_dw");
    if (rc) return rc;
    const int nch = dw_chunks(R, B);
    DIC_REQUIRE(workspace_bytes >= (size_t)nch * DW_OUT * sizeof(float), DIC_ERR_WORKSPACE, "lstm_dw: workspace %zu < %zu", workspace_bytes,
                (size_t)nch * DW_OUT * sizeof(float));
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)lstm_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DW_LDS);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_dw: cannot reserve %d B of LDS: %s", DW_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    DwArgs a{(const __bf16*)dgx, (const __bf16*)out, (const __bf16*)x, h0, (float*)workspace, R, B};
    hipLaunchKernelGGL(lstm_dw_kernel, dim3(nch, 2), dim3(256), DW_LDS, st, a);
    hipLaunchKernelGGL(lstm_dw_finalize, dim3((DW_OUT + 31) / 32), dim3(256), 0, st, (const float*)workspace, nch, I, g,
                       accumulate ? 1.0f : 0.0f);
    return check_launch("lstm_dw");
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

}  // extern "C"
