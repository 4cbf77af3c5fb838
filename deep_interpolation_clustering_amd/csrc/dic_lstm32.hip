// Recurrence kernels of the bidirectional LSTM (hidden 128; clustering_interp.py:14-41, nn.LSTM semantics, gate order i,f,g,o)
// for the two regimes the 64-row software-pipelined bf16 kernels of dic_lstm.hip do not serve:
//
//   T = float   the f32 step -- the configuration of the 1e-5 parity tests and of the reference itself.  Round 1 left it on
//               MIOpen's nn.LSTM, ~90 % of a 25 ms step at B = 4096 (profiles/r1_step_f32_B4096_kernel_stats.csv).  Here the
//               recurrent product runs on v_mfma_f32_32x32x2_f32: exact f32 (a k-ordered fmaf chain, one rounding per product),
//               at the f32 vector rate, with the wave's W_hh slice (4 gates x 32 units x 128 = 256 f32 registers per lane)
//               resident for all R steps -- no per-step weight traffic, no per-step launches.
//   T = bf16    small batches (the reference's own B = 256, p1_pretrain_main.py:43): the 64-row kernels put B/64 x 2 workgroups on
//               256 CUs and walk two 32-row halves per step; these take ONE 32-row tile per workgroup, so twice the workgroups
//               each finish a step in about half the time.
//
// One 256-thread workgroup owns 32 batch rows of one direction for the whole sequence.  The MFMA is issued transposed, as in
// dic_lstm.hip: D[gate unit][batch] = W[gate unit][k] . h^T[k][batch], so wave w's A operand is the W_hh slice of hidden units
// [32w, 32w+32) and the four gates of a unit land in the same lane and register: the gate math is register-local.
// Saved state is plain row-major: gates (R,B,2,4,H) and cell states (R,B,2,H) in T -- the layouts the GEMMs / tests read.
#include "dic_common.h"

namespace dic {

constexpr int SH = 128;             // hidden size
constexpr int S4 = 4 * SH;
constexpr int SROWS = 32;           // batch rows per workgroup

typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sbf16x4 __attribute__((ext_vector_type(4)));
typedef float sf32x16 __attribute__((ext_vector_type(16)));
typedef float sf32x4 __attribute__((ext_vector_type(4)));
typedef float sf32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ size_t sstate_off(int bm, int dir, int b, int B) { return (bm ? (size_t)b * 2 + dir : (size_t)dir * B + b) * SH; }
// gate non-linearities: full-precision library forms for the f32 step (it is bound by its 256 f32 MFMAs per step, not by these),
// v_exp_f32 / v_rcp_f32 forms for bf16 operands (as dic_lstm.hip)
template <typename T> __device__ __forceinline__ float sigmoid_acc(float x);
template <typename T> __device__ __forceinline__ float tanh_acc(float x);
template <> __device__ __forceinline__ float sigmoid_acc<float>(float x) { return 1.0f / (1.0f + expf(-x)); }
template <> __device__ __forceinline__ float tanh_acc<float>(float x) { return tanhf(x); }
template <> __device__ __forceinline__ float sigmoid_acc<__bf16>(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp2(-kLog2e * x)); }
template <> __device__ __forceinline__ float tanh_acc<__bf16>(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + fast_exp2(2.0f * kLog2e * x)); }

template <typename T> struct Rec;
// ---- f32: v_mfma_f32_32x32x2_f32.  MFMA n = 2m + e multiplies k = 4m + 2 (lane >> 5) + e: a lane's two operands of an MFMA pair
// are adjacent floats (one 8-byte load for A at setup, one ds_read_b64 for B per pair).
template <> struct Rec<float> {
    static constexpr int PITCH(int K) { return K + 2; }       // LDS row pitch in elements: == 2 (mod 64) words -> conflict-free b64 reads over 32 rows
    template <int K> struct Frag { float v[K / 2]; };       // [m*2 + e], m < K/4
    template <int K> __device__ static void load_a(Frag<K>& f, const float* row, int hh, int stride) {
#pragma unroll
        for (int m = 0; m < K / 4; ++m) {
            f.v[2 * m] = row[(size_t)(4 * m + 2 * hh) * stride];
            f.v[2 * m + 1] = row[(size_t)(4 * m + 2 * hh + 1) * stride];
        }
    }
    template <int K> __device__ static sf32x16 mma(const Frag<K>& a, const float* brow, int hh, sf32x16 acc) {
#pragma unroll
        for (int m = 0; m < K / 4; ++m) {
            const sf32x2 b = *reinterpret_cast<const sf32x2*>(brow + 4 * m + 2 * hh);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[2 * m], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[2 * m + 1], b[1], acc, 0, 0, 0);
            if ((m & 7) == 7) asm volatile("" ::: "memory");      // keeps the scheduler from hoisting all K/4 LDS reads at once (spills)
        }
        return acc;
    }
};
// ---- bf16: v_mfma_f32_32x32x16_bf16, k = 16 ks + 8 (lane >> 5) + j
template <> struct Rec<__bf16> {
    static constexpr int PITCH(int K) { return K + 8; }       // 272-B / 1040-B rows: conflict-free ds_read_b128
    template <int K> struct Frag { sbf16x8 v[K / 16]; };
    template <int K> __device__ static void load_a(Frag<K>& f, const __bf16* row, int hh, int stride) {
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks) {
            if (stride == 1) {
                f.v[ks] = *reinterpret_cast<const sbf16x8*>(row + ks * 16 + 8 * hh);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) f.v[ks][j] = row[(size_t)(ks * 16 + 8 * hh + j) * stride];
            }
        }
    }
    template <int K> __device__ static sf32x16 mma(const Frag<K>& a, const __bf16* brow, int hh, sf32x16 acc) {
#pragma unroll
        for (int ks = 0; ks < K / 16; ++ks)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v[ks], *reinterpret_cast<const sbf16x8*>(brow + ks * 16 + 8 * hh), acc, 0, 0, 0);
        return acc;
    }
};

template <typename T> struct Vec4;
template <> struct Vec4<float> { typedef sf32x4 type; };
template <> struct Vec4<__bf16> { typedef sbf16x4 type; };

template <typename T>
struct RecFwdArgs {
    const T* gx;           // (R,B,2,4,H) input projection + both biases
    const T* whh;          // (2,4H,H)
    const float* h0; const float* c0;      // state layout per `bm`, or NULL
    T* out;                // (R,B,2H)
    float* hn; float* cn;
    T* gates;              // (R,B,2,4,H) post-activation i,f,g,o or NULL
    T* cs;                 // (R,B,2,H) cell states or NULL
    int R, B, bm;
};

template <typename T>
__global__ __launch_bounds__(256, 1) void lstm_rec_fwd_kernel(RecFwdArgs<T> a) {
    typedef typename Vec4<T>::type V4;
    constexpr int HP = Rec<T>::PITCH(SH);
    __shared__ __align__(16) T hbuf[2][SROWS * HP];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int dir = blockIdx.y, b0 = blockIdx.x * SROWS, B = a.B, R = a.R;
    const int b = b0 + r;
    const bool ok = b < B;
    const int bc = min(b, B - 1);

    typename Rec<T>::template Frag<SH> wf[4];            // this wave's W_hh rows: gate g, hidden units 32w + (lane & 31)
#pragma unroll
    for (int g = 0; g < 4; ++g) Rec<T>::template load_a<SH>(wf[g], a.whh + ((size_t)(dir * 4 + g) * SH + 32 * w + r) * SH, hh, 1);

    // lane owns batch row b and hidden units u(q) = 32w + 8q + 4hh + {0..3}, q = 0..3 (accumulator register k = 4q + j)
    float c[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int u = 32 * w + 8 * q + 4 * hh;
        sf32x4 hv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            if (a.h0) hv = *reinterpret_cast<const sf32x4*>(a.h0 + sstate_off(a.bm, dir, b, B) + u);
            if (a.c0) cv = *reinterpret_cast<const sf32x4*>(a.c0 + sstate_off(a.bm, dir, b, B) + u);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { hbuf[0][r * HP + u + j] = (T)hv[j]; c[4 * q + j] = cv[j]; }
    }
    // the step's input projection arrives directly in the accumulator layout: 16 loads of 4 adjacent units per lane
    V4 gnext[4][4];
    auto load_gx = [&](int step) {
        const int t = dir ? R - 1 - step : step;
        const T* base = a.gx + (((size_t)t * B + bc) * 2 + dir) * S4 + 32 * w + 4 * hh;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q) gnext[g][q] = *reinterpret_cast<const V4*>(base + g * SH + 8 * q);
    };
    load_gx(0);
    __syncthreads();

    for (int step = 0; step < R; ++step) {
        const int t = dir ? R - 1 - step : step;
        const int cur = step & 1;
        sf32x16 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[g][4 * q + j] = (float)gnext[g][q][j];
        if (step + 1 < R) load_gx(step + 1);             // in flight across the MFMAs and the gate math
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = Rec<T>::template mma<SH>(wf[g], &hbuf[cur][r * HP], hh, acc[g]);
        const bool last = step == R - 1;
        const size_t row = (size_t)t * B + bc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = 32 * w + 8 * q + 4 * hh;
            V4 hb, ib, fb, gb, ob, cb;
            sf32x4 cv, hv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * q + j;
                const float ig = sigmoid_acc<T>(acc[0][k]), fg = sigmoid_acc<T>(acc[1][k]), gg = tanh_acc<T>(acc[2][k]), og = sigmoid_acc<T>(acc[3][k]);
                const float cn = fmaf(fg, c[k], ig * gg);
                const float hn = og * tanh_acc<T>(cn);
                c[k] = cn;
                cv[j] = cn; hv[j] = hn;
                hb[j] = (T)hn; ib[j] = (T)ig; fb[j] = (T)fg; gb[j] = (T)gg; ob[j] = (T)og; cb[j] = (T)cn;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) hbuf[cur ^ 1][r * HP + u + j] = hb[j];
            if (ok) {
                *reinterpret_cast<V4*>(a.out + row * 2 * SH + dir * SH + u) = hb;
                if (a.gates) {
                    T* gp = a.gates + (row * 2 + dir) * S4 + u;
                    *reinterpret_cast<V4*>(gp) = ib;
                    *reinterpret_cast<V4*>(gp + SH) = fb;
                    *reinterpret_cast<V4*>(gp + 2 * SH) = gb;
                    *reinterpret_cast<V4*>(gp + 3 * SH) = ob;
                    *reinterpret_cast<V4*>(a.cs + (row * 2 + dir) * SH + u) = cb;
                }
                if (last) {
                    *reinterpret_cast<sf32x4*>(a.hn + sstate_off(a.bm, dir, b, B) + u) = hv;
                    *reinterpret_cast<sf32x4*>(a.cn + sstate_off(a.bm, dir, b, B) + u) = cv;
                }
            }
        }
        __syncthreads();
    }
}

template <typename T>
struct RecBwdArgs {
    const T* whh;          // (2,4H,H) -- read transposed (strided) once at start-up;  or whh_t (2,H,4H) when `transposed`
    const T* gates;        // (R,B,2,4,H)
    const T* cs;           // (R,B,2,H)
    const float* c0;
    const T* dout;         // (R,B,2H) or NULL
    const float* dhn; const float* dcn;
    T* dgx;                // (R,B,2,4,H)
    float* dh0; float* dc0;
    float* dbias_part;     // (gridDim.x, 2, 4H) or NULL
    int R, B, bm, transposed;
};

template <typename T>
__global__ __launch_bounds__(256, 1) void lstm_rec_bwd_kernel(RecBwdArgs<T> a) {
    typedef typename Vec4<T>::type V4;
    constexpr int GP = Rec<T>::PITCH(S4);
    extern __shared__ __align__(16) unsigned char rsm[];
    T* dgt = reinterpret_cast<T*>(rsm);                 // [32][GP] gate gradients of the current step
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int dir = blockIdx.y, b0 = blockIdx.x * SROWS, B = a.B, R = a.R;
    const int b = b0 + r;
    const bool ok = b < B;
    const int bc = min(b, B - 1);

    // A operand: row u = 32w + (lane & 31) of W_hh^T over all 4H gate columns
    typename Rec<T>::template Frag<S4> wt;
    if (a.transposed) Rec<T>::template load_a<S4>(wt, a.whh + ((size_t)dir * SH + 32 * w + r) * S4, hh, 1);
    else Rec<T>::template load_a<S4>(wt, a.whh + (size_t)dir * S4 * SH + 32 * w + r, hh, SH);

    sf32x16 dh;
    float dc[16], ccar[16];
    float bsum0 = 0.f, bsum1 = 0.f;                      // bias gradient: columns tid and tid + 256 of dG, summed over rows and steps
    {
        const int t0 = dir ? 0 : R - 1;                  // first visited step = last forward step
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = 32 * w + 8 * q + 4 * hh;
            sf32x4 hv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                if (a.dhn) hv = *reinterpret_cast<const sf32x4*>(a.dhn + sstate_off(a.bm, dir, b, B) + u);
                if (a.dcn) cv = *reinterpret_cast<const sf32x4*>(a.dcn + sstate_off(a.bm, dir, b, B) + u);
            }
            const V4 ct = *reinterpret_cast<const V4*>(a.cs + (((size_t)t0 * B + bc) * 2 + dir) * SH + u);
#pragma unroll
            for (int j = 0; j < 4; ++j) { dh[4 * q + j] = hv[j]; dc[4 * q + j] = cv[j]; ccar[4 * q + j] = (float)ct[j]; }
        }
    }
    for (int step = 0; step < R; ++step) {
        const int t = dir ? step : R - 1 - step;           // reverse of the forward visiting order
        const bool first_fwd = step == R - 1;
        const int tp = dir ? t + 1 : t - 1;                 // forward predecessor
        const size_t row = (size_t)t * B + bc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = 32 * w + 8 * q + 4 * hh;
            const T* gp = a.gates + (row * 2 + dir) * S4 + u;
            const V4 ib = *reinterpret_cast<const V4*>(gp), fb = *reinterpret_cast<const V4*>(gp + SH);
            const V4 gb = *reinterpret_cast<const V4*>(gp + 2 * SH), ob = *reinterpret_cast<const V4*>(gp + 3 * SH);
            sf32x4 cp = {0.f, 0.f, 0.f, 0.f};
            if (!first_fwd) {
                const V4 cpv = *reinterpret_cast<const V4*>(a.cs + (((size_t)tp * B + bc) * 2 + dir) * SH + u);
#pragma unroll
                for (int j = 0; j < 4; ++j) cp[j] = (float)cpv[j];
            } else if (a.c0 && ok) {
                const sf32x4 c0v = *reinterpret_cast<const sf32x4*>(a.c0 + sstate_off(a.bm, dir, b, B) + u);
#pragma unroll
                for (int j = 0; j < 4; ++j) cp[j] = (float)(T)c0v[j];
            }
            sf32x4 go = {0.f, 0.f, 0.f, 0.f};
            if (a.dout) {
                const V4 gov = *reinterpret_cast<const V4*>(a.dout + row * 2 * SH + dir * SH + u);
#pragma unroll
                for (int j = 0; j < 4; ++j) go[j] = (float)gov[j];
            }
            V4 di, df, dg, dO;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 4 * q + j;
                const float ig = (float)ib[j], fg = (float)fb[j], gg = (float)gb[j], og = (float)ob[j];
                const float dht = dh[k] + go[j];
                const float tc = tanh_acc<T>(ccar[k]);
                const float dct = fmaf(dht * og, 1.0f - tc * tc, dc[k]);
                const float vi = dct * gg * ig * (1.0f - ig), vf = dct * cp[j] * fg * (1.0f - fg);
                const float vg = dct * ig * (1.0f - gg * gg), vo = dht * tc * og * (1.0f - og);
                const bool live = ok;
                di[j] = (T)(live ? vi : 0.f); df[j] = (T)(live ? vf : 0.f); dg[j] = (T)(live ? vg : 0.f); dO[j] = (T)(live ? vo : 0.f);
                dc[k] = dct * fg;
                ccar[k] = cp[j];
            }
            T* lp = dgt + r * GP + u;
#pragma unroll
            for (int j = 0; j < 4; ++j) { lp[j] = di[j]; lp[SH + j] = df[j]; lp[2 * SH + j] = dg[j]; lp[3 * SH + j] = dO[j]; }
            if (ok) {
                T* op = a.dgx + (row * 2 + dir) * S4 + u;
                *reinterpret_cast<V4*>(op) = di;
                *reinterpret_cast<V4*>(op + SH) = df;
                *reinterpret_cast<V4*>(op + 2 * SH) = dg;
                *reinterpret_cast<V4*>(op + 3 * SH) = dO;
            }
        }
        __syncthreads();                                   // the dG tile of this step is complete
        if (a.dbias_part) {
#pragma unroll 8
            for (int i = 0; i < SROWS; ++i) { bsum0 += (float)dgt[i * GP + tid]; bsum1 += (float)dgt[i * GP + 256 + tid]; }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) dh[k] = 0.f;
        dh = Rec<T>::template mma<S4>(wt, dgt + r * GP, hh, dh);          // dh_{prev}[u][b] = sum_n W_hh[n][u] dG[b][n]
        __syncthreads();                                   // every wave is done reading the tile
    }
    if (a.dbias_part) {
        float* o = a.dbias_part + ((size_t)blockIdx.x * 2 + dir) * S4;
        o[tid] = bsum0;
        o[256 + tid] = bsum1;
    }
    if (ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u = 32 * w + 8 * q + 4 * hh;
            sf32x4 hv, cv;
#pragma unroll
            for (int j = 0; j < 4; ++j) { hv[j] = dh[4 * q + j]; cv[j] = dc[4 * q + j]; }
            *reinterpret_cast<sf32x4*>(a.dh0 + sstate_off(a.bm, dir, b, B) + u) = hv;
            *reinterpret_cast<sf32x4*>(a.dc0 + sstate_off(a.bm, dir, b, B) + u) = cv;
        }
    }
}

__global__ __launch_bounds__(256) void lstm_rec_dbias_finalize(const float* partials, int nblk, float* dbias) {
    __shared__ double red[256];
    const int n = 2 * S4;
    const double s = reduce_partials_32x8(partials, nblk, n, blockIdx.x * 32, red);
    const int i = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x < 32 && i < n) dbias[i] = (float)s;
}

template <typename T>
static int rec_fwd(const void* gx, const void* whh, const float* h0, const float* c0, int R, int B, void* out, float* hn, float* cn,
                   void* gates, void* cs, int bm, hipStream_t st) {
    RecFwdArgs<T> a{(const T*)gx, (const T*)whh, h0, c0, (T*)out, hn, cn, (T*)gates, (T*)cs, R, B, bm != 0};
    hipLaunchKernelGGL(lstm_rec_fwd_kernel<T>, dim3((B + SROWS - 1) / SROWS, 2), dim3(256), 0, st, a);
    return check_launch("lstm_rec_fwd");
}

template <typename T>
static int rec_bwd(const void* whh, int transposed, const void* gates, const void* cs, const float* c0, const void* dout, const float* dhn,
                   const float* dcn, int R, int B, void* dgx, float* dh0, float* dc0, float* dbias, void* workspace, int bm, hipStream_t st) {
    const size_t lds = (size_t)SROWS * Rec<T>::PITCH(S4) * sizeof(T);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)lstm_rec_bwd_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "lstm_rec_bwd: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        attr_set = true;
    }
    const int nwg = (B + SROWS - 1) / SROWS;
    RecBwdArgs<T> a{(const T*)whh, (const T*)gates, (const T*)cs, c0, (const T*)dout, dhn, dcn, (T*)dgx, dh0, dc0,
                    dbias ? (float*)workspace : nullptr, R, B, bm != 0, transposed};
    hipLaunchKernelGGL(lstm_rec_bwd_kernel<T>, dim3(nwg, 2), dim3(256), lds, st, a);
    if (dbias) hipLaunchKernelGGL(lstm_rec_dbias_finalize, dim3(2 * S4 / 32), dim3(256), 0, st, (const float*)workspace, nwg, dbias);
    return check_launch("lstm_rec_bwd");
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_lstm_rec_fwd(int dtype, const void* gx, const void* whh, const float* h0, const float* c0, int R, int B, int H, void* out,
                     float* hn, float* cn, void* gates, void* cs, int state_batch_major, dic_stream_t stream) {
    DIC_REQUIRE(R > 0 && B > 0, DIC_ERR_INVALID_ARG, "lstm_rec_fwd: non-positive size");
    DIC_REQUIRE(H == SH, DIC_ERR_UNSUPPORTED, "lstm_rec_fwd: hidden size %d (compiled for %d)", H, SH);
    DIC_REQUIRE(dtype == DIC_DTYPE_F32 || dtype == DIC_DTYPE_BF16, DIC_ERR_INVALID_ARG, "lstm_rec_fwd: dtype %d", dtype);
    DIC_REQUIRE(gx && whh && out && hn && cn, DIC_ERR_INVALID_ARG, "lstm_rec_fwd: NULL pointer");
    DIC_REQUIRE((gates == nullptr) == (cs == nullptr), DIC_ERR_INVALID_ARG, "lstm_rec_fwd: gates and cs go together");
    if (dtype == DIC_DTYPE_F32) return rec_fwd<float>(gx, whh, h0, c0, R, B, out, hn, cn, gates, cs, state_batch_major, (hipStream_t)stream);
    return rec_fwd<__bf16>(gx, whh, h0, c0, R, B, out, hn, cn, gates, cs, state_batch_major, (hipStream_t)stream);
}

size_t dic_lstm_rec_bwd_workspace(int B) { return B > 0 ? (size_t)((B + SROWS - 1) / SROWS) * 2 * S4 * sizeof(float) : 0; }

int dic_lstm_rec_bwd(int dtype, const void* whh, int whh_is_transposed, const void* gates, const void* cs, const float* c0, const void* dout,
                     const float* dhn, const float* dcn, int R, int B, int H, void* dgx, float* dh0, float* dc0, float* dbias,
                     void* workspace, size_t workspace_bytes, int state_batch_major, dic_stream_t stream) {
    DIC_REQUIRE(R > 0 && B > 0, DIC_ERR_INVALID_ARG, "lstm_rec_bwd: non-positive size");
    DIC_REQUIRE(H == SH, DIC_ERR_UNSUPPORTED, "lstm_rec_bwd: hidden size %d (compiled for %d)", H, SH);
    DIC_REQUIRE(dtype == DIC_DTYPE_F32 || dtype == DIC_DTYPE_BF16, DIC_ERR_INVALID_ARG, "lstm_rec_bwd: dtype %d", dtype);
    DIC_REQUIRE(whh && gates && cs && dgx && dh0 && dc0, DIC_ERR_INVALID_ARG, "lstm_rec_bwd: NULL pointer");
    DIC_REQUIRE(!dbias || (workspace && workspace_bytes >= dic_lstm_rec_bwd_workspace(B)), DIC_ERR_WORKSPACE,
                "lstm_rec_bwd: dbias needs %zu B of workspace", dic_lstm_rec_bwd_workspace(B));
    if (dtype == DIC_DTYPE_F32)
        return rec_bwd<float>(whh, whh_is_transposed, gates, cs, c0, dout, dhn, dcn, R, B, dgx, dh0, dc0, dbias, workspace, state_batch_major, (hipStream_t)stream);
    return rec_bwd<__bf16>(whh, whh_is_transposed, gates, cs, c0, dout, dhn, dcn, R, B, dgx, dh0, dc0, dbias, workspace, state_batch_major, (hipStream_t)stream);
}

}  // extern "C"
