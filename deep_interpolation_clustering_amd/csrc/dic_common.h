// Shared host/device helpers for the gfx950 kernels of the DIC hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dic_hip.h"

// Streaming (nontemporal) accesses for data a launch touches exactly once in whole cache lines (16 B per lane, or 8 B per lane over a
// contiguous 512-B piece): they would only push each other out of L2.  Partial-line stores must NOT use it (measured 2x slower).
// -DDIC_NO_NT: plain accesses everywhere (A/B builds).
#ifndef DIC_NO_NT
#define DIC_NT_LOAD(T, p) __builtin_nontemporal_load(reinterpret_cast<const T*>(p))
#define DIC_NT_STORE(T, p, v) __builtin_nontemporal_store((v), reinterpret_cast<T*>(p))
#else
#define DIC_NT_LOAD(T, p) (*reinterpret_cast<const T*>(p))
#define DIC_NT_STORE(T, p, v) (*reinterpret_cast<T*>(p) = (v))
#endif

namespace dic {


constexpr int kWave = 64;            // CDNA4 wavefront
constexpr int kBlock = 256;          // default workgroup: 4 waves, one per SIMD
constexpr int kNumCU = 256;          // MI355X
constexpr float kLog2e = 1.4426950408889634f;

// host side ---------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_launch(const char* what);

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// dic_kmeans_mfma.hip: E-step + partial M-step of a Lloyd iteration on the matrix cores, for 8 < K <= 32
int kmeans_mfma_blocks(int N, int K, int n_runs);
bool kmeans_use_mfma(int K, int n_runs);
int kmeans_assign_mfma_launch(const float* X, const float* xnorm, int N, int D, int K, int n_runs, const float* centers, int32_t* labels,
                              const float* status, float* mind, float* psum, int* pcnt, hipStream_t st);

#define DIC_REQUIRE(cond, code, ...)                \
    do {                                            \
        if (!(cond)) {                              \
            ::dic::set_error(__VA_ARGS__);          \
            return (code);                          \
        }                                           \
    } while (0)

// device side -------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// 2^x on the transcendental unit (v_exp_f32).  Callers pass x <= 0 (max-subtracted logits),
// so the missing denormal handling only flushes weights below 2^-126 to zero.
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// e^x and ln x on the transcendental unit (one multiply + v_exp_f32 / v_log_f32 + one multiply: ~1 ulp of the hardware op plus the
// rounding of the scaled argument, i.e. a relative error of |x| * 6e-8 for e^x) instead of libm's ~25-instruction sequences.  Used
// where the argument is O(10) at most and the result feeds 3e-5-tolerance outputs: the cross-channel epilogue of k1 spends as many
// vector instructions in expf / logf as a third of its streaming passes.  Arguments of fast_log are >= 1 there (sums whose largest
// term is 1), so its missing denormal handling never shows; e^-inf = 0, NaN propagates.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * kLog2e); }
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }

// log(1 + e^k): the positive bandwidth upstream derives from its raw `kernel` parameter
// (interpolation_layer.py:51, rbf.py:78).
__device__ __forceinline__ float softplus_raw(float k) { return logf(1.0f + expf(k)); }
__device__ __forceinline__ float sigmoidf(float k) { return 1.0f / (1.0f + expf(-k)); }

// Fixed-order (deterministic) second-stage reduction: out_i = sum_b partials[b*n + i].
// Call from a 256-thread workgroup; it covers outputs i0 .. i0+31 as W outputs x 256/W slices of b, W = 32 (coalesced 128-B reads
// per slice row), or W = 8 / 16 when there are no more outputs than that (then i0 = 0): a handful of outputs over thousands of
// partial rows is a chain of dependent loads, and 32 / 16 slices shorten it 4x / 2x (rbf_bwd's 6 outputs: 20 -> 6 us).
// The result is valid in threads 0..31 (for i = i0 + tid).
template <typename T>
__device__ __forceinline__ double reduce_partials_32x8(const T* partials, int nblk, int n, int i0, double* lds256) {
    const int W = n <= 8 ? 8 : (n <= 16 ? 16 : 32), nsl = 256 / W;
    const int o = threadIdx.x & (W - 1), sl = threadIdx.x / W, i = i0 + o;
    double acc = 0.0;
    if (i < n) {
        // eight independent chains so that eight loads are in flight per thread (the loop is latency-bound otherwise: a few dozen
        // workgroups on the whole chip, every load a trip to HBM)
        double ch[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) ch[k] = 0.0;
        int b = sl;
        for (; b + 7 * nsl < nblk; b += 8 * nsl) {
            T v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = partials[(size_t)(b + k * nsl) * n + i];
#pragma unroll
            for (int k = 0; k < 8; ++k) ch[k] += (double)v[k];
        }
        for (; b < nblk; b += nsl) ch[0] += (double)partials[(size_t)b * n + i];
        acc = ((ch[0] + ch[1]) + (ch[2] + ch[3])) + ((ch[4] + ch[5]) + (ch[6] + ch[7]));
    }
    lds256[threadIdx.x] = acc;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x < W)
        for (int k = 0; k < nsl; ++k) r += lds256[k * W + threadIdx.x];
    return r;
}

// Per-channel sums over a tile: returns sum over (e, r) of buf[(e*C + c)*R + r] for channel c = tid / LPC in EVERY lane of
// that channel's group (LPC = 32 lanes per channel for C <= 8, 16 for C <= 16; lanes beyond C*LPC return 0).  Call from all
// 256 threads.  Replaces `if (tid < C) for e, r: s += ...` -- a serial chain of Ev*R dependent LDS reads on C threads that cost
// as much as the tile's whole compute phase.  Fixed summation order.
__device__ __forceinline__ int channel_group_lanes(int C) { return C <= 8 ? 32 : 16; }
__device__ __forceinline__ float channel_sum_er(const float* buf, int Ev, int C, int R, int tid) {
    const int lpc = channel_group_lanes(C);
    const int c = tid / lpc, j = tid - c * lpc;
    float s = 0.f;
    if (c < C)
        for (int i = j; i < Ev * R; i += lpc) {
            const int e = i / R, r = i - e * R;
            s += buf[(e * C + c) * R + r];
        }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1)
        if (m < lpc) s += __shfl_xor(s, m);
    return s;
}

// A ragged encounter store in HBM (the device-resident form of dataloader.py:54-79's feed_data): for every
// (encounter, channel) row only the observed samples, packed back to back.  `enc_idx` maps the rows of a batch to store encounters
// (NULL: the batch is encounters 0..B-1), so a shuffled batch is read in place -- no gather pass, no padded (B,4C,T) copy.
// The packed arrays carry >= 64 readable elements behind the last row (clamped look-ahead reads stay in bounds).
struct StoreSrc {
    const float* t_pk;               // time stamps (h)
    const float* v_pk;               // values: ob * mask, rescaled
    const unsigned char* hold_pk;    // optional hold-out flags (plane 3): a denoising step feeds v * hold to the model, the target stays v
    const int64_t* row_off;          // (N*C + 1)
    const int32_t* enc_idx;          // (B) or NULL
};
__device__ __forceinline__ int64_t store_row_off(const StoreSrc& s, int e, int c, int C) {
    return s.row_off[(size_t)(s.enc_idx ? s.enc_idx[e] : e) * C + c];
}

// XCD-aware block remap (8 XCDs, round-robin dispatch): consecutive logical tiles land on the
// same XCD so neighbouring rows share that XCD's L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

}  // namespace dic
