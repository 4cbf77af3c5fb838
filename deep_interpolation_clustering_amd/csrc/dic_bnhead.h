// Per-column constants and the in-kernel dropout mask of the CompressFC tail (BatchNorm1d(128) -> ReLU -> Dropout -> Linear(128, C),
// rbf.py:111-125), shared by the streaming kernels of dic_bnhead.hip and the fused backward of dic_fcgrad.hip.
#pragma once
#include "dic_common.h"

namespace dic {

constexpr int BK = 128;       // BatchNorm width / Linear in_features (compiled in)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Dropout between the activation and the Linear (nn.Dropout(p) in CompressFC / the heads, rbf.py:120, clustering_interp.py:51):
// the keep mask is a counter-based hash of (seed, call counter, element index), recomputed in the backward -- no mask tensor.
// One 64-bit murmur3 finaliser yields the decisions of two neighbouring columns (its low and high words).
struct Drop {
    float scale;              // 1 / (1 - p); 1 when dropout is off
    unsigned int thresh;      // keep iff hash >= thresh = p * 2^32; 0 = keep all
    unsigned long long key;   // seed ^ golden-ratio * counter
};
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}
// keep-and-scale factors of the 8 columns kc*8 .. kc*8+7 of `row`
__device__ __forceinline__ void drop_factors(const Drop& d, long row, int kc, float (&f)[8]) {
    if (d.thresh == 0u) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 1.f;
        return;
    }
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) {
        const unsigned long long hsh = mix64(d.key + (unsigned long long)(row * 64 + kc * 4 + e2));      // one hash per column pair
        f[2 * e2] = (unsigned int)hsh >= d.thresh ? d.scale : 0.f;
        f[2 * e2 + 1] = (unsigned int)(hsh >> 32) >= d.thresh ? d.scale : 0.f;
    }
}
__device__ __forceinline__ Drop load_drop(float p, const unsigned long long* rng) {
    Drop d;
    d.scale = 1.f; d.thresh = 0u; d.key = 0ull;
    if (p > 0.f && rng) {
        d.scale = 1.0f / (1.0f - p);
        d.thresh = (unsigned int)fminf(p * 4294967296.0f, 4294967040.0f);
        d.key = rng[0] ^ (rng[1] * 0x9E3779B97F4A7C15ULL);
    }
    return d;
}

struct ColParams {            // per-lane constants for its 8 columns
    float scale[8], shift[8], mean[8], rstd[8];
};

__device__ __forceinline__ ColParams load_cols(const float* mean, const float* rstd, const float* gamma, const float* beta, int kc) {
    ColParams p;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int k = kc * 8 + e;
        p.mean[e] = mean[k];
        p.rstd[e] = rstd[k];
        p.scale[e] = gamma[k] * p.rstd[e];
        p.shift[e] = beta[k] - p.mean[e] * p.scale[e];
    }
    return p;
}

}  // namespace dic
