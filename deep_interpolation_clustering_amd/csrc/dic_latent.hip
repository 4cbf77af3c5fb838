// k3 / k4: latent-row vs centroid kernels -- DEC Student-t soft assignment (fwd/bwd, KL) and the
// k-means Lloyd iteration (assign + partial sums, update, relocation, convergence), k-means
// prediction and the k-means++ candidate step.
//
// Replaces ClusterAssignment.forward / target_distribution (dec.py:49-76), Net.kl_loss
// (clustering_interp.py:205-207) and the scikit-learn KMeans calls at clustering_trainer.py:75-82,
// p2_clustering_optK.py:260-389, p4_clustering_final.py:159-174 (sklearn 1.7.2
// _k_means_lloyd.pyx:23-218, _k_means_common.pyx:167-311, _kmeans.py:218-243,624-753).
//
// Shared structure (HBM-bound integer/f32 streaming, no MFMA): ONE WAVE PER LATENT ROW.  A 256-wide
// f32 row is exactly one 1 KiB wave-wide float4 load; the K centroids live in LDS and are read as
// float4 (all lanes of a wave read consecutive 16-B slots: conflict-free).  The K per-lane partial
// distances are reduced across the wave with a value-halving butterfly (about K + 6 shuffles
// instead of 6 K), after which lane l holds the distance to centroid l >> (6 - log2 KP).
// Reductions across rows (column sums, centroid sums, dL/dmu) are per-lane register accumulators,
// combined per workgroup through LDS and across workgroups by a fixed-order second stage
// (deterministic: no float atomics).
#include "dic_common.h"

namespace dic {

constexpr int kLatBlock = 256;
constexpr int kLatWaves = kLatBlock / kWave;

// value-halving butterfly: v[0..N) per lane -> v[0] = sum over the wave of v[kidx],
// kidx = lane >> (6 - log2 N).  MASK starts at 32.
template <int N, int MASK>
__device__ __forceinline__ void fold(float* v, int lane) {
    if constexpr (N > 1) {
        constexpr int H = N / 2;
        const bool hi = (lane & MASK) != 0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const float send = hi ? v[i] : v[i + H];
            const float keep = hi ? v[i + H] : v[i];
            v[i] = keep + __shfl_xor(send, MASK);
        }
        fold<H, MASK / 2>(v, lane);
    } else {
#pragma unroll
        for (int m = MASK; m >= 1; m >>= 1) v[0] += __shfl_xor(v[0], m);
    }
}

template <int KP> struct KLog2;
template <> struct KLog2<2> { static constexpr int v = 1; };
template <> struct KLog2<4> { static constexpr int v = 2; };
template <> struct KLog2<8> { static constexpr int v = 3; };
template <> struct KLog2<16> { static constexpr int v = 4; };
template <> struct KLog2<32> { static constexpr int v = 5; };

__device__ __forceinline__ float4 load_row4(const float* X, long row, int D, int lane) {
    if (lane * 4 < D) return *reinterpret_cast<const float4*>(X + row * D + lane * 4);
    return make_float4(0.f, 0.f, 0.f, 0.f);
}

// stage K centroid rows (D wide) into LDS rows of 256 floats, zero-padded to KP rows
template <int KP>
__device__ __forceinline__ void stage_centers(float* cl, const float* centers, int K, int D) {
    for (int i = threadIdx.x; i < KP * 256; i += blockDim.x) {
        const int k = i >> 8, d = i & 255;
        cl[i] = (k < K && d < D) ? centers[k * D + d] : 0.f;
    }
}

// =============================================================================== DEC forward
template <int KP>
__global__ __launch_bounds__(kLatBlock) void dec_fwd_kernel(const float* z, const float* centers, int B, int D, int K,
                                                           float alpha, float* q, float* tsaved, float* partials) {
    __shared__ __align__(16) float cl[KP * 256];
    __shared__ float colred[kLatWaves][KP];
    constexpr int LG = KLog2<KP>::v;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_centers<KP>(cl, centers, K, D);
    __syncthreads();
    const int kidx = lane >> (6 - LG);
    const bool writer = (lane & ((64 >> LG) - 1)) == 0 && kidx < K;
    const float power = 0.5f * (alpha + 1.0f), inv_alpha = 1.0f / alpha;
    float colacc = 0.f;
    const long nw = (long)gridDim.x * kLatWaves;
    for (long row = (long)blockIdx.x * kLatWaves + wave; row < B; row += nw) {
        const float4 x = load_row4(z, row, D, lane);
        float v[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const float4 c = *reinterpret_cast<const float4*>(cl + k * 256 + lane * 4);
            const float d0 = x.x - c.x, d1 = x.y - c.y, d2 = x.z - c.z, d3 = x.w - c.w;
            v[k] = fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, d0 * d0)));
        }
        fold<KP, 32>(v, lane);
        const float t = 1.0f / (1.0f + v[0] * inv_alpha);                     // dec.py:57
        float n = (alpha == 1.0f) ? t : exp2f(power * log2f(t));               // dec.py:58-60
        if (kidx >= K) n = 0.f;
        float s = n;
#pragma unroll
        for (int m = 32; m >= (64 >> LG); m >>= 1) s += __shfl_xor(s, m);      // sum over the K distinct kidx
        const float qv = n / s;                                                // dec.py:61
        if (writer) {
            q[row * K + kidx] = qv;
            if (tsaved) tsaved[row * K + kidx] = t;
        }
        colacc += qv;
    }
    if ((lane & ((64 >> LG) - 1)) == 0) colred[wave][kidx] = colacc;
    __syncthreads();
    if (partials && threadIdx.x < K) {
        float s = 0.f;
        for (int w = 0; w < kLatWaves; ++w) s += colred[w][threadIdx.x];
        partials[(size_t)blockIdx.x * K + threadIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void colsum_finalize(const float* partials, int nblk, int K, float* colsum) {
    __shared__ double red[256];
    const double s = reduce_partials_32x8(partials, nblk, K, 0, red);
    if (threadIdx.x < K) colsum[threadIdx.x] = (float)s;
}

// target_distribution (dec.py:73-74): p = (q^2/f) / sum_j (q^2/f)
__global__ void dec_target_kernel(const float* q, const float* colsum, int B, int K, float* p) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= B) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) { const float qq = q[row * K + k]; s += qq * qq / colsum[k]; }
    for (int k = 0; k < K; ++k) { const float qq = q[row * K + k]; p[row * K + k] = (qq * qq / colsum[k]) / s; }
}

// =============================================================================== DEC backward
template <int KP>
__global__ __launch_bounds__(kLatBlock) void dec_bwd_kernel(const float* z, const float* centers, const float* tsaved,
                                                           const float* grad_q, int B, int D, int K, float alpha,
                                                           float* grad_z, float* partials) {
    __shared__ __align__(16) float cl[KP * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_centers<KP>(cl, centers, K, D);
    __syncthreads();
    const float power = 0.5f * (alpha + 1.0f);
    const float dn_scale = -(alpha + 1.0f) / (2.0f * alpha);
    float4 acc[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const long nw = (long)gridDim.x * kLatWaves;
    for (long row = (long)blockIdx.x * kLatWaves + wave; row < B; row += nw) {
        const float4 x = load_row4(z, row, D, lane);
        float tt[KP], nn[KP], gg[KP];
        float s = 0.f, gq = 0.f;
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const bool live = k < K;
            tt[k] = live ? tsaved[row * K + k] : 0.f;
            gg[k] = live ? grad_q[row * K + k] : 0.f;
            nn[k] = (alpha == 1.0f) ? tt[k] : (live ? exp2f(power * log2f(tt[k])) : 0.f);
            s += nn[k];
        }
        const float inv_s = 1.0f / s;
#pragma unroll
        for (int k = 0; k < KP; ++k) gq = fmaf(gg[k], nn[k] * inv_s, gq);     // sum_k g_k q_k
        float4 gz = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            // coef = 2 * dn/dd2 * (g - <g,q>) / s,  dn/dd2 = -((alpha+1)/(2 alpha)) n t
            const float coef = 2.0f * dn_scale * nn[k] * tt[k] * (gg[k] - gq) * inv_s;
            const float4 c = *reinterpret_cast<const float4*>(cl + k * 256 + lane * 4);
            const float d0 = x.x - c.x, d1 = x.y - c.y, d2 = x.z - c.z, d3 = x.w - c.w;
            gz.x = fmaf(coef, d0, gz.x); gz.y = fmaf(coef, d1, gz.y); gz.z = fmaf(coef, d2, gz.z); gz.w = fmaf(coef, d3, gz.w);
            acc[k].x = fmaf(coef, d0, acc[k].x); acc[k].y = fmaf(coef, d1, acc[k].y);
            acc[k].z = fmaf(coef, d2, acc[k].z); acc[k].w = fmaf(coef, d3, acc[k].w);
        }
        if (lane * 4 < D) *reinterpret_cast<float4*>(grad_z + row * D + lane * 4) = gz;
    }
    // combine the 4 waves through LDS (reusing the centroid tile), then one partial per workgroup
    __syncthreads();
    for (int w = 0; w < kLatWaves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                float4* slot = reinterpret_cast<float4*>(cl + k * 256 + lane * 4);
                if (w == 0) *slot = acc[k];
                else { float4 o = *slot; o.x += acc[k].x; o.y += acc[k].y; o.z += acc[k].z; o.w += acc[k].w; *slot = o; }
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < K * D; i += kLatBlock) {
        const int k = i / D, d = i - k * D;
        partials[(size_t)blockIdx.x * K * D + i] = cl[k * 256 + d];
    }
}

__global__ __launch_bounds__(256) void dec_bwd_finalize(const float* partials, int nblk, int n, float sign, float* out) {
    __shared__ double red[256];
    const double s = reduce_partials_32x8(partials, nblk, n, blockIdx.x * 32, red);
    const int i = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x < 32 && i < n) out[i] = (float)(sign * s);
}

// fused KL(p||q)/batch_div and d/dq
__global__ __launch_bounds__(kLatBlock) void dec_kl_kernel(const float* q, const float* p, long n, float batch_div,
                                                          float gscale, double* partials, float* grad_q) {
    __shared__ double red[kLatWaves];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * kLatBlock + threadIdx.x; i < n; i += (long)gridDim.x * kLatBlock) {
        const float pv = p[i], qv = q[i];
        if (pv > 0.f) acc += pv * (logf(pv) - logf(qv));       // F.kl_div: 0 where target == 0
        if (grad_q) grad_q[i] = -gscale * pv / (batch_div * qv);
    }
    const double w = wave_sum((double)acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0;
        for (int i = 0; i < kLatWaves; ++i) s += red[i];
        partials[blockIdx.x] = s;
    }
}
__global__ __launch_bounds__(256) void dec_kl_finalize(const double* partials, int nblk, float batch_div, float* out) {
    __shared__ double red[256];
    const double s = reduce_partials_32x8(partials, nblk, 1, 0, red);
    if (threadIdx.x == 0) out[0] = (float)(s / (double)batch_div);
}

// =============================================================================== k-means
struct KmWs {   // workspace carve-up (per launch; identical in host sizing and kernels)
    size_t psum, pcnt, pchg, sums, mind, total;
};
__host__ __device__ inline KmWs km_ws(int N, int D, int K, int n_runs, int nblk) {
    KmWs w;
    size_t o = 0;
    w.psum = o; o += (size_t)n_runs * nblk * K * D * sizeof(float);
    w.pcnt = o; o += (size_t)n_runs * nblk * (K + 1) * sizeof(int);     // K counts + #changed per workgroup
    w.pchg = o;
    o = (o + 15) & ~(size_t)15;
    w.sums = o; o += (size_t)n_runs * K * D * sizeof(double);
    w.mind = o; o += (size_t)n_runs * N * sizeof(float);
    w.total = (o + 15) & ~(size_t)15;
    return w;
}
static int km_blocks(int N) { return (int)max(1L, min(((long)N + 4 * kLatWaves - 1) / (4 * kLatWaves), (long)2 * kNumCU)); }

// E-step (+ partial M-step): argmin_j(||c_j||^2 - 2 x.c_j), first minimum wins (_k_means_lloyd.pyx:193-213)
template <int KP, bool UPDATE>
__global__ __launch_bounds__(kLatBlock) void kmeans_assign_kernel(const float* X, const float* xnorm, int N, int D, int K,
                                                                 const float* centers_all, int32_t* labels_all,
                                                                 const float* status_all, float* mind_all,
                                                                 float* psum_all, int* pcnt_all, int* pchg_all) {
    __shared__ __align__(16) float cl[KP * 256];
    __shared__ float cnorm[KP];
    __shared__ int cntred[kLatWaves][KP];
    __shared__ int chgred[kLatWaves];
    constexpr int LG = KLog2<KP>::v;
    const int run = blockIdx.y, nblk = gridDim.x;
    if (UPDATE && status_all[run * DIC_KM_STATUS_WORDS] != 0.f) return;     // this run has converged
    const float* centers = centers_all + (size_t)run * K * D;
    int32_t* labels = labels_all + (size_t)run * N;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stage_centers<KP>(cl, centers, K, D);
    __syncthreads();
    if (wave == 0) {     // ||c_k||^2 per centroid (row_norms, _k_means_lloyd.pyx:99)
        for (int k = 0; k < KP; ++k) {
            const float4 c = *reinterpret_cast<const float4*>(cl + k * 256 + lane * 4);
            const float s = wave_sum(fmaf(c.w, c.w, fmaf(c.z, c.z, fmaf(c.y, c.y, c.x * c.x))));
            if (lane == 0) cnorm[k] = s;
        }
    }
    __syncthreads();
    const int kidx = lane >> (6 - LG);
    const float my_cnorm = kidx < K ? cnorm[kidx] : 0.f;
    float4 acc[KP];
    int cnt[KP];
#pragma unroll
    for (int k = 0; k < KP; ++k) { acc[k] = make_float4(0.f, 0.f, 0.f, 0.f); cnt[k] = 0; }
    int changed = 0;
    const long nw = (long)nblk * kLatWaves;
    // the row of the NEXT trip is requested before this trip's shuffle chain: at K >= 8 the accumulators leave two waves
    // per SIMD, too few to hide the load latency by occupancy alone
    long row = (long)blockIdx.x * kLatWaves + wave;
    float4 xn = row < N ? load_row4(X, row, D, lane) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (; row < N; row += nw) {
        const float4 x = xn;
        if (row + nw < N) xn = load_row4(X, row + nw, D, lane);
        float v[KP];
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const float4 c = *reinterpret_cast<const float4*>(cl + k * 256 + lane * 4);
            v[k] = fmaf(x.w, c.w, fmaf(x.z, c.z, fmaf(x.y, c.y, x.x * c.x)));
        }
        fold<KP, 32>(v, lane);
        float best = kidx < K ? fmaf(-2.0f, v[0], my_cnorm) : INFINITY;
        int lab = kidx;
#pragma unroll
        for (int m = 32; m >= (64 >> LG); m >>= 1) {
            const float ov = __shfl_xor(best, m);
            const int oi = __shfl_xor(lab, m);
            if (ov < best || (ov == best && oi < lab)) { best = ov; lab = oi; }
        }
        const int label = __builtin_amdgcn_readfirstlane(lab);
        if (UPDATE) {
            const int old = labels[row];
            changed += (old != label) ? 1 : 0;
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                if (label == k) {
                    acc[k].x += x.x; acc[k].y += x.y; acc[k].z += x.z; acc[k].w += x.w;
                    cnt[k] += 1;
                }
            }
            if (lane == 0) {
                labels[row] = label;
                mind_all[(size_t)run * N + row] = fmaxf(0.f, xnorm[row] + best);
            }
        } else {
            // exact ||x - c_label||^2 for inertia / mindist (_k_means_common.pyx _inertia_dense)
            const float4 c = *reinterpret_cast<const float4*>(cl + label * 256 + lane * 4);
            const float d0 = x.x - c.x, d1 = x.y - c.y, d2 = x.z - c.z, d3 = x.w - c.w;
            const float dist = wave_sum(fmaf(d3, d3, fmaf(d2, d2, fmaf(d1, d1, d0 * d0))));
            if (lane == 0) {
                labels[row] = label;
                if (mind_all) mind_all[(size_t)run * N + row] = dist;
            }
        }
    }
    if (!UPDATE) return;
    // workgroup partials: centroid sums through LDS (reuse the centroid tile), counts, #changed
    __syncthreads();
    for (int w = 0; w < kLatWaves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                float4* slot = reinterpret_cast<float4*>(cl + k * 256 + lane * 4);
                if (w == 0) *slot = acc[k];
                else { float4 o = *slot; o.x += acc[k].x; o.y += acc[k].y; o.z += acc[k].z; o.w += acc[k].w; *slot = o; }
            }
        }
        __syncthreads();
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < KP; ++k) cntred[wave][k] = cnt[k];
        chgred[wave] = changed;
    }
    __syncthreads();
    const size_t pb = (size_t)run * nblk + blockIdx.x;
    for (int i = threadIdx.x; i < K * D; i += kLatBlock) {
        const int k = i / D, d = i - k * D;
        psum_all[pb * K * D + i] = cl[k * 256 + d];
    }
    if (threadIdx.x < K) {
        int s = 0;
        for (int w = 0; w < kLatWaves; ++w) s += cntred[w][threadIdx.x];
        pcnt_all[pb * (K + 1) + threadIdx.x] = s;
    }
    if (threadIdx.x == 0) {
        int s = 0;
        for (int w = 0; w < kLatWaves; ++w) s += chgred[w];
        pcnt_all[pb * (K + 1) + K] = s;
    }
    (void)pchg_all;
}

// stage A of the update: fixed-order f64 reduction of the workgroup partial sums; grid (K*D/32, n_runs)
__global__ __launch_bounds__(256) void kmeans_reduce_kernel(const float* psum_all, int nblk, int K, int D,
                                                           const float* status_all, double* sums_all) {
    __shared__ double red[256];
    const int run = blockIdx.y, n = K * D;
    if (status_all[run * DIC_KM_STATUS_WORDS] != 0.f) return;
    const double s = reduce_partials_32x8(psum_all + (size_t)run * nblk * n, nblk, n, blockIdx.x * 32, red);
    const int i = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x < 32 && i < n) sums_all[(size_t)run * n + i] = s;
}

// K sums over the block's 256 threads (terms[k * 256 + tid] in LDS) -> out[k], bit for bit what the fold-down tree
// `for (st = 128; st; st >>= 1) v[tid] += v[tid + st]` gives (the first two levels by hand, the other six ARE the xor butterfly of
// wave_sum: the same pairs are added at every level) -- but with two barriers in all instead of nine per centre (K = 20: 21 -> 8 us
// per Lloyd iteration of the update kernel, 1.2 s of the p2 sweep).
__device__ __forceinline__ void block_sums_256(int K, const float* terms, float* out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    for (int k = w; k < K; k += 4) {
        const float* t = terms + k * 256 + lane;
        const float v = wave_sum((t[0] + t[128]) + (t[64] + t[192]));
        if (lane == 0) out[k] = v;
    }
    __syncthreads();
}

// stage B: counts, empty-cluster relocation, averaging, centre shift, convergence; one block per run
__global__ __launch_bounds__(256) void kmeans_update_kernel(const float* X, int N, int D, int K, int nblk,
                                                           const int* pcnt_all, const int* pchg_all, double* sums_all,
                                                           float* mind_all, const int32_t* labels_all,
                                                           float* centers_all, float* status_all) {
    __shared__ float cntf[DIC_MAX_CLUSTERS];
    __shared__ int s_changed, s_far, s_relocs;
    __shared__ float s_fard;
    __shared__ float redv[256];
    __shared__ int redi[256];
    __shared__ float shift2[DIC_MAX_CLUSTERS];
    __shared__ float shk[DIC_MAX_CLUSTERS * 256];
    const int run = blockIdx.x, tid = threadIdx.x;
    float* status = status_all + run * DIC_KM_STATUS_WORDS;
    if (status[0] != 0.f) return;
    double* sums = sums_all + (size_t)run * K * D;
    float* centers = centers_all + (size_t)run * K * D;
    float* mind = mind_all + (size_t)run * N;
    const int32_t* labels = labels_all + (size_t)run * N;

    {   // counts (exact in f64) and #changed: K+1 <= 33 outputs -> two passes of the 32-wide reducer
        __shared__ double red[256];
        const int* pc = pcnt_all + (size_t)run * nblk * (K + 1);
        const double c0 = reduce_partials_32x8(pc, nblk, K + 1, 0, red);
        if (tid < 32 && tid < K) cntf[tid] = (float)c0;
        if (tid < 32 && tid == K) s_changed = (int)c0;
        __syncthreads();
        if (K + 1 > 32) {
            const double c1 = reduce_partials_32x8(pc, nblk, K + 1, 32, red);
            if (tid == 0) s_changed = (int)c1;       // K == 32: output 32 is #changed
        }
        if (tid == 0) s_relocs = 0;
        (void)pchg_all;
    }
    __syncthreads();

    // _relocate_empty_clusters_dense (_k_means_common.pyx:167-211): each empty cluster, in
    // increasing id, takes the point farthest from its assigned centre (farthest first).
    for (int k = 0; k < K; ++k) {
        if (cntf[k] != 0.f) continue;              // uniform across the block
        float bd = -1.f; int bi = -1;
        for (int i = tid; i < N; i += 256) {
            const float dv = mind[i];
            if (dv > bd) { bd = dv; bi = i; }
        }
        redv[tid] = bd; redi[tid] = bi;
        __syncthreads();
        for (int st = 128; st >= 1; st >>= 1) {
            if (tid < st) {
                const float ov = redv[tid + st]; const int oi = redi[tid + st];
                if (ov > redv[tid] || (ov == redv[tid] && oi >= 0 && (redi[tid] < 0 || oi < redi[tid]))) { redv[tid] = ov; redi[tid] = oi; }
            }
            __syncthreads();
        }
        if (tid == 0) { s_fard = redv[0]; s_far = redi[0]; }
        __syncthreads();
        if (!(s_fard > 0.f) || s_far < 0) break;   // max distance 0: relocating is pointless (:193-196)
        const int far = s_far, old = labels[far];
        if (tid < D) {
            const double xv = (double)X[(size_t)far * D + tid];
            sums[(size_t)old * D + tid] -= xv;
            sums[(size_t)k * D + tid] = xv;
        }
        __syncthreads();
        if (tid == 0) { cntf[k] = 1.f; cntf[old] -= 1.f; mind[far] = -1.f; s_relocs += 1; }
        __syncthreads();
    }

    // _average_centers (:274-295) incl. its in-place quirk for a still-empty cluster, then _center_shift
    int amax = 0;
    for (int k = 1; k < K; ++k) if (cntf[k] > cntf[amax]) amax = k;
    float* ss_local = shk + tid;                  // [k][256]: this thread's term of centre k's squared shift
    if (tid < D) {
        float newc[DIC_MAX_CLUSTERS];
        for (int k = 0; k < K; ++k) newc[k] = (float)sums[(size_t)k * D + tid];
        for (int k = 0; k < K; ++k) {
            if (cntf[k] > 0.f) newc[k] *= 1.0f / cntf[k];
            else newc[k] = newc[amax];
        }
        for (int k = 0; k < K; ++k) {
            const float dlt = newc[k] - centers[(size_t)k * D + tid];
            ss_local[k * 256] = dlt * dlt;
            centers[(size_t)k * D + tid] = newc[k];
        }
    } else {
        for (int k = 0; k < K; ++k) ss_local[k * 256] = 0.f;
    }
    block_sums_256(K, shk, shift2);
    if (tid == 0) {
        float tot = 0.f;
        for (int k = 0; k < K; ++k) { const float sh = sqrtf(shift2[k]); tot += sh * sh; }   // (center_shift**2).sum()
        const float iters = status[1] + 1.f;
        status[1] = iters; status[2] = tot; status[3] = (float)s_changed; status[5] += (float)s_relocs;
        if (s_changed == 0) { status[4] = 1.f; status[0] = 1.f; }             // strict convergence (_kmeans.py:717-721)
        else if (tot <= status[6]) status[0] = 1.f;                           // tol stop (:724-733)
        if (iters >= status[7]) status[0] = 1.f;                              // max_iter
    }
}

// ---- Lloyd with the points sharded over ranks (one process per GPU): stage A stops at per-rank partial statistics,
// the caller sums them over ranks (RCCL all-reduce of one f64 buffer), stage B finishes the iteration identically on every
// rank.  stats[run] = [ sums (K*D) | counts (K) | #labels changed (1) ] in f64.
__global__ __launch_bounds__(256) void kmeans_pack_counts_kernel(const int* pcnt_all, int nblk, int K, int D, const float* status_all,
                                                                double* stats_all) {
    __shared__ double red[256];
    const int run = blockIdx.x, tid = threadIdx.x;
    if (status_all[run * DIC_KM_STATUS_WORDS] != 0.f) return;
    const int* pc = pcnt_all + (size_t)run * nblk * (K + 1);
    double* out = stats_all + (size_t)run * ((size_t)K * D + K + 1) + (size_t)K * D;
    for (int i0 = 0; i0 < K + 1; i0 += 32) {                     // K + 1 <= 33 outputs: at most two passes
        const double c = reduce_partials_32x8(pc, nblk, K + 1, i0, red);
        if (tid < 32 && i0 + tid < K + 1) out[i0 + tid] = c;
        __syncthreads();
    }
}

// stage B on globally summed statistics: averaging (with scikit-learn's in-place quirk for a still-empty cluster), centre
// shift, convergence -- the tail of kmeans_update_kernel.  An EMPTY cluster cannot be relocated here (the farthest point
// may live on another rank): the run is halted with status[0] = 2 and the caller re-runs it unsharded.
__global__ __launch_bounds__(256) void kmeans_finish_kernel(int D, int K, const double* stats_all, float* centers_all, float* status_all) {
    __shared__ float cntf[DIC_MAX_CLUSTERS];
    __shared__ float shift2[DIC_MAX_CLUSTERS];
    __shared__ float shk[DIC_MAX_CLUSTERS * 256];
    __shared__ int s_empty;
    const int run = blockIdx.x, tid = threadIdx.x;
    float* status = status_all + run * DIC_KM_STATUS_WORDS;
    if (status[0] != 0.f) return;
    const double* stats = stats_all + (size_t)run * ((size_t)K * D + K + 1);
    const double* sums = stats;
    float* centers = centers_all + (size_t)run * K * D;
    if (tid == 0) s_empty = 0;
    __syncthreads();
    if (tid < K) {
        cntf[tid] = (float)stats[(size_t)K * D + tid];
        if (cntf[tid] == 0.f) s_empty = 1;
    }
    __syncthreads();
    if (s_empty) {
        if (tid == 0) status[0] = 2.f;
        return;
    }
    const int changed = (int)stats[(size_t)K * D + K];
    float* ss_local = shk + tid;                  // [k][256]: this thread's term of centre k's squared shift
    if (tid < D) {
        for (int k = 0; k < K; ++k) {
            const float nc = (float)sums[(size_t)k * D + tid] * (1.0f / cntf[k]);
            const float dlt = nc - centers[(size_t)k * D + tid];
            ss_local[k * 256] = dlt * dlt;
            centers[(size_t)k * D + tid] = nc;
        }
    } else {
        for (int k = 0; k < K; ++k) ss_local[k * 256] = 0.f;
    }
    block_sums_256(K, shk, shift2);
    if (tid == 0) {
        float tot = 0.f;
        for (int k = 0; k < K; ++k) { const float sh = sqrtf(shift2[k]); tot += sh * sh; }
        const float iters = status[1] + 1.f;
        status[1] = iters; status[2] = tot; status[3] = (float)changed;
        if (changed == 0) { status[4] = 1.f; status[0] = 1.f; }
        else if (tot <= status[6]) status[0] = 1.f;
        if (iters >= status[7]) status[0] = 1.f;
    }
}

__global__ __launch_bounds__(256) void inertia_kernel(const float* mind_all, int N, float* inertia) {
    __shared__ double red[4];
    const int run = blockIdx.x;
    double s = 0.0;
    for (int i = threadIdx.x; i < N; i += 256) s += (double)mind_all[(size_t)run * N + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) inertia[run] = (float)(red[0] + red[1] + red[2] + red[3]);
}

// k-means++ inner step (sklearn/cluster/_kmeans.py:218-243): exact (f64) squared distances of every row to L candidate rows, min with the restart's running
// `closest`, the candidates' potentials.  Round 6: ONE pass over the rows for up to LC = 64 candidates (grid.y = further groups of LC).  The candidate rows sit in
// LDS; a wave takes a row (lane = 4 features), forms its per-lane partial of all LC distances in f64 and folds the LC partials across the 64 lanes with the
// halving exchange of `fold` (63 exchanges for 64 sums instead of 6 each: lane l ends up with the distance to candidate l), so each lane then finishes ONE
// (row, candidate): min with closest, store, potential.  Until then one workgroup COLUMN per candidate re-read X and paid a six-step f64 butterfly per (row,
// candidate): 357 us per launch for 40 candidates on 75 000 x 256 (0.9 s of p2's sweep).
template <int N_, int MASK>
__device__ __forceinline__ void fold_f64(double* v, int lane) {
    if constexpr (N_ > 1) {
        constexpr int H = N_ / 2;
        const bool hi = (lane & MASK) != 0;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const double send = hi ? v[i] : v[i + H];
            const double keep = hi ? v[i + H] : v[i];
            v[i] = keep + __shfl_xor(send, MASK);
        }
        fold_f64<H, MASK / 2>(v, lane);
    } else {
#pragma unroll
        for (int m = MASK; m >= 1; m >>= 1) v[0] += __shfl_xor(v[0], m);
    }
}

template <int LC>
__global__ __launch_bounds__(kLatBlock) void kmeans_pp_kernel(const float* X, int N, int D, int row_lo, int row_hi, const int64_t* cand, int L, int group,
                                                             const float* closest_all, float* dist_out, double* ppart) {
    extern __shared__ __align__(16) float pp_cl[];                  // [LC][256] candidate rows, zero beyond D / L
    __shared__ double red[kLatWaves][LC];
    constexpr int LG = LC == 64 ? 6 : LC == 32 ? 5 : 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l0 = blockIdx.y * LC;
    for (int i = threadIdx.x; i < LC * 64; i += kLatBlock) {
        const int l = i >> 6, q = i & 63;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (l0 + l < L) v = load_row4(X, cand[l0 + l], D, q);
        *reinterpret_cast<float4*>(pp_cl + l * 256 + 4 * q) = v;
    }
    __syncthreads();
    const int lidx = lane >> (6 - LG), my_l = l0 + lidx;          // the candidate this lane finishes (fold: index = the lane's top LG bits)
    const bool writer = (lane & ((64 >> LG) - 1)) == 0 && my_l < L;
    const float* closest = closest_all + (size_t)(writer ? my_l / group : 0) * N;
    double pot = 0.0;
    const long nw = (long)gridDim.x * kLatWaves;
    const int n_here = min(LC, L - l0);
    for (long row = (long)row_lo + (long)blockIdx.x * kLatWaves + wave; row < row_hi; row += nw) {      // this rank's rows of the (.,N) arrays
        const float4 x = load_row4(X, row, D, lane);
        asm volatile("" ::: "memory");          // (the candidate rows are re-read from LDS for every row: hoisted out of this loop as doubles they need 512 registers)
        double p[LC];
#pragma unroll
        for (int l = 0; l < LC; ++l) {
            p[l] = 0.0;
            if (l < n_here) {                                      // (uniform: the slots behind the last candidate cost nothing)
                const float4 c = *reinterpret_cast<const float4*>(pp_cl + l * 256 + 4 * lane);
                const double d0 = (double)x.x - c.x, d1 = (double)x.y - c.y, d2 = (double)x.z - c.z, d3 = (double)x.w - c.w;
                p[l] = d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
            }
        }
        fold_f64<LC, 32>(p, lane);
        if (writer) {
            const float m = fminf(closest[row], (float)p[0]);
            dist_out[(size_t)my_l * N + row] = m;
            pot += (double)m;
        }
    }
    if ((lane & ((64 >> LG) - 1)) == 0) red[wave][lidx] = pot;
    __syncthreads();
    if (threadIdx.x < LC && l0 + threadIdx.x < L)
        ppart[(size_t)blockIdx.x * L + l0 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ __launch_bounds__(256) void kmeans_pp_finalize(const double* ppart, int nblk, int L, double* pot) {
    __shared__ double red[256];
    const double s = reduce_partials_32x8(ppart, nblk, L, blockIdx.x * 32, red);
    const int l = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x < 32 && l < L) pot[l] = s;
}

static int pad_k(int K) { int kp = 2; while (kp < K) kp <<= 1; return kp; }
static int lat_blocks(int rows) { return (int)max(1L, min(((long)rows + 2 * kLatWaves - 1) / (2 * kLatWaves), (long)4 * kNumCU)); }

static int latent_check(const char* who, int rows, int D, int K) {
    DIC_REQUIRE(rows > 0 && D > 0 && K > 0, DIC_ERR_INVALID_ARG, "%s: non-positive size", who);
    DIC_REQUIRE(D % 4 == 0 && D <= DIC_LATENT_MAX_DIM, DIC_ERR_UNSUPPORTED, "%s: D=%d must be a multiple of 4 and <= %d", who, D,
                DIC_LATENT_MAX_DIM);
    DIC_REQUIRE(K <= DIC_MAX_CLUSTERS, DIC_ERR_UNSUPPORTED, "%s: K=%d > %d", who, K, DIC_MAX_CLUSTERS);
    return DIC_OK;
}

#define DIC_DISPATCH_KP(KPV, ...)                    \
    switch (KPV) {                                   \
        case 2: { constexpr int KP = 2; __VA_ARGS__; } break;   \
        case 4: { constexpr int KP = 4; __VA_ARGS__; } break;   \
        case 8: { constexpr int KP = 8; __VA_ARGS__; } break;   \
        case 16: { constexpr int KP = 16; __VA_ARGS__; } break; \
        default: { constexpr int KP = 32; __VA_ARGS__; } break; \
    }

// (dic_cumsum_f64)
__global__ __launch_bounds__(1024) void cumsum_f64_kernel(const float* x, int n, double* out) {
    __shared__ double wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* xr = x + (size_t)blockIdx.x * n;
    double* orow = out + (size_t)blockIdx.x * n;
    const int chunk = ((n + 15) / 16 + 63) / 64 * 64;
    const int lo = min(n, w * chunk), hi = min(n, lo + chunk);
    double s = 0.0;
    for (int i = lo + lane; i < hi; i += 64) s += (double)xr[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if (lane == 0) wsum[w] = s;
    __syncthreads();
    double carry = 0.0;
    for (int j = 0; j < w; ++j) carry += wsum[j];
    for (int base = lo; base < hi; base += 64) {
        const int i = base + lane;
        double v = i < hi ? (double)xr[i] : 0.0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const double t = __shfl_up(v, d);
            if (lane >= d) v += t;
        }
        v += carry;
        if (i < hi) orow[i] = v;
        carry = __shfl(v, 63);
    }
}

}  // namespace dic

using namespace dic;

extern "C" {

size_t dic_dec_fwd_workspace(int B, int D, int K) {
    if (B <= 0 || K <= 0) return 0;
    return (size_t)lat_blocks(B) * K * sizeof(float);
}

int dic_dec_fwd(const float* z, const float* centers, int B, int D, int K, float alpha, float* q, float* tsaved,
                float* colsum, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    int rc = latent_check("dec_fwd", B, D, K);
    if (rc) return rc;
    DIC_REQUIRE(z && centers && q, DIC_ERR_INVALID_ARG, "dec_fwd: NULL pointer");
    DIC_REQUIRE(alpha > 0.f, DIC_ERR_INVALID_ARG, "dec_fwd: alpha must be positive");
    const int nblk = lat_blocks(B);
    DIC_REQUIRE(!colsum || (workspace && workspace_bytes >= (size_t)nblk * K * sizeof(float)), DIC_ERR_WORKSPACE,
                "dec_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* part = colsum ? (float*)workspace : nullptr;
    DIC_DISPATCH_KP(pad_k(K), hipLaunchKernelGGL(dec_fwd_kernel<KP>, dim3(nblk), dim3(kLatBlock), 0, st, z, centers, B, D, K,
                                                 alpha, q, tsaved, part));
    if (colsum) hipLaunchKernelGGL(colsum_finalize, dim3(1), dim3(256), 0, st, (const float*)part, nblk, K, colsum);
    return check_launch("dec_fwd");
}

int dic_dec_target(const float* q, const float* colsum, int B, int K, float* p, dic_stream_t stream) {
    DIC_REQUIRE(B > 0 && K > 0, DIC_ERR_INVALID_ARG, "dec_target: non-positive size");
    DIC_REQUIRE(q && colsum && p, DIC_ERR_INVALID_ARG, "dec_target: NULL pointer");
    hipLaunchKernelGGL(dec_target_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, q, colsum, B, K, p);
    return check_launch("dec_target");
}

static int dec_bwd_blocks(int B, int K) { return min(lat_blocks(B), K > 8 ? kNumCU : 2 * kNumCU); }

size_t dic_dec_bwd_workspace(int B, int D, int K) {
    if (B <= 0 || D <= 0 || K <= 0) return 0;
    return (size_t)dec_bwd_blocks(B, K) * K * D * sizeof(float);
}

int dic_dec_bwd(const float* z, const float* centers, const float* q, const float* tsaved, const float* grad_q, int B,
                int D, int K, float alpha, float* grad_z, float* grad_centers, void* workspace, size_t workspace_bytes,
                dic_stream_t stream) {
    (void)q;
    int rc = latent_check("dec_bwd", B, D, K);
    if (rc) return rc;
    DIC_REQUIRE(z && centers && tsaved && grad_q && grad_z && grad_centers && workspace, DIC_ERR_INVALID_ARG,
                "dec_bwd: NULL pointer");
    const int nblk = dec_bwd_blocks(B, K);
    DIC_REQUIRE(workspace_bytes >= (size_t)nblk * K * D * sizeof(float), DIC_ERR_WORKSPACE, "dec_bwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    DIC_DISPATCH_KP(pad_k(K), hipLaunchKernelGGL(dec_bwd_kernel<KP>, dim3(nblk), dim3(kLatBlock), 0, st, z, centers, tsaved,
                                                 grad_q, B, D, K, alpha, grad_z, (float*)workspace));
    const int n = K * D;
    hipLaunchKernelGGL(dec_bwd_finalize, dim3((n + 31) / 32), dim3(256), 0, st, (const float*)workspace, nblk, n, -1.0f,
                       grad_centers);
    return check_launch("dec_bwd");
}

static int kl_blocks(long n) { return (int)max(1L, min((n + kLatBlock - 1) / kLatBlock, (long)kNumCU)); }

size_t dic_dec_kl_workspace(int B, int K) {
    if (B <= 0 || K <= 0) return 0;
    return (size_t)kl_blocks((long)B * K) * sizeof(double);
}

int dic_dec_kl(const float* q, const float* p, int B, int K, float batch_div, float gscale, float* kl_out, float* grad_q,
               void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(B > 0 && K > 0 && batch_div > 0.f, DIC_ERR_INVALID_ARG, "dec_kl: bad size");
    DIC_REQUIRE(q && p && kl_out && workspace, DIC_ERR_INVALID_ARG, "dec_kl: NULL pointer");
    const long n = (long)B * K;
    const int nblk = kl_blocks(n);
    DIC_REQUIRE(workspace_bytes >= (size_t)nblk * sizeof(double), DIC_ERR_WORKSPACE, "dec_kl: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dec_kl_kernel, dim3(nblk), dim3(kLatBlock), 0, st, q, p, n, batch_div, gscale, (double*)workspace,
                       grad_q);
    hipLaunchKernelGGL(dec_kl_finalize, dim3(1), dim3(256), 0, st, (const double*)workspace, nblk, batch_div, kl_out);
    return check_launch("dec_kl");
}

size_t dic_kmeans_pp_workspace(int N, int L) {
    if (N <= 0 || L <= 0) return 0;
    return (size_t)L * km_blocks(N) * sizeof(double);
}

size_t dic_kmeans_workspace(int N, int D, int K, int n_runs) {
    if (N <= 0 || D <= 0 || K <= 0 || n_runs <= 0) return 0;
    const size_t a = km_ws(N, D, K, n_runs, km_blocks(N)).total;
    return a;
}

int dic_kmeans_lloyd_iter(const float* X, const float* xnorm, int N, int D, int K, int n_runs, float* centers,
                          int32_t* labels, float* status, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    int rc = latent_check("kmeans_lloyd_iter", N, D, K);
    if (rc) return rc;
    DIC_REQUIRE(n_runs > 0 && n_runs <= 65535, DIC_ERR_INVALID_ARG, "kmeans_lloyd_iter: n_runs=%d", n_runs);
    DIC_REQUIRE(X && xnorm && centers && labels && status && workspace, DIC_ERR_INVALID_ARG, "kmeans_lloyd_iter: NULL pointer");
    const int nblk = km_blocks(N);
    const KmWs w = km_ws(N, D, K, n_runs, nblk);
    DIC_REQUIRE(workspace_bytes >= w.total, DIC_ERR_WORKSPACE, "kmeans_lloyd_iter: workspace %zu < %zu", workspace_bytes, w.total);
    char* ws = (char*)workspace;
    float* psum = (float*)(ws + w.psum); int* pcnt = (int*)(ws + w.pcnt); int* pchg = (int*)(ws + w.pchg);
    double* sums = (double*)(ws + w.sums); float* mind = (float*)(ws + w.mind);
    hipStream_t st = (hipStream_t)stream;
    int np = nblk;                                    // partial slots per restart actually written
    if (kmeans_use_mfma(K, n_runs)) {                         // matrix-core E-step (dic_kmeans_mfma.hip); fewer, fatter workgroups
        np = kmeans_mfma_blocks(N, K, n_runs);
        rc = kmeans_assign_mfma_launch(X, xnorm, N, D, K, n_runs, centers, labels, status, mind, psum, pcnt, st);
        if (rc) return rc;
    } else {
        DIC_DISPATCH_KP(pad_k(K), hipLaunchKernelGGL((kmeans_assign_kernel<KP, true>), dim3(nblk, n_runs), dim3(kLatBlock), 0, st, X,
                                                     xnorm, N, D, K, (const float*)centers, labels, (const float*)status, mind, psum,
                                                     pcnt, pchg));
    }
    hipLaunchKernelGGL(kmeans_reduce_kernel, dim3((K * D + 31) / 32, n_runs), dim3(256), 0, st, (const float*)psum, np, K, D,
                       (const float*)status, sums);
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(n_runs), dim3(256), 0, st, X, N, D, K, np, (const int*)pcnt,
                       (const int*)pchg, sums, mind, (const int32_t*)labels, centers, status);
    return check_launch("kmeans_lloyd_iter");
}

size_t dic_kmeans_stats_words(int D, int K) { return (D > 0 && K > 0) ? (size_t)K * D + K + 1 : 0; }

int dic_kmeans_lloyd_partial(const float* X, const float* xnorm, int N, int D, int K, int n_runs, const float* centers,
                             int32_t* labels, const float* status, double* stats, void* workspace, size_t workspace_bytes,
                             dic_stream_t stream) {
    int rc = latent_check("kmeans_lloyd_partial", N, D, K);
    if (rc) return rc;
    DIC_REQUIRE(n_runs > 0 && n_runs <= 65535, DIC_ERR_INVALID_ARG, "kmeans_lloyd_partial: n_runs=%d", n_runs);
    DIC_REQUIRE(X && xnorm && centers && labels && status && stats && workspace, DIC_ERR_INVALID_ARG, "kmeans_lloyd_partial: NULL pointer");
    const int nblk = km_blocks(N);
    const KmWs w = km_ws(N, D, K, n_runs, nblk);
    DIC_REQUIRE(workspace_bytes >= w.total, DIC_ERR_WORKSPACE, "kmeans_lloyd_partial: workspace %zu < %zu", workspace_bytes, w.total);
    char* ws = (char*)workspace;
    float* psum = (float*)(ws + w.psum); int* pcnt = (int*)(ws + w.pcnt); int* pchg = (int*)(ws + w.pchg);
    double* sums = (double*)(ws + w.sums); float* mind = (float*)(ws + w.mind);
    hipStream_t st = (hipStream_t)stream;
    const size_t words = dic_kmeans_stats_words(D, K);
    int np = nblk;
    if (kmeans_use_mfma(K, n_runs)) {
        np = kmeans_mfma_blocks(N, K, n_runs);
        rc = kmeans_assign_mfma_launch(X, xnorm, N, D, K, n_runs, centers, labels, status, mind, psum, pcnt, st);
        if (rc) return rc;
    } else {
        DIC_DISPATCH_KP(pad_k(K), hipLaunchKernelGGL((kmeans_assign_kernel<KP, true>), dim3(nblk, n_runs), dim3(kLatBlock), 0, st, X,
                                                     xnorm, N, D, K, centers, labels, status, mind, psum, pcnt, pchg));
    }
    hipLaunchKernelGGL(kmeans_reduce_kernel, dim3((K * D + 31) / 32, n_runs), dim3(256), 0, st, (const float*)psum, np, K, D,
                       status, sums);
    // sums -> the head of each run's stats record (strided device copy), counts / #changed behind them
    hipError_t e = hipMemcpy2DAsync(stats, words * sizeof(double), sums, (size_t)K * D * sizeof(double), (size_t)K * D * sizeof(double),
                                    (size_t)n_runs, hipMemcpyDeviceToDevice, st);
    DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "kmeans_lloyd_partial: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kmeans_pack_counts_kernel, dim3(n_runs), dim3(256), 0, st, (const int*)pcnt, np, K, D, status, stats);
    return check_launch("kmeans_lloyd_partial");
}

int dic_kmeans_lloyd_finish(int D, int K, int n_runs, const double* stats, float* centers, float* status, dic_stream_t stream) {
    DIC_REQUIRE(D > 0 && D <= DIC_LATENT_MAX_DIM && D % 4 == 0 && K > 0 && K <= DIC_MAX_CLUSTERS, DIC_ERR_UNSUPPORTED,
                "kmeans_lloyd_finish: D=%d K=%d", D, K);
    DIC_REQUIRE(n_runs > 0 && n_runs <= 65535, DIC_ERR_INVALID_ARG, "kmeans_lloyd_finish: n_runs=%d", n_runs);
    DIC_REQUIRE(stats && centers && status, DIC_ERR_INVALID_ARG, "kmeans_lloyd_finish: NULL pointer");
    hipLaunchKernelGGL(kmeans_finish_kernel, dim3(n_runs), dim3(256), 0, (hipStream_t)stream, D, K, stats, centers, status);
    return check_launch("kmeans_lloyd_finish");
}

int dic_kmeans_predict(const float* X, int N, int D, int K, int n_runs, const float* centers, int32_t* labels,
                       float* mindist, float* inertia, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    int rc = latent_check("kmeans_predict", N, D, K);
    if (rc) return rc;
    DIC_REQUIRE(n_runs > 0 && n_runs <= 65535, DIC_ERR_INVALID_ARG, "kmeans_predict: n_runs=%d", n_runs);
    DIC_REQUIRE(X && centers && labels, DIC_ERR_INVALID_ARG, "kmeans_predict: NULL pointer");
    float* mind = mindist;
    if (inertia && !mind) {
        DIC_REQUIRE(workspace && workspace_bytes >= (size_t)n_runs * N * sizeof(float), DIC_ERR_WORKSPACE,
                    "kmeans_predict: inertia needs n_runs*N floats of workspace");
        mind = (float*)workspace;
    }
    const int nblk = km_blocks(N);
    hipStream_t st = (hipStream_t)stream;
    DIC_DISPATCH_KP(pad_k(K), hipLaunchKernelGGL((kmeans_assign_kernel<KP, false>), dim3(nblk, n_runs), dim3(kLatBlock), 0, st, X,
                                                 (const float*)nullptr, N, D, K, centers, labels, (const float*)nullptr, mind,
                                                 (float*)nullptr, (int*)nullptr, (int*)nullptr));
    if (inertia) hipLaunchKernelGGL(inertia_kernel, dim3(n_runs), dim3(256), 0, st, (const float*)mind, N, inertia);
    return check_launch("kmeans_predict");
}

static int pp_candidates(const float* X, int N, int D, int row_lo, int row_hi, const int64_t* cand, int L, int group,
                         const float* closest, float* dist_out, double* pot_out, void* workspace,
                         size_t workspace_bytes, dic_stream_t stream) {
    int rc = latent_check("kmeans_pp_candidates", N, D, 1);
    if (rc) return rc;
    DIC_REQUIRE(L > 0 && L <= 65535 && group > 0 && L % group == 0, DIC_ERR_INVALID_ARG, "kmeans_pp_candidates: L=%d group=%d", L, group);
    DIC_REQUIRE(0 <= row_lo && row_lo < row_hi && row_hi <= N, DIC_ERR_INVALID_ARG, "kmeans_pp_candidates: rows [%d, %d) of %d", row_lo, row_hi, N);
    DIC_REQUIRE(X && cand && closest && dist_out && pot_out && workspace, DIC_ERR_INVALID_ARG, "kmeans_pp_candidates: NULL pointer");
    const int nblk = km_blocks(row_hi - row_lo);
    DIC_REQUIRE(workspace_bytes >= (size_t)L * nblk * sizeof(double), DIC_ERR_WORKSPACE, "kmeans_pp_candidates: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    // candidates per pass over the rows: 16 / 32 / 64 (the LDS image of 64 candidate rows is 64 KB)
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kmeans_pp_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "kmeans_pp_candidates: cannot reserve 64 KB of LDS: %s", hipGetErrorString(e));
        attr_set = true;
    }
    if (L <= 16)
        hipLaunchKernelGGL(kmeans_pp_kernel<16>, dim3(nblk, 1), dim3(kLatBlock), 16 * 1024, st, X, N, D, row_lo, row_hi, cand, L, group, closest, dist_out,
                           (double*)workspace);
    else if (L <= 32)
        hipLaunchKernelGGL(kmeans_pp_kernel<32>, dim3(nblk, 1), dim3(kLatBlock), 32 * 1024, st, X, N, D, row_lo, row_hi, cand, L, group, closest, dist_out,
                           (double*)workspace);
    else
        hipLaunchKernelGGL(kmeans_pp_kernel<64>, dim3(nblk, (L + 63) / 64), dim3(kLatBlock), 64 * 1024, st, X, N, D, row_lo, row_hi, cand, L, group, closest,
                           dist_out, (double*)workspace);
    hipLaunchKernelGGL(kmeans_pp_finalize, dim3((L + 31) / 32), dim3(256), 0, st, (const double*)workspace, nblk, L, pot_out);
    return check_launch("kmeans_pp_candidates");
}

int dic_kmeans_pp_candidates(const float* X, int N, int D, const int64_t* cand, int L, int group,
                             const float* closest, float* dist_out, double* pot_out, void* workspace,
                             size_t workspace_bytes, dic_stream_t stream) {
    return pp_candidates(X, N, D, 0, N, cand, L, group, closest, dist_out, pot_out, workspace, workspace_bytes, stream);
}

int dic_kmeans_pp_candidates_rows(const float* X, int N, int D, int row_lo, int row_hi, const int64_t* cand, int L, int group,
                                  const float* closest, float* dist_out, double* pot_out, void* workspace,
                                  size_t workspace_bytes, dic_stream_t stream) {
    return pp_candidates(X, N, D, row_lo, row_hi, cand, L, group, closest, dist_out, pot_out, workspace, workspace_bytes, stream);
}

// f64 inclusive running sums along the rows of an f32 matrix: k-means++ draws its next centre where stable_cumsum(closest_dist_sq) crosses the sampled
// thresholds (sklearn/cluster/_kmeans.py:218-243; the reference's call sites: clustering_trainer.py:75-82, p2_clustering_optK.py:260-389).  One workgroup of
// 16 waves per row: every wave sums its contiguous chunk, the chunk totals are prefixed in a fixed order, then the chunk is scanned 64 elements at a time
// with the carry running along -- coalesced reads and writes, a deterministic summation order.
int dic_cumsum_f64(const float* x, int n_rows, int n, double* out, dic_stream_t stream) {
    DIC_REQUIRE(x && out, DIC_ERR_INVALID_ARG, "cumsum_f64: NULL pointer");
    DIC_REQUIRE(n_rows > 0 && n_rows <= 65535 && n > 0, DIC_ERR_INVALID_ARG, "cumsum_f64: %d rows of %d", n_rows, n);
    hipLaunchKernelGGL(cumsum_f64_kernel, dim3(n_rows), dim3(1024), 0, (hipStream_t)stream, x, n, out);
    return check_launch("cumsum_f64");
}

}  // extern "C"
