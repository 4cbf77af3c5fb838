// k2: Gaussian RBF de-interpolation from the reference grid back to the observed time stamps,
// its backward, and the masked reconstruction loss.
//
// Replaces RBF.forward minus compress_fc (rbf.py:57-108, gaussian rbf.py:129-131) and
// Net.rec_loss (clustering_interp.py:197-203); math per SURVEY.md Appendix A (k2).
//
// Forward: one lane per observation slot; the encounter tile's v (C x R grid values), the grid
// and the bandwidths sit in LDS (every lane of a row reads the same word = broadcast).
// Backward: same tiling as k1 -- per-slot (t, g*m/(N+eps), S/(N+eps)) triples are staged in LDS
// once, work items (row, grid point[, split]) stream them, so dL/dv needs no cross-lane
// reduction and dL/dbeta needs one per workgroup.
#include "dic_common.h"

namespace dic {

constexpr float kRbfEps = 1e-10f;   // rbf.py:107

struct RbfArgs {
    const float* x; const int32_t* lengths; int B, C, T, R, E;
    const float* ref_grid; const float* rbf_kernel; const float* v;
    float* y; float* norm;                       // forward outputs; norm (optional) = 1 / (sum_r phi + eps) per valid slot
    int prefix_only;                             // lengths given: write only the first n slots of each row (the padding stays as the caller allocated it)
    int v_rbc;                                   // v is laid out (R,B,C) -- the row order CompressFC produces it in -- instead of (B,C,R)
    const float* ob;                             // optional (B,C,T) observations: the reconstruction loss rides along (Net.rec_loss,
    double* sse_part;                            //   clustering_interp.py:197-203): per-workgroup [sum (y - ob)^2, #valid] over the valid slots
    StoreSrc st;                                 // st.row_off != NULL: time stamps (and, with `ob` set, the observations = st.v_pk) are read in
                                                 //   place from the ragged encounter store instead of x / ob; y and norm stay (B,C,T)
};

__host__ __device__ inline int rbf_fwd_words(int E, int C, int R) { return ((E * C * R + R + C + E * C + 1) & ~1) + 2 * E * C; }

// CT / RT: compile-time channel / grid-point counts (0 = run time).  The kernels decode (encounter, channel, grid point, slot) from flat
// indices everywhere; with the reference's shapes (C = 6 or 12, R = 24) as constants those divisions become multiply-shifts.
template <int CT, int RT>
__global__ __launch_bounds__(kBlock) void rbf_fwd_kernel(RbfArgs a) {
    extern __shared__ __align__(16) float smem[];
    const int C = CT ? CT : a.C, R = RT ? RT : a.R, T = a.T, E = a.E;
    float* vs = smem;                  // [E][C][R]
    float* refg = vs + E * C * R;      // [R]
    float* nbeta = refg + R;           // [C]  -beta*log2(e)
    int* cnt = reinterpret_cast<int*>(nbeta + C);   // [E*C]
    int64_t* roff = reinterpret_cast<int64_t*>(smem + ((E * C * R + R + C + E * C + 1) & ~1));      // [E*C] row offsets in a ragged store
    const int tid = threadIdx.x, e0 = blockIdx.x * E, Ev = min(E, a.B - e0), nrows = Ev * C;

    if (a.v_rbc) {
        for (int i = tid; i < nrows * R; i += kBlock) {      // i = r * nrows + (e * C + c): consecutive lanes read consecutive floats of a grid point's rows
            const int r = i / nrows, ec = i - r * nrows;
            vs[ec * R + r] = a.v[((size_t)r * a.B + e0) * C + ec];
        }
    } else {
        for (int i = tid; i < nrows * R; i += kBlock) vs[i] = a.v[(size_t)e0 * C * R + i];
    }
    for (int i = tid; i < R; i += kBlock) refg[i] = a.ref_grid[i];
    for (int i = tid; i < C; i += kBlock) nbeta[i] = -softplus_raw(a.rbf_kernel[i]) * kLog2e;
    for (int i = tid; i < nrows; i += kBlock) {
        cnt[i] = a.lengths ? max(0, min(a.lengths[(size_t)e0 * C + i], T)) : T;
        if (a.st.row_off) roff[i] = store_row_off(a.st, e0 + i / C, i % C, C);
    }
    __syncthreads();

    const int nchunk = (T + kWave - 1) / kWave, units = nrows * nchunk;
    const int wave = tid >> 6, lane = tid & 63;
    float sse = 0.f, nvalid = 0.f;
#pragma unroll 2
    for (int u = wave; u < units; u += kBlock / kWave) {
        const int row = u / nchunk;
        const int i = (u - row * nchunk) * kWave + lane;
        if (i >= T) continue;
        if (a.prefix_only && (u - row * nchunk) * kWave >= cnt[row]) continue;      // wave-uniform: the whole chunk is padding
        const int e = row / C, c = row - e * C;
        bool valid = i < cnt[row];
        float t = 0.f;
        const float* obrow;                          // the row's observations (dereferenced on valid slots of the fused loss only)
        if (a.st.row_off) {
            const int64_t off = roff[row];
            if (valid) t = a.st.t_pk[off + i];
            obrow = a.st.v_pk + off;
        } else {
            const float* base = a.x + (size_t)(e0 + e) * 4 * C * T;
            if (valid) {
                t = base[(size_t)(2 * C + c) * T + i];
                if (!a.lengths) valid = base[(size_t)(C + c) * T + i] != 0.f;
            }
            obrow = a.ob + ((size_t)(e0 + e) * C + c) * T;
        }
        float N = 0.f, S = 0.f;
        if (valid) {
            const float nb = nbeta[c];
            const float* vr = vs + row * R;
            for (int r = 0; r < R; ++r) {
                const float d = t - refg[r];
                const float phi = fast_exp2(nb * (d * d));
                N += phi;
                S = fmaf(phi, vr[r], S);
            }
        }
        if (a.prefix_only && !valid) continue;
        const size_t o = ((size_t)(e0 + e) * C + c) * T + i;
        const float inv = 1.0f / (N + kRbfEps);
        a.y[o] = valid ? S * inv : 0.f;
        if (a.norm) a.norm[o] = valid ? inv : 0.f;
        if (a.ob && valid) {
            const float d = S * inv - obrow[i];
            sse = fmaf(d, d, sse);
            nvalid += 1.f;
        }
    }
    if (a.ob) {          // fixed-order sums: lanes -> waves -> one pair per workgroup
        __syncthreads();                             // (the v tile is dead: its first words carry the wave sums)
        const double ws = wave_sum((double)sse), wc = wave_sum((double)nvalid);
        double* red = reinterpret_cast<double*>(smem);
        if (lane == 0) { red[wave] = ws; red[kBlock / kWave + wave] = wc; }
        __syncthreads();
        if (tid == 0) {
            double s2 = 0, c2 = 0;
            for (int w2 = 0; w2 < kBlock / kWave; ++w2) { s2 += red[w2]; c2 += red[kBlock / kWave + w2]; }
            a.sse_part[2 * blockIdx.x] = s2;
            a.sse_part[2 * blockIdx.x + 1] = c2;
        }
    }
}

// The forward for prefix masks (lengths given), a ROW PER WAVE (round 4).  The tile kernel above decodes (row, chunk) from a flat unit index for every
// 64-slot chunk, keeps the tile's v in LDS behind a workgroup barrier and spends 13.5 vector instructions per (slot, grid point) wave-pair where the arithmetic
// is 6 (SQ_INSTS_VALU, profiles/r4_kernels_B32768_pmc_sq.json: its vector pipes are 93 % busy -- with bookkeeping).  Here a wave walks whole rows: the row's
// grid values go through R words of wave-private LDS (no barrier: a wave's LDS operations execute in order), lane = slot, and the only per-chunk work besides
// the R-step loop is one clamped load and the stores.  Same expression order per slot as the tile kernel: y and 1/(N + eps) are bit-identical to it.
template <int RT, bool STORE>
__global__ __launch_bounds__(kBlock) void rbf_fwd_row_kernel(RbfArgs a) {
    constexpr int NW = kBlock / kWave, RP = DIC_MAX_REFPOINTS;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C, T = a.T, B = a.B, R = RT ? RT : a.R;
    float* lv = smem + wave * RP;          // [NW][RP] this wave's row of grid values
    float* refg = smem + NW * RP;          // [RP]
    float* nbeta = refg + RP;              // [C]  -beta*log2(e)
    for (int i = tid; i < R; i += kBlock) refg[i] = a.ref_grid[i];
    for (int i = tid; i < C; i += kBlock) nbeta[i] = -softplus_raw(a.rbf_kernel[i]) * kLog2e;
    __syncthreads();
    float sse = 0.f, nvalid = 0.f;
    const int nrows = B * C, nwaves = gridDim.x * NW;
    for (int row = blockIdx.x * NW + wave; row < nrows; row += nwaves) {
        const int e = row / C, c = row - e * C;
        const int n = max(0, min(a.lengths[row], T));
        const float nb = nbeta[c];
        if (lane < R) lv[lane] = a.v[a.v_rbc ? ((size_t)lane * B + e) * C + c : (size_t)row * R + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float* tp;
        const float* obp;
        if (STORE) {
            const int64_t so = store_row_off(a.st, e, c, C);
            tp = a.st.t_pk + so;
            obp = a.st.v_pk + so;
        } else {
            tp = a.x + ((size_t)e * 4 * C + 2 * C + c) * T;
            obp = a.ob + (size_t)row * T;
        }
        const size_t o0 = (size_t)row * T;
        const int nslots = a.prefix_only ? n : T;              // prefix_only: the padding stays as the caller allocated it
        for (int i0 = 0; i0 < nslots; i0 += kWave) {
            const int i = i0 + lane;
            const bool valid = i < n;
            float N = 0.f, S = 0.f;
            if (i0 < n) {                                      // (wave-uniform)
                const float t = tp[min(i, n - 1)];
                if (RT) {
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        const float d = t - refg[r];
                        const float phi = fast_exp2(nb * (d * d));
                        N += phi;
                        S = fmaf(phi, lv[r], S);
                    }
                } else {
                    for (int r = 0; r < R; ++r) {
                        const float d = t - refg[r];
                        const float phi = fast_exp2(nb * (d * d));
                        N += phi;
                        S = fmaf(phi, lv[r], S);
                    }
                }
            }
            if (i < nslots) {
                const float inv = 1.0f / (N + kRbfEps);
                a.y[o0 + i] = valid ? S * inv : 0.f;
                if (a.norm) a.norm[o0 + i] = valid ? inv : 0.f;
                if (a.ob && valid) {
                    const float d = S * inv - obp[i];
                    sse = fmaf(d, d, sse);
                    nvalid += 1.f;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // the next row's grid values overwrite lv only after this row's reads
        __builtin_amdgcn_wave_barrier();
    }
    if (a.ob) {          // fixed-order sums: lanes -> waves -> one pair per workgroup
        __syncthreads();
        const double ws = wave_sum((double)sse), wc = wave_sum((double)nvalid);
        double* red = reinterpret_cast<double*>(smem);
        if (lane == 0) { red[wave] = ws; red[NW + wave] = wc; }
        __syncthreads();
        if (tid == 0) {
            double s2 = 0, c2 = 0;
            for (int w2 = 0; w2 < NW; ++w2) { s2 += red[w2]; c2 += red[NW + w2]; }
            a.sse_part[2 * blockIdx.x] = s2;
            a.sse_part[2 * blockIdx.x + 1] = c2;
        }
    }
}

// thousands of per-workgroup [sse, count] pairs -> out2: 512 slices x 2 columns, fixed order (f64)
__global__ __launch_bounds__(1024) void sse_pairs_finalize(const double* partials, int nblk, float* out2) {
    __shared__ double red[1024];
    const int col = threadIdx.x & 1, sl = threadIdx.x >> 1;
    double acc = 0.0;
    for (int b = sl; b < nblk; b += 512) acc += partials[2 * (size_t)b + col];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int m = 512; m >= 2; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x < 2) out2[threadIdx.x] = (float)red[threadIdx.x];
}

// ------------------------------------------------------------------------------- backward
struct RbfBwdArgs {
    const float* x; const int32_t* lengths; int B, C, T, R, E, S, logS, nblk;
    const float* ref_grid; const float* rbf_kernel; const float* v;
    const float* y; const float* norm; const float* grad_y;
    float* grad_v; float* partials;
    int v_rbc;                                   // v and grad_v are laid out (R,B,C) instead of (B,C,R)
    // the fused reconstruction loss (dic_rbf_bwd_loss): grad_y is not materialised, dL/dy = 2 grad_loss (y - ob) / #valid on the valid slots
    const float* ob; const float* sse_count; const float* grad_loss;
    StoreSrc st;                                 // st.row_off != NULL: time stamps (and the observations of the fused loss) come from the store
};

struct RbfBwdLayout { int cnt, refg, nbeta, vs, gbeta, obs, stride, total_words; };
constexpr int kRbfMaxSplit = 16;
// odd stride (in float4 elements) + room for the loop-tail padding: see interp_row_stride
__host__ __device__ inline int rbf_row_stride(int T) { return (T + kRbfMaxSplit) | 1; }
__host__ __device__ inline RbfBwdLayout rbf_bwd_layout(int E, int C, int R, int T) {
    RbfBwdLayout L;
    int o = 0;
    L.cnt = o;   o += E * C + 1;      // +1: tile maximum
    L.refg = o;  o += R;
    L.nbeta = o; o += C;
    L.vs = o;    o += E * C * R;
    L.gbeta = o; o += E * C * R;      // per-item dL/dbeta terms
    o = (o + 3) & ~3;                 // float4 alignment
    L.stride = rbf_row_stride(T);
    L.obs = o;   o += 4 * E * C * L.stride;
    L.total_words = o;
    return L;
}

template <int S, int CT, int RT>
__global__ __launch_bounds__(kBlock) void rbf_bwd_kernel(RbfBwdArgs a) {
    constexpr int U = S >= 16 ? 1 : S >= 4 ? 2 : 4 / S;     // slots per trip and lane: U * S <= kRbfMaxSplit (the zero-weight padding behind each row)
    constexpr int LOGS = S == 1 ? 0 : S == 2 ? 1 : S == 4 ? 2 : S == 8 ? 3 : 4;
    extern __shared__ __align__(16) float smem[];
    const int C = CT ? CT : a.C, R = RT ? RT : a.R, T = a.T, E = a.E;
    const RbfBwdLayout L = rbf_bwd_layout(E, C, R, T);
    const int stride = L.stride;
    int* cnt = reinterpret_cast<int*>(smem + L.cnt);
    int* tile_max = cnt + E * C;
    float* refg = smem + L.refg; float* nbeta = smem + L.nbeta; float* vs = smem + L.vs; float* gb = smem + L.gbeta;
    float4* obs = reinterpret_cast<float4*>(smem + L.obs);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    for (int i = tid; i < R; i += kBlock) refg[i] = a.ref_grid[i];
    for (int i = tid; i < C; i += kBlock) nbeta[i] = -softplus_raw(a.rbf_kernel[i]) * kLog2e;
    const float gscale = a.ob ? 2.0f * a.grad_loss[0] / a.sse_count[1] : 0.f;
    float gbeta_acc = 0.f;          // channel tid / lanes-per-channel (every lane of the group carries it)

    for (int e0 = blockIdx.x * E; e0 < a.B; e0 += a.nblk * E) {
        const int Ev = min(E, a.B - e0), nrows = Ev * C;
        __syncthreads();
        if (tid == 0) *tile_max = 0;
        __syncthreads();
        if (a.v_rbc) {
            for (int i = tid; i < nrows * R; i += kBlock) {
                const int r = i / nrows, ec = i - r * nrows;
                vs[ec * R + r] = a.v[((size_t)r * a.B + e0) * C + ec];
            }
        } else {
            for (int i = tid; i < nrows * R; i += kBlock) vs[i] = a.v[(size_t)e0 * C * R + i];
        }
        for (int i = tid; i < nrows; i += kBlock) {
            const int n = a.lengths ? max(0, min(a.lengths[(size_t)e0 * C + i], T)) : T;
            cnt[i] = n;
            atomicMax(tile_max, n);
        }
        __syncthreads();
        // stage (t, a = g*m/(N+eps), a*y, -) per slot (y = S/(N+eps) for a valid slot), 4 chunks in flight per wave; masked slots and
        // the padding up to the tile maximum get weight 0 (so the streaming loop needs no bounds checks)
        const int npad = min(*tile_max + kRbfMaxSplit, stride);
        const int nchunk = (npad + kWave - 1) / kWave, units = nrows * nchunk;
        constexpr int NW = kBlock / kWave, G = 4;
        if (a.lengths) {
            // prefix masks: every address is known up front -- the 16 loads of a trip are issued back to back from clamped (always
            // valid) addresses and the padding is selected away afterwards (it may hold anything: prefix-only producers never write it)
            for (int u0 = wave * G; u0 < units; u0 += NW * G) {
                float tv[G], gy[G], nm[G], yv[G];
                int dst[G];
                bool valid[G];
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const int u = min(u0 + k, units - 1);
                    const int row = u / nchunk;
                    const int i = (u - row * nchunk) * kWave + lane;
                    dst[k] = (u0 + k < units && i < npad) ? row * stride + i : -1;
                    valid[k] = i < cnt[row];
                    const int e = row / C, c = row - e * C, ic = min(i, T - 1);
                    const size_t o = ((size_t)(e0 + e) * C + c) * T + ic;
#ifdef DIC_K2_EXP_NOSTAGE     // experiment: no global loads in the staging phase
                    tv[k] = (float)ic; gy[k] = 1.f; nm[k] = 0.5f; yv[k] = (float)o;
#else
                    if (a.st.row_off) {           // (clamped to the row: the store keeps 64 readable elements behind its last row for empty rows)
                        const int64_t so = store_row_off(a.st, e0 + e, c, C) + min(ic, max(cnt[row] - 1, 0));
                        tv[k] = a.st.t_pk[so];
                        gy[k] = a.ob ? a.st.v_pk[so] : a.grad_y[o];
                    } else {
                        tv[k] = a.x[((size_t)(e0 + e) * 4 * C + 2 * C + c) * T + ic];
                        gy[k] = a.ob ? a.ob[o] : a.grad_y[o];
                    }
                    nm[k] = a.norm[o]; yv[k] = a.y[o];
#endif
                }
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    if (a.ob) gy[k] = gscale * (yv[k] - gy[k]);
                    const float w = valid[k] ? gy[k] * nm[k] : 0.f;      // dL/dS = g/den (the forward saved 1/den)
                    if (dst[k] >= 0) obs[dst[k]] = make_float4(valid[k] ? tv[k] : 0.f, w, valid[k] ? w * yv[k] : 0.f, 0.f);
                }
            }
        } else {
            for (int u0 = wave * G; u0 < units; u0 += NW * G) {
                float4 val[G];
                int dst[G];
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const int u = u0 + k;
                    dst[k] = -1;
                    val[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (u < units) {
                        const int row = u / nchunk;
                        const int i = (u - row * nchunk) * kWave + lane;
                        if (i < npad) {
                            dst[k] = row * stride + i;
                            if (i < cnt[row]) {
                                const int e = row / C, c = row - e * C;
                                const float* base = a.x + (size_t)(e0 + e) * 4 * C * T;
                                const size_t o = ((size_t)(e0 + e) * C + c) * T + i;
                                const float t = base[(size_t)(2 * C + c) * T + i];
                                const float m = base[(size_t)(C + c) * T + i];
                                if (m != 0.f) {
                                    const float w = a.grad_y[o] * a.norm[o];
                                    val[k] = make_float4(t, w, w * a.y[o], 0.f);
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < G; ++k)
                    if (dst[k] >= 0) obs[dst[k]] = val[k];
            }
        }
        __syncthreads();
        const int nitems = nrows * R * S;
        for (int base = 0; base < nitems; base += kBlock) {
            const int item = base + tid;
            const bool live = item < nitems;
            const int it = live ? item : 0;
            const int s = it & (S - 1), q = it >> LOGS;
            const int row = q / R, r = q - row * R, c = row % C;
            const float4* p = obs + row * stride + s;
            const float ref = refg[r], nb = nbeta[c], vr = vs[row * R + r];
            int nw = live ? cnt[row] : 0;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) nw = max(nw, __shfl_xor(nw, m));
            nw = __builtin_amdgcn_readfirstlane(nw);
#ifdef DIC_K2_EXP_NOLOOP      // experiment (scripts/k2_experiments.sh): everything but the (slot, grid point) loop
            const int nj = a.B < 0 ? 2 : 0;
#else
            const int nj = ((nw + S - 1) / S + U - 1) / U * U;
#endif
            // dL/dv_r = sum_t a*phi;  dL/dbeta = -sum_t a*phi*u*(y - v_r), u = (t - ref)^2, kept as two sums (a*y was staged):
            // 7 vector operations + 1 exp per (slot, grid point)
            float gv = 0.f, q1 = 0.f, q2 = 0.f;
            for (int j = 0; j < nj; j += U) {
                float4 o[U];
#pragma unroll
                for (int k = 0; k < U; ++k) o[k] = p[(j + k) * S];
#pragma unroll
                for (int k = 0; k < U; ++k) {
                    const float d = o[k].x - ref;
                    const float u = d * d;
                    const float e = fast_exp2(nb * u);
                    gv = fmaf(o[k].y, e, gv);
                    const float eu = e * u;
                    q2 = fmaf(o[k].y, eu, q2);
                    q1 = fmaf(o[k].z, eu, q1);
                }
            }
            float gbt = fmaf(-vr, q2, q1);
#pragma unroll
            for (int m = 1; m < S; m <<= 1) { gv += __shfl_xor(gv, m); gbt += __shfl_xor(gbt, m); }
            if (live && s == 0) {
                if (a.v_rbc) a.grad_v[((size_t)r * a.B + e0) * C + row] = gv;
                else a.grad_v[((size_t)e0 * C + row) * R + r] = gv;
                gb[row * R + r] = gbt;
            }
        }
        __syncthreads();
        gbeta_acc += channel_sum_er(gb, Ev, C, R, tid);
    }
    const int lpc = channel_group_lanes(C);
    if (tid % lpc == 0 && tid / lpc < C) a.partials[(size_t)blockIdx.x * C + tid / lpc] = gbeta_acc;
}

// The same backward for the reference's own shape (C = 6 channels, R = 24 grid points, prefix masks), one ENCOUNTER PER WAVE:
// the generic kernel above spends more vector instructions on index arithmetic, tile bookkeeping and barrier phases than on the
// (slot, grid point) math (scripts/k2_experiments.sh: 104 of its 217 us remain with both the loop and the loads removed).  Here
//   * an encounter is C*R*4 = 576 (row, grid point, split) items = exactly 9 rounds of one wave, so the (row, grid point) of a lane in
//     round k never changes: its LDS offset, grid point, bandwidth and the offset of its v / grad_v element are computed once per
//     kernel, and each round's trip count is the larger of two SCALAR row lengths;
//   * waves share nothing -- no workgroup barrier inside the encounter loop (LDS operations of one wave execute in order), so the
//     waves of a CU drift apart and one wave's load latency is another's compute;
//   * the next encounter's rows (first 64 slots of t, g, 1/den, y per channel), its v elements and lengths are requested right
//     after the current one has been staged, and land during its 9 rounds;
//   * dL/dbeta stays in 9 per-lane accumulators over all encounters of the wave and is reduced once at the end (fixed order).
constexpr int kRbfWaveSplit = 4;
#ifndef DIC_K2_WAVE_U
#define DIC_K2_WAVE_U 2
#endif
constexpr int WU = DIC_K2_WAVE_U;      // slots per lane and trip of the (slot, grid point) loop: the LDS reads of one trip are issued together
// (Packed arithmetic was tried here -- slot planes read in pairs by ds_read2_b32, 7 v_pk_*_f32 + 2 v_exp_f32 per slot pair: 140 us
//  against 122 us for this scalar loop at the same occupancy.  v_pk_*_f32 saves instruction slots on gfx950, not cycles, and the
//  pair operands cost registers, i.e. occupancy.)
// STORE: time stamps (and the observations of the fused loss) come from the ragged encounter store (a.st) -- a variant of its own, so
// that the dense variant keeps its scalar-register budget (the kernel lives at the SGPR limit).
template <int C, int R, bool STORE>
__global__ __launch_bounds__(kBlock) void rbf_bwd_wave_kernel(RbfBwdArgs a) {
    constexpr int S = kRbfWaveSplit, ROUNDS = C * R * S / kWave, QPR = kWave / S;       // QPR (row, grid point) pairs per round
    static_assert(C * R * S % kWave == 0, "an encounter must be a whole number of wave rounds");
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, s = lane & (S - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = a.T, B = a.B, stride = rbf_row_stride(T);
    float4* obs = reinterpret_cast<float4*>(smem) + wave * C * stride;       // [C][stride] x (t, a = g/den, a*y, -)
    const int nwaves = a.nblk * (kBlock / kWave);

    int poff[ROUNDS], voff[ROUNDS], rowl[ROUNDS];
    float ref[ROUNDS], nb[ROUNDS], gacc[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int q = QPR * k + (lane >> 2), row = q / R, r = q - row * R;
        rowl[k] = row;
        poff[k] = row * stride + s;
        voff[k] = a.v_rbc ? r * B * C + row : q;
        ref[k] = a.ref_grid[r];
        nb[k] = -softplus_raw(a.rbf_kernel[row]) * kLog2e;
        gacc[k] = 0.f;
    }
    const int ic = min(lane, T - 1);
    const size_t vstep = a.v_rbc ? (size_t)C : (size_t)C * R;       // v / grad_v offset of encounter e = e * vstep + voff[k]
    const float* gsrc = a.ob ? a.ob : a.grad_y;                     // fused loss: the observations stand in for the incoming gradient
    const float gscale = a.ob ? 2.0f * a.grad_loss[0] / a.sse_count[1] : 0.f;

    float pt[C], pg[C], pn[C], py[C], pv[ROUNDS];
    int plen[C];
    // store mode: the C rows of an encounter lie back to back in the packed arrays: one 64-bit base + 32-bit deltas per staged encounter;
    // the encounter index of the NEXT request is fetched one encounter ahead (index -> offsets -> samples is a chain of dependent loads)
    int64_t pbase = 0;
    int pdelta[C];
    int enc_next = 0;
    constexpr bool from_store = STORE;
    auto store_enc = [&](int e) { return a.st.enc_idx ? a.st.enc_idx[e] : e; };
    auto request = [&](int e) {
        if (STORE) {
            const int64_t* ro = a.st.row_off + (size_t)enc_next * C;
            pbase = ro[0];
#pragma unroll
            for (int c = 0; c < C; ++c) pdelta[c] = (int)(ro[c] - pbase);
            if (e + nwaves < B) enc_next = store_enc(e + nwaves);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const size_t o = ((size_t)e * C + c) * T + ic;
#ifdef DIC_K2_EXP_NOSTAGE
            pt[c] = (float)ic; pg[c] = 1.f; pn[c] = 0.5f; py[c] = (float)o;
#else
            if (STORE) {                   // lane <= 63 slots past the row start: inside the store's padding even for its last row
                pt[c] = a.st.t_pk[pbase + pdelta[c] + ic];
                pg[c] = a.ob ? a.st.v_pk[pbase + pdelta[c] + ic] : gsrc[o];
            } else {
                pt[c] = a.x[((size_t)e * 4 * C + 2 * C + c) * T + ic];
                pg[c] = gsrc[o];
            }
            pn[c] = a.norm[o]; py[c] = a.y[o];
#endif
            plen[c] = a.lengths[(size_t)e * C + c];
        }
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) pv[k] = a.v[(size_t)e * vstep + voff[k]];
    };
    const int e_first = blockIdx.x * (kBlock / kWave) + wave;
    if (STORE && e_first < B) enc_next = store_enc(e_first);
    if (e_first < B) request(e_first);

    for (int e = e_first; e < B; e += nwaves) {
        int n[C], maxn = 0;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            n[c] = __builtin_amdgcn_readfirstlane(max(0, min(plen[c], T)));
            maxn = max(maxn, n[c]);
        }
        // slots 0..63 of every row: (t, a = g/den, a*y, -), zero weight behind the row's length
        if (lane < stride) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const bool valid = lane < n[c];
                const float g = a.ob ? gscale * (py[c] - pg[c]) : pg[c];
                const float w = valid ? g * pn[c] : 0.f;
                obs[c * stride + lane] = make_float4(valid ? pt[c] : 0.f, w, valid ? w * py[c] : 0.f, 0.f);
            }
        }
        if (maxn + kRbfMaxSplit > kWave) {        // (wave-uniform, rare at the reference's ~50 samples per channel) the rest of the rows
#pragma unroll
            for (int c = 0; c < C; ++c)
                for (int i = kWave + lane; i < stride; i += kWave) {
                    float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (i < n[c]) {
                        const size_t o = ((size_t)e * C + c) * T + i;
                        const float yo = a.y[o];
                        const float gsv = (from_store && a.ob) ? a.st.v_pk[pbase + pdelta[c] + i] : gsrc[o];
                        const float w = (a.ob ? gscale * (yo - gsv) : gsv) * a.norm[o];
                        const float tt = from_store ? a.st.t_pk[pbase + pdelta[c] + i] : a.x[((size_t)e * 4 * C + 2 * C + c) * T + i];
                        val = make_float4(tt, w, w * yo, 0.f);
                    }
                    obs[c * stride + i] = val;
                }
        }
        float vr[ROUNDS];
#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) vr[k] = pv[k];
        if (e + nwaves < B) request(e + nwaves);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

#pragma unroll
        for (int k = 0; k < ROUNDS; ++k) {
            const int row_lo = (QPR * k) / R, row_hi = (QPR * k + QPR - 1) / R;
            const int nw = max(n[row_lo], n[row_hi]);
#ifdef DIC_K2_EXP_NOLOOP
            const int nj = a.B < 0 ? 2 : 0;
#else
            const int nj = ((nw + S - 1) / S + WU - 1) / WU * WU;
#endif
            // dL/dv_r = sum_t a*phi;  dL/dbeta = -sum_t a*phi*u*(y - v_r), u = (t - ref)^2, kept as two sums (a*y was staged):
            // 7 vector operations + 1 exp per (slot, grid point)
            const float4* p = obs + poff[k];
            float gv = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll 1
            for (int j = 0; j < nj; j += WU) {
                float4 o[WU];
#pragma unroll
                for (int i = 0; i < WU; ++i) o[i] = p[(j + i) * S];     // (at most slot nw + 10: zero weight)
#pragma unroll
                for (int i = 0; i < WU; ++i) {
                    const float d = o[i].x - ref[k];
                    const float u = d * d;
                    const float ex = fast_exp2(nb[k] * u);
                    gv = fmaf(o[i].y, ex, gv);
                    const float eu = ex * u;
                    q2 = fmaf(o[i].y, eu, q2);
                    q1 = fmaf(o[i].z, eu, q1);
                }
            }
            gacc[k] += fmaf(-vr[k], q2, q1);
            // sum over the 4 splits of a (row, grid point): two quad permutes on the vector ALU (no LDS round trip)
            gv += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gv), 0xB1, 0xF, 0xF, false));
            gv += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, gv), 0x4E, 0xF, 0xF, false));
            if (s == 0) a.grad_v[(size_t)e * vstep + voff[k]] = gv;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // per-channel sums of this wave's dL/dbeta terms, then the 4 waves of the workgroup: one partial row per workgroup
    float chan[C];
#pragma unroll
    for (int c = 0; c < C; ++c) chan[c] = 0.f;
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int row_lo = (QPR * k) / R, row_hi = (QPR * k + QPR - 1) / R;
        chan[row_lo] += rowl[k] == row_lo ? gacc[k] : 0.f;
        if (row_hi != row_lo) chan[row_hi] += rowl[k] == row_hi ? gacc[k] : 0.f;
    }
    float* red = smem + (kBlock / kWave) * C * stride * 4;       // behind the 4 waves' slot buffers
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float t = wave_sum(chan[c]);
        if (lane == 0) red[wave * C + c] = t;
    }
    __syncthreads();
    if (tid < C) a.partials[(size_t)blockIdx.x * C + tid] = (red[tid] + red[C + tid]) + (red[2 * C + tid] + red[3 * C + tid]);
}

// ---- the backward with the SLOTS on the lanes (round 4; prefix masks, any C, R <= 4 RQ).  The two kernels above put (row, grid point[, split]) items on the
// lanes and stream the row's slots from an LDS copy: a staging phase under a barrier, 16 B of LDS read per (slot, grid point), whole rows of T + 16 slots
// resident (58 KB per encounter at C = 12, T = 288: two workgroups per CU), index decode per item.  At configs[3] (n = 200 slots per row) that is 37
// vector-instruction slots per (slot, grid point) where the arithmetic needs 11.  Here a wave owns a ROW at a time and nothing is staged: lane (q, s) =
// (lane >> 4, lane & 15) takes slot 16 k + s of chunk k straight from global memory into registers (four lanes share an address: one request) and grid
// points RQ q .. RQ q + RQ - 1, whose three sums it keeps in 3 RQ registers for the whole row -- 7 vector operations + 1 exp per (slot, grid point), no LDS,
// no barrier, no per-item decode; a row's tail wastes at most 15 slots (4 % at n = 200).  At the end of a row the sums of the 16 slot lanes are added by four
// DPP steps inside the 16-lane row (quad permutes, half mirror, mirror: 12 RQ instructions per row), RQ lanes per quarter store dL/dv, and the row's
// dL/dbeta term goes to the wave's per-channel accumulator in LDS.  ~60 registers: eight waves per SIMD hide the load latency of the 16-slot chunks.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float row16_sum(float x) {      // every lane of a 16-lane row ends up with the row's sum
    x += dpp_move<0xB1>(x);       // quad_perm [1,0,3,2]
    x += dpp_move<0x4E>(x);       // quad_perm [2,3,0,1]
    x += dpp_move<0x141>(x);      // row_half_mirror
    x += dpp_move<0x140>(x);      // row_mirror
    return x;
}

template <int RQ, bool STORE>
__global__ __launch_bounds__(kBlock) void rbf_bwd_slot_kernel(RbfBwdArgs a) {
    constexpr int NW = kBlock / kWave;
    extern __shared__ __align__(16) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, s = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C, T = a.T, B = a.B, R = a.R;
    float* nbeta = smem;               // [C]
    float* gb = smem + C;              // [NW][C] dL/dbeta terms of this wave's rows
    for (int i = tid; i < C; i += kBlock) nbeta[i] = -softplus_raw(a.rbf_kernel[i]) * kLog2e;
    for (int i = tid; i < NW * C; i += kBlock) gb[i] = 0.f;
    __syncthreads();
    float ref[RQ];
    bool rok[RQ];
#pragma unroll
    for (int j = 0; j < RQ; ++j) { rok[j] = RQ * q + j < R; ref[j] = a.ref_grid[min(RQ * q + j, R - 1)]; }
    const float gscale = a.ob ? 2.0f * a.grad_loss[0] / a.sse_count[1] : 0.f;
    const int nrows = B * C, nwaves = a.nblk * NW;
    for (int row = blockIdx.x * NW + wave; row < nrows; row += nwaves) {
        const int e = row / C, c = row - e * C;
        const int n = max(0, min(a.lengths[row], T));
        const float nb = nbeta[c];
        const size_t o0 = (size_t)row * T;
        const float* tp;
        const float* gp;
        if (STORE) {
            const int64_t so = store_row_off(a.st, e, c, C);
            tp = a.st.t_pk + so;
            gp = a.ob ? a.st.v_pk + so : a.grad_y + o0;
        } else {
            tp = a.x + ((size_t)e * 4 * C + 2 * C + c) * T;
            gp = (a.ob ? a.ob : a.grad_y) + o0;
        }
        const float* np = a.norm + o0;
        const float* yp = a.y + o0;
        float vr[RQ], gv[RQ], q1[RQ], q2[RQ];
#pragma unroll
        for (int j = 0; j < RQ; ++j) {
            const int r = min(RQ * q + j, R - 1);
            vr[j] = a.v[a.v_rbc ? ((size_t)r * B + e) * C + c : (size_t)row * R + r];
            gv[j] = q1[j] = q2[j] = 0.f;
        }
        const int last = max(n - 1, 0), nchunks = (n + 15) >> 4;
        // exp(-beta u) = exp2(-(sc d)^2), sc = sqrt(beta log2 e): the bandwidth rides on the time stamps (one multiplication per slot and RQ per row
        // instead of one per (slot, grid point)); the two u-weighted sums come out scaled by sc^2 and are scaled back once per row
        // (a bandwidth whose softplus underflows to 0 -- raw parameter below about -100 -- gives sc = 0 and all-zero scaled sums: no rescale then,
        //  0 * inf would put a NaN into dL/dbeta, whose factor sigmoid(raw) is 0 there anyway)
        const float sc = __builtin_sqrtf(-nb), unscale = nb < 0.f ? __builtin_amdgcn_rcpf(-nb) : 0.f;
        float refs[RQ];
#pragma unroll
        for (int j = 0; j < RQ; ++j) refs[j] = sc * ref[j];
        // The next chunk's four loads stay in flight across this chunk's arithmetic -- behind a compiler barrier: without one the loop is rotated, the
        // loads sink to the top of the iteration that uses them and every chunk exposes a load round trip.  Indices are clamped into the row: the slots
        // behind it may hold anything -- they get weight 0.  (Unsigned byte offsets: scalar base + 32-bit vector offset addressing.)
        auto at = [](const float* base, unsigned byte_off) { return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off); };
        unsigned off = 4u * (unsigned)min(s, last);
        float tv = at(tp, off), gy = at(gp, off), nm = at(np, off), yv = at(yp, off);
        for (int k = 0; k < nchunks; ++k) {
            const int i = 16 * k + s;
            off = 4u * (unsigned)min(i + 16, last);
            const float tv2 = at(tp, off), gy2 = at(gp, off), nm2 = at(np, off), yv2 = at(yp, off);
            asm volatile("" ::: "memory");
            const float g = a.ob ? gscale * (yv - gy) : gy;
            const float w = i < n ? g * nm : 0.f;      // dL/dS = g/den (the forward saved 1/den)
            const float wy = w * yv;
            const float ts = sc * tv;
#pragma unroll
            for (int j = 0; j < RQ; ++j) {
                const float d = ts - refs[j];
                const float u = d * d;
                const float ex = fast_exp2(-u);
                gv[j] = fmaf(w, ex, gv[j]);
                const float eu = ex * u;
                q2[j] = fmaf(w, eu, q2[j]);
                q1[j] = fmaf(wy, eu, q1[j]);
            }
            tv = tv2; gy = gy2; nm = nm2; yv = yv2;
        }
        float gbt = 0.f, out = 0.f;
#pragma unroll
        for (int j = 0; j < RQ; ++j) {
            gv[j] = row16_sum(gv[j]);
            const float t1 = row16_sum(q1[j]), t2 = row16_sum(q2[j]);
            gbt += rok[j] ? unscale * fmaf(-vr[j], t2, t1) : 0.f;
            out = s == j ? gv[j] : out;
        }
        gbt += __shfl_xor(gbt, 16);
        gbt += __shfl_xor(gbt, 32);
        if (lane == 0) gb[wave * C + c] += gbt;
        const int r = RQ * q + s;
        if (s < RQ && r < R) a.grad_v[a.v_rbc ? ((size_t)r * B + e) * C + c : (size_t)row * R + r] = out;
    }
    __syncthreads();
    if (tid < C) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += gb[w * C + tid];
        a.partials[(size_t)blockIdx.x * C + tid] = sum;
    }
}

__global__ __launch_bounds__(256) void rbf_bwd_finalize(const float* partials, int nblk, int C, const float* rbf_kernel,
                                                       float* grad_kernel) {
    __shared__ double red[256];
    const double s = reduce_partials_32x8(partials, nblk, C, blockIdx.x * 32, red);
    const int c = blockIdx.x * 32 + threadIdx.x;
    if (threadIdx.x < 32 && c < C) grad_kernel[c] = (float)(s * (double)sigmoidf(rbf_kernel[c]));
}

// ------------------------------------------------------------------------------- rec loss
__global__ __launch_bounds__(kBlock) void masked_sse_kernel(const float* ob, const float* rec, const float* mask,
                                                           const int32_t* lengths, int rows, int T, int nblk,
                                                           double* partials) {
    __shared__ double red[2][kBlock / kWave];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nchunk = (T + kWave - 1) / kWave;
    const long units = (long)rows * nchunk;
    float sse = 0.f, cntv = 0.f;
    // four (row, 64-slot chunk) units per trip, all loads issued before any is used: one unit per trip is a chain of
    // dependent loads (length -> values) and ran at 0.75 TB/s
    const long stride = (long)nblk * (kBlock / kWave);
    for (long u0 = (long)blockIdx.x * (kBlock / kWave) + wave; u0 < units; u0 += 4 * stride) {
        size_t o[4];
        bool live[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long u = u0 + k * stride;
            const long row = min(u, units - 1) / nchunk;
            const int i = (int)(min(u, units - 1) - row * nchunk) * kWave + lane;
            const int n = lengths ? max(0, min(lengths[row], T)) : T;
            live[k] = u < units && i < n;
            o[k] = (size_t)row * T + min(i, T - 1);
        }
        float m[4], r[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m[k] = (mask && live[k]) ? mask[o[k]] : 1.f;
            r[k] = live[k] ? rec[o[k]] : 0.f;
            b[k] = live[k] ? ob[o[k]] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (live[k]) {
                const float d = r[k] * m[k] - b[k] * m[k];
                sse = fmaf(d, d, sse);
                cntv += (m[k] == 1.f) ? 1.f : 0.f;
            }
        }
    }
    const double ws = wave_sum((double)sse), wc = wave_sum((double)cntv);
    if (lane == 0) { red[0][wave] = ws; red[1][wave] = wc; }
    __syncthreads();
    if (tid == 0) {
        double s = 0, c = 0;
        for (int w = 0; w < kBlock / kWave; ++w) { s += red[0][w]; c += red[1][w]; }
        partials[2 * blockIdx.x] = s;
        partials[2 * blockIdx.x + 1] = c;
    }
}

__global__ __launch_bounds__(256) void masked_sse_finalize(const double* partials, int nblk, float* out2) {
    __shared__ double red[256];
    const double s = reduce_partials_32x8(partials, nblk, 2, 0, red);
    if (threadIdx.x < 2) out2[threadIdx.x] = (float)s;
}

__global__ __launch_bounds__(kBlock) void masked_sse_bwd_kernel(const float* ob, const float* rec, const float* mask,
                                                               const int32_t* lengths, long total, int T,
                                                               const float* sse_count, const float* grad_loss,
                                                               float* grad_rec, int prefix_only) {
    const float scale = 2.0f * grad_loss[0] / sse_count[1];
    for (long o = (long)blockIdx.x * kBlock + threadIdx.x; o < total; o += (long)gridDim.x * kBlock) {
        float m;
        if (lengths) {
            const long row = o / T;
            m = ((int)(o - row * T) < lengths[row]) ? 1.f : 0.f;
            if (prefix_only && m == 0.f) continue;          // padding: never read by the de-interpolation backward
            if (mask && m != 0.f) m = mask[o];
        } else {
            m = mask[o];
        }
        grad_rec[o] = (m != 0.f) ? scale * m * (rec[o] * m - ob[o] * m) : 0.f;
    }
}

static int rbf_tile(int B, int per_enc_words, int fixed_words, int budget_bytes) {
    int E = (budget_bytes / 4 - fixed_words) / per_enc_words;
    E = max(1, min(E, 16));
    return max(1, min(E, max(1, B / (8 * kNumCU))));
}

static int sse_blocks(int rows, int T) {
    const long units = (long)rows * ((T + kWave - 1) / kWave);
    return (int)max(1L, min((units + 15) / 16, (long)4 * kNumCU));
}

}  // namespace dic

using namespace dic;

extern "C" {

static int rbf_fwd_grid(int B, int C, int R, int* E) {
    const int per_enc = rbf_fwd_words(2, C, R) - rbf_fwd_words(1, C, R);
    *E = rbf_tile(B, per_enc, rbf_fwd_words(1, C, R) - per_enc, 32 * 1024);
    return (B + *E - 1) / *E;
}

// row-per-wave variant (rbf_fwd_row_kernel; prefix masks): DIC_RBF_FWD_ROW=0 keeps the tile kernel (A/B)
static int rbf_fwd_row_grid(int B, int C) {
    const int wpb = kBlock / kWave;
    return (int)min(((long)B * C + wpb - 1) / wpb, (long)8 * kNumCU);
}
static bool rbf_fwd_row_on() {
    const char* e = getenv("DIC_RBF_FWD_ROW");
    return !(e && e[0] == '0');
}

static int rbf_fwd_launch(RbfArgs a, float* out2, hipStream_t st) {
    const int B = a.B, C = a.C, T = a.T, R = a.R;
    DIC_REQUIRE(B > 0 && C > 0 && T > 0 && R > 0, DIC_ERR_INVALID_ARG, "rbf_fwd: non-positive size");
    DIC_REQUIRE(C <= DIC_MAX_CHANNELS && R <= DIC_MAX_REFPOINTS, DIC_ERR_UNSUPPORTED, "rbf_fwd: C=%d R=%d", C, R);
    DIC_REQUIRE((a.x || a.st.row_off) && a.ref_grid && a.rbf_kernel && a.v && a.y, DIC_ERR_INVALID_ARG, "rbf_fwd: NULL pointer");
    if (a.lengths && rbf_fwd_row_on()) {
        const dim3 rgrid(rbf_fwd_row_grid(B, C));
        const size_t rlds = (size_t)((kBlock / kWave + 1) * DIC_MAX_REFPOINTS + C) * sizeof(float);
        const bool store = a.st.row_off != nullptr;
        if (R == 24) {
            if (store) hipLaunchKernelGGL((rbf_fwd_row_kernel<24, true>), rgrid, dim3(kBlock), rlds, st, a);
            else hipLaunchKernelGGL((rbf_fwd_row_kernel<24, false>), rgrid, dim3(kBlock), rlds, st, a);
        } else {
            if (store) hipLaunchKernelGGL((rbf_fwd_row_kernel<0, true>), rgrid, dim3(kBlock), rlds, st, a);
            else hipLaunchKernelGGL((rbf_fwd_row_kernel<0, false>), rgrid, dim3(kBlock), rlds, st, a);
        }
        if (a.ob) hipLaunchKernelGGL(sse_pairs_finalize, dim3(1), dim3(1024), 0, st, (const double*)a.sse_part, (int)rgrid.x, out2);
        return check_launch("rbf_fwd");
    }
    const dim3 grid(rbf_fwd_grid(B, C, R, &a.E));
    const size_t lds = (size_t)rbf_fwd_words(a.E, C, R) * 4;
    if (C == 6 && R == 24) hipLaunchKernelGGL((rbf_fwd_kernel<6, 24>), grid, dim3(kBlock), lds, st, a);
    else if (C == 12 && R == 24) hipLaunchKernelGGL((rbf_fwd_kernel<12, 24>), grid, dim3(kBlock), lds, st, a);
    else hipLaunchKernelGGL((rbf_fwd_kernel<0, 0>), grid, dim3(kBlock), lds, st, a);
    if (a.ob) hipLaunchKernelGGL(sse_pairs_finalize, dim3(1), dim3(1024), 0, st, (const double*)a.sse_part, (int)grid.x, out2);
    return check_launch("rbf_fwd");
}

int dic_rbf_fwd(const float* x, const int32_t* lengths, int B, int C, int T, int R, const float* ref_grid,
                const float* rbf_kernel, const float* v, int v_time_major, float* y, float* norm, int prefix_only, dic_stream_t stream) {
    RbfArgs a{x, lengths, B, C, T, R, 1, ref_grid, rbf_kernel, v, y, norm, (prefix_only && lengths) ? 1 : 0, v_time_major != 0, nullptr, nullptr};
    return rbf_fwd_launch(a, nullptr, (hipStream_t)stream);
}

size_t dic_rbf_fwd_loss_workspace(int B, int C, int T, int R) {
    if (B <= 0 || C <= 0 || T <= 0 || R <= 0) return 0;
    int E;
    return (size_t)max(rbf_fwd_grid(B, C, R, &E), rbf_fwd_row_grid(B, C)) * 2 * sizeof(double);
}

int dic_rbf_fwd_loss(const float* x, const int32_t* lengths, int B, int C, int T, int R, const float* ref_grid, const float* rbf_kernel,
                     const float* v, int v_time_major, const float* ob, float* y, float* norm, int prefix_only, float* out2,
                     void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(ob && out2 && workspace, DIC_ERR_INVALID_ARG, "rbf_fwd_loss: NULL pointer");
    DIC_REQUIRE(workspace_bytes >= dic_rbf_fwd_loss_workspace(B, C, T, R), DIC_ERR_WORKSPACE, "rbf_fwd_loss: workspace too small");
    RbfArgs a{x, lengths, B, C, T, R, 1, ref_grid, rbf_kernel, v, y, norm, (prefix_only && lengths) ? 1 : 0, v_time_major != 0, ob, (double*)workspace};
    return rbf_fwd_launch(a, out2, (hipStream_t)stream);
}

int dic_rbf_fwd_store(const float* t_pk, const float* v_pk, const int64_t* row_off, const int32_t* enc_idx, const int32_t* lengths,
                      int B, int C, int T, int R, const float* ref_grid, const float* rbf_kernel, const float* v, int v_time_major,
                      int with_loss, float* y, float* norm, int prefix_only, float* out2, void* workspace, size_t workspace_bytes,
                      dic_stream_t stream) {
    DIC_REQUIRE(t_pk && v_pk && row_off && lengths, DIC_ERR_INVALID_ARG, "rbf_fwd_store: NULL store pointer / lengths");
    DIC_REQUIRE(!with_loss || (out2 && workspace), DIC_ERR_INVALID_ARG, "rbf_fwd_store: the fused loss needs out2 and a workspace");
    DIC_REQUIRE(!with_loss || workspace_bytes >= dic_rbf_fwd_loss_workspace(B, C, T, R), DIC_ERR_WORKSPACE, "rbf_fwd_store: workspace too small");
    RbfArgs a{nullptr, lengths, B, C, T, R, 1, ref_grid, rbf_kernel, v, y, norm, prefix_only ? 1 : 0, v_time_major != 0,
              with_loss ? v_pk : nullptr, with_loss ? (double*)workspace : nullptr, StoreSrc{t_pk, v_pk, nullptr, row_off, enc_idx}};
    return rbf_fwd_launch(a, with_loss ? out2 : nullptr, (hipStream_t)stream);
}

static void rbf_bwd_geometry(int B, int C, int T, int R, int* E, int* nblk, size_t* lds) {
    const int one = rbf_bwd_layout(1, C, R, T).total_words, two = rbf_bwd_layout(2, C, R, T).total_words;
    *E = rbf_tile(B, two - one, one - (two - one), 24 * 1024);
    *nblk = min((B + *E - 1) / *E, 8 * kNumCU);
    *lds = (size_t)rbf_bwd_layout(*E, C, R, T).total_words * 4;
}

// wave-per-encounter variant (rbf_bwd_wave_kernel): workgroups of 4 waves, as many as stay resident, trimmed so that every wave
// gets the same number of encounters (to within one)
static bool rbf_bwd_wave_geometry(int B, int C, int T, int R, int* nblk, size_t* lds) {
    if (!(C == 6 && R == 24)) return false;
    const int wpb = kBlock / kWave;
    *lds = (size_t)wpb * C * rbf_row_stride(T) * sizeof(float4) + (size_t)wpb * C * sizeof(float);
    if (*lds > 48 * 1024) return false;
    const int resident = (int)min((size_t)8, (size_t)160 * 1024 / *lds) * kNumCU;
    int n = min((B + wpb - 1) / wpb, resident);
    const int per = (B + n * wpb - 1) / (n * wpb);         // encounters per wave
    *nblk = (B + per * wpb - 1) / (per * wpb);
    return true;
}

// slots-on-lanes variant (rbf_bwd_slot_kernel): one row per wave at a time; DIC_RBF_BWD_SLOT=0 switches it off, =2 prefers it to the
// wave-per-encounter kernel where both apply (A/B)
static int rbf_bwd_slot_mode() {
    const char* e = getenv("DIC_RBF_BWD_SLOT");
    return (e && e[0]) ? atoi(e) : 1;
}
static bool rbf_bwd_slot_geometry(int B, int C, int R, int* nblk) {
    if (R > 24 || rbf_bwd_slot_mode() == 0) return false;
    const int wpb = kBlock / kWave;
    *nblk = (int)min(((long)B * C + wpb - 1) / wpb, (long)5 * kNumCU);      // 84 registers: five waves per SIMD = five workgroups per CU, one round
    return true;
}

size_t dic_rbf_bwd_workspace(int B, int C, int T, int R) {
    if (B <= 0 || C <= 0 || T <= 0 || R <= 0) return 0;
    int E, nblk, nblk2 = 0, nblk3 = 0; size_t lds;
    rbf_bwd_geometry(B, C, T, R, &E, &nblk, &lds);
    rbf_bwd_wave_geometry(B, C, T, R, &nblk2, &lds);
    rbf_bwd_slot_geometry(B, C, R, &nblk3);
    return (size_t)max(max(nblk, nblk2), nblk3) * C * sizeof(float);
}

static int rbf_bwd_launch(const float* x, const int32_t* lengths, int B, int C, int T, int R, const float* ref_grid,
                          const float* rbf_kernel, const float* v, int v_time_major, const float* y, const float* norm, const float* grad_y,
                          const float* ob, const float* sse_count, const float* grad_loss,
                          float* grad_v, float* grad_rbf_kernel, void* workspace, size_t workspace_bytes, dic_stream_t stream,
                          const StoreSrc* store = nullptr) {
    DIC_REQUIRE(B > 0 && C > 0 && T > 0 && R > 0, DIC_ERR_INVALID_ARG, "rbf_bwd: non-positive size");
    DIC_REQUIRE(C <= DIC_MAX_CHANNELS && R <= DIC_MAX_REFPOINTS, DIC_ERR_UNSUPPORTED, "rbf_bwd: C=%d R=%d", C, R);
    DIC_REQUIRE((x || store) && ref_grid && rbf_kernel && v && y && norm && (grad_y || ob) && grad_v && grad_rbf_kernel && workspace,
                DIC_ERR_INVALID_ARG, "rbf_bwd: NULL pointer");
    DIC_REQUIRE(!ob || (lengths && sse_count && grad_loss), DIC_ERR_INVALID_ARG, "rbf_bwd_loss: needs lengths, sse_count and grad_loss");
    RbfBwdArgs a{};
    a.x = x; a.lengths = lengths; a.B = B; a.C = C; a.T = T; a.R = R;
    a.ref_grid = ref_grid; a.rbf_kernel = rbf_kernel; a.v = v; a.y = y; a.norm = norm; a.grad_y = grad_y;
    a.grad_v = grad_v; a.partials = (float*)workspace; a.v_rbc = v_time_major != 0;
    a.ob = ob; a.sse_count = sse_count; a.grad_loss = grad_loss;
    if (store) a.st = *store;
    size_t lds;
    hipStream_t st = (hipStream_t)stream;
    const bool wave_ok = lengths && (size_t)B * C * R < ((size_t)1 << 31) && rbf_bwd_wave_geometry(B, C, T, R, &a.nblk, &lds);
    if (lengths && (!wave_ok || rbf_bwd_slot_mode() == 2) && rbf_bwd_slot_geometry(B, C, R, &a.nblk)) {
        DIC_REQUIRE(workspace_bytes >= (size_t)a.nblk * C * sizeof(float), DIC_ERR_WORKSPACE, "rbf_bwd: workspace too small");
        const size_t slds = (size_t)(1 + kBlock / kWave) * C * sizeof(float);
        if (R <= 8) {
            if (store) hipLaunchKernelGGL((rbf_bwd_slot_kernel<2, true>), dim3(a.nblk), dim3(kBlock), slds, st, a);
            else hipLaunchKernelGGL((rbf_bwd_slot_kernel<2, false>), dim3(a.nblk), dim3(kBlock), slds, st, a);
        } else {
            if (store) hipLaunchKernelGGL((rbf_bwd_slot_kernel<6, true>), dim3(a.nblk), dim3(kBlock), slds, st, a);
            else hipLaunchKernelGGL((rbf_bwd_slot_kernel<6, false>), dim3(a.nblk), dim3(kBlock), slds, st, a);
        }
        hipLaunchKernelGGL(rbf_bwd_finalize, dim3((C + 31) / 32), dim3(256), 0, st, (const float*)workspace, a.nblk, C, rbf_kernel,
                           grad_rbf_kernel);
        return check_launch("rbf_bwd");
    }
    if (wave_ok) {
        rbf_bwd_wave_geometry(B, C, T, R, &a.nblk, &lds);
        DIC_REQUIRE(workspace_bytes >= (size_t)a.nblk * C * sizeof(float), DIC_ERR_WORKSPACE, "rbf_bwd: workspace too small");
        if (store) hipLaunchKernelGGL((rbf_bwd_wave_kernel<6, 24, true>), dim3(a.nblk), dim3(kBlock), lds, st, a);
        else hipLaunchKernelGGL((rbf_bwd_wave_kernel<6, 24, false>), dim3(a.nblk), dim3(kBlock), lds, st, a);
        hipLaunchKernelGGL(rbf_bwd_finalize, dim3((C + 31) / 32), dim3(256), 0, st, (const float*)workspace, a.nblk, C, rbf_kernel,
                           grad_rbf_kernel);
        return check_launch("rbf_bwd");
    }
    rbf_bwd_geometry(B, C, T, R, &a.E, &a.nblk, &lds);
    DIC_REQUIRE(lds <= 64 * 1024, DIC_ERR_UNSUPPORTED, "rbf_bwd: one encounter needs %zu B of LDS", lds);
    DIC_REQUIRE(workspace_bytes >= (size_t)a.nblk * C * sizeof(float), DIC_ERR_WORKSPACE, "rbf_bwd: workspace too small");
    {   // lanes per (row, grid point) item: see sci_cci_fwd
        const int base_items = a.E * C * R;
        const int nest = max(8, T / 2);
        double best = 1e30;
        a.S = 1; a.logS = 0;
        for (int cand = 1, lg = 0; cand <= kRbfMaxSplit; cand <<= 1, ++lg) {
            const int rounds = (base_items * cand + kBlock - 1) / kBlock;
            const double cost = rounds * ((double)nest / cand * 12.0 + 6.0 * lg + 25.0);
            if (cost < best * 0.97) { best = cost; a.S = cand; a.logS = lg; }
        }
    }
    switch (a.S) {
#define DIC_RBF_BWD(SS)                                                                                                   \
        if (C == 6 && R == 24) hipLaunchKernelGGL((rbf_bwd_kernel<SS, 6, 24>), dim3(a.nblk), dim3(kBlock), lds, st, a);       \
        else if (C == 12 && R == 24) hipLaunchKernelGGL((rbf_bwd_kernel<SS, 12, 24>), dim3(a.nblk), dim3(kBlock), lds, st, a); \
        else hipLaunchKernelGGL((rbf_bwd_kernel<SS, 0, 0>), dim3(a.nblk), dim3(kBlock), lds, st, a);                          \
        break;
        case 1: DIC_RBF_BWD(1)
        case 2: DIC_RBF_BWD(2)
        case 4: DIC_RBF_BWD(4)
        case 8: DIC_RBF_BWD(8)
        default: DIC_RBF_BWD(16)
#undef DIC_RBF_BWD
    }
    hipLaunchKernelGGL(rbf_bwd_finalize, dim3((C + 31) / 32), dim3(256), 0, st, (const float*)workspace, a.nblk, C, rbf_kernel,
                       grad_rbf_kernel);
    return check_launch("rbf_bwd");
}

int dic_rbf_bwd(const float* x, const int32_t* lengths, int B, int C, int T, int R, const float* ref_grid,
                const float* rbf_kernel, const float* v, int v_time_major, const float* y, const float* norm, const float* grad_y,
                float* grad_v, float* grad_rbf_kernel, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(grad_y, DIC_ERR_INVALID_ARG, "rbf_bwd: grad_y is NULL");
    return rbf_bwd_launch(x, lengths, B, C, T, R, ref_grid, rbf_kernel, v, v_time_major, y, norm, grad_y, nullptr, nullptr, nullptr, grad_v,
                          grad_rbf_kernel, workspace, workspace_bytes, stream);
}

int dic_rbf_bwd_loss(const float* x, const int32_t* lengths, int B, int C, int T, int R, const float* ref_grid, const float* rbf_kernel,
                     const float* v, int v_time_major, const float* y, const float* norm, const float* ob, const float* sse_count,
                     const float* grad_loss, float* grad_v, float* grad_rbf_kernel, void* workspace, size_t workspace_bytes,
                     dic_stream_t stream) {
    DIC_REQUIRE(ob, DIC_ERR_INVALID_ARG, "rbf_bwd_loss: ob is NULL");
    return rbf_bwd_launch(x, lengths, B, C, T, R, ref_grid, rbf_kernel, v, v_time_major, y, norm, nullptr, ob, sse_count, grad_loss, grad_v,
                          grad_rbf_kernel, workspace, workspace_bytes, stream);
}

int dic_rbf_bwd_store(const float* t_pk, const float* v_pk, const int64_t* row_off, const int32_t* enc_idx, const int32_t* lengths,
                      int B, int C, int T, int R, const float* ref_grid, const float* rbf_kernel, const float* v, int v_time_major,
                      const float* y, const float* norm, const float* grad_y, const float* sse_count, const float* grad_loss,
                      float* grad_v, float* grad_rbf_kernel, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(t_pk && v_pk && row_off && lengths, DIC_ERR_INVALID_ARG, "rbf_bwd_store: NULL store pointer / lengths");
    const StoreSrc st{t_pk, v_pk, nullptr, row_off, enc_idx};
    // grad_y == NULL: the fused reconstruction loss -- the observations are the store's values
    return rbf_bwd_launch(nullptr, lengths, B, C, T, R, ref_grid, rbf_kernel, v, v_time_major, y, norm, grad_y, grad_y ? nullptr : v_pk,
                          sse_count, grad_loss, grad_v, grad_rbf_kernel, workspace, workspace_bytes, stream, &st);
}

size_t dic_masked_sse_workspace(int B, int C, int T) {
    if (B <= 0 || C <= 0 || T <= 0) return 0;
    return (size_t)sse_blocks(B * C, T) * 2 * sizeof(double);
}

int dic_masked_sse_fwd(const float* ob, const float* rec, const float* mask, const int32_t* lengths, int B, int C, int T,
                       float* out2, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(B > 0 && C > 0 && T > 0, DIC_ERR_INVALID_ARG, "masked_sse: non-positive size");
    DIC_REQUIRE(ob && rec && out2 && workspace && (mask || lengths), DIC_ERR_INVALID_ARG, "masked_sse: NULL pointer");
    const int nblk = sse_blocks(B * C, T);
    DIC_REQUIRE(workspace_bytes >= (size_t)nblk * 2 * sizeof(double), DIC_ERR_WORKSPACE, "masked_sse: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(masked_sse_kernel, dim3(nblk), dim3(kBlock), 0, st, ob, rec, mask, lengths, B * C, T, nblk,
                       (double*)workspace);
    hipLaunchKernelGGL(masked_sse_finalize, dim3(1), dim3(256), 0, st, (const double*)workspace, nblk, out2);
    return check_launch("masked_sse_fwd");
}

int dic_masked_sse_bwd(const float* ob, const float* rec, const float* mask, const int32_t* lengths, int B, int C, int T,
                       const float* sse_count, const float* grad_loss, float* grad_rec, int prefix_only, dic_stream_t stream) {
    DIC_REQUIRE(B > 0 && C > 0 && T > 0, DIC_ERR_INVALID_ARG, "masked_sse_bwd: non-positive size");
    DIC_REQUIRE(ob && rec && sse_count && grad_loss && grad_rec && (mask || lengths), DIC_ERR_INVALID_ARG,
                "masked_sse_bwd: NULL pointer");
    const long total = (long)B * C * T;
    const int grid = (int)min((total + kBlock - 1) / kBlock, (long)16 * kNumCU);
    hipLaunchKernelGGL(masked_sse_bwd_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, ob, rec, mask, lengths,
                       total, T, sse_count, grad_loss, grad_rec, (prefix_only && lengths) ? 1 : 0);
    return check_launch("masked_sse_bwd");
}

}  // extern "C"
