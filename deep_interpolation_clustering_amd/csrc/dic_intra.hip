// The gap statistic's "inertia" of one clustering (p2_clustering_optK.py:334-351): for every cluster c the sum over its ordered point pairs of the Euclidean
// distance, T[c] = sum_{i, j in c} ||x_i - x_j|| (= np.sum(pairwise_distances(X[a == c]))) -- the ONLY thing the 10 reference sets of every K need from the
// point pairs (190 of the 209 pair passes of a K = 2..20 sweep).  Round 6: on the matrix cores.  dic_cluster_intra_sums (dic_pairdist.hip: the difference
// form on the VALU, every ordered pair, 11.6 ms per reference set, 2.2 s of the 16 s sweep) stays for the per-point sums and for widths above 256.
//
// d^2(i, j) = ||x_i||^2 + ||x_j||^2 - 2 x_i . x_j is ONE inner product of two augmented rows,
//     a_i = [ x_i (256) | n_i as three bf16 pieces | 1 1 1 | 0 ... ]         b_j = [ -2 x_j (256) | 1 1 1 | n_j as three bf16 pieces | 0 ... ]      (288 columns)
// with x taken RELATIVE TO ITS CLUSTER'S CENTROID (distances do not change, the cancellation in the norm form shrinks to the cluster's own spread) and every
// coordinate as two bf16 planes (x = hi + lo, good to 2^-17) multiplied as hi.hi + lo.hi + hi.lo with f32 accumulation -- the "split products" of DESIGN.md
// section 3; the norms n = sum x^2 are formed in f32 and enter exactly (three bf16 pieces carry all 24 bits, their partner columns are 1).  Measured against the
// f64 sum on the golden clusterings: <= 2e-7 relative on T[c] (the VALU kernel: <= 1e-7); the reference itself works in f32 (tests: rtol 1e-5).
//
// The machine is dic_lstm_dx_tile_x3's (dic_dxproj.hip): one persistent 8-wave workgroup per CU walks 256 x 256 point-pair tiles, both operands' 32-column slabs
// (hi | lo: 32 KB) stream through LDS-DMA rings (3 + 2 slots = all 160 KB), one raw barrier and one counted vmcnt wait per slab, the transposed product (lane =
// point i, registers = points j: the row sum over j is in-lane).  Only the tile pairs (I, J >= I) of one cluster are visited, an off-diagonal tile counts twice:
// sum_c n_c^2 / 2 of the N^2 pairs.  A tile's 128 accumulators per lane end as sqrt(max(d^2, 0)) summed in a fixed order in f32, tiles in f64 per lane; every wave
// writes ONE f64 partial per cluster it met (the tile list is sorted by cluster) and a second kernel adds the waves' partials in order: no atomics, deterministic.
#include "dic_common.h"

namespace dic {

typedef __bf16 qbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 qbf16x4 __attribute__((ext_vector_type(4)));
typedef float qf32x16 __attribute__((ext_vector_type(16)));
typedef float qf32x4 __attribute__((ext_vector_type(4)));

constexpr int QD = 256;                               // coordinates (narrower inputs are zero-padded)
constexpr int QLD = 288;                              // columns of an augmented row
constexpr int QT = 256;                               // points per tile edge
constexpr int QK = 32;                                // columns per slab
constexpr int Q_ROWB = QK * 2;                        // 64 B
constexpr int Q_PLANE = QT * Q_ROWB;                  // 16 KB: one plane of a slab
constexpr int Q_SLOT = 2 * Q_PLANE;                   // 32 KB: hi | lo
constexpr int Q_SLABS = QLD / QK;                     // 9
constexpr int Q_NI = 3, Q_NJ = 2;                     // ring depths of the two operands
constexpr int Q_LDS = (Q_NI + Q_NJ) * Q_SLOT;         // 163 840 B
constexpr int Q_WAVES = 8;
static_assert(Q_LDS <= 160 * 1024, "intra_x3: LDS budget");

// LDS-DMA as asm (not counted by the compiler: see dic_lstmgrad.hip): 64 lanes x 16 B -> 1 KiB at lds_dst
__device__ __forceinline__ void qdma16(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// The augmented rows.  One wave per point (sorted by cluster): lane l holds coordinates 4 l .. 4 l + 3.
struct IntraPrepArgs {
    const float* xs; int ldx;      // (n, d) points sorted by cluster, row stride ldx
    const int* seg;                // (K + 1) cluster boundaries
    const float* mu;               // (K, d) a point near each cluster (its centroid)
    __bf16* pa; __bf16* pb;        // (rows, 288) hi planes of a / b; the lo planes `plane` elements behind
    long plane;
    int n, d, K;
};

__device__ __forceinline__ void split2(float v, __bf16& hi, __bf16& lo) {
    hi = (__bf16)v;
    lo = (__bf16)(v - (float)hi);
}

__global__ __launch_bounds__(256) void intra_prep_kernel(IntraPrepArgs a) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= a.n) return;
    int lo = 0, hi = a.K;                               // seg[lo] <= row < seg[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a.seg[mid] <= row) lo = mid; else hi = mid;
    }
    const int col = 4 * lane;
    qf32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (col < a.d) {
        const qf32x4 x = *reinterpret_cast<const qf32x4*>(a.xs + (size_t)row * a.ldx + col);
        const qf32x4 m = *reinterpret_cast<const qf32x4*>(a.mu + (size_t)lo * a.d + col);
        v = x - m;
    }
    float nrm = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) nrm += __shfl_xor(nrm, o);
    qbf16x4 ah, al, bh, bl;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        __bf16 h, l;
        split2(v[e], h, l);
        ah[e] = h; al[e] = l;
        bh[e] = (__bf16)(-2.f * (float)h); bl[e] = (__bf16)(-2.f * (float)l);          // exact
    }
    const size_t at = (size_t)row * QLD + col;
    *reinterpret_cast<qbf16x4*>(a.pa + at) = ah;
    *reinterpret_cast<qbf16x4*>(a.pa + a.plane + at) = al;
    *reinterpret_cast<qbf16x4*>(a.pb + at) = bh;
    *reinterpret_cast<qbf16x4*>(a.pb + a.plane + at) = bl;
    if (lane < 8) {                                     // columns 256 + 4 lane ..: [n n n 1 | 1 1 0 0 | 0 ..] and [1 1 1 n | n n 0 0 | 0 ..]
        const __bf16 n0 = (__bf16)nrm;
        const float r1 = nrm - (float)n0;
        const __bf16 n1 = (__bf16)r1;
        const __bf16 n2 = (__bf16)(r1 - (float)n1);
        const __bf16 one = (__bf16)1.f, z = (__bf16)0.f;
        qbf16x4 ea = {z, z, z, z}, eb = {z, z, z, z};
        if (lane == 0) { ea = qbf16x4{n0, n1, n2, one}; eb = qbf16x4{one, one, one, n0}; }
        if (lane == 1) { ea = qbf16x4{one, one, z, z}; eb = qbf16x4{n1, n2, z, z}; }
        const size_t et = (size_t)row * QLD + QD + 4 * lane;
        const qbf16x4 zz = {z, z, z, z};
        *reinterpret_cast<qbf16x4*>(a.pa + et) = ea;
        *reinterpret_cast<qbf16x4*>(a.pa + a.plane + et) = zz;
        *reinterpret_cast<qbf16x4*>(a.pb + et) = eb;
        *reinterpret_cast<qbf16x4*>(a.pb + a.plane + et) = zz;
    }
}

// A tile-list entry through the scalar cache (the compiler would use a vector load, counted in vmcnt, and drain the DMA ring for it)
typedef int qi32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ qi32x4 sload_tile(const int4* p) {
    qi32x4 r;
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(p) : "memory");
    return r;
}

// ------------------------------------------------------------------------------------------------------------------------------------------
struct IntraX3Args {
    const __bf16* pa; const __bf16* pb; long plane;
    const int4* tiles; int ntiles;      // totals: (first row of I, first row of J, end row of the cluster, cluster), sorted by cluster
    double* partial; int K;             //         (workgroups x 8 waves, K), zeroed by the caller
    float* rows_out; int n_rows;        // row sums (ROWS): tiles = (first row of I, first row of J, end row of J's cluster, output slot), rows_out (slots, 2, 256)
};

// ROWS = false: the per-cluster totals (above).  ROWS = true: the ROW SUMS of the same distance tiles over ALL point pairs, S[i][c] = sum_{j in c} ||x_i - x_j||
// (the silhouette's mean distances, internal_eval.py:112-123): a tile is 256 rows of ANY cluster against 256 rows of one cluster c; a workgroup walks a CONTIGUOUS
// range of the tile list (sorted by row block, then cluster), and every maximal run of one (row block, cluster) inside a range is an output "slot": the lane's two
// row sums (its two 32-row blocks, 64 columns each per tile: a tile's 64 terms first, then the running sum -- f32 throughout) are written once per slot, the two
// column halves (wn) side by side; rows_reduce_kernel adds a (row block, cluster)'s slots in order.  No atomics, no weights, i == j masked wherever the ranges meet.
template <bool ROWS>
__global__ __launch_bounds__(512, 1) void intra_x3_kernel(IntraX3Args a) {
    extern __shared__ __align__(16) unsigned char qsm[];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, l31 = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 3, wn = w >> 2;
    const int nch = gridDim.x;
    const int per = (a.ntiles + nch - 1) / nch;          // (ROWS: contiguous ranges)
    const int my_tiles = ROWS ? max(0, min(per, a.ntiles - (int)blockIdx.x * per))
                              : ((int)blockIdx.x < a.ntiles ? (a.ntiles - 1 - (int)blockIdx.x) / nch + 1 : 0);
    const int S = my_tiles * Q_SLABS;
    if (S == 0) return;
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)qsm);
    const unsigned ldsI = lds0, ldsJ = lds0 + Q_NI * Q_SLOT;
    // DMA: a plane of a slab = 16 instructions of 16 rows; wave w issues c = w, w + 8, w + 16, w + 24 of the 32 (plane c >> 4, row group c & 15); lane L -> row
    // 16 rg + (L >> 2), physical 16-B piece L & 3 = logical piece (L & 3) ^ ((row >> 2) & 3), and (row >> 2) & 3 = (L >> 4) & 3 for every row group
    const unsigned v_dma = (unsigned)(lane >> 2) * (QLD * 2) + (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) * 16);
    // the entries of the tile being multiplied and of the next one (whose first slabs are requested two iterations ahead) live in scalar registers
    auto tile = [&](int i) {
        const int t = min(i, my_tiles - 1);
        return sload_tile(a.tiles + __builtin_amdgcn_readfirstlane(ROWS ? (int)blockIdx.x * per + t : (int)blockIdx.x + t * nch));
    };
    qi32x4 e_cur = tile(0), e_nxt = tile(1), e_prev = e_cur;
    int cur_tile = 0;
    auto issue = [&](const __bf16* mat, int row0, int ks, unsigned dst) {
        const __bf16* src = mat + (size_t)row0 * QLD + ks * QK;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = w + 8 * j, pl = c >> 4, rg = c & 15;
            qdma16(src + (size_t)pl * a.plane + (size_t)(16 * rg) * QLD, v_dma, dst + pl * Q_PLANE + rg * 1024);
        }
    };
    auto issue_i = [&](int s) { issue(a.pa, s / Q_SLABS == cur_tile ? e_cur[0] : e_nxt[0], s % Q_SLABS, ldsI + (s % Q_NI) * Q_SLOT); };
    auto issue_j = [&](int s) { issue(a.pb, s / Q_SLABS == cur_tile ? e_cur[1] : e_nxt[1], s % Q_SLABS, ldsJ + (s % Q_NJ) * Q_SLOT); };
    // fragment reads: row (32 block + l31), logical piece 2 kk + hh -> physical piece ^ ((l31 >> 2) & 3)
    const int sw = (l31 >> 2) & 3;
    int poff[QK / 16];
#pragma unroll
    for (int kk = 0; kk < QK / 16; ++kk) poff[kk] = ((2 * kk + hh) ^ sw) * 16;
    const int j_row = (128 * wn + l31) * Q_ROWB;          // b rows (points j) of this wave's 4 column blocks (+ 32 nb rows): the MFMA's A operand
    const int i_row = (64 * wm + l31) * Q_ROWB;           // a rows (points i) of this wave's 2 row blocks (+ 32 mb rows): the MFMA's B operand

#pragma unroll
    for (int it = 1 - Q_NI; it < 0; ++it) {
        if (it + Q_NJ - 1 >= 0 && it + Q_NJ - 1 < S) issue_j(it + Q_NJ - 1);
        if (it + Q_NI - 1 < S) issue_i(it + Q_NI - 1);
    }
    qf32x16 acc[4][2];
    double dsum = 0.0;
    int cur_c = -1;
    auto flush = [&]() {
        if (cur_c < 0) return;
        double v = dsum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) v += __shfl_xor(v, o);
        if (lane == 0) a.partial[((size_t)blockIdx.x * Q_WAVES + w) * a.K + cur_c] = v;
        dsum = 0.0;
    };
    // a finished tile: D^T layout, lane (i = 64 wm + 32 mb + l31, hh), register k of block nb -> j = 128 wn + 32 nb + (k & 3) + 8 (k >> 2) + 4 hh
    float rs[2] = {0.f, 0.f};
    auto flush_rows = [&]() {
        if (cur_c < 0) return;
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const float v = rs[mb] + __shfl_xor(rs[mb], 32);
            if (hh == 0) a.rows_out[((size_t)cur_c * 2 + wn) * QT + 64 * wm + 32 * mb + l31] = v;
            rs[mb] = 0.f;
        }
    };
    auto finish_rows = [&](qi32x4 e) {
        const int I0 = e[0], J0 = e[1], end = e[2];
        if (e[3] != cur_c) {
            flush_rows();
            cur_c = e[3];
        }
        const bool apart = I0 + QT <= J0 || J0 + QT <= I0;
        if (apart && I0 + QT <= a.n_rows && J0 + QT <= end) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                float t = 0.f;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int k = 0; k < 16; ++k) t += __builtin_amdgcn_sqrtf(fmaxf(acc[nb][mb][k], 0.f));
                rs[mb] += t;
            }
        } else {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int gi = I0 + 64 * wm + 32 * mb + l31;
                const bool iv = gi < a.n_rows;
                float t = 0.f;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const int jb = J0 + 128 * wn + 32 * nb + 4 * hh;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int gj = jb + (k & 3) + 8 * (k >> 2);
                        const bool ok = iv && gj < end && gi != gj;
                        t += ok ? __builtin_amdgcn_sqrtf(fmaxf(acc[nb][mb][k], 0.f)) : 0.f;
                    }
                }
                rs[mb] += t;
            }
        }
    };
    auto finish_totals = [&](qi32x4 e) {
        const int I0 = e[0], J0 = e[1], end = e[2];
        if (e[3] != cur_c) {
            flush();
            cur_c = e[3];
        }
        const bool diag = I0 == J0;
        float t = 0.f;
        if (!diag && I0 + QT <= end && J0 + QT <= end) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int k = 0; k < 16; ++k) t += __builtin_amdgcn_sqrtf(fmaxf(acc[nb][mb][k], 0.f));
        } else {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int gi = I0 + 64 * wm + 32 * mb + l31;          // (global rows: a tile-relative i == j test is loop-invariant, and the compiler keeps all 128 of its
                const bool iv = gi < end;                               //  lane masks in scalar registers across the whole slab loop)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const int jb = J0 + 128 * wn + 32 * nb + 4 * hh;
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int gj = jb + (k & 3) + 8 * (k >> 2);
                        const bool ok = iv && gj < end && gi != gj;
                        t += ok ? __builtin_amdgcn_sqrtf(fmaxf(acc[nb][mb][k], 0.f)) : 0.f;
                    }
                }
            }
        }
        dsum += (double)t * (diag ? 1.0 : 2.0);
    };
    for (int s = 0; s < S; ++s) {
        const int ks = s % Q_SLABS;
        if (ks == 0 && s > 0) {          // (before this iteration's requests: they may belong to the tile after this one)
            e_prev = e_cur;
            e_cur = e_nxt;
            ++cur_tile;
            e_nxt = tile(cur_tile + 1);
        }
        // counted wait: at the top of iteration s the DMA instructions still allowed in flight are those of I-slab s + 1 (4 per wave), issued last in iteration s - 1
        if (S - 1 - s >= Q_NI - 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();          // everybody's pieces of slab s are in; everybody is done reading slab s - 1 (its two slots are free)
        const unsigned char* A = qsm + Q_NI * Q_SLOT + (s % Q_NJ) * Q_SLOT + j_row;
        const unsigned char* Bm = qsm + (s % Q_NI) * Q_SLOT + i_row;
        qbf16x8 ah[2][4], bh[2][2], al[4], bl[2];
        auto load_hi = [&](int kk, int set) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) bh[set][mb] = *reinterpret_cast<const qbf16x8*>(Bm + mb * 32 * Q_ROWB + poff[kk]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) ah[set][nb] = *reinterpret_cast<const qbf16x8*>(A + nb * 32 * Q_ROWB + poff[kk]);
        };
        auto load_lo = [&](int kk) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) bl[mb] = *reinterpret_cast<const qbf16x8*>(Bm + Q_PLANE + mb * 32 * Q_ROWB + poff[kk]);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) al[nb] = *reinterpret_cast<const qbf16x8*>(A + Q_PLANE + nb * 32 * Q_ROWB + poff[kk]);
        };
        load_hi(0, 0);
        __builtin_amdgcn_sched_barrier(0);          // (the first fragment reads go out BEFORE the DMA instructions)
        if (s + Q_NJ - 1 < S) issue_j(s + Q_NJ - 1);
        if (s + Q_NI - 1 < S) issue_i(s + Q_NI - 1);
        __builtin_amdgcn_sched_barrier(0);
        if (ks == 0) {
            if (s > 0) {
                if constexpr (ROWS) finish_rows(e_prev);
                else finish_totals(e_prev);
            }
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[nb][mb][k] = 0.f;
        }
        const bool coords = ks < QD / QK;          // the augmentation slab has empty lo planes
#pragma unroll
        for (int kk = 0; kk < QK / 16; ++kk) {
            if (coords) load_lo(kk);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk & 1][nb], bh[kk & 1][mb], acc[nb][mb], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 1 < QK / 16) load_hi(kk + 1, (kk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (coords) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[nb], bh[kk & 1][mb], acc[nb][mb], 0, 0, 0);
                        acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[kk & 1][nb], bl[mb], acc[nb][mb], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (ROWS) {
        finish_rows(e_cur);
        flush_rows();
    } else {
        finish_totals(e_cur);
        flush();
    }
}

// S[i][c] = the sum of the slots of (row block of i, c), both column halves, in slot order.  grid (row blocks, K).
__global__ __launch_bounds__(QT) void rows_reduce_kernel(const float* rows_out, const int* group_start, int n_rows, int K, float* S) {
    const int g = blockIdx.x * K + blockIdx.y, i = blockIdx.x * QT + threadIdx.x;
    if (i >= n_rows) return;
    float s = 0.f;
    for (int slot = group_start[g]; slot < group_start[g + 1]; ++slot)
        s += rows_out[((size_t)slot * 2) * QT + threadIdx.x] + rows_out[((size_t)slot * 2 + 1) * QT + threadIdx.x];
    S[(size_t)i * K + blockIdx.y] = s;
}

// one wave per cluster: lane l adds rows l, l + 64, ... in order, then the 64 lane sums are added by a fixed butterfly
__global__ __launch_bounds__(64) void intra_finalize_kernel(const double* partial, int rows, int K, double* out) {
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int r = lane; r < rows; r += 64) s += partial[(size_t)r * K + c];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[c] = s;
}

static int intra_chunks(int ntiles) { return max(1, min(ntiles, kNumCU)); }
static size_t intra_plane_elems(int64_t n) { return (size_t)(n + QT) * QLD; }          // a tile may start up to 255 rows before the end: QT rows of padding
static size_t intra_partial_bytes(int K) { return (size_t)kNumCU * Q_WAVES * K * sizeof(double); }

}  // namespace dic

using namespace dic;

extern "C" {

size_t dic_cluster_pair_rowsums_workspace(int64_t N, int n_slots) {
    if (N <= 0 || n_slots < 0) return 0;
    return 4 * intra_plane_elems(N) * sizeof(__bf16) + (size_t)max(1, n_slots) * 2 * QT * sizeof(float);
}

static int intra_reserve_lds(const char* who) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)intra_x3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)intra_x3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, Q_LDS);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "%s: cannot reserve %d B of LDS: %s", who, Q_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    return DIC_OK;
}

// planes of the augmented rows (+ zeroed padding rows) at the head of the workspace; returns the first byte behind them
static int intra_planes(const char* who, const float* X, long ldx, const int32_t* seg, const float* centres, int64_t N, int D, int K, void* workspace, hipStream_t st,
                        __bf16** pa_out, __bf16** pb_out) {
    const size_t plane = intra_plane_elems(N);
    __bf16* pa = (__bf16*)workspace;
    __bf16* pb = pa + 2 * plane;
    // the padding rows behind the last point are read by the last tiles (and masked): zero, not stale, so that no NaN pattern ever enters an accumulator
    hipError_t e = hipMemsetAsync(pa + (size_t)N * QLD, 0, (size_t)QT * QLD * sizeof(__bf16), st);
    if (e == hipSuccess) e = hipMemsetAsync(pa + plane + (size_t)N * QLD, 0, (size_t)QT * QLD * sizeof(__bf16), st);
    if (e == hipSuccess) e = hipMemsetAsync(pb + (size_t)N * QLD, 0, (size_t)QT * QLD * sizeof(__bf16), st);
    if (e == hipSuccess) e = hipMemsetAsync(pb + plane + (size_t)N * QLD, 0, (size_t)QT * QLD * sizeof(__bf16), st);
    DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "%s: memset: %s", who, hipGetErrorString(e));
    IntraPrepArgs p{X, (int)ldx, seg, centres, pa, pb, (long)plane, (int)N, D, K};
    hipLaunchKernelGGL(intra_prep_kernel, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, st, p);
    *pa_out = pa;
    *pb_out = pb;
    return DIC_OK;
}

int dic_cluster_pair_rowsums(const float* X, long ldx, const float* centre, int64_t N, int D, int K, const int32_t* tiles, int ntiles, const int32_t* group_start,
                             int n_slots, float* S, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(N > 0 && D > 0 && K > 0 && ldx >= D, DIC_ERR_INVALID_ARG, "cluster_pair_rowsums: N=%lld D=%d K=%d ldx=%ld", (long long)N, D, K, ldx);
    DIC_REQUIRE(D <= QD && D % 4 == 0 && ldx % 4 == 0, DIC_ERR_UNSUPPORTED, "cluster_pair_rowsums: D=%d (row stride %ld): at most %d, multiples of 4", D, ldx, QD);
    DIC_REQUIRE(N < (1 << 30) && K <= 65535, DIC_ERR_UNSUPPORTED, "cluster_pair_rowsums: N=%lld K=%d", (long long)N, K);
    DIC_REQUIRE(X && centre && tiles && group_start && S && workspace && ntiles > 0 && n_slots > 0, DIC_ERR_INVALID_ARG, "cluster_pair_rowsums: NULL pointer / empty list");
    DIC_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)centre & 15) == 0 && ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)tiles & 15) == 0, DIC_ERR_UNSUPPORTED,
                "cluster_pair_rowsums: operands must be 16-B aligned");
    DIC_REQUIRE(workspace_bytes >= dic_cluster_pair_rowsums_workspace(N, n_slots), DIC_ERR_WORKSPACE, "cluster_pair_rowsums: workspace %zu < %zu", workspace_bytes,
                dic_cluster_pair_rowsums_workspace(N, n_slots));
    int rc = intra_reserve_lds("cluster_pair_rowsums");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    __bf16 *pa, *pb;
    // ONE centre for every point (pairs cross clusters): "cluster" 0 = all rows; the prep kernel only needs seg[0] = 0 <= row, which K = 1 never reads
    rc = intra_planes("cluster_pair_rowsums", X, ldx, (const int32_t*)nullptr, centre, N, D, 1, workspace, st, &pa, &pb);
    if (rc) return rc;
    const size_t plane = intra_plane_elems(N);
    float* rows_out = (float*)(pb + 2 * plane);
    IntraX3Args a{pa, pb, (long)plane, (const int4*)tiles, ntiles, nullptr, K, rows_out, (int)N};
    hipLaunchKernelGGL(intra_x3_kernel<true>, dim3(intra_chunks(ntiles)), dim3(512), Q_LDS, st, a);
    hipLaunchKernelGGL(rows_reduce_kernel, dim3((unsigned)((N + QT - 1) / QT), K), dim3(QT), 0, st, (const float*)rows_out, group_start, (int)N, K, S);
    return check_launch("cluster_pair_rowsums");
}

size_t dic_cluster_intra_totals_workspace(int64_t N, int K) {
    if (N <= 0 || K <= 0) return 0;
    return 4 * intra_plane_elems(N) * sizeof(__bf16) + intra_partial_bytes(K);
}

int dic_cluster_intra_totals(const float* X, long ldx, const int32_t* seg, const float* centres, int64_t N, int D, int K, const int32_t* tiles, int ntiles,
                             double* totals, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(N > 0 && D > 0 && K > 0 && ldx >= D, DIC_ERR_INVALID_ARG, "cluster_intra_totals: N=%lld D=%d K=%d ldx=%ld", (long long)N, D, K, ldx);
    DIC_REQUIRE(D <= QD && D % 4 == 0 && ldx % 4 == 0, DIC_ERR_UNSUPPORTED, "cluster_intra_totals: D=%d (row stride %ld): at most %d, multiples of 4", D, ldx, QD);
    DIC_REQUIRE(N < (1 << 30), DIC_ERR_UNSUPPORTED, "cluster_intra_totals: N=%lld", (long long)N);
    DIC_REQUIRE(X && seg && centres && totals && workspace && (tiles || ntiles == 0) && ntiles >= 0, DIC_ERR_INVALID_ARG, "cluster_intra_totals: NULL pointer");
    DIC_REQUIRE(((uintptr_t)X & 15) == 0 && ((uintptr_t)centres & 15) == 0 && ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)tiles & 15) == 0, DIC_ERR_UNSUPPORTED,
                "cluster_intra_totals: operands must be 16-B aligned");
    DIC_REQUIRE(workspace_bytes >= dic_cluster_intra_totals_workspace(N, K), DIC_ERR_WORKSPACE, "cluster_intra_totals: workspace %zu < %zu", workspace_bytes,
                dic_cluster_intra_totals_workspace(N, K));
    int rc = intra_reserve_lds("cluster_intra_totals");
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    __bf16 *pa, *pb;
    rc = intra_planes("cluster_intra_totals", X, ldx, seg, centres, N, D, K, workspace, st, &pa, &pb);
    if (rc) return rc;
    const size_t plane = intra_plane_elems(N);
    double* partial = (double*)(pb + 2 * plane);
    hipError_t e = hipMemsetAsync(partial, 0, intra_partial_bytes(K), st);
    DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "cluster_intra_totals: memset: %s", hipGetErrorString(e));
    if (ntiles > 0) {
        IntraX3Args a{pa, pb, (long)plane, (const int4*)tiles, ntiles, partial, K, nullptr, (int)N};
        hipLaunchKernelGGL(intra_x3_kernel<false>, dim3(intra_chunks(ntiles)), dim3(512), Q_LDS, st, a);
    }
    hipLaunchKernelGGL(intra_finalize_kernel, dim3(K), dim3(64), 0, st, partial, kNumCU * Q_WAVES, K, totals);
    return check_launch("cluster_intra_totals");
}

}  // extern "C"
