// k4 at K >= 9: the Lloyd E-step + partial M-step (sklearn _k_means_lloyd.pyx:167-218: pairwise = ||c||^2 - 2 X C^T by sgemm,
// first-minimum label, per-cluster sums / counts) on the matrix cores.
//
// Why: the one-wave-per-row kernel of dic_latent.hip reduces K per-lane partial distances across the wave with a shuffle
// butterfly; at K = 16 / 32 that is a long dependent chain with 172 registers per lane (two waves per SIMD): 0.56 / 1.7 ms per
// iteration of 10 restarts on 75 000 x 256 latents, 55 % of the p2 K-sweep's 68 s (rocprofv3, round 2).  Both halves of the
// iteration are GEMMs:
//     scores  D'[centroid][row] = C . X^T                (v_mfma_f32_32x32x2_f32: exact f32 products, f32 accumulation, as sgemm)
//     sums    S[centroid][d]    = onehot(label)^T . X     (the A operand is 0 / 1: the MFMA adds exactly the assigned rows)
// One workgroup serves the 32 centroid columns of 32/KP restarts at once (the restarts share every X tile), keeps those
// centroids in LDS, and gives each of its 4 waves its own 32-row tiles.  The score product is issued transposed so that a LANE
// owns a row and the centroids sit in its registers: the argmin is 15 in-lane compares + one cross-half exchange, no butterfly.
// Outputs are the per-workgroup partials the existing reduce / update kernels consume (fixed-order f64 second stage).
#include "dic_common.h"

namespace dic {

constexpr int MCP = 256 + 4;         // LDS pitch of a centroid row (floats): 65 16-B slots -> the 16 rows of a ds_read_b128 lane group fall on 16 different slots
constexpr int XCH = 64;              // features per staged X chunk
constexpr int XP = XCH + 4;          // LDS pitch of a staged X row (floats): 17 16-B slots, the same property
// Both LDS images keep every aligned group of 8 features in the order [0 1 4 5 | 2 3 6 7]: one 16-B read at 8 m + 4 hh then hands the
// lower half-wave (hh = 0) features {0 1 4 5} and the upper one {2 3 6 7}, i.e. the MFMA k pairs (0|2) (1|3) (4|6) (5|7) -- the order
// in which round 2's kernel (8-B reads at 4 m + 2 hh) accumulated, so the scores are bit-identical to it.
__device__ __forceinline__ int km_perm8(int d4) { return (d4 >> 3) * 8 + ((d4 >> 2) & 1) * 2; }      // position of the LOW pair of the float4 at feature d4; the high pair sits 4 further

typedef float kf32x16 __attribute__((ext_vector_type(16)));
typedef float kf32x2 __attribute__((ext_vector_type(2)));
typedef float kf32x4 __attribute__((ext_vector_type(4)));

struct KmMfmaArgs {
    const float* X; const float* xnorm; int N, D, K, n_runs, nblk;
    const float* centers;      // (n_runs,K,D)
    int32_t* labels;           // (n_runs,N)
    const float* status;       // (n_runs,8): status[0] != 0 -> converged, skipped
    float* mind;               // (n_runs,N)
    float* psum;               // (n_runs,nblk,K*D)
    int* pcnt;                 // (n_runs,nblk,K+1): counts | #changed
};

template <int KP, int NMB>
__global__ __launch_bounds__(256, NMB == 1 ? 2 : 1) void kmeans_assign_mfma_kernel(KmMfmaArgs a) {
    constexpr int MCOLS = 32 * NMB;
    constexpr int G = MCOLS / KP;                       // restarts per workgroup
    static_assert(G >= 1, "a restart's centroids must fit the workgroup's columns");
    extern __shared__ __align__(16) unsigned char ksm[];
    float* cent = reinterpret_cast<float*>(ksm);                         // [64][MCP]; reused for the cross-wave sum at the end
    float* cnorm = cent + MCOLS * MCP;                                   // [64]
    int* cnt = reinterpret_cast<int*>(cnorm + MCOLS);                    // [64] members per centroid column
    int* chg = cnt + MCOLS;                                              // [G] labels changed per restart
    unsigned char* lab8 = reinterpret_cast<unsigned char*>(chg + 4);     // [4 waves][G][32]  (G <= 4: KP >= 8)
    float* xbuf = reinterpret_cast<float*>(lab8 + 4 * G * 32);           // [4 waves][32 rows][XP]: the wave's X tile, one 64-feature chunk at a time
    static_assert(G <= 4, "chg[] has four slots");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 31, hh = lane >> 5;
    const int N = a.N, D = a.D, K = a.K;
    const int run0 = blockIdx.y * G;
    bool any = false;
#pragma unroll
    for (int g = 0; g < G; ++g) any = any || (run0 + g < a.n_runs && a.status[(run0 + g) * DIC_KM_STATUS_WORDS] == 0.f);
    if (!any) return;                                    // uniform over the workgroup

    // ---- centroids of the G restarts -> LDS (zero rows for k >= K / finished restarts), their squared norms
    for (int i = tid; i < MCOLS * 64; i += 256) {        // float4 pieces
        const int col = i >> 6, d4 = (i & 63) * 4;
        const int g = col / KP, k = col - g * KP, run = run0 + g;
        kf32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (run < a.n_runs && k < K && d4 < D) v = *reinterpret_cast<const kf32x4*>(a.centers + ((size_t)run * K + k) * D + d4);
        float* dst = cent + col * MCP + km_perm8(d4);
        *reinterpret_cast<kf32x2*>(dst) = kf32x2{v[0], v[1]};
        *reinterpret_cast<kf32x2*>(dst + 4) = kf32x2{v[2], v[3]};
    }
    if (tid < MCOLS) cnt[tid] = 0;
    if (tid < G) chg[tid] = 0;
    __syncthreads();
    for (int col = w; col < MCOLS; col += 4) {           // row_norms (_k_means_lloyd.pyx:99), one wave per centroid
        const float* cp = cent + col * MCP + km_perm8(4 * lane);     // features 4 lane .. 4 lane + 3, in that order
        const kf32x4 c = kf32x4{cp[0], cp[1], cp[4], cp[5]};
        const float s = wave_sum(fmaf(c[3], c[3], fmaf(c[2], c[2], fmaf(c[1], c[1], c[0] * c[0]))));
        const int g = col / KP, k = col - g * KP;
        if (lane == 0) cnorm[col] = (k < K) ? s : INFINITY;          // padding columns can never win the argmin
    }
    __syncthreads();
    // the centroid of accumulator register `reg` of M-block mb in this lane: m = 32 mb + (reg & 3) + 8 (reg >> 2) + 4 hh; its squared norm is read
    // from LDS where the argmin needs it (two addresses per wave: broadcast reads) -- held in 16 registers it made the kernel spill
    kf32x16 S[NMB][8];                                   // sums[centroid column block][d block]: column jj of d-block nb is d = 128 (nb >> 2) + 4 jj + (nb & 3)
#pragma unroll
    for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb)
#pragma unroll
            for (int k = 0; k < 16; ++k) S[mb][nb][k] = 0.f;

    const int ntiles = (N + 31) / 32;
    // staging of X chunks (scores phase): lane -> (16-B piece, row of 4) such that a 16-lane group is 2 rows x 128 B
    float* xb = xbuf + w * 32 * XP;
    const int lq = (lane & 7) | (((lane >> 4) & 1) << 3), lrow = ((lane >> 3) & 1) | (((lane >> 5) & 1) << 1);
    float* xdst = xb + lrow * XP + km_perm8(4 * lq);
    kf32x4 v[8];
    auto load_chunk = [&](int r0_, int c) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v[i] = kf32x4{0.f, 0.f, 0.f, 0.f};
            if (XCH * c + 4 * lq < D) v[i] = *reinterpret_cast<const kf32x4*>(a.X + ((size_t)min(r0_ + 4 * i + lrow, N - 1) * D + (unsigned)(XCH * c + 4 * lq)));
        }
    };
    for (int tile = blockIdx.x * 4 + w; tile < ntiles; tile += a.nblk * 4) {
        const int r0 = tile * 32, row = r0 + j;
        const bool valid = row < N;
        // ---- scores: D'[centroid][row] = sum_k C[centroid][k] X[row][k].  The tile's rows come in as whole 128-B lines (a load instruction
        // = 4 rows x 256 B), go through the wave's own LDS chunk and come back as the B operand (lane = row): round 2's per-lane 8-B loads
        // straight from the row (32 rows x 16 B per instruction, 64 instructions per tile) kept the texture path busier than the matrix
        // cores (rocprofv3: SQ_VALU_MFMA_BUSY_CYCLES 35 % of the kernel's time).  The next chunk's loads fly during this chunk's MFMAs.
        kf32x16 dacc[NMB];
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
            for (int k = 0; k < 16; ++k) dacc[mb][k] = 0.f;
        {
            load_chunk(r0, 0);
#pragma unroll
            for (int c = 0; c < 256 / XCH; ++c) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {        // (this wave's previous reads of the chunk were issued before: the LDS queue keeps the order)
                    *reinterpret_cast<kf32x2*>(xdst + 4 * i * XP) = kf32x2{v[i][0], v[i][1]};
                    *reinterpret_cast<kf32x2*>(xdst + 4 * i * XP + 4) = kf32x2{v[i][2], v[i][3]};
                }
                __builtin_amdgcn_wave_barrier();
                if (c + 1 < 256 / XCH) load_chunk(r0, c + 1);
#pragma unroll
                for (int mm = 0; mm < XCH / 8; ++mm) {
                    const kf32x4 x4 = *reinterpret_cast<const kf32x4*>(xb + j * XP + 8 * mm + 4 * hh);
#pragma unroll
                    for (int mb = 0; mb < NMB; ++mb) {
                        const kf32x4 c4 = *reinterpret_cast<const kf32x4*>(cent + (32 * mb + j) * MCP + XCH * c + 8 * mm + 4 * hh);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dacc[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(c4[e], x4[e], dacc[mb], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        // the sums phase's B operand (rows 2n + hh as they lie: whole 512-B half rows per half-wave) in a ring of four row pairs that
        // reuses the staging registers: the first four pairs are requested here, before the argmin, pair n + 4 once pair n is consumed
        auto load_rows = [&](int n) {
            const float* xrow = a.X + (size_t)min(r0 + 2 * n + hh, N - 1) * D;
            v[n & 3] = kf32x4{0.f, 0.f, 0.f, 0.f};
            v[4 + (n & 3)] = kf32x4{0.f, 0.f, 0.f, 0.f};
            if (4 * j < D) v[n & 3] = *reinterpret_cast<const kf32x4*>(xrow + 4 * j);
            if (128 + 4 * j < D) v[4 + (n & 3)] = *reinterpret_cast<const kf32x4*>(xrow + 128 + 4 * j);
        };
#pragma unroll
        for (int n = 0; n < 4; ++n) load_rows(n);
        // ---- argmin per restart: in-lane over this half's centroids (ascending index, strict <: first minimum wins), then across halves
        const float xn = valid && a.xnorm ? a.xnorm[row] : 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int run = run0 + g;
            float best = INFINITY;
            int bi = 1 << 20;
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int mcol = 32 * mb + (reg & 3) + 8 * (reg >> 2);      // + 4 hh
                    if (mcol / KP != g) continue;                               // compile-time: (mcol + 4 hh) / KP == mcol / KP for KP >= 8
                    const float s = fmaf(-2.0f, dacc[mb][reg], cnorm[32 * mb + (reg & 3) + 8 * (reg >> 2) + 4 * hh]);
                    const int kidx = mcol + 4 * hh - g * KP;
                    if (s < best) { best = s; bi = kidx; }
                }
            const float ob = __shfl_xor(best, 32);
            const int oi = __shfl_xor(bi, 32);
            if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            const bool active = run < a.n_runs && a.status[run * DIC_KM_STATUS_WORDS] == 0.f;
            if (hh == 0) {
                unsigned char code = 255;                                       // rows past the end / finished restarts add to no sum
                if (valid && active) {
                    int32_t* lp = a.labels + (size_t)run * N + row;
                    if (*lp != bi) atomicAdd(&chg[g], 1);
                    *lp = bi;
                    a.mind[(size_t)run * N + row] = fmaxf(0.f, xn + best);
                    atomicAdd(&cnt[g * KP + bi], 1);
                    code = (unsigned char)bi;
                }
                lab8[(w * G + g) * 32 + j] = code;
            }
        }
        // (lab8 of this wave is written and read by this wave only: wave-local ordering through the LDS queue)
        // ---- sums: S[col][d] += sum_rows onehot[col][row] X[row][d]; MFMA n multiplies rows 2n + hh
        const unsigned* lwp[NMB];                            // packed labels of the 32 rows for the restart of this lane's centroid column(s): read from LDS
#pragma unroll                                              // per row pair below (broadcast reads) -- eight registers of them made the K <= 32 kernel spill
        for (int mb = 0; mb < NMB; ++mb) lwp[mb] = reinterpret_cast<const unsigned*>(lab8 + (w * G + (32 * mb + j) / KP) * 32);
        int kc[NMB];
#pragma unroll
        for (int mb = 0; mb < NMB; ++mb) kc[mb] = (32 * mb + j) % KP;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const kf32x4 xlo = v[n & 3], xhi = v[4 + (n & 3)];
            // byte rr of the packed labels: dword rr >> 2, byte rr & 3 (rr = 2n + hh)
            float onehot[NMB];
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb) {
                const unsigned lwv = lwp[mb][n >> 1];
                const unsigned b = (hh ? (lwv >> (16 * (n & 1) + 8)) : (lwv >> (16 * (n & 1)))) & 255u;
                onehot[mb] = b == (unsigned)kc[mb] ? 1.0f : 0.0f;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mb = 0; mb < NMB; ++mb) {
                    S[mb][e] = __builtin_amdgcn_mfma_f32_32x32x2f32(onehot[mb], xlo[e], S[mb][e], 0, 0, 0);
                    S[mb][4 + e] = __builtin_amdgcn_mfma_f32_32x32x2f32(onehot[mb], xhi[e], S[mb][4 + e], 0, 0, 0);
                }
            if (n + 4 < 16) load_rows(n + 4);
        }
    }
    // ---- the 4 waves' sums -> one partial per workgroup (through the centroid area), counts, #changed
    __syncthreads();
    float* red = cent;                                   // [64 cols][256 d]
    for (int ww = 0; ww < 4; ++ww) {
        if (w == ww) {
#pragma unroll
            for (int mb = 0; mb < NMB; ++mb)
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {          // d = 128 half + 4 j + (nb & 3): the lane's four d-blocks of a half are one 16-B piece
                        const int col = 32 * mb + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
                        kf32x4* p = reinterpret_cast<kf32x4*>(red + col * 256 + 128 * half + 4 * j);
                        const kf32x4 sv = {S[mb][4 * half][reg], S[mb][4 * half + 1][reg], S[mb][4 * half + 2][reg], S[mb][4 * half + 3][reg]};
                        *p = (ww == 0) ? sv : *p + sv;
                    }
        }
        __syncthreads();
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int run = run0 + g;
        if (run >= a.n_runs || a.status[run * DIC_KM_STATUS_WORDS] != 0.f) continue;
        const size_t pb = (size_t)run * a.nblk + blockIdx.x;
        for (int i = tid; i < K * D; i += 256) {
            const int k = i / D, d = i - k * D;
            a.psum[pb * K * D + i] = red[(g * KP + k) * 256 + d];
        }
        if (tid < K) a.pcnt[pb * (K + 1) + tid] = cnt[g * KP + tid];
        if (tid == 0) a.pcnt[pb * (K + 1) + K] = chg[g];
    }
}

// Row-chunk workgroups per restart group: one workgroup per CU over ALL groups (69 KB of LDS, ~500 registers: one resident workgroup
// per CU), so that each pays its centroid staging and its cross-wave reduction once and walks several tiles per wave
// (0.34 -> 0.2 ms per iteration of 10 restarts at K = 16 against three rounds of 256 workgroups).
// One MFMA row block (32 centroid columns) per workgroup: 32 / KP restarts share its X tiles.  (Two row blocks = 64 columns per
// workgroup was the first layout: it needs all 512 registers -- one workgroup per CU, spilling -- and pads 10 restarts of K = 16 to
// 192 columns where 32-column groups need 160; with half the accumulators two workgroups fit a CU: 284 -> 254 us per iteration.)
// A/B, round 6 (scripts/kmeans_bench.py with DIC_AB_LIB): placing the restart groups of one row block on ONE XCD (workgroup id -> (row block, group) such that
// ids b and b + 8 walk the same X tiles; 8 x floor(64 / groups) row blocks so that no XCD gets a 65th workgroup) leaves the iteration where it was -- K = 16 x 10
// restarts 160.9 against 158.6 us, K = 8 109.3 / 109.7, K = 20 265.6 / 266.1: the kernel does not wait for X out of the Infinity Cache (4.8 TB/s over all CUs);
// its time is the two f32 MFMA chains (2 x 128 v_mfma_f32_32x32x2 per 32-row tile and wave, two waves per SIMD: ~95 us of matrix-core time at K = 16 at the
// clock the part sustains under MFMA load) plus the argmin / label phase between them.  Not kept.
static int kmeans_mfma_groups(int K, int n_runs) {
    const int G = 32 / (K <= 8 ? 8 : K <= 16 ? 16 : 32);
    return (n_runs + G - 1) / G;
}
// K <= 8 too (round 3): the one-wave-per-row kernel of dic_latent.hip walks X once per RESTART (grid.y = restart: 20 x 77 MB out of L2 /
// Infinity Cache per Lloyd iteration of KMeans(4, n_init=20), 0.41 ms); here four restarts of up to 8 centroids share every X tile.
// Only where there is something to share -- two or more restarts in the launch; a single restart (fixed init, n_init = 1) stays on the
// one-pass wave-per-row kernel.  DIC_KMEANS_SMALLK_MFMA=0 keeps the round-2 kernel for every K <= 8 launch.
bool kmeans_use_mfma(int K, int n_runs) {
    static const bool small_k = [] { const char* e = getenv("DIC_KMEANS_SMALLK_MFMA"); return !(e && e[0] == '0'); }();
    return K > 8 || (small_k && n_runs >= 2);
}
int kmeans_mfma_blocks(int N, int K, int n_runs) {
    return (int)max(1L, min(((long)N + 127) / 128, (long)max(1, 2 * kNumCU / kmeans_mfma_groups(K, n_runs))));      // two workgroups per CU
}

template <int KP, int NMB>
static int kmeans_assign_mfma_launch_t(const KmMfmaArgs& a, int groups, hipStream_t st) {
    constexpr int MCOLS = 32 * NMB, G = MCOLS / KP;
    const size_t lds = (size_t)MCOLS * MCP * 4 + MCOLS * 4 + MCOLS * 4 + 16 + 4 * G * 32 + 4 * 32 * XP * 4;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)kmeans_assign_mfma_kernel<KP, NMB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "kmeans_assign_mfma: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL((kmeans_assign_mfma_kernel<KP, NMB>), dim3(a.nblk, groups), dim3(256), lds, st, a);
    return DIC_OK;
}

// Launches the E-step + partial M-step for K in (8, 32].  nblk = kmeans_mfma_blocks(N, K, n_runs) partial slots per restart are written.
int kmeans_assign_mfma_launch(const float* X, const float* xnorm, int N, int D, int K, int n_runs, const float* centers, int32_t* labels,
                              const float* status, float* mind, float* psum, int* pcnt, hipStream_t st) {
    KmMfmaArgs a{X, xnorm, N, D, K, n_runs, kmeans_mfma_blocks(N, K, n_runs), centers, labels, status, mind, psum, pcnt};
    const int groups = kmeans_mfma_groups(K, n_runs);
    const int rc = K <= 8 ? kmeans_assign_mfma_launch_t<8, 1>(a, groups, st)
                 : K <= 16 ? kmeans_assign_mfma_launch_t<16, 1>(a, groups, st) : kmeans_assign_mfma_launch_t<32, 1>(a, groups, st);
    return rc ? rc : check_launch("kmeans_assign_mfma");
}

}  // namespace dic
