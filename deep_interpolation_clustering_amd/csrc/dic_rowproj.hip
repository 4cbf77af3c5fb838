// Row-wise projections with a 256-wide input applied to every (time step, encounter) row:
//     out[row][n] = bias[n] + sum_k x[row][k] W[n][k]          x (N,256) bf16, W (Nout,256) bf16, out (N,Nout) bf16
// i.e. the decoder LSTM's input projection gx = relu(enc_out) . W_ih^T + (b_ih + b_hh) (clustering_interp.py:47-59 through nn.LSTM:
// N = 786 432 rows, Nout = 1024, 412 GFLOP, 1.6 GB of output at B = 32 768).  The library GEMM for this shape has K = 256 only --
// four k-iterations per 256x256 macro-tile, so its prologue / epilogue never overlap -- and ran at 0.58-0.60 ms (3.4 TB/s of its
// own traffic).  Here the weights never move: wave w of a workgroup keeps W[n][0..255] for its 32 output columns in 64 registers
// (the B operand of all 16 MFMA k-steps), the workgroup streams 32-row tiles of x through LDS (global -> registers -> LDS, one
// tile ahead, two images, one barrier per tile), and the 32 x 256 output block leaves through an LDS staging tile as whole
// 512-B row segments.  grid (chunks, Nout / 256): the workgroups of a row chunk sit on one XCD (chunk count a multiple of 8), so
// x is fetched from HBM about once and served to the other column stripes by that XCD's L2.
#include "dic_common.h"

namespace dic {

constexpr int PK = 256;                              // input width (K)
constexpr int PT = 32;                               // rows per tile
constexpr int PX_PITCH = PK * 2 + 48;                // 560 B: the 32 rows of a straight 16-B read fall on disjoint bank groups
constexpr int PS_PITCH = 256 * 2 + 16;               // 528 B: staging rows of the 256-column output block
constexpr int P_TILE = PT * PX_PITCH, P_STAGE = PT * PS_PITCH;
constexpr int P_LDS = 2 * P_TILE + 2 * P_STAGE;      // 69 632 B: two workgroups per CU

typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x16 __attribute__((ext_vector_type(16)));

struct RowProjArgs {
    const __bf16* x;       // (N, 256)
    const __bf16* w;       // (Nout, 256)
    const __bf16* bias;    // (Nout) or NULL
    __bf16* out;           // (N, Nout)
    long N;
    int Nout;
};

__global__ __launch_bounds__(512, 2) void row_proj_kernel(RowProjArgs a) {
    extern __shared__ __align__(16) unsigned char psm[];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long N = a.N;
    const int ntiles = (int)((N + PT - 1) / PT), nch = gridDim.x;
    const int n0 = blockIdx.y * 256, ncol = n0 + 32 * w + (lane & 31);
    unsigned char* stage = psm + 2 * P_TILE;

    pbf16x8 wreg[PK / 16];                 // B operand: W[ncol][16 ks + 8 hh .. + 7]
#pragma unroll
    for (int ks = 0; ks < PK / 16; ++ks) wreg[ks] = *reinterpret_cast<const pbf16x8*>(a.w + (size_t)ncol * PK + 16 * ks + 8 * hh);
    const float bn = a.bias ? (float)a.bias[ncol] : 0.f;

    const int xrow = tid >> 5, xpc = tid & 31;               // rows xrow, xrow + 16; 32 pieces of 16 B per row
    uint4 p0, p1;
    auto request = [&](int tile) {           // (clamped, always valid addresses; rows past the end are never stored)
        const long r0 = (long)tile * PT;
        p0 = *reinterpret_cast<const uint4*>(a.x + (size_t)min(r0 + xrow, N - 1) * PK + xpc * 8);
        p1 = *reinterpret_cast<const uint4*>(a.x + (size_t)min(r0 + xrow + 16, N - 1) * PK + xpc * 8);
    };
    auto land = [&](int slot) {
        unsigned char* base = psm + slot * P_TILE;
        *reinterpret_cast<uint4*>(base + xrow * PX_PITCH + xpc * 16) = p0;
        *reinterpret_cast<uint4*>(base + (xrow + 16) * PX_PITCH + xpc * 16) = p1;
    };
    const int a_off = (lane & 31) * PX_PITCH + hh * 16;       // A operand: row (lane & 31), 16-B piece 2 ks + hh

    int tile = blockIdx.x;
    if (tile < ntiles) { request(tile); land(0); }
    if (tile + nch < ntiles) request(tile + nch);
    __syncthreads();
    int slot = 0;
    for (; tile < ntiles; tile += nch) {
        const unsigned char* base = psm + slot * P_TILE;
        pf32x16 acc;
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = bn;
#pragma unroll
        for (int ks = 0; ks < PK / 16; ++ks) {
            const pbf16x8 af = *reinterpret_cast<const pbf16x8*>(base + a_off + ks * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, wreg[ks], acc, 0, 0, 0);
        }
        if (tile + nch < ntiles) land(slot ^ 1);
        if (tile + 2 * nch < ntiles) request(tile + 2 * nch);
        // C/D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        unsigned char* sb = stage + slot * P_STAGE;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int m = (k & 3) + 8 * (k >> 2) + 4 * hh;
            *reinterpret_cast<__bf16*>(sb + m * PS_PITCH + (32 * w + (lane & 31)) * 2) = (__bf16)acc[k];
        }
        __syncthreads();
        const long r0 = (long)tile * PT;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int p = tid + 512 * k, row = p >> 5, pc = p & 31;
            if (r0 + row < N)
                *reinterpret_cast<uint4*>(a.out + (size_t)(r0 + row) * a.Nout + n0 + pc * 8) = *reinterpret_cast<const uint4*>(sb + row * PS_PITCH + pc * 16);
        }
        slot ^= 1;
    }
}

}  // namespace dic

using namespace dic;

extern "C" {

int dic_row_proj(const void* x, const void* w, const void* bias, int64_t N, int in_features, int out_features, void* out, dic_stream_t stream) {
    DIC_REQUIRE(N > 0, DIC_ERR_INVALID_ARG, "row_proj: non-positive row count");
    DIC_REQUIRE(in_features == PK && out_features > 0 && out_features % 256 == 0, DIC_ERR_UNSUPPORTED,
                "row_proj: (%d -> %d) (compiled for 256 inputs and a multiple of 256 outputs)", in_features, out_features);
    DIC_REQUIRE(x && w && out, DIC_ERR_INVALID_ARG, "row_proj: NULL pointer");
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)row_proj_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
        DIC_REQUIRE(e == hipSuccess, DIC_ERR_LAUNCH, "row_proj: cannot reserve %d B of LDS: %s", P_LDS, hipGetErrorString(e));
        attr_set = true;
    }
    const int stripes = out_features / 256;
    const int ntiles = (int)((N + PT - 1) / PT);
    int nch = max(1, min(ntiles, 2 * kNumCU / stripes));          // two workgroups per CU
    if (nch >= 8) nch = nch / 8 * 8;                             // a multiple of 8: the stripes of a row chunk share an XCD (and its L2)
    RowProjArgs a{(const __bf16*)x, (const __bf16*)w, (const __bf16*)bias, (__bf16*)out, (long)N, out_features};
    hipLaunchKernelGGL(row_proj_kernel, dim3(nch, stripes), dim3(512), P_LDS, (hipStream_t)stream, a);
    return check_launch("row_proj");
}

}  // extern "C"
