// Tail of the joint step (pretrain_trainer.py:228-229 / clustering_trainer.py:278-279): clip_grad_norm_(grad_clip)
// followed by Adam(amsgrad=True, weight_decay = L2) (utils.py:83) over the ONE flat parameter / gradient bucket of
// dist.FlatParams, as a single element-wise kernel: 5 streams x 2.3 MB.  torch's fused multi-tensor Adam spends
// 0.11 ms per step here (25 small tensors -> a handful of workgroups); this is ~10 us.
//
// Update rule = torch.optim.Adam's (torch/optim/adam.py, single-tensor path), in its operation order:
//   g <- coef * g  (written back, as clip_grad_norm_ does) ;  g += wd * p
//   m = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2 ;  vmax = max(vmax, v)
//   denom = sqrt(vmax) / sqrt(1 - b2^t) + eps ;  p -= lr / (1 - b1^t) * m / denom
// `step` (t, already incremented by the caller), `coef` and -- when `hyper` = [lr, beta1, beta2, eps, weight_decay] is given -- the
// hyper-parameters are read from device memory, so that ONE captured hipGraph serves every learning rate a scheduler sets.
#include "dic_common.h"

namespace dic {

__global__ __launch_bounds__(256) void adam_amsgrad_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                           float* __restrict__ v, float* __restrict__ vmax, long n, float lr, float b1,
                                                           float b2, float eps, float wd, const float* __restrict__ step,
                                                           const float* __restrict__ coef, const unsigned char* __restrict__ active,
                                                           const float* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; b1 = hyper[1]; b2 = hyper[2]; eps = hyper[3]; wd = hyper[4]; }      // device-resident: a captured graph follows the scheduler
    const float t = step[0];
    const float c = coef ? coef[0] : 1.0f;
    const float bc1 = 1.0f - powf(b1, t), bc2 = 1.0f - powf(b2, t);
    const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        if (active && !active[i]) continue;       // no gradient reached this parameter: torch.optim skips it (no decay, no state)
        const float pi = p[i];
        float gi = g[i] * c;
        g[i] = gi;
        gi = fmaf(wd, pi, gi);
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);          // lerp_, as torch writes it
        const float vi = fmaf(b2, v[i], (1.0f - b2) * gi * gi);
        const float vm = fmaxf(vmax[i], vi);
        m[i] = mi;
        v[i] = vi;
        vmax[i] = vm;
        p[i] = pi - step_size * (mi / (sqrtf(vm) / bc2_sqrt + eps));
    }
}

// clip_grad_norm_'s two scalars over the flat bucket: total = ||g||_2 (f64 accumulation of per-workgroup partials, fixed order),
// coef = min(1, max_norm / (total + 1e-6)) (torch/nn/utils/clip_grad.py) -- two launches instead of norm + add + div + clamp
__global__ __launch_bounds__(256) void grad_sqsum_kernel(const float* g, long n, double* partials) {
    __shared__ double red[4];
    float acc = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) acc = fmaf(g[i], g[i], acc);
    const double w = wave_sum((double)acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ __launch_bounds__(256) void grad_norm_finalize(const double* partials, int nblk, float max_norm, float* out2) {
    __shared__ double red[256];
    const double s = reduce_partials_32x8(partials, nblk, 1, 0, red);
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(s);
        out2[0] = total;
        out2[1] = fminf(1.0f, max_norm / (total + 1e-6f));
    }
}

static int norm_blocks(long n) { return (int)max(1L, min((n + 1023) / 1024, (long)2 * kNumCU)); }

}  // namespace dic

using namespace dic;

// dst[j][0..n[j]) += src[j][0..n[j]) for up to 16 small tensors in ONE launch: the parameter gradients the hot-path kernels deliver as
// separate little tensors (interpolation / RBF bandwidths, cross-channel matrix, centroids, CompressFC's six) go into their slots of the
// flat gradient bucket this way instead of one 5-us AccumulateGrad add each (pretrain_trainer.py:223-231: backward, then the optimizer).
constexpr int kAccumMax = 16;
struct AccumMany {
    const float* src[kAccumMax];
    float* dst[kAccumMax];
    int n[kAccumMax];
};
__global__ __launch_bounds__(256) void accumulate_many_kernel(AccumMany a) {
    const int j = blockIdx.x;
    const float* s = a.src[j];
    float* d = a.dst[j];
    for (int i = blockIdx.y * 256 + threadIdx.x; i < a.n[j]; i += gridDim.y * 256) d[i] += s[i];
}

extern "C" {

int dic_adam_amsgrad_step(float* p, float* g, float* m, float* v, float* vmax, int64_t n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, const float* step, const float* grad_scale, const unsigned char* active,
                          const float* hyper, dic_stream_t stream) {
    DIC_REQUIRE(n > 0, DIC_ERR_INVALID_ARG, "adam_amsgrad_step: non-positive size");
    DIC_REQUIRE(p && g && m && v && vmax && step, DIC_ERR_INVALID_ARG, "adam_amsgrad_step: NULL pointer");
    const int grid = (int)max(1L, min(((long)n + 255) / 256, (long)8 * kNumCU));
    hipLaunchKernelGGL(adam_amsgrad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax, (long)n, lr, beta1, beta2,
                       eps, weight_decay, step, grad_scale, active, hyper);
    return check_launch("adam_amsgrad_step");
}

size_t dic_grad_norm_workspace(int64_t n) { return n > 0 ? (size_t)norm_blocks((long)n) * sizeof(double) : 0; }

int dic_grad_norm_clip(const float* g, int64_t n, float max_norm, float* out2, void* workspace, size_t workspace_bytes, dic_stream_t stream) {
    DIC_REQUIRE(n > 0, DIC_ERR_INVALID_ARG, "grad_norm_clip: non-positive size");
    DIC_REQUIRE(g && out2 && workspace, DIC_ERR_INVALID_ARG, "grad_norm_clip: NULL pointer");
    const int nblk = norm_blocks((long)n);
    DIC_REQUIRE(workspace_bytes >= (size_t)nblk * sizeof(double), DIC_ERR_WORKSPACE, "grad_norm_clip: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(grad_sqsum_kernel, dim3(nblk), dim3(256), 0, st, g, (long)n, (double*)workspace);
    hipLaunchKernelGGL(grad_norm_finalize, dim3(1), dim3(256), 0, st, (const double*)workspace, nblk, max_norm, out2);
    return check_launch("grad_norm_clip");
}

int dic_accumulate_many(const float* const* src, float* const* dst, const int* n, int count, dic_stream_t stream) {
    DIC_REQUIRE(count >= 0, DIC_ERR_INVALID_ARG, "accumulate_many: negative count");
    DIC_REQUIRE(count == 0 || (src && dst && n), DIC_ERR_INVALID_ARG, "accumulate_many: NULL pointer");
    for (int base = 0; base < count; base += kAccumMax) {
        AccumMany a{};
        const int m = min(kAccumMax, count - base);
        int nmax = 0;
        for (int j = 0; j < m; ++j) {
            DIC_REQUIRE(src[base + j] && dst[base + j] && n[base + j] > 0, DIC_ERR_INVALID_ARG, "accumulate_many: entry %d is empty", base + j);
            a.src[j] = src[base + j]; a.dst[j] = dst[base + j]; a.n[j] = n[base + j];
            nmax = max(nmax, n[base + j]);
        }
        hipLaunchKernelGGL(accumulate_many_kernel, dim3(m, max(1, min(64, (nmax + 2047) / 2048))), dim3(256), 0, (hipStream_t)stream, a);
    }
    return check_launch("accumulate_many");
}

}  // extern "C"
