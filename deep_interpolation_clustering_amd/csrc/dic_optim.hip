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

}  // namespace dic

using namespace dic;

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

}  // extern "C"
