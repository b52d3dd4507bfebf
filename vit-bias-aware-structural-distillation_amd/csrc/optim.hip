// Fused Schedule-Free AdamW step over one flat fp32 parameter buffer
// (schedulefree 1.4.1 AdamWScheduleFree, train mode; reference call site
// src/training/trainer.py:54-58,158-159).  HBM-bound: reads y, g, z, v and
// writes y, z, v once (28 B per parameter), 16 B per lane.
#include "basd_common.h"

namespace basd {

__global__ __launch_bounds__(256) void sf_adamw_kernel(float* __restrict__ y, const float* __restrict__ g,
                                                       float* __restrict__ z, float* __restrict__ v,
                                                       int64_t n4, int64_t n, float lr, float y_step,
                                                       float beta2, float omb2, float eps, float wd,
                                                       float ckp1, float inv_bc2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 yy = reinterpret_cast<float4*>(y)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 zz = reinterpret_cast<float4*>(z)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* py = &yy.x; const float* pg = &gg.x; float* pz = &zz.x; float* pv = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float grad = pg[k];
      const float vn = beta2 * pv[k] + omb2 * grad * grad;
      const float denom = sqrtf(vn * inv_bc2) + eps;
      const float gn = grad / denom + wd * py[k];        // weight decay evaluated at y
      float yn = py[k] + ckp1 * (pz[k] - py[k]);         // y.lerp_(z, ckp1)
      yn = fmaf(y_step, gn, yn);
      pz[k] = fmaf(-lr, gn, pz[k]);
      py[k] = yn;
      pv[k] = vn;
    }
    reinterpret_cast<float4*>(y)[i] = yy;
    reinterpret_cast<float4*>(z)[i] = zz;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n % 4 elements)
  if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
    const int64_t i = n4 * 4 + threadIdx.x;
    const float grad = g[i];
    const float vn = beta2 * v[i] + omb2 * grad * grad;
    const float denom = sqrtf(vn * inv_bc2) + eps;
    const float gn = grad / denom + wd * y[i];
    float yn = y[i] + ckp1 * (z[i] - y[i]);
    yn = fmaf(y_step, gn, yn);
    z[i] = fmaf(-lr, gn, z[i]);
    y[i] = yn;
    v[i] = vn;
  }
}

// y <- y + w (z - y): the train()/eval() mode switch of schedule-free optimisers
__global__ __launch_bounds__(256) void lerp_kernel(float* __restrict__ y, const float* __restrict__ z, int64_t n,
                                                   float w) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = y[i] + w * (z[i] - y[i]);
}

}  // namespace basd

extern "C" int basd_sf_adamw_step(float* y, const float* g, float* z, float* v, int64_t n, double lr,
                                  double beta1, double beta2, double eps, double weight_decay, double ckp1,
                                  double bias_correction2, void* stream) {
  using namespace basd;
  if (n <= 0) return BASD_OK;
  if (((uintptr_t)y | (uintptr_t)g | (uintptr_t)z | (uintptr_t)v) & 15)
    return fail(BASD_ERR_SHAPE, "sf_adamw_step: buffers must be 16-byte aligned");
  const int64_t n4 = n / 4;
  int64_t grid = (n4 + 255) / 256;
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  // derived scalars are formed in double on the host (1 - 0.999f is off by 5e-5 relative in fp32)
  hipLaunchKernelGGL(sf_adamw_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, y, g, z, v, n4, n,
                     (float)lr, (float)(lr * (beta1 * (1.0 - ckp1) - 1.0)), (float)beta2, (float)(1.0 - beta2),
                     (float)eps, (float)weight_decay, (float)ckp1, (float)(1.0 / bias_correction2));
  return check_launch("sf_adamw_step");
}

extern "C" int basd_lerp(float* y, const float* z, int64_t n, float w, void* stream) {
  using namespace basd;
  if (n <= 0) return BASD_OK;
  int64_t grid = (n + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(lerp_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, y, z, n, w);
  return check_launch("lerp");
}
