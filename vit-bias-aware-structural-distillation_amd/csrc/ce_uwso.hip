// Cross-entropy on (soft or hard) targets with label smoothing, its gradient, and the UW-SO combination with the
// Procrustes term, fused (reference src/losses/combined.py:57,78-85 + nn.CrossEntropyLoss(label_smoothing) built at
// src/training/trainer.py:47):
//     t'_bc = (1 - s) t_bc + s / C,   ce = mean_b( -sum_c t'_bc log_softmax(z_b)_c ),
//     w_ce = (1 / ce) / (1 / ce + 1 / geo),  w_geo = 1 - w_ce   (both detached, clamped at eps like the reference),
//     total = w_ce ce + w_geo geo,   d total / d z_bc = w_ce (softmax(z_b)_c sum_c' t'_bc' - t'_bc) / B.
// geo is a DEVICE scalar (the loss never synchronises with the host).  Two launches: one workgroup per row (row loss
// + unscaled gradient), then a scaling pass in which every workgroup re-derives the two weights from the B row losses
// (B <= a few thousand floats, cheaper than a third launch); torch runs ~25 pointwise / reduction launches for this.
#include "basd_common.h"

namespace basd {

__device__ __forceinline__ float block_reduce_256(float v, float* red, bool is_max) {
  // DPP wave reduction, then the four waves through LDS
  if (is_max) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  } else {
    v = wave_sum(v);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return is_max ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const float* __restrict__ soft,
                                                      const int64_t* __restrict__ labels, int C, float smoothing,
                                                      float* __restrict__ row_loss, float* __restrict__ dlogits) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* z = logits + (size_t)b * C;
  float mx = -3.0e38f;
  for (int c = tid; c < C; c += 256) mx = fmaxf(mx, z[c]);
  mx = block_reduce_256(mx, red, true);
  float se = 0.f;
  for (int c = tid; c < C; c += 256) se += __expf(z[c] - mx);
  se = block_reduce_256(se, red, false);
  const float lse = mx + __logf(se);
  const int64_t lab = labels ? labels[b] : -1;
  const float off = smoothing / (float)C, on = 1.f - smoothing;
  float loss = 0.f, tsum = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float t = on * (soft ? soft[(size_t)b * C + c] : (c == lab ? 1.f : 0.f)) + off;
    loss = fmaf(-t, z[c] - lse, loss);
    tsum += t;
  }
  loss = block_reduce_256(loss, red, false);
  tsum = block_reduce_256(tsum, red, false);         // 1 for proper distributions; kept general (soft targets may not sum to 1)
  if (tid == 0) row_loss[b] = loss;
  for (int c = tid; c < C; c += 256) {
    const float t = on * (soft ? soft[(size_t)b * C + c] : (c == lab ? 1.f : 0.f)) + off;
    dlogits[(size_t)b * C + c] = __expf(z[c] - lse) * tsum - t;
  }
}

__global__ __launch_bounds__(256) void ce_uwso_finish_kernel(const float* __restrict__ row_loss, int B, int64_t n,
                                                             const float* __restrict__ geo, float* __restrict__ dlogits,
                                                             float* __restrict__ out4) {
  __shared__ float red[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) s += row_loss[b];
  s = block_reduce_256(s, red, false);
  const float eps = 1.1920929e-07f;                  // torch.finfo(float32).eps
  const float ce = s / (float)B, g = geo ? *geo : 0.f;
  float w_ce = 1.f, w_geo = 0.f;
  if (geo) {
    const float ic = 1.f / fmaxf(ce, eps), ig = 1.f / fmaxf(g, eps);
    w_ce = ic / (ic + ig);
    w_geo = ig / (ic + ig);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out4[0] = w_ce * ce + w_geo * g;
    out4[1] = ce;
    out4[2] = w_ce;
    out4[3] = w_geo;
  }
  const float scale = w_ce / (float)B;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dlogits[i] *= scale;
}

}  // namespace basd

extern "C" int basd_ce_uwso(const float* logits, const float* soft_targets, const int64_t* labels, int B, int C,
                            float smoothing, const float* geo, float* row_loss, float* dlogits, float* out4,
                            void* stream) {
  using namespace basd;
  if (B <= 0 || C <= 0) return BASD_OK;
  if ((soft_targets == nullptr) == (labels == nullptr))
    return fail(BASD_ERR_SHAPE, "ce_uwso: exactly one of soft_targets [B, C] / labels [B] must be given");
  if (row_loss == nullptr || dlogits == nullptr || out4 == nullptr)
    return fail(BASD_ERR_SHAPE, "ce_uwso: row_loss [B], dlogits [B, C] and out4 [4] are required");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ce_rows_kernel, dim3(B), dim3(256), 0, st, logits, soft_targets, labels, C, smoothing, row_loss,
                     dlogits);
  const int64_t n = (int64_t)B * C;
  int grid = (int)((n + 256 * 8 - 1) / (256 * 8));
  if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(ce_uwso_finish_kernel, dim3(grid), dim3(256), 0, st, row_loss, B, n, geo, dlogits, out4);
  return check_launch("ce_uwso");
}
