// Spectrally weighted principal-angle distances and layer mixing weights of the Grassmannian selector, fused
// (reference src/losses/layer_selector.py:100-108: acos / clamp / weighted sum / softmax over teacher layers, a
// dozen pointwise launches per extraction point there, ~25 torch launches in round 1 of this build):
//     theta_m = acos(min(sigma_ijm, 1 - eps)),   d2_ij = sum_m sw_jm theta_m^2 / sum_m sw_jm,
//     pre_ij = -d2_ij / softplus(log_temp_i),    w_i = softmax_j(pre_ij)
// plus the per-direction coefficient the backward seeds need,
//     coef_ijm = [d(d2_ij) / d(sigma_ijm)] / sigma_ijm  (/ sigma_ijm^2 more when the caller keeps UN-normalised
//     singular vectors sigma_m u_m: Phi = W^T diag(coef) W then equals U^T diag(gsig / sigma) U).
// One workgroup per extraction point i; teacher layers are walked in turn, the D directions are spread over the
// 256 threads.  sw is zero beyond the layer's MP rank (masked), so rank-masked directions drop out of every sum.
#include "basd_common.h"

namespace basd {

__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void angle_weights_kernel(const float* __restrict__ sigma, const float* __restrict__ sw,
                                                            const float* __restrict__ log_temp, int L, int D,
                                                            int unnormalised, float* __restrict__ d2_out,
                                                            float* __restrict__ pre_out, float* __restrict__ w_out,
                                                            float* __restrict__ coef) {
  __shared__ float red[4];
  __shared__ float s_pre[64];
  const int i = blockIdx.x, tid = threadIdx.x;
  const float eps = 1.1920929e-07f;                  // torch.finfo(float32).eps: the reference's clamp(max = 1 - eps)
  const float lt = log_temp[i];
  const float tau = lt > 20.f ? lt : log1pf(expf(lt));     // softplus, torch's threshold
  for (int j = 0; j < L; ++j) {
    const float* sg = sigma + ((size_t)i * L + j) * D;
    const float* swj = sw + (size_t)j * D;
    float den = 0.f;
    for (int m = tid; m < D; m += 256) den += swj[m];
    den = block_sum_256(den, red);                   // 0 at MP rank 0: NaN weights like the reference (flagged by mp_rank)
    float num = 0.f;
    for (int m = tid; m < D; m += 256) {
      const float s = sg[m], w = swj[m];
      const float sc = fminf(s, 1.f - eps);
      const float th = acosf(sc);
      num = fmaf(w * th, th, num);
      // d(d2) / d(sigma) = sw 2 theta (-1 / sqrt(1 - sigma^2)) / den, zero where the clamp is active
      float c = 0.f;
      // cosines below 1e-9 (1e-12 for normalised vectors) are numerical zeros: their 1 / sigma^3 would overflow fp32
      if (s <= 1.f - eps && s > (unnormalised ? 1e-9f : 1e-12f)) {
        const float gs = w * 2.f * th * (-1.f / sqrtf(1.f - sc * sc)) / den;
        c = unnormalised ? gs / (s * s * s) : gs / s;
      }
      coef[((size_t)i * L + j) * D + m] = c;
    }
    num = block_sum_256(num, red);
    if (tid == 0) {
      const float d2 = num / den;
      d2_out[i * L + j] = d2;
      s_pre[j] = -d2 / tau;
      pre_out[i * L + j] = s_pre[j];
    }
  }
  __syncthreads();
  if (tid == 0) {                                    // softmax over the (at most 64) teacher layers
    float mx = -3.0e38f;
    for (int j = 0; j < L; ++j) mx = fmaxf(mx, s_pre[j]);
    float sum = 0.f;
    for (int j = 0; j < L; ++j) sum += expf(s_pre[j] - mx);
    for (int j = 0; j < L; ++j) w_out[i * L + j] = expf(s_pre[j] - mx) / sum;
  }
}

}  // namespace basd

extern "C" int basd_angle_weights(const float* sigma, const float* sw, const float* log_temp, int E, int L, int D,
                                  int unnormalised, float* d2, float* pre, float* weights, float* coef, void* stream) {
  using namespace basd;
  if (E <= 0 || L <= 0) return BASD_OK;
  if (L > 64 || D < 1) return fail(BASD_ERR_SHAPE, "angle_weights: L=%d (<= 64) D=%d", L, D);
  hipLaunchKernelGGL(angle_weights_kernel, dim3(E), dim3(256), 0, (hipStream_t)stream, sigma, sw, log_temp, L, D,
                     unnormalised, d2, pre, weights, coef);
  return check_launch("angle_weights");
}

// ---------------------------------------------------------------------------------------------
// Raise a status bit if any of n fp64 values is NOT <= tol (so a NaN raises it too), with an atomic OR: the health word
// is shared with the kernels of the other stream, which atomicOr into it concurrently (a torch bitwise_or_ is a plain
// read-modify-write and can lose their flag).
namespace basd {
__global__ __launch_bounds__(256) void flag_if_exceeds_kernel(const double* __restrict__ v, int64_t n, double tol,
                                                              int bit, int* __restrict__ status) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) bad |= !(v[i] <= tol);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(status, bit);
}
}  // namespace basd

extern "C" int basd_flag_if_exceeds_f64(const double* values, int64_t n, double tol, int bit, int32_t* status,
                                        void* stream) {
  using namespace basd;
  if (n <= 0 || status == nullptr) return BASD_OK;
  const int grid = (int)((n + 255) / 256 < 256 ? (n + 255) / 256 : 256);
  hipLaunchKernelGGL(flag_if_exceeds_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, values, n, tol, bit,
                     (int*)status);
  return check_launch("flag_if_exceeds");
}

