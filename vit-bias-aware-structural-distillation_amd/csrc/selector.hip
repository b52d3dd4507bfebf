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
// Backward of the selector weights (autograd of src/losses/layer_selector.py:86-108 w.r.t. the student tokens and the
// temperatures) as ONE C entry.  From the forward's seeds T_ij = A_full A_bar^T Phi (rows b >= k_j) it forms, per
// extraction point i,
//     g_pre = w (g_w - <w, g_w>) (+ g_pre_out),   g_d2 = -g_pre / tau,   g_lt = sigmoid(lt) sum_j g_pre d2 / tau^2
//     C_i   = sum_j g_d2_ij T_ij,   K_i[b, a] = C_i[b, a] / (lam_a - lam_b)      (0 where the gap is 0)
//     G_i   = V_i^T K_i V_i,        W_i = P^T (G_i + G_i^T) P
// so that d loss / d s_i = (s_i - mean) W_i.  One reduction kernel + four batched fp64 GEMMs (basd_bgemm_f64) + one
// symmetrising store on a caller workspace; rounds 1 - 3 ran this block as ~15 torch launches.
namespace basd {

__global__ __launch_bounds__(256) void selector_bwd_seed_kernel(
    const float* __restrict__ g_w, const float* __restrict__ g_pre_out, const float* __restrict__ wts,
    const float* __restrict__ d2, const float* __restrict__ log_temp, const float* __restrict__ t_seed,
    const double* __restrict__ lam, int L, int D, float* __restrict__ g_lt, double* __restrict__ k_out) {
  __shared__ float s_gd2[64];
  __shared__ float s_tau;
  const int i = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  if (tid == 0) {
    const float lt = log_temp[i];
    const float tau = lt > 20.f ? lt : log1pf(expf(lt));
    float dotw = 0.f;
    for (int j = 0; j < L; ++j) dotw += wts[i * L + j] * g_w[i * L + j];
    float g_tau = 0.f;
    for (int j = 0; j < L; ++j) {
      float gp = wts[i * L + j] * (g_w[i * L + j] - dotw);
      if (g_pre_out != nullptr) gp += g_pre_out[i * L + j];
      s_gd2[j] = -gp / tau;
      g_tau += gp * d2[i * L + j];
    }
    g_tau /= tau * tau;
    if (b == 0) g_lt[i] = g_tau / (1.f + expf(-lt));
    s_tau = tau;
  }
  __syncthreads();
  const double lb = lam[(size_t)i * D + b];
  for (int a = tid; a < D; a += 256) {
    double c = 0.0;
    for (int j = 0; j < L; ++j) c += (double)(s_gd2[j] * t_seed[(((size_t)i * L + j) * D + b) * D + a]);
    const double gap = lam[(size_t)i * D + a] - lb;
    k_out[((size_t)i * D + b) * D + a] = gap != 0.0 ? c / gap : 0.0;
  }
}

__global__ __launch_bounds__(256) void symmetrise_store_kernel(const double* __restrict__ h, int n, float* __restrict__ out) {
  const int i = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n * n) {
    const int r = e / n, c = e - r * n;
    out[(size_t)i * n * n + e] = (float)(h[(size_t)i * n * n + e] + h[(size_t)i * n * n + (size_t)c * n + r]);
  }
}

}  // namespace basd

extern "C" int64_t basd_angle_weights_bwd_workspace_bytes(int E, int D, int D_s) {
  return (int64_t)E * ((int64_t)3 * D * D + (int64_t)D * D_s + (int64_t)D_s * D_s) * 8 + 1024;
}

extern "C" int basd_angle_weights_bwd(const float* g_w, const float* g_pre_out, const float* weights, const float* d2,
                                      const float* log_temp, const float* t_seed, const float* v_s, const double* lam_s,
                                      const float* proj_s, int E, int L, int D, int D_s, float* g_log_temp, float* w_tok,
                                      void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace basd;
  if (E <= 0 || L <= 0) return BASD_OK;
  if (L > 64 || D < 1 || D_s < 1) return fail(BASD_ERR_SHAPE, "angle_weights_bwd: L=%d (<= 64) D=%d D_s=%d", L, D, D_s);
  if (workspace == nullptr || workspace_bytes < basd_angle_weights_bwd_workspace_bytes(E, D, D_s))
    return fail(BASD_ERR_SHAPE, "angle_weights_bwd: workspace of %lld bytes, need %lld", (long long)workspace_bytes,
                (long long)basd_angle_weights_bwd_workspace_bytes(E, D, D_s));
  double* k = reinterpret_cast<double*>((((uintptr_t)workspace) + 255) & ~(uintptr_t)255);
  double* kv = k + (size_t)E * D * D;            // K V
  double* g = kv + (size_t)E * D * D;            // V^T K V
  double* gp = g + (size_t)E * D * D;            // G P          [E, D, D_s]
  double* h = gp + (size_t)E * D * D_s;          // P^T G P      [E, D_s, D_s]
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(selector_bwd_seed_kernel, dim3(E, D), dim3(256), 0, st, g_w, g_pre_out, weights, d2, log_temp,
                     t_seed, lam_s, L, D, g_log_temp, k);
  int rc = check_launch("angle_weights_bwd (seed reduction)");
  if (rc) return rc;
  const int64_t dd = (int64_t)D * D;
  // K V:  [D, D] f64 x [D, D] f32
  rc = basd_bgemm_f64(k, BASD_DTYPE_F64, dd, D, 0, v_s, BASD_DTYPE_F32, dd, D, 0, kv, BASD_DTYPE_F64, dd, D, E, D, D, D, 0, stream);
  if (rc) return rc;
  // V^T (K V)
  rc = basd_bgemm_f64(v_s, BASD_DTYPE_F32, dd, D, 1, kv, BASD_DTYPE_F64, dd, D, 0, g, BASD_DTYPE_F64, dd, D, E, D, D, D, 0, stream);
  if (rc) return rc;
  // G P:  proj_s [D, D_s] shared by the batch (stride 0)
  rc = basd_bgemm_f64(g, BASD_DTYPE_F64, dd, D, 0, proj_s, BASD_DTYPE_F32, 0, D_s, 0, gp, BASD_DTYPE_F64, (int64_t)D * D_s, D_s,
                      E, D, D_s, D, 0, stream);
  if (rc) return rc;
  // P^T (G P)
  rc = basd_bgemm_f64(proj_s, BASD_DTYPE_F32, 0, D_s, 1, gp, BASD_DTYPE_F64, (int64_t)D * D_s, D_s, 0, h, BASD_DTYPE_F64,
                      (int64_t)D_s * D_s, D_s, E, D_s, D_s, D, 0, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(symmetrise_store_kernel, dim3((D_s * D_s + 255) / 256, E), dim3(256), 0, st, h, D_s, w_tok);
  return check_launch("angle_weights_bwd (symmetrise)");
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

