// LayerNorm for the ViT blocks: bf16 activations in/out, fp32 gamma/beta, fp32 statistics.
//
// Under bf16 autocast torch upcasts layer_norm inputs to fp32 (a copy in, a copy out per call) and
// its ROCm kernel refuses bf16 activations with fp32 parameters; backward is three more kernels.
// Here forward is one pass (x read once, y written once, mean / rstd saved) and backward is one
// pass that also accumulates d gamma / d beta (per-thread column partials, one fp32 atomic per
// column per workgroup -- straight into the flat gradient buffer when the caller passes it).
// The result equals torch's fp32 layer_norm followed by the bf16 rounding the next Linear applies.
//
// Layout: each thread owns 8 contiguous columns (one 16-byte load); a row is covered by
// TPR = D / 8 threads (D % 8 == 0, D <= 2048); a 256-thread workgroup processes
// RPB = 256 / TPR rows per iteration and grid-strides over the rows.
#include "basd_common.h"

namespace basd {

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
  const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(w[i] << 16);
    f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ unsigned int pack2(float a, float b) {
  __hip_bfloat16 x = __float2bfloat16(a), y = __float2bfloat16(b);    // round to nearest even, NaN safe
  return (unsigned int)(*reinterpret_cast<unsigned short*>(&x)) |
         ((unsigned int)(*reinterpret_cast<unsigned short*>(&y)) << 16);
}

// sum over the TPR threads of a row (row groups are contiguous thread ranges) through LDS.
// Called by EVERY thread of the workgroup (the barriers must not sit in a branch that splits a
// wave: with D = 192 the last wave holds both worker and idle lanes); idle threads only synchronise.
template <int NV>
__device__ __forceinline__ void row_reduce(float (&v)[NV], float* red, int row_in_blk, int t_in_row, int tpr,
                                           bool worker) {
  // red: [RPB][NV][TPR] -> every thread of the row sums the TPR partials
  if (worker) {
#pragma unroll
    for (int k = 0; k < NV; ++k) red[(row_in_blk * NV + k) * tpr + t_in_row] = v[k];
  }
  __syncthreads();
  if (worker) {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      float s = 0.f;
      const float* p = red + (row_in_blk * NV + k) * tpr;
      for (int j = 0; j < tpr; ++j) s += p[j];
      v[k] = s;
    }
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void ln_fwd_kernel(const unsigned short* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, int64_t rows, int D, float eps,
                                                     unsigned short* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd) {
  extern __shared__ float red[];
  const int tpr = D >> 3, rpb = 256 / tpr;
  const int tid = threadIdx.x;
  const int row_in_blk = tid / tpr, t_in_row = tid - row_in_blk * tpr;
  const bool worker = row_in_blk < rpb;
  float g[8], b[8];
  if (worker) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { g[i] = gamma[t_in_row * 8 + i]; b[i] = beta[t_in_row * 8 + i]; }
  }
  for (int64_t r0 = (int64_t)blockIdx.x * rpb; r0 < rows; r0 += (int64_t)gridDim.x * rpb) {
    const int64_t r = r0 + row_in_blk;
    const bool live = worker && r < rows;
    float f[8];
    if (live) unpack8(*reinterpret_cast<const uint4*>(x + r * D + t_in_row * 8), f);
    else {
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] = 0.f;
    }
    float s[1] = {0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) s[0] += f[i];
    row_reduce<1>(s, red, row_in_blk, t_in_row, tpr, worker);
    const float mu = s[0] / (float)D;
    float q[1] = {0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float d = f[i] - mu; q[0] = fmaf(d, d, q[0]); }   // two-pass variance
    row_reduce<1>(q, red, row_in_blk, t_in_row, tpr, worker);
    const float rs = rsqrtf(q[0] / (float)D + eps);
    if (live) {
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = fmaf((f[i] - mu) * rs, g[i], b[i]);
      uint4 out;
      out.x = pack2(o[0], o[1]); out.y = pack2(o[2], o[3]); out.z = pack2(o[4], o[5]); out.w = pack2(o[6], o[7]);
      *reinterpret_cast<uint4*>(y + r * D + t_in_row * 8) = out;
      if (t_in_row == 0) { mean[r] = mu; rstd[r] = rs; }
    }
  }
}

__global__ __launch_bounds__(256) void ln_bwd_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, int64_t rows, int D,
                                                     unsigned short* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta) {
  extern __shared__ float red[];
  const int tpr = D >> 3, rpb = 256 / tpr;
  const int tid = threadIdx.x;
  const int row_in_blk = tid / tpr, t_in_row = tid - row_in_blk * tpr;
  const bool worker = row_in_blk < rpb;
  float g[8], dg[8], db[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { g[i] = worker ? gamma[t_in_row * 8 + i] : 0.f; dg[i] = 0.f; db[i] = 0.f; }
  for (int64_t r0 = (int64_t)blockIdx.x * rpb; r0 < rows; r0 += (int64_t)gridDim.x * rpb) {
    const int64_t r = r0 + row_in_blk;
    const bool live = worker && r < rows;
    float fx[8], fdy[8], xh[8], gg[8];
    float mu = 0.f, rs = 0.f;
    if (live) {
      unpack8(*reinterpret_cast<const uint4*>(x + r * D + t_in_row * 8), fx);
      unpack8(*reinterpret_cast<const uint4*>(dy + r * D + t_in_row * 8), fdy);
      mu = mean[r]; rs = rstd[r];
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) { fx[i] = 0.f; fdy[i] = 0.f; }
    }
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      xh[i] = (fx[i] - mu) * rs;
      gg[i] = fdy[i] * g[i];
      s[0] += gg[i];
      s[1] = fmaf(gg[i], xh[i], s[1]);
      dg[i] = fmaf(fdy[i], xh[i], dg[i]);
      db[i] += fdy[i];
    }
    row_reduce<2>(s, red, row_in_blk, t_in_row, tpr, worker);
    if (live) {
      const float m1 = s[0] / (float)D, m2 = s[1] / (float)D;
      float o[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) o[i] = rs * (gg[i] - m1 - xh[i] * m2);
      uint4 out;
      out.x = pack2(o[0], o[1]); out.y = pack2(o[2], o[3]); out.z = pack2(o[4], o[5]); out.w = pack2(o[6], o[7]);
      *reinterpret_cast<uint4*>(dx + r * D + t_in_row * 8) = out;
    }
  }
  // column partials: combine the RPB row groups of the workgroup in LDS, then one atomic per column
  if (dgamma != nullptr) {
    float* acc = red;                                  // [2][D]
    for (int i = tid; i < 2 * D; i += 256) acc[i] = 0.f;
    __syncthreads();
    if (worker) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        atomicAdd(&acc[t_in_row * 8 + i], dg[i]);
        atomicAdd(&acc[D + t_in_row * 8 + i], db[i]);
      }
    }
    __syncthreads();
    for (int i = tid; i < D; i += 256) {
      atomicAdd(&dgamma[i], acc[i]);
      atomicAdd(&dbeta[i], acc[D + i]);
    }
  }
}

static int ln_grid(int64_t rows, int rpb) {
  int64_t g = (rows + rpb - 1) / rpb;
  if (g > 1024) g = 1024;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace basd

extern "C" int basd_layernorm_fwd_bf16(const void* x, const float* gamma, const float* beta, int64_t rows, int D,
                                       float eps, void* y, float* mean, float* rstd, void* stream) {
  using namespace basd;
  if (rows <= 0) return BASD_OK;
  if (D % 8 || D < 8 || D > 2048) return fail(BASD_ERR_SHAPE, "layernorm_fwd_bf16: D %% 8 != 0 or D > 2048 (%d)", D);
  const int tpr = D / 8, rpb = 256 / tpr;
  const size_t lds = (size_t)rpb * 2 * tpr * 4 + 64;
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(ln_grid(rows, rpb)), dim3(256), lds, (hipStream_t)stream,
                     (const unsigned short*)x, gamma, beta, rows, D, eps, (unsigned short*)y, mean, rstd);
  return check_launch("layernorm_fwd_bf16");
}

extern "C" int basd_layernorm_bwd_bf16(const void* dy, const void* x, const float* gamma, const float* mean,
                                       const float* rstd, int64_t rows, int D, void* dx, float* dgamma, float* dbeta,
                                       void* stream) {
  using namespace basd;
  if (rows <= 0) return BASD_OK;
  if (D % 8 || D < 8 || D > 2048) return fail(BASD_ERR_SHAPE, "layernorm_bwd_bf16: D %% 8 != 0 or D > 2048 (%d)", D);
  const int tpr = D / 8, rpb = 256 / tpr;
  size_t lds = (size_t)rpb * 2 * tpr * 4;
  if (lds < (size_t)2 * D * 4) lds = (size_t)2 * D * 4;
  lds += 64;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(ln_grid(rows, rpb)), dim3(256), lds, (hipStream_t)stream,
                     (const unsigned short*)dy, (const unsigned short*)x, gamma, mean, rstd, rows, D,
                     (unsigned short*)dx, dgamma, dbeta);
  return check_launch("layernorm_bwd_bf16");
}
