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

// Transposed bf16 images of the weight matrices for the input-gradient GEMMs (dX = dY W runs as the "NT" kernel on
// W^T): all matrices of the model in ONE launch, read from the fp32 master buffer.  A 32 x 32 tile per workgroup goes
// through LDS (padded rows) so that both the fp32 reads and the bf16 writes are along rows.
constexpr int TT_MAX = 64;
struct TransposeTable {
  int64_t src[TT_MAX], dst[TT_MAX];
  int rows[TT_MAX], cols[TT_MAX], tile0[TT_MAX + 1];
  int n;
};

__global__ __launch_bounds__(256) void transpose_table_kernel(const float* __restrict__ master,
                                                              unsigned short* __restrict__ out, TransposeTable t) {
  __shared__ float tile[32][33];
  const int b = blockIdx.x;
  int e = 0;
  while (e + 1 < t.n && b >= t.tile0[e + 1]) ++e;            // <= 64 entries, wave-uniform
  const int rows = t.rows[e], cols = t.cols[e];
  const int tiles_c = (cols + 31) / 32;
  const int local = b - t.tile0[e];
  const int r0 = (local / tiles_c) * 32, c0 = (local % tiles_c) * 32;
  const float* src = master + t.src[e];
  unsigned short* dst = out + t.dst[e];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;                // output row = source column
    if (c < cols && r < rows) dst[(int64_t)c * rows + r] = __builtin_bit_cast(unsigned short, (__bf16)tile[tx][ty + 8 * i]);
  }
}

}  // namespace basd

extern "C" int basd_transpose_bf16_table(const float* master, void* out, const int64_t* table, int n_entries,
                                         void* stream) {
  using namespace basd;
  if (n_entries < 0) return fail(BASD_ERR_SHAPE, "transpose_bf16_table: n_entries = %d", n_entries);
  for (int base = 0; base < n_entries; base += TT_MAX) {
    TransposeTable t;
    t.n = n_entries - base < TT_MAX ? n_entries - base : TT_MAX;
    int tiles = 0;
    for (int i = 0; i < t.n; ++i) {
      const int64_t* e = table + 4 * (int64_t)(base + i);
      if (e[2] <= 0 || e[3] <= 0 || e[2] > 0x7fffffff || e[3] > 0x7fffffff)
        return fail(BASD_ERR_SHAPE, "transpose_bf16_table: entry %d has shape %lld x %lld", base + i, (long long)e[2],
                    (long long)e[3]);
      t.src[i] = e[0]; t.dst[i] = e[1]; t.rows[i] = (int)e[2]; t.cols[i] = (int)e[3];
      t.tile0[i] = tiles;
      tiles += (int)((e[2] + 31) / 32) * (int)((e[3] + 31) / 32);
    }
    t.tile0[t.n] = tiles;
    hipLaunchKernelGGL(transpose_table_kernel, dim3(tiles), dim3(256), 0, (hipStream_t)stream, master,
                       (unsigned short*)out, t);
  }
  return check_launch("transpose_bf16_table");
}

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
