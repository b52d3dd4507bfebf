// Teacher attention tap: head-averaged CLS-row attention  importance[b, t-1] =
// mean_h softmax_t( bf16(q_cls . k_t) * scale )   for t = 1 .. T-1,
// read straight from the packed qkv activations [B, T, 3, H, hd] (bf16) of the block.
// This is the only part of the teacher's attention map the loss consumes (reference
// src/models/teacher.py:33-37 builds softmax(QK^T / sqrt(hd)) for every head and query,
// src/losses/relational.py:22-27 keeps the CLS row averaged over heads).  The torch
// formulation (q[:, :, :1] @ k^T) first makes K^T contiguous: a 77 MB copy plus a
// 3072-batch GEMV per teacher layer, 2.1 ms per step; this kernel streams K once.
#include "basd_common.h"

namespace basd {

// one workgroup per batch element, one wave per head (round robin), lanes over tokens
template <int HD>
__global__ __launch_bounds__(256) void cls_importance_kernel(const unsigned short* __restrict__ qkv, int T, int H,
                                                             float scale, float* __restrict__ out) {
  extern __shared__ float s_acc[];                 // [nwaves][T]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const size_t row = (size_t)3 * H * HD;           // elements per token
  const unsigned short* base = qkv + (size_t)blockIdx.x * T * row;
  float* acc = s_acc + (size_t)wave * T;
  for (int t = lane; t < T; t += 64) acc[t] = 0.f;
  constexpr int MAXT = 5;                          // tokens per lane: T <= 320
  for (int h = wave; h < H; h += nw) {
    // q of the CLS token, this head (same for every lane: scalar loads)
    float q[HD];
    const unsigned short* qp = base + (size_t)h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) q[d] = bf16_bits_to_f32(qp[d]);
    float logit[MAXT];
    float mx = -3.0e38f;
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
      const int t = lane + 64 * u;
      logit[u] = -3.0e38f;
      if (t < T) {
        const uint4* kp = reinterpret_cast<const uint4*>(base + (size_t)t * row + (size_t)(H + h) * HD);
        float dot = 0.f;
#pragma unroll
        for (int v = 0; v < HD / 8; ++v) {
          const uint4 w = kp[v];
          const unsigned int ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            dot = fmaf(q[8 * v + 2 * e], __uint_as_float(ww[e] << 16), dot);
            dot = fmaf(q[8 * v + 2 * e + 1], __uint_as_float(ww[e] & 0xffff0000u), dot);
          }
        }
        // the reference's autocast matmul returns bf16: round to nearest even, then scale in fp32
        unsigned int bits = __float_as_uint(dot);
        bits += 0x7fffu + ((bits >> 16) & 1u);
        logit[u] = __uint_as_float(bits & 0xffff0000u) * scale;
        mx = fmaxf(mx, logit[u]);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
    float p[MAXT], sum = 0.f;
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
      p[u] = (lane + 64 * u < T) ? expf(logit[u] - mx) : 0.f;
      sum += p[u];
    }
    sum = wave_sum(sum);
    const float inv = 1.f / (sum * (float)H);
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
      const int t = lane + 64 * u;
      if (t < T) acc[t] += p[u] * inv;
    }
  }
  __syncthreads();
  for (int t = 1 + tid; t < T; t += blockDim.x) {
    float s = 0.f;
    for (int w = 0; w < nw; ++w) s += s_acc[(size_t)w * T + t];
    out[(size_t)blockIdx.x * (T - 1) + t - 1] = s;
  }
}

}  // namespace basd

extern "C" int basd_cls_importance_bf16(const void* qkv, int B, int T, int H, int hd, float scale, float* out,
                                        void* stream) {
  using namespace basd;
  if (B <= 0) return BASD_OK;
  if (T < 2 || T > 320 || H < 1 || (hd != 32 && hd != 64 && hd != 80))
    return fail(BASD_ERR_SHAPE, "cls_importance: T=%d H=%d hd=%d unsupported (2 <= T <= 320, hd 32|64|80)", T, H, hd);
  const size_t lds = (size_t)4 * T * sizeof(float);
  const unsigned short* p = (const unsigned short*)qkv;
  if (hd == 64)
    hipLaunchKernelGGL(cls_importance_kernel<64>, dim3(B), dim3(256), lds, (hipStream_t)stream, p, T, H, scale, out);
  else if (hd == 80)
    hipLaunchKernelGGL(cls_importance_kernel<80>, dim3(B), dim3(256), lds, (hipStream_t)stream, p, T, H, scale, out);
  else
    hipLaunchKernelGGL(cls_importance_kernel<32>, dim3(B), dim3(256), lds, (hipStream_t)stream, p, T, H, scale, out);
  return check_launch("cls_importance");
}
