// Fused tall-skinny statistics: z = X P^T (fp32 MFMA 16x16x4), gram += z^T z and
// colsum += 1^T z accumulated in fp64 (MFMA f64 16x16x4), one pass over X.
// Replaces, per layer, the projection GEMM done twice by the reference
// (src/losses/layer_selector.py:72 and :135), the covariance GEMM (:13) and the
// column mean (:35); the [M, D_s] matrix z is never written to HBM.
//
// Workgroup = 512 threads (8 waves), 64-row tiles, grid-stride; the fp64 Gram
// accumulators (lower-triangular 16x16 tiles) stay in registers across all the
// tiles of a workgroup and are added to HBM once (fp64 atomics, <= 256 WGs).
#include "basd_common.h"

namespace basd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64;       // rows per tile
constexpr int KC = 32;       // K chunk of the projection
constexpr int XS = KC + 2;   // LDS row stride of the X / P chunks (conflict-free fragment reads)
constexpr int MAXCT = 8;     // projection column tiles per wave (d_out <= 256)

__device__ __forceinline__ int nct_tiles(int d_out) { const int n = d_out >> 4; return n * (n + 1) / 2; }

template <typename T>
__device__ __forceinline__ float ld_as_f32(const T* p);
template <> __device__ __forceinline__ float ld_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_as_f32<unsigned short>(const unsigned short* p) {
  return bf16_bits_to_f32(*p);
}

template <typename T, int MAXGT>   // MAXGT = Gram tiles per wave: 10 (d_out <= 192) or 17 (<= 256)
__global__ __launch_bounds__(512) void token_gram_kernel(const T* __restrict__ x, int64_t rows, int d_in,
                                                         const float* __restrict__ proj, int d_out,
                                                         double* __restrict__ gram,
                                                         double* __restrict__ colsum) {
  extern __shared__ __align__(16) float sm[];
  const int ldz = d_out + 16;
  float* Xs = sm;                      // [TM][XS]
  float* Ps = Xs + TM * XS;            // [d_out][XS]
  float* Z = Ps + d_out * XS;          // [TM][ldz]
  __shared__ unsigned char s_it[136], s_jt[136];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nct = d_out >> 4;                  // 16-wide column tiles
  const int rt = wave & 3;                     // projection row tile of this wave
  const int ct0 = (wave >> 2) * ((nct + 1) >> 1);
  const int ct1 = (wave >> 2) ? nct : ((nct + 1) >> 1);
  const int ngt = nct * (nct + 1) / 2;         // lower-triangular Gram tiles

  for (int g = threadIdx.x; g < nct_tiles(d_out); g += 512) {
    int it = 0;
    while ((it + 1) * (it + 2) / 2 <= g) ++it;
    s_it[g] = (unsigned char)it;
    s_jt[g] = (unsigned char)(g - it * (it + 1) / 2);
  }
  __syncthreads();
  f64x4 gacc[MAXGT];
#pragma unroll
  for (int i = 0; i < MAXGT; ++i) gacc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  double csum = 0.0;

  const int64_t ntiles = (rows + TM - 1) / TM;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * TM;
    f32x4 zacc[MAXCT];
#pragma unroll
    for (int i = 0; i < MAXCT; ++i) zacc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < d_in; k0 += KC) {
      __syncthreads();
      // stage X chunk [TM x KC] and P chunk [d_out x KC]
      for (int i = tid; i < TM * KC; i += 512) {
        const int r = i / KC, k = i - r * KC;
        const int64_t gr = r0 + r;
        Xs[r * XS + k] = (gr < rows) ? ld_as_f32<T>(x + gr * d_in + k0 + k) : 0.f;
      }
      for (int i = tid; i < d_out * KC; i += 512) {
        const int r = i / KC, k = i - r * KC;
        Ps[r * XS + k] = proj[(size_t)r * d_in + k0 + k];
      }
      __syncthreads();
#pragma unroll 2
      for (int kk = 0; kk < KC / 4; ++kk) {
        const float a = Xs[(rt * 16 + (lane & 15)) * XS + kk * 4 + (lane >> 4)];
#pragma unroll
        for (int c = 0; c < MAXCT; ++c) {
          const int ct = ct0 + c;
          if (ct < ct1) {
            const float b = Ps[(ct * 16 + (lane & 15)) * XS + kk * 4 + (lane >> 4)];
            zacc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, zacc[c], 0, 0, 0);
          }
        }
      }
    }
    // z tile -> LDS (C/D layout: col = lane & 15, row = (lane >> 4) * 4 + reg)
#pragma unroll
    for (int c = 0; c < MAXCT; ++c) {
      const int ct = ct0 + c;
      if (ct < ct1) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          Z[(rt * 16 + (lane >> 4) * 4 + reg) * ldz + ct * 16 + (lane & 15)] = zacc[c][reg];
      }
    }
    __syncthreads();
    // column sums (rows beyond `rows` were staged as zeros)
    if (tid < d_out) {
      double s = 0.0;
      for (int r = 0; r < TM; ++r) s += (double)Z[r * ldz + tid];
      csum += s;
    }
    // Gram: lower-triangular tiles, round-robin over the 8 waves
#pragma unroll
    for (int gi = 0; gi < MAXGT; ++gi) {
      const int g = wave + gi * 8;
      if (g < ngt) {
        const int it = s_it[g], jt = s_jt[g];   // jt <= it
#pragma unroll 2
        for (int kk = 0; kk < TM / 4; ++kk) {
          const float* zr = Z + (kk * 4 + (lane >> 4)) * ldz + (lane & 15);
          const double a = (double)zr[it * 16];
          const double b = (double)zr[jt * 16];
          gacc[gi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, gacc[gi], 0, 0, 0);
        }
      }
    }
  }
  // flush (f64 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg)
#pragma unroll
  for (int gi = 0; gi < MAXGT; ++gi) {
    const int g = wave + gi * 8;
    if (g < ngt) {
      const int it = s_it[g], jt = s_jt[g];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = it * 16 + (lane >> 4) + 4 * reg;
        const int j = jt * 16 + (lane & 15);
        const double v = gacc[gi][reg];
        atomicAdd(&gram[(size_t)i * d_out + j], v);
        if (it != jt) atomicAdd(&gram[(size_t)j * d_out + i], v);
      }
    }
  }
  if (tid < d_out) atomicAdd(&colsum[tid], csum);
}

}  // namespace basd

extern "C" int basd_token_gram(const void* x, int x_dtype, int64_t rows, int d_in, const float* proj,
                               int d_out, double* gram, double* colsum, void* stream) {
  using namespace basd;
  if (rows <= 0) return BASD_OK;
  if (d_out < 16 || d_out > 256 || d_out % 16 || d_in < KC || d_in % KC)
    return fail(BASD_ERR_SHAPE, "token_gram: need d_out %% 16 == 0, 16 <= d_out <= 256, d_in %% %d == 0 (got %d, %d)",
                KC, d_out, d_in);
  const size_t lds = (size_t)(TM * XS + d_out * XS + TM * (d_out + 16)) * 4;
  const int64_t ntiles = (rows + TM - 1) / TM;
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  hipStream_t st = (hipStream_t)stream;
#define BASD_TG_LAUNCH(T, G)                                                                        \
  do {                                                                                              \
    hipFuncSetAttribute((const void*)token_gram_kernel<T, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL((token_gram_kernel<T, G>), dim3(grid), dim3(512), lds, st, (const T*)x, rows, d_in, \
                       proj, d_out, gram, colsum);                                                  \
  } while (0)
  if (x_dtype == BASD_DTYPE_F32) {
    if (d_out <= 192) BASD_TG_LAUNCH(float, 10); else BASD_TG_LAUNCH(float, 17);
  } else if (x_dtype == BASD_DTYPE_BF16) {
    if (d_out <= 192) BASD_TG_LAUNCH(unsigned short, 10); else BASD_TG_LAUNCH(unsigned short, 17);
  } else {
    return fail(BASD_ERR_DTYPE, "token_gram: dtype %d", x_dtype);
  }
#undef BASD_TG_LAUNCH
  return check_launch("token_gram");
}
