// Fused tall-skinny statistics: z = X P^T (fp32 MFMA 16x16x4), gram += z^T z and
// colsum += 1^T z accumulated in fp64 (MFMA f64 16x16x4), one pass over X.
// Replaces, per layer, the projection GEMM done twice by the reference
// (src/losses/layer_selector.py:72 and :135), the covariance GEMM (:13) and the
// column mean (:35); the [M, D_s] matrix z is never written to HBM.
//
// Workgroup = 512 threads (8 waves), 64-row tiles, grid-stride; the fp64 Gram
// accumulators (lower-triangular 16x16 tiles) stay in registers across all the
// tiles of a workgroup and are added to HBM once (fp64 atomics, <= 256 WGs).
#include <stdlib.h>
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
                                                         int rows_per_batch, int64_t batch_stride,
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
        // row gr of the logical [rows, d_in] matrix lives at batch gr / N, token gr % N of a strided view
        const int64_t gb = gr / rows_per_batch;
        Xs[r * XS + k] = (gr < rows) ? ld_as_f32<T>(x + gb * batch_stride + (gr - gb * rows_per_batch) * d_in + k0 + k) : 0.f;
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
      }
    }
  }
  if (tid < d_out) atomicAdd(&colsum[tid], csum);
}

}  // namespace basd

extern "C" int basd_token_gram(const void* x, int x_dtype, int64_t rows, int d_in, int rows_per_batch,
                               int64_t batch_stride, const float* proj, int d_out, double* gram, double* colsum,
                               void* stream) {
  using namespace basd;
  if (rows <= 0) return BASD_OK;
  if (rows_per_batch < 1) return fail(BASD_ERR_SHAPE, "token_gram: rows_per_batch < 1");
  if (d_out < 16 || d_out > 256 || d_out % 16 || d_in < KC || d_in % KC)
    return fail(BASD_ERR_SHAPE, "token_gram: need d_out %% 16 == 0, 16 <= d_out <= 256, d_in %% %d == 0 (got %d, %d)",
                KC, d_out, d_in);
  const size_t lds = (size_t)(TM * XS + d_out * XS + TM * (d_out + 16)) * 4;
  const int64_t ntiles = (rows + TM - 1) / TM;
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  hipStream_t st = (hipStream_t)stream;
#define BASD_TG_LAUNCH(T, G)                                                                        \
  do {                                                                                              \
    allow_full_lds((const void*)token_gram_kernel<T, G>);                                           \
    hipLaunchKernelGGL((token_gram_kernel<T, G>), dim3(grid), dim3(512), lds, st, (const T*)x, rows, d_in, \
                       rows_per_batch, batch_stride, proj, d_out, gram, colsum);                                                  \
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

// --------------------------------------------------------------------------------------------
// bf16-input fast path.  The tokens a ViT emits under bf16 autocast are EXACT bf16 values; the
// fp32 projection P is split into three bf16 terms P = Ph + Pm + Pl (24 significant bits), so
//     z = X Ph + X Pm + X Pl
// is evaluated with bf16 MFMAs (v_mfma_f32_16x16x32_bf16, exact products, fp32 accumulate) at
// 16x the fp32-MFMA rate for 3x the instructions.  Same accuracy class as the fp32 path
// (|dP/P| <= 2^-24), checked against fp64 in tests/test_kernels_gpu.py.
//
// Workgroup = 256 threads = 4 waves (1 wave per SIMD: the whole 512-register file), tile =
// 128 rows (32 per wave).  Per K chunk of 32 the three P splits are staged once in LDS
// ([split][col][32 k], 80-byte rows) and every B fragment read serves two row groups.  The z
// tile then goes to LDS (overlaying the P buffer) for the fp64 Gram MFMAs, whose accumulators
// (20 lower-triangular tiles per wave) persist in registers across all the tiles of a workgroup.
namespace basd {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TM2 = 128;          // rows per workgroup tile
constexpr int KC2 = 32;           // K chunk
constexpr int PROW = 80;          // bytes per (split, col) row of the staged P chunk (64 + 16 pad)

template <int NCT>   // NCT = d_out / 16 (12 for d_out = 192)
__global__ __launch_bounds__(256) void token_gram_bf16x3_kernel(const unsigned short* __restrict__ x, int64_t rows,
                                                                int d_in, int rows_per_batch, int64_t batch_stride,
                                                                const unsigned short* __restrict__ psplit,
                                                                double* __restrict__ gram,
                                                                double* __restrict__ colsum) {
  constexpr int D_OUT = NCT * 16;
  constexpr int LDZ = D_OUT + 16;
  constexpr int NGT = NCT * (NCT + 1) / 2;
  constexpr int GT_PER_WAVE = (NGT + 3) / 4;
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* Pl = smem;                                   // [3][D_OUT][PROW] bytes   (projection phase)
  float* Z = reinterpret_cast<float*>(smem);                  // [TM2][LDZ] floats        (Gram phase)
  __shared__ unsigned char s_it[NGT], s_jt[NGT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int g = tid; g < NGT; g += 256) {
    int it = 0;
    while ((it + 1) * (it + 2) / 2 <= g) ++it;
    s_it[g] = (unsigned char)it;
    s_jt[g] = (unsigned char)(g - it * (it + 1) / 2);
  }
  f64x4 gacc[GT_PER_WAVE];
#pragma unroll
  for (int i = 0; i < GT_PER_WAVE; ++i) gacc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
  double csum = 0.0;
  const size_t split_stride = (size_t)D_OUT * d_in;

  const int64_t ntiles = (rows + TM2 - 1) / TM2;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t r0 = tile * TM2 + wave * 32;
    f32x4 zacc[2][NCT];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int c = 0; c < NCT; ++c) zacc[g][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int64_t ra = r0 + (lane & 15), rb = ra + 16;
    const int64_t ba = ra / rows_per_batch, bb = rb / rows_per_batch;      // strided [B, N, D] views
    const unsigned short* xa = x + ba * batch_stride + (ra - ba * rows_per_batch) * d_in + 8 * (lane >> 4);
    const unsigned short* xb = x + bb * batch_stride + (rb - bb * rows_per_batch) * d_in + 8 * (lane >> 4);
    // software pipeline over K chunks: the global loads of chunk c+1 (three P splits -> registers,
    // A fragments) are issued before the MFMAs of chunk c and land in the OTHER LDS buffer after
    // them; one barrier per chunk.  (The unpipelined loop spent ~5 us per chunk waiting on L2.)
    constexpr int PVEC = (3 * D_OUT * 4 + 255) / 256;          // uint4 per thread per chunk
    uint4 pre[PVEC];
    bf16x8 a0n, a1n;
    auto fetch = [&](int k0) {
#pragma unroll
      for (int v = 0; v < PVEC; ++v) {
        const int e = tid + 256 * v;
        const int q = e & 3, rc = e >> 2;
        const int sp = rc / D_OUT, col = rc - sp * D_OUT;
        pre[v] = (rc < 3 * D_OUT)
                     ? *reinterpret_cast<const uint4*>(psplit + sp * split_stride + (size_t)col * d_in + k0 + 8 * q)
                     : make_uint4(0, 0, 0, 0);
      }
      a0n = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
      a1n = a0n;
      if (ra < rows) a0n = *reinterpret_cast<const bf16x8*>(xa + k0);
      if (rb < rows) a1n = *reinterpret_cast<const bf16x8*>(xb + k0);
    };
    auto stash = [&](unsigned char* buf) {
#pragma unroll
      for (int v = 0; v < PVEC; ++v) {
        const int e = tid + 256 * v;
        const int q = e & 3, rc = e >> 2;
        if (rc < 3 * D_OUT) *reinterpret_cast<uint4*>(buf + (size_t)rc * PROW + 16 * q) = pre[v];
      }
    };
    constexpr size_t PBUF = (size_t)3 * D_OUT * PROW;
    __syncthreads();                           // the Gram phase of the previous tile no longer reads Z
    fetch(0);
    stash(Pl);
    bf16x8 a0 = a0n, a1 = a1n;
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < d_in; k0 += KC2) {
      const bool more = k0 + KC2 < d_in;
      if (more) fetch(k0 + KC2);
      const unsigned char* Pc = Pl + cur * PBUF;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(
              Pc + (size_t)(s * D_OUT + c * 16 + (lane & 15)) * PROW + 16 * (lane >> 4));
          zacc[0][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b, zacc[0][c], 0, 0, 0);
          zacc[1][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b, zacc[1][c], 0, 0, 0);
        }
      }
      if (more) {
        stash(Pl + (cur ^ 1) * PBUF);
        a0 = a0n; a1 = a1n;
      }
      cur ^= 1;
      __syncthreads();
    }
    __syncthreads();                          // all fragment reads done before Z overlays the P buffer
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          Z[(wave * 32 + g * 16 + (lane >> 4) * 4 + reg) * LDZ + c * 16 + (lane & 15)] = zacc[g][c][reg];
    __syncthreads();
    // ---- Gram of the tile.  With s = the tile's exact column sums (fp64), n its rows and mu ANY fp32 vector,
    //     sum_rows z z^T = G_c + mu w^T + w mu^T,   G_c = sum (z - mu)(z - mu)^T,   w = s - (n / 2) mu.
    // mu = the fp32-rounded column mean keeps the entries of z - mu small (and the subtraction exact): G_c runs on the
    // fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 sums over 128 rows), the two rank-1 terms are
    // one fp64 MFMA per Gram tile, everything is added up in fp64.  The caller forms the centred Gram as unc - s s^T / m:
    // with an fp32 Gram of the UNcentred tokens that subtraction cancels the leading digits whenever the token mean
    // dominates their spread; the all-fp64 Gram before it ran at half the MFMA rate and made this phase (41 k cycles per
    // tile) longer than the three-split projection (28 k).
    const int64_t rows_left = rows - tile * TM2;
    const int nvalid = rows_left < TM2 ? (int)rows_left : TM2;
    float* mu = Z + (size_t)TM2 * LDZ;                     // [D_OUT] fp32 tile mean (behind the z tile)
    double* wv = reinterpret_cast<double*>(mu + D_OUT);    // [D_OUT] fp64 w
    if (tid < D_OUT) {
      double sacc = 0.0;
      for (int r = 0; r < TM2; ++r) sacc += (double)Z[r * LDZ + tid];      // rows beyond `rows` are zeros
      const float m_ = (float)(sacc / (double)nvalid);
      mu[tid] = m_;
      wv[tid] = sacc - 0.5 * (double)nvalid * (double)m_;
      csum += sacc;
    }
    __syncthreads();
    for (int e = tid; e < nvalid * D_OUT; e += 256) {
      const int r = e / D_OUT, c = e - r * D_OUT;
      Z[r * LDZ + c] -= mu[c];
    }
    __syncthreads();
#pragma unroll
    for (int gi = 0; gi < GT_PER_WAVE; ++gi) {
      const int g = wave + gi * 4;
      if (g < NGT) {
        const int it = s_it[g], jt = s_jt[g];
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int kk = 0; kk < TM2 / 4; ++kk) {
          const float* zr = Z + (kk * 4 + (lane >> 4)) * LDZ + (lane & 15);
          t = __builtin_amdgcn_mfma_f32_16x16x4f32(zr[it * 16], zr[jt * 16], t, 0, 0, 0);
        }
        // the fp32 tile holds rows 4 q + reg (q = lane >> 4), the fp64 accumulator rows q + 4 reg: a 4 x 4 transpose
        // between lane group and register -- two register <-> lane butterflies
        {
          typedef unsigned int tg_u32x2 __attribute__((ext_vector_type(2)));
          tg_u32x2 a02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[0]), __float_as_uint(t[2]), false, false);
          tg_u32x2 a13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[1]), __float_as_uint(t[3]), false, false);
          tg_u32x2 b01 = __builtin_amdgcn_permlane16_swap(a02[0], a13[0], false, false);
          tg_u32x2 b23 = __builtin_amdgcn_permlane16_swap(a02[1], a13[1], false, false);
          gacc[gi][0] += (double)__uint_as_float(b01[0]);
          gacc[gi][1] += (double)__uint_as_float(b01[1]);
          gacc[gi][2] += (double)__uint_as_float(b23[0]);
          gacc[gi][3] += (double)__uint_as_float(b23[1]);
        }
        // + mu w^T + w mu^T: k slot 0 carries (mu_i, w_j), k slot 1 (w_i, mu_j), the other two are zero
        const int ks = lane >> 4;
        const double mi = (double)mu[it * 16 + (lane & 15)], mj = (double)mu[jt * 16 + (lane & 15)];
        const double wi = wv[it * 16 + (lane & 15)], wj = wv[jt * 16 + (lane & 15)];
        const double ai = ks == 0 ? mi : ks == 1 ? wi : 0.0;
        const double bj = ks == 0 ? wj : ks == 1 ? mj : 0.0;
        gacc[gi] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bj, gacc[gi], 0, 0, 0);
      }
    }
  }
  // Measured / tried and rejected (round 4).
  // (a) Two K chunks in flight (a second register set for the three P splits
  // and the A fragments, loop unrolled by two): the kernel already fills 256 VGPRs + 154 AGPRs (96 projection + 160 Gram
  // accumulators); the 44 extra registers make the allocator spill 237 -- not built further.  The projection loop stays
  // bound by the L2 round trip of a 36 KiB chunk that has only the 72 MFMAs of ONE chunk (0.55 us) to hide under:
  // 174 us per launch at 50 176 x 768 -> 192 against ~60 us of matrix work.  What would fix it: the P chunk through
  // LDS-DMA (no staging registers), which needs the unpadded, source-swizzled LDS image of gemm_bf16.hip.
  // (b) the partial tiles as plain stores into a workspace + a second launch that adds
  // the 256 partial sets up (as the weight-gradient kernel does): 221 vs 174 us at 50 176 x 768 -> 192, 143 vs 100 us at
  // d_in = 192 -- the fp64 atomics below are NOT what this kernel waits for -- and a scratch buffer shared by the
  // teacher's and the student's launches chains the two streams of the step together.
#pragma unroll
  for (int gi = 0; gi < GT_PER_WAVE; ++gi) {
    const int g = wave + gi * 4;
    if (g < NGT) {
      const int it = s_it[g], jt = s_jt[g];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = it * 16 + (lane >> 4) + 4 * reg;
        const int j = jt * 16 + (lane & 15);
        const double v = gacc[gi][reg];
        atomicAdd(&gram[(size_t)i * D_OUT + j], v);
      }
    }
  }
  if (tid < D_OUT) atomicAdd(&colsum[tid], csum);
}

// 256-row tiles (round 4).  The projection loop above is bound by the L2 round trip of a P chunk, whatever the number of
// MFMAs under it, and 50 176 rows are 392 tiles of 128 on 256 workgroups: 136 of them run two tiles, i.e. two times 24
// round trips.  Here a wave carries FOUR row groups of 16 (tile = 256 rows: 196 workgroups, one tile each, one set of
// round trips; every B fragment read serves four row groups), and the Gram phase runs twice on 128-row halves of the
// tile (the z half tile + its means fit the 106 KiB the P buffers leave; the other half waits in the accumulators).
// The Gram accumulators only live from the first Gram pass to the end of the tile (flushed with the fp64 atomics per
// tile), so the projection has the register file to itself.
template <int NCT>   // NCT = d_out / 16 (12 for d_out = 192)
__global__ __launch_bounds__(256) void token_gram_bf16x3_t256_kernel(const unsigned short* __restrict__ x, int64_t rows,
                                                                int d_in, int rows_per_batch, int64_t batch_stride,
                                                                const unsigned short* __restrict__ psplit,
                                                                double* __restrict__ gram,
                                                                double* __restrict__ colsum) {
  constexpr int D_OUT = NCT * 16;
  constexpr int LDZ = D_OUT + 16;
  constexpr int NGT = NCT * (NCT + 1) / 2;
  constexpr int GT_PER_WAVE = (NGT + 3) / 4;
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned char* Pl = smem;                                   // [3][D_OUT][PROW] bytes   (projection phase)
  float* Z = reinterpret_cast<float*>(smem);                  // [TM2][LDZ] floats        (Gram phase)
  __shared__ unsigned char s_it[NGT], s_jt[NGT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int g = tid; g < NGT; g += 256) {
    int it = 0;
    while ((it + 1) * (it + 2) / 2 <= g) ++it;
    s_it[g] = (unsigned char)it;
    s_jt[g] = (unsigned char)(g - it * (it + 1) / 2);
  }
  double csum = 0.0;
  const size_t split_stride = (size_t)D_OUT * d_in;
  __syncthreads();                              // s_it / s_jt

  const int64_t ntiles = (rows + 2 * TM2 - 1) / (2 * TM2);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // row group g of wave w: rows tile * 256 + (g >> 1) * 128 + w * 32 + (g & 1) * 16 + (lane & 15): the groups {0, 1}
    // of the four waves are the first 128 rows of the tile (Gram pass 0), the groups {2, 3} the second 128
    f32x4 zacc[4][NCT];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int c = 0; c < NCT; ++c) zacc[g][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned short* xr[4];
    bool rok[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t r = tile * (2 * TM2) + (g >> 1) * TM2 + wave * 32 + (g & 1) * 16 + (lane & 15);
      rok[g] = r < rows;
      const int64_t rr = rok[g] ? r : 0;
      const int64_t bq = rr / rows_per_batch;                              // strided [B, N, D] views
      xr[g] = x + bq * batch_stride + (rr - bq * rows_per_batch) * d_in + 8 * (lane >> 4);
    }
    // software pipeline over K chunks: the global loads of chunk c+1 (three P splits -> registers,
    // A fragments) are issued before the MFMAs of chunk c and land in the OTHER LDS buffer after
    // them; one barrier per chunk.  (The unpipelined loop spent ~5 us per chunk waiting on L2.)
    constexpr int PVEC = (3 * D_OUT * 4 + 255) / 256;          // uint4 per thread per chunk
    // (Measured and rejected: a second register set so that two chunks are in flight -- the loop unrolled by two spills
    // ~100 scratch operations per chunk pair even with the Gram accumulators out of the way: 393 vs 150 us.)
    uint4 pre[PVEC];
    bf16x8 an[4];
    auto fetch = [&](int k0) {
#pragma unroll
      for (int v = 0; v < PVEC; ++v) {
        const int e = tid + 256 * v;
        const int q = e & 3, rc = e >> 2;
        const int sp = rc / D_OUT, col = rc - sp * D_OUT;
        pre[v] = (rc < 3 * D_OUT)
                     ? *reinterpret_cast<const uint4*>(psplit + sp * split_stride + (size_t)col * d_in + k0 + 8 * q)
                     : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        an[g] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
        if (rok[g]) an[g] = *reinterpret_cast<const bf16x8*>(xr[g] + k0);
      }
    };
    auto stash = [&](unsigned char* buf) {
#pragma unroll
      for (int v = 0; v < PVEC; ++v) {
        const int e = tid + 256 * v;
        const int q = e & 3, rc = e >> 2;
        if (rc < 3 * D_OUT) *reinterpret_cast<uint4*>(buf + (size_t)rc * PROW + 16 * q) = pre[v];
      }
    };
    constexpr size_t PBUF = (size_t)3 * D_OUT * PROW;
    __syncthreads();                           // the Gram phase of the previous tile no longer reads Z
    fetch(0);
    stash(Pl);
    bf16x8 ac[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) ac[g] = an[g];
    __syncthreads();
    int cur = 0;
    for (int k0 = 0; k0 < d_in; k0 += KC2) {
      const bool more = k0 + KC2 < d_in;
      if (more) fetch(k0 + KC2);
      const unsigned char* Pc = Pl + cur * PBUF;
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) {
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
          const bf16x8 b = *reinterpret_cast<const bf16x8*>(
              Pc + (size_t)(s_ * D_OUT + c * 16 + (lane & 15)) * PROW + 16 * (lane >> 4));
#pragma unroll
          for (int g = 0; g < 4; ++g) zacc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ac[g], b, zacc[g][c], 0, 0, 0);
        }
      }
      if (more) {
        stash(Pl + (cur ^ 1) * PBUF);
#pragma unroll
        for (int g = 0; g < 4; ++g) ac[g] = an[g];
      }
      cur ^= 1;
      __syncthreads();
    }
    f64x4 gacc[GT_PER_WAVE];
#pragma unroll
    for (int i = 0; i < GT_PER_WAVE; ++i) gacc[i] = (f64x4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
    const int64_t rows_left = rows - (tile * (2 * TM2) + pass * TM2);
    if (rows_left <= 0) break;                 // uniform
    __syncthreads();                          // all fragment reads (pass 0) / Gram reads of pass 0 (pass 1) are done
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int c = 0; c < NCT; ++c)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg)
          Z[(wave * 32 + g * 16 + (lane >> 4) * 4 + reg) * LDZ + c * 16 + (lane & 15)] = zacc[2 * pass + g][c][reg];
    __syncthreads();
    // ---- Gram of the tile.  With s = the tile's exact column sums (fp64), n its rows and mu ANY fp32 vector,
    //     sum_rows z z^T = G_c + mu w^T + w mu^T,   G_c = sum (z - mu)(z - mu)^T,   w = s - (n / 2) mu.
    // mu = the fp32-rounded column mean keeps the entries of z - mu small (and the subtraction exact): G_c runs on the
    // fp32 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 sums over 128 rows), the two rank-1 terms are
    // one fp64 MFMA per Gram tile, everything is added up in fp64.  The caller forms the centred Gram as unc - s s^T / m:
    // with an fp32 Gram of the UNcentred tokens that subtraction cancels the leading digits whenever the token mean
    // dominates their spread; the all-fp64 Gram before it ran at half the MFMA rate and made this phase (41 k cycles per
    // tile) longer than the three-split projection (28 k).
    const int nvalid = rows_left < TM2 ? (int)rows_left : TM2;
    float* mu = Z + (size_t)TM2 * LDZ;                     // [D_OUT] fp32 tile mean (behind the z tile)
    double* wv = reinterpret_cast<double*>(mu + D_OUT);    // [D_OUT] fp64 w
    if (tid < D_OUT) {
      double sacc = 0.0;
      for (int r = 0; r < TM2; ++r) sacc += (double)Z[r * LDZ + tid];      // rows beyond `rows` are zeros
      const float m_ = (float)(sacc / (double)nvalid);
      mu[tid] = m_;
      wv[tid] = sacc - 0.5 * (double)nvalid * (double)m_;
      csum += sacc;
    }
    __syncthreads();
    for (int e = tid; e < nvalid * D_OUT; e += 256) {
      const int r = e / D_OUT, c = e - r * D_OUT;
      Z[r * LDZ + c] -= mu[c];
    }
    __syncthreads();
#pragma unroll
    for (int gi = 0; gi < GT_PER_WAVE; ++gi) {
      const int g = wave + gi * 4;
      if (g < NGT) {
        const int it = s_it[g], jt = s_jt[g];
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int kk = 0; kk < TM2 / 4; ++kk) {
          const float* zr = Z + (kk * 4 + (lane >> 4)) * LDZ + (lane & 15);
          t = __builtin_amdgcn_mfma_f32_16x16x4f32(zr[it * 16], zr[jt * 16], t, 0, 0, 0);
        }
        // the fp32 tile holds rows 4 q + reg (q = lane >> 4), the fp64 accumulator rows q + 4 reg: a 4 x 4 transpose
        // between lane group and register -- two register <-> lane butterflies
        {
          typedef unsigned int tg_u32x2 __attribute__((ext_vector_type(2)));
          tg_u32x2 a02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[0]), __float_as_uint(t[2]), false, false);
          tg_u32x2 a13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(t[1]), __float_as_uint(t[3]), false, false);
          tg_u32x2 b01 = __builtin_amdgcn_permlane16_swap(a02[0], a13[0], false, false);
          tg_u32x2 b23 = __builtin_amdgcn_permlane16_swap(a02[1], a13[1], false, false);
          gacc[gi][0] += (double)__uint_as_float(b01[0]);
          gacc[gi][1] += (double)__uint_as_float(b01[1]);
          gacc[gi][2] += (double)__uint_as_float(b23[0]);
          gacc[gi][3] += (double)__uint_as_float(b23[1]);
        }
        // + mu w^T + w mu^T: k slot 0 carries (mu_i, w_j), k slot 1 (w_i, mu_j), the other two are zero
        const int ks = lane >> 4;
        const double mi = (double)mu[it * 16 + (lane & 15)], mj = (double)mu[jt * 16 + (lane & 15)];
        const double wi = wv[it * 16 + (lane & 15)], wj = wv[jt * 16 + (lane & 15)];
        const double ai = ks == 0 ? mi : ks == 1 ? wi : 0.0;
        const double bj = ks == 0 ? wj : ks == 1 ? mj : 0.0;
        gacc[gi] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, bj, gacc[gi], 0, 0, 0);
      }
    }
    }   // pass
#pragma unroll
    for (int gi = 0; gi < GT_PER_WAVE; ++gi) {
      const int g = wave + gi * 4;
      if (g < NGT) {
        const int it = s_it[g], jt = s_jt[g];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int i = it * 16 + (lane >> 4) + 4 * reg;
          const int j = jt * 16 + (lane & 15);
          atomicAdd(&gram[(size_t)i * D_OUT + j], gacc[gi][reg]);
        }
      }
    }
  }
  if (tid < D_OUT) atomicAdd(&colsum[tid], csum);
}

template <int NCT>
static void launch_tg_bf16x3(const void* x, int64_t rows, int d_in, int rows_per_batch, int64_t batch_stride,
                             const void* psplit, double* gram, double* colsum, hipStream_t st) {
  constexpr int D_OUT = NCT * 16;
  const size_t p_bytes = (size_t)2 * 3 * D_OUT * PROW;      // double-buffered P chunk
  const size_t z_bytes = (size_t)TM2 * (D_OUT + 16) * 4 + (size_t)D_OUT * 12;     // z tile + fp32 mean + fp64 w
  const size_t lds = p_bytes > z_bytes ? p_bytes : z_bytes;
  const int64_t ntiles = (rows + TM2 - 1) / TM2;
  // 256-row tiles once every CU would get more than one 128-row tile (BASD_TOKEN_GRAM_T256=0: always 128, A/B timing)
  const char* env = getenv("BASD_TOKEN_GRAM_T256");
  if (ntiles > 256 && !(env && env[0] == '0')) {
    const int64_t nt2 = (rows + 2 * TM2 - 1) / (2 * TM2);
    const int grid2 = (int)(nt2 < 256 ? nt2 : 256);
    allow_full_lds((const void*)token_gram_bf16x3_t256_kernel<NCT>);
    hipLaunchKernelGGL((token_gram_bf16x3_t256_kernel<NCT>), dim3(grid2), dim3(256), lds, st, (const unsigned short*)x, rows,
                       d_in, rows_per_batch, batch_stride, (const unsigned short*)psplit, gram, colsum);
    return;
  }
  // (measured and rejected: one workgroup per 128-row tile, so that a side-stream launch holds a CU for one tile only --
  // pipelined c2 step 33.01 / 32.94 ms against 32.75 / 32.68 with at most 256 grid-striding workgroups)
  const int grid = (int)(ntiles < 256 ? ntiles : 256);
  allow_full_lds((const void*)token_gram_bf16x3_kernel<NCT>);
  hipLaunchKernelGGL((token_gram_bf16x3_kernel<NCT>), dim3(grid), dim3(256), lds, st, (const unsigned short*)x, rows,
                     d_in, rows_per_batch, batch_stride, (const unsigned short*)psplit, gram, colsum);
}

}  // namespace basd

extern "C" int basd_token_gram_bf16x3(const void* x, int64_t rows, int d_in, int rows_per_batch, int64_t batch_stride,
                                      const void* proj_split, int d_out, double* gram, double* colsum,
                                      void* stream) {
  using namespace basd;
  if (rows <= 0) return BASD_OK;
  if (rows_per_batch < 1) return fail(BASD_ERR_SHAPE, "token_gram_bf16x3: rows_per_batch < 1");
  if (d_in % KC2 || d_in < KC2) return fail(BASD_ERR_SHAPE, "token_gram_bf16x3: d_in %% 32 != 0 (%d)", d_in);
  hipStream_t st = (hipStream_t)stream;
  switch (d_out) {
    case 32: launch_tg_bf16x3<2>(x, rows, d_in, rows_per_batch, batch_stride, proj_split, gram, colsum, st); break;
    case 64: launch_tg_bf16x3<4>(x, rows, d_in, rows_per_batch, batch_stride, proj_split, gram, colsum, st); break;
    case 128: launch_tg_bf16x3<8>(x, rows, d_in, rows_per_batch, batch_stride, proj_split, gram, colsum, st); break;
    case 192: launch_tg_bf16x3<12>(x, rows, d_in, rows_per_batch, batch_stride, proj_split, gram, colsum, st); break;
    default: return fail(BASD_ERR_SHAPE, "token_gram_bf16x3: d_out %d not in {32, 64, 128, 192}", d_out);
  }
  return check_launch("token_gram_bf16x3");
}

// --------------------------------------------------------------------------------------------
// Gram of a tall fp32 matrix at student widths beyond the fused kernel's 256 columns (round 4; BASELINE c4 / c5:
// z = X P^T [rows, 384 | 768] is materialised by basd_gemm_bf16x3_f32, z^T z ran as a split-K batched fp64-MFMA GEMM:
// 1.5 ms per layer at c5, 54 ms per step).  Same arithmetic as the Gram phase of token_gram_bf16x3_kernel: per tile of
// 64 rows the column means are subtracted, the centred tile is multiplied on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32, exact products, fp32 sums over 64 rows), and the exact uncentred statistics are restored
// with the fp64 rank-1 terms  sum z z^T = G_c + mu w^T + w mu^T,  w = s - (n / 2) mu;  everything is added up in fp64.
//
// A workgroup (4 waves) owns a 128 x 128 block (I, J) of the lower block triangle and a slice of the rows; per row tile
// it stages the two 64 x 128 column panels in LDS (prefetched into registers one tile ahead), every thread sums one of
// the 256 staged columns (fp64), wave w multiplies the sub-tile rows {2 w, 2 w + 1} x all eight sub-tile columns (the
// means are subtracted on the fragment reads: no centring pass), and the 16 fp64 accumulator tiles per wave (128 VGPRs)
// persist over the whole row slice; one fp64 atomic per element and workgroup at the end (row slices: ~12 per block).
namespace basd {

constexpr int GW_TR = 64;             // rows per tile
constexpr int GW_BC = 128;            // columns per panel
constexpr int GW_LD = GW_BC + 16;     // floats per staged row: 144 = 16 (mod 32): the four k rows of a fragment read
                                      // fall on different halves of the 32 banks (132 made every read 2-way)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gram_wide_f32_kernel(const float* __restrict__ z, int64_t rows, int d, int nbc,
                                                            int rsplit, double* __restrict__ gram,
                                                            double* __restrict__ colsum) {
  extern __shared__ __align__(16) float gw_lds[];
  float* Pi = gw_lds;                                  // [GW_TR][GW_LD] panel I (output rows)
  float* Pj = Pi + GW_TR * GW_LD;                      // [GW_TR][GW_LD] panel J (output columns); == Pi on the diagonal
  float* mu = Pj + GW_TR * GW_LD;                      // [256] fp32 tile means (I then J)
  double* wv = reinterpret_cast<double*>(mu + 256);    // [256] fp64 w = s - (n / 2) mu
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // block (bi >= bj) of the lower block triangle and the row slice of this workgroup
  const int blk = blockIdx.x / rsplit, rs = blockIdx.x - blk * rsplit;
  int bi = 0;
  while ((bi + 1) * (bi + 2) / 2 <= blk) ++bi;
  const int bj = blk - bi * (bi + 1) / 2;
  const bool diag = bi == bj;
  const int ci0 = bi * GW_BC, cj0 = bj * GW_BC;
  const int64_t ntile = (rows + GW_TR - 1) / GW_TR;
  const int64_t t_lo = ntile * rs / rsplit, t_hi = ntile * (rs + 1) / rsplit;
  f64x4 acc[2][8];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = (f64x4){0.0, 0.0, 0.0, 0.0};
  double csum = 0.0;
  // staging: thread -> (row = tid >> 5 (+ 8 per pass), column quad = tid & 31) of a panel: 8 passes of 16-byte loads
  const int sr = tid >> 5, sq = tid & 31;
  float4 pre_i[8], pre_j[8];
  auto fetch = [&](int64_t tile) {
    const int64_t r0 = tile * GW_TR;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int64_t r = r0 + sr + 8 * p;
      const bool ok = r < rows;
      const float* src = z + (ok ? r : 0) * (int64_t)d;
      const bool oki = ok && ci0 + 4 * sq < d, okj = ok && cj0 + 4 * sq < d;
      pre_i[p] = oki ? *reinterpret_cast<const float4*>(src + ci0 + 4 * sq) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (!diag) pre_j[p] = okj ? *reinterpret_cast<const float4*>(src + cj0 + 4 * sq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  for (int64_t tile = t_lo; tile < t_hi; ++tile) {
    // two workgroups per CU: while this one waits for its panels (global -> registers -> LDS) the other one multiplies
    // (a register prefetch across the MFMA phase costs 64 VGPRs, i.e. the second workgroup)
    fetch(tile);
    __syncthreads();                                   // the previous tile's fragment reads are done
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      *reinterpret_cast<float4*>(Pi + (sr + 8 * p) * GW_LD + 4 * sq) = pre_i[p];
      if (!diag) *reinterpret_cast<float4*>(Pj + (sr + 8 * p) * GW_LD + 4 * sq) = pre_j[p];
    }
    __syncthreads();
    const int64_t left = rows - tile * GW_TR;
    const int nvalid = left < GW_TR ? (int)left : GW_TR;
    {
      // thread t < 128: column t of panel I, t >= 128: column t - 128 of panel J (the same panel on the diagonal)
      const float* col = (tid < 128 ? Pi : (diag ? Pi : Pj)) + (tid & 127);
      double s = 0.0;
#pragma unroll 8
      for (int r = 0; r < GW_TR; ++r) s += (double)col[r * GW_LD];
      const float m_ = (float)(s / (double)nvalid);
      mu[tid] = m_;
      wv[tid] = s - 0.5 * (double)nvalid * (double)m_;
      if (diag && tid < 128) csum += s;
    }
    __syncthreads();
    const float* Pjj = diag ? Pi : Pj;
    const int li = lane & 15, kq = lane >> 4;
    const bool partial = nvalid < GW_TR;               // uniform: only the last tile of the matrix
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int it = 2 * wave + a;                     // sub-tile row of the block
      const float mi = mu[it * 16 + li];
      float af[GW_TR / 4];
#pragma unroll
      for (int kk = 0; kk < GW_TR / 4; ++kk) {
        const int r = kk * 4 + kq;
        const float v = Pi[r * GW_LD + it * 16 + li] - mi;
        af[kk] = (partial && r >= nvalid) ? 0.f : v;
      }
      double mi_r[4], wi_r[4];                         // rank-1 terms of this lane's four rows of sub-tile row `it`
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) { mi_r[reg] = (double)mu[it * 16 + 4 * kq + reg]; wi_r[reg] = wv[it * 16 + 4 * kq + reg]; }
#pragma unroll
      for (int jp = 0; jp < 4; ++jp) {                 // sub-tile columns in pairs: two independent MFMA chains
        const int j0 = 2 * jp, j1 = 2 * jp + 1;
        if (diag && j0 > it) continue;                 // strictly upper sub-tiles of a diagonal block: never stored
        const bool use1 = !(diag && j1 > it);
        const float mj0 = mu[128 + j0 * 16 + li], mj1 = mu[128 + j1 * 16 + li];
        f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < GW_TR / 4; ++kk) {
          const int r = kk * 4 + kq;
          float b0 = Pjj[r * GW_LD + j0 * 16 + li] - mj0, b1 = Pjj[r * GW_LD + j1 * 16 + li] - mj1;
          if (partial && r >= nvalid) { b0 = 0.f; b1 = 0.f; }
          t0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk], b0, t0, 0, 0, 0);
          t1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[kk], b1, t1, 0, 0, 0);
        }
        // t[reg] = G_c[row 4 kq + reg of sub-tile it][column li of sub-tile j]; + mu_i w_j + w_i mu_j in fp64
        const double wj0 = wv[128 + j0 * 16 + li], wj1 = wv[128 + j1 * 16 + li];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          acc[a][j0][reg] += (double)t0[reg] + mi_r[reg] * wj0 + wi_r[reg] * (double)mj0;
          if (use1) acc[a][j1][reg] += (double)t1[reg] + mi_r[reg] * wj1 + wi_r[reg] * (double)mj1;
        }
      }
    }
  }
  // ---- add the workgroup's block into the caller's (zeroed / accumulating) Gram: lower 16 x 16 tiles only
  const int li = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int it = 2 * wave + a;
#pragma unroll
    for (int jt = 0; jt < 8; ++jt) {
      if (diag && jt > it) continue;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int gi = ci0 + it * 16 + 4 * kq + reg, gj = cj0 + jt * 16 + li;
        if (gi < d && gj < d) atomicAdd(&gram[(size_t)gi * d + gj], acc[a][jt][reg]);
      }
    }
  }
  if (diag && tid < 128 && ci0 + tid < d) atomicAdd(&colsum[ci0 + tid], csum);
}

}  // namespace basd

extern "C" int basd_gram_f32_centred(const float* z, int64_t rows, int d, double* gram, double* colsum, void* stream) {
  using namespace basd;
  if (rows <= 0) return BASD_OK;
  if (d < 16 || d % 16 || d > 4096 || (((uintptr_t)z) & 15))
    return fail(BASD_ERR_SHAPE, "gram_f32_centred: need d %% 16 == 0, 16 <= d <= 4096, 16-byte aligned z (d=%d)", d);
  const int nbc = (d + GW_BC - 1) / GW_BC;
  const int nblk = nbc * (nbc + 1) / 2;
  const int64_t ntile = (rows + GW_TR - 1) / GW_TR;
  int rsplit = 512 / nblk;                              // two workgroups per CU, ONE round (273 workgroups on 256 CUs at one
                                                        // per CU ran the launch twice as long as 252)
  if ((int64_t)rsplit > ntile / 8) rsplit = (int)(ntile / 8);   // >= 8 row tiles per slice: a slice ends in 16 k fp64 atomics
  if (rsplit < 1) rsplit = 1;
  const size_t lds = (size_t)2 * GW_TR * GW_LD * 4 + 256 * 4 + 256 * 8;
  allow_full_lds((const void*)gram_wide_f32_kernel);
  hipLaunchKernelGGL(gram_wide_f32_kernel, dim3(nblk * rsplit), dim3(256), lds, (hipStream_t)stream, z, rows, d, nbc,
                     rsplit, gram, colsum);
  return check_launch("gram_f32_centred");
}
