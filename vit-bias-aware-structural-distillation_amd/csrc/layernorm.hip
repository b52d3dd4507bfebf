// LayerNorm for the ViT blocks: bf16 activations in/out, fp32 gamma/beta, fp32 statistics.
//
// Under bf16 autocast torch upcasts layer_norm inputs to fp32 (a copy in, a copy out per call) and
// its ROCm kernel refuses bf16 activations with fp32 parameters; backward is three more kernels.
// Here forward is one pass (x read once, y written once, mean / rstd saved) and backward is one
// pass that also accumulates d gamma / d beta (per-lane column partials, one fp32 atomic per
// column per workgroup -- straight into the flat gradient buffer when the caller passes it).
// The result equals torch's fp32 layer_norm followed by the bf16 rounding the next Linear applies.
// The forward optionally fuses the residual add that precedes the norm in a pre-norm block
// (s = bf16(x + r) written out, y = LN(s)): one kernel instead of add + norm for the frozen teacher.
//
// Layout: a row is owned by a GROUP of G lanes of one wave (G = 32 for D <= 256, else 64; two
// rows per wave when G = 32) and every lane holds NCH chunks of 8 contiguous columns (one 16-byte
// load each: chunk = lane_in_group + c * G).  Row statistics are reduced inside the wave with DPP
// row operations and v_permlane16/32_swap: no LDS, no barrier in the row loop (the first version
// of this kernel reduced through LDS with four barriers per row and ran at 1.9 TB/s).
#include "basd_common.h"

namespace basd {

typedef unsigned int ln_u32x2 __attribute__((ext_vector_type(2)));

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
__device__ __forceinline__ float round_bf16(float a) {
  __hip_bfloat16 x = __float2bfloat16(a);
  return bf16_bits_to_f32(*reinterpret_cast<unsigned short*>(&x));
}

// sum over the G lanes of a row group, result in every lane of the group
template <int G>
__device__ __forceinline__ float group_allsum(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true));   // row_mirror
  ln_u32x2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(r.x) + __uint_as_float(r.y);                                                 // rows 0+1, 2+3
  if (G == 64) {
    r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
  }
  return v;
}

// FUSE_ADD: s = bf16(x + res) is written to `sum_out` and normalised; otherwise x is normalised.
// U rows per group are in flight at a time (all their loads are issued before the first reduction): with one row per
// group and iteration a CU holds ~18 KB of loads in flight and the narrow student rows (D = 192: 384 bytes) ran at
// 2.0 - 2.7 TB/s.
template <int G, int NCH, bool FUSE_ADD, int U>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const unsigned short* __restrict__ x,
                                                     const unsigned short* __restrict__ res,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     int64_t rows, int D, float eps, unsigned short* __restrict__ sum_out,
                                                     unsigned short* __restrict__ y, float* __restrict__ mean,
                                                     float* __restrict__ rstd, const float* __restrict__ row_scale,
                                                     int rows_per_scale) {
  constexpr int RPW = 64 / G;                       // rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lg = lane & (G - 1), sub = lane / G;
  const int nchunk = D >> 3;
  float g[NCH][8], b[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lg + c * G;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      g[c][i] = ch < nchunk ? gamma[ch * 8 + i] : 0.f;
      b[c][i] = ch < nchunk ? beta[ch * 8 + i] : 0.f;
    }
  }
  const float inv_d = 1.f / (float)D;
  const int64_t wstride = (int64_t)gridDim.x * 4 * RPW;
  // a lane leaves the loop when its FIRST row is out of range; rows u > 0 beyond the end are skipped lane by lane (the
  // cross-lane reductions only combine lanes of the same row group, which agree on r)
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * RPW + sub; r0 < rows; r0 += wstride * U) {
    uint4 xv[U][NCH], rv[U][NCH];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * wstride;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lg + c * G;
        xv[u][c] = make_uint4(0, 0, 0, 0);
        rv[u][c] = make_uint4(0, 0, 0, 0);
        if (ch < nchunk && r < rows) {
          xv[u][c] = *reinterpret_cast<const uint4*>(x + r * D + ch * 8);
          if (FUSE_ADD) rv[u][c] = *reinterpret_cast<const uint4*>(res + r * D + ch * 8);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * wstride;
      const bool live = r < rows;
      float f[NCH][8];
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lg + c * G;
        unpack8(xv[u][c], f[c]);
        if (FUSE_ADD && ch < nchunk && live) {
          float fr[8];
          unpack8(rv[u][c], fr);
          // stochastic depth of a trained block: the branch x is scaled per SAMPLE (keep mask / keep probability)
          const float sc = row_scale ? row_scale[r / rows_per_scale] : 1.f;
#pragma unroll
          for (int i = 0; i < 8; ++i) f[c][i] = round_bf16(fmaf(sc, f[c][i], fr[i]));
          uint4 o;
          o.x = pack2(f[c][0], f[c][1]); o.y = pack2(f[c][2], f[c][3]); o.z = pack2(f[c][4], f[c][5]); o.w = pack2(f[c][6], f[c][7]);
          *reinterpret_cast<uint4*>(sum_out + r * D + ch * 8) = o;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += f[c][i];
      }
      const float mu = group_allsum<G>(s) * inv_d;
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lg + c * G;
        if (ch < nchunk) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { const float d = f[c][i] - mu; q = fmaf(d, d, q); }   // two-pass variance
        }
      }
      const float rs = rsqrtf(group_allsum<G>(q) * inv_d + eps);
      if (live) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int ch = lg + c * G;
          if (ch < nchunk) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fmaf((f[c][i] - mu) * rs, g[c][i], b[c][i]);
            uint4 out;
            out.x = pack2(o[0], o[1]); out.y = pack2(o[2], o[3]); out.z = pack2(o[4], o[5]); out.w = pack2(o[6], o[7]);
            *reinterpret_cast<uint4*>(y + r * D + ch * 8) = out;
          }
        }
        if (lg == 0 && mean != nullptr) { mean[r] = mu; rstd[r] = rs; }
      }
    }
  }
}

template <int G, int NCH, int U>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const unsigned short* __restrict__ dy, const unsigned short* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, int64_t rows, int D,
                                                     unsigned short* __restrict__ dx, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, const unsigned short* __restrict__ dres,
                                                     unsigned short* __restrict__ dbranch,
                                                     const float* __restrict__ row_scale, int rows_per_scale) {
  extern __shared__ float acc[];                    // [4 waves][2][D] column partials
  constexpr int RPW = 64 / G;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lg = lane & (G - 1), sub = lane / G;
  const int nchunk = D >> 3;
  float g[NCH][8], dg[NCH][8], db[NCH][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = lg + c * G;
#pragma unroll
    for (int i = 0; i < 8; ++i) { g[c][i] = ch < nchunk ? gamma[ch * 8 + i] : 0.f; dg[c][i] = 0.f; db[c][i] = 0.f; }
  }
  const float inv_d = 1.f / (float)D;
  const int64_t wstride = (int64_t)gridDim.x * 4 * RPW;
  // U rows per group in flight (see ln_fwd_kernel)
  for (int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * RPW + sub; r0 < rows; r0 += wstride * U) {
    uint4 xv[U][NCH], yv[U][NCH], rv[U][NCH];
    float mus[U], rss[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * wstride;
      const bool live = r < rows;
      mus[u] = live ? mean[r] : 0.f;
      rss[u] = live ? rstd[r] : 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lg + c * G;
        xv[u][c] = make_uint4(0, 0, 0, 0);
        yv[u][c] = make_uint4(0, 0, 0, 0);
        rv[u][c] = make_uint4(0, 0, 0, 0);
        if (ch < nchunk && live) {
          xv[u][c] = *reinterpret_cast<const uint4*>(x + r * D + ch * 8);
          yv[u][c] = *reinterpret_cast<const uint4*>(dy + r * D + ch * 8);
          if (dres != nullptr) rv[u][c] = *reinterpret_cast<const uint4*>(dres + r * D + ch * 8);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t r = r0 + u * wstride;
      const bool live = r < rows;
      const float mu = mus[u], rs = rss[u];
      float xh[NCH][8], gg[NCH][8];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = lg + c * G;
        if (ch < nchunk && live) {
          float fx[8], fdy[8];
          unpack8(xv[u][c], fx);
          unpack8(yv[u][c], fdy);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            xh[c][i] = (fx[i] - mu) * rs;
            gg[c][i] = fdy[i] * g[c][i];
            s1 += gg[c][i];
            s2 = fmaf(gg[c][i], xh[c][i], s2);
            dg[c][i] = fmaf(fdy[i], xh[c][i], dg[c][i]);
            db[c][i] += fdy[i];
          }
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) { xh[c][i] = 0.f; gg[c][i] = 0.f; }
        }
      }
      const float m1 = group_allsum<G>(s1) * inv_d, m2 = group_allsum<G>(s2) * inv_d;
      if (live) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const int ch = lg + c * G;
          if (ch < nchunk) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = rs * (gg[c][i] - m1 - xh[c][i] * m2);
            if (dres != nullptr) {                      // gradient arriving through the residual path of a pre-norm block
              float fr[8];
              unpack8(rv[u][c], fr);
#pragma unroll
              for (int i = 0; i < 8; ++i) o[i] += fr[i];
            }
            uint4 out;
            out.x = pack2(o[0], o[1]); out.y = pack2(o[2], o[3]); out.z = pack2(o[4], o[5]); out.w = pack2(o[6], o[7]);
            *reinterpret_cast<uint4*>(dx + r * D + ch * 8) = out;
            if (dbranch != nullptr) {                   // gradient of the (stochastic-depth scaled) branch input
              const float sc = row_scale ? row_scale[r / rows_per_scale] : 1.f;
              out.x = pack2(sc * o[0], sc * o[1]); out.y = pack2(sc * o[2], sc * o[3]);
              out.z = pack2(sc * o[4], sc * o[5]); out.w = pack2(sc * o[6], sc * o[7]);
              *reinterpret_cast<uint4*>(dbranch + r * D + ch * 8) = out;
            }
          }
        }
      }
    }
  }
  // column partials: the two row groups of a wave (G = 32) combine through v_permlane32_swap, the four
  // waves through their own LDS rows (plain stores: LDS float atomics serialise per lane, 14 us of a
  // 42 us call), then one global atomic per column per workgroup
  if (dgamma != nullptr) {
    float* mine = acc + (size_t)wave * 2 * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int ch = lg + c * G;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float a = dg[c][i], b = db[c][i];
        if (G == 32) {
          ln_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
          a = __uint_as_float(r.x) + __uint_as_float(r.y);
          r = __builtin_amdgcn_permlane32_swap(__float_as_uint(b), __float_as_uint(b), false, false);
          b = __uint_as_float(r.x) + __uint_as_float(r.y);
        }
        if (ch < nchunk && sub == 0) { mine[ch * 8 + i] = a; mine[D + ch * 8 + i] = b; }
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += 256) {
      const float v = (acc[i] + acc[2 * D + i]) + (acc[4 * D + i] + acc[6 * D + i]);
      atomicAdd(i < D ? &dgamma[i] : &dbeta[i - D], v);
    }
  }
}

static int ln_grid(int64_t rows, int rows_per_wg, int cap = 2048) {
  int64_t g = (rows + rows_per_wg - 1) / rows_per_wg;
  if (g > cap) g = cap;
  return (int)(g < 1 ? 1 : g);
}

template <bool FUSE>
static int launch_ln_fwd(const void* x, const void* res, const float* gamma, const float* beta, int64_t rows, int D,
                         float eps, void* sum_out, void* y, float* mean, float* rstd, const float* row_scale,
                         int rows_per_scale, hipStream_t st) {
  const int nchunk = D / 8;
  const unsigned short* xp = (const unsigned short*)x;
  const unsigned short* rp = (const unsigned short*)res;
  unsigned short* sp = (unsigned short*)sum_out;
  unsigned short* yp = (unsigned short*)y;
#define BASD_LN_FWD(G, NCH, U)                                                                                  \
  hipLaunchKernelGGL((ln_fwd_kernel<G, NCH, FUSE, U>), dim3(ln_grid(rows, 4 * (64 / G) * U)), dim3(256), 0, st, xp, rp, \
                     gamma, beta, rows, D, eps, sp, yp, mean, rstd, row_scale, rows_per_scale)
  if (nchunk <= 32) BASD_LN_FWD(32, 1, 4);
  else if (nchunk <= 64) BASD_LN_FWD(64, 1, 4);
  else if (nchunk <= 128) BASD_LN_FWD(64, 2, 2);
  else if (nchunk <= 192) BASD_LN_FWD(64, 3, 1);
  else BASD_LN_FWD(64, 4, 1);
#undef BASD_LN_FWD
  return check_launch(FUSE ? "add_layernorm_fwd_bf16" : "layernorm_fwd_bf16");
}

}  // namespace basd

extern "C" int basd_layernorm_fwd_bf16(const void* x, const float* gamma, const float* beta, int64_t rows, int D,
                                       float eps, void* y, float* mean, float* rstd, void* stream) {
  using namespace basd;
  if (rows <= 0) return BASD_OK;
  if (D % 8 || D < 8 || D > 2048) return fail(BASD_ERR_SHAPE, "layernorm_fwd_bf16: D %% 8 != 0 or D > 2048 (%d)", D);
  return launch_ln_fwd<false>(x, nullptr, gamma, beta, rows, D, eps, nullptr, y, mean, rstd, nullptr, 1,
                              (hipStream_t)stream);
}

extern "C" int basd_add_layernorm_fwd_bf16(const void* x, const void* residual, const float* gamma, const float* beta,
                                           int64_t rows, int D, float eps, void* sum_out, void* y, float* mean,
                                           float* rstd, const float* row_scale, int rows_per_scale, void* stream) {
  using namespace basd;
  if (rows <= 0) return BASD_OK;
  if (D % 8 || D < 8 || D > 2048) return fail(BASD_ERR_SHAPE, "add_layernorm_fwd_bf16: D %% 8 != 0 or D > 2048 (%d)", D);
  if (residual == nullptr || sum_out == nullptr) return fail(BASD_ERR_SHAPE, "add_layernorm_fwd_bf16: null residual / sum_out");
  if (row_scale != nullptr && rows_per_scale < 1) return fail(BASD_ERR_SHAPE, "add_layernorm_fwd_bf16: rows_per_scale < 1");
  return launch_ln_fwd<true>(x, residual, gamma, beta, rows, D, eps, sum_out, y, mean, rstd, row_scale,
                             rows_per_scale < 1 ? 1 : rows_per_scale, (hipStream_t)stream);
}

extern "C" int basd_layernorm_bwd_bf16(const void* dy, const void* x, const float* gamma, const float* mean,
                                       const float* rstd, int64_t rows, int D, void* dx, float* dgamma, float* dbeta,
                                       const void* dres, void* dbranch, const float* row_scale, int rows_per_scale,
                                       void* stream) {
  using namespace basd;
  if (rows <= 0) return BASD_OK;
  if (D % 8 || D < 8 || D > 2048) return fail(BASD_ERR_SHAPE, "layernorm_bwd_bf16: D %% 8 != 0 or D > 2048 (%d)", D);
  const int nchunk = D / 8;
  const size_t lds = (size_t)4 * 2 * D * 4;
  hipStream_t st = (hipStream_t)stream;
  // every workgroup ends with one fp32 atomic per column into the same 2 D addresses, which the memory side
  // serialises (~20 ns per workgroup, measured): 512 workgroups balance that tail against load parallelism
  const int bwd_cap = 512;
#define BASD_LN_BWD(G, NCH, U)                                                                                    \
  hipLaunchKernelGGL((ln_bwd_kernel<G, NCH, U>), dim3(ln_grid(rows, 4 * (64 / G) * 4, bwd_cap)), dim3(256), lds, st, \
                     (const unsigned short*)dy, (const unsigned short*)x, gamma, mean, rstd, rows, D,               \
                     (unsigned short*)dx, dgamma, dbeta, (const unsigned short*)dres, (unsigned short*)dbranch,      \
                     row_scale, rows_per_scale < 1 ? 1 : rows_per_scale)
  if (nchunk <= 32) BASD_LN_BWD(32, 1, 4);
  else if (nchunk <= 64) BASD_LN_BWD(64, 1, 4);
  else if (nchunk <= 128) BASD_LN_BWD(64, 2, 2);
  else if (nchunk <= 192) BASD_LN_BWD(64, 3, 1);
  else BASD_LN_BWD(64, 4, 1);
#undef BASD_LN_BWD
  return check_launch("layernorm_bwd_bf16");
}
