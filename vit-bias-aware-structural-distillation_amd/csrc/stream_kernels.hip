// HBM-streaming kernels of the BASD loss path:
//   mix_tokens        all E softmax-weighted mixes of the L teacher layers from ONE read of
//                     each layer (reference src/losses/layer_selector.py:110-112 re-reads the
//                     [L,B,N,D] stack E times and materialises a same-size temporary)
//   mix_grad_dots     d loss / d w[i,j] = <g_i, x_j> for all (i,j) from one read of each layer
//   procrustes_prep   resample + importance normalise + weighted centring + sqrt-weighting
//                     (src/losses/relational.py:29-46, src/losses/combined.py:9-14)
// Loads are 16 B per lane (8 bf16 / 4 fp32), grid-stride, <= 2048 workgroups.
#include <stdlib.h>
#include "basd_common.h"

namespace basd {

constexpr int kMaxE = 8;
constexpr int kMaxL = 64;

// the L layer pointers travel in the kernel argument buffer (512 B): no device-side table, no
// host-to-device copy per call (which would also break hipGraph capture)
struct LayerPtrs { const void* p[kMaxL]; };

template <typename T> struct Vec;
template <> struct Vec<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
};
template <> struct Vec<unsigned short> {   // bf16 bits
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const unsigned short* p, float (&v)[8]) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const unsigned int w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
};

template <typename T, int E>
__global__ __launch_bounds__(256) void mix_tokens_kernel(LayerPtrs layers, int L,
                                                         const float* __restrict__ w, int64_t nvec,
                                                         float* __restrict__ out, int64_t elems,
                                                         int64_t per_batch, int64_t batch_stride) {
  constexpr int N = Vec<T>::N;
  __shared__ float s_w[kMaxE * kMaxL];
  __shared__ const T* s_ptr[kMaxL];
  for (int i = threadIdx.x; i < E * L; i += blockDim.x) s_w[i] = w[i];
  for (int i = threadIdx.x; i < L; i += blockDim.x) s_ptr[i] = static_cast<const T*>(layers.p[i]);
  __syncthreads();
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec;
       v += (int64_t)gridDim.x * blockDim.x) {
    float acc[E][N];
#pragma unroll
    for (int i = 0; i < E; ++i)
#pragma unroll
      for (int k = 0; k < N; ++k) acc[i][k] = 0.f;
    // element v*N of the logical contiguous tensor inside a batch-strided source view
    const int64_t e0 = v * N, sb = e0 / per_batch;
    const int64_t src_off = sb * batch_stride + (e0 - sb * per_batch);
    for (int j = 0; j < L; ++j) {
      float x[N];
      Vec<T>::load(s_ptr[j] + src_off, x);
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const float wij = s_w[i * L + j];
#pragma unroll
        for (int k = 0; k < N; ++k) acc[i][k] = fmaf(wij, x[k], acc[i][k]);
      }
    }
#pragma unroll
    for (int i = 0; i < E; ++i) {
      float* o = out + (int64_t)i * elems + v * N;
#pragma unroll
      for (int k = 0; k < N; k += 4)
        *reinterpret_cast<float4*>(o + k) = make_float4(acc[i][k], acc[i][k + 1], acc[i][k + 2], acc[i][k + 3]);
    }
  }
}

template <typename T, int E>
__global__ __launch_bounds__(256) void mix_grad_dots_kernel(LayerPtrs layers, int L,
                                                            const float* __restrict__ g, int64_t nvec,
                                                            int64_t elems, double* __restrict__ dots,
                                                            int64_t per_batch, int64_t batch_stride) {
  constexpr int N = Vec<T>::N;
  __shared__ const T* s_ptr[kMaxL];
  __shared__ double s_acc[kMaxE * kMaxL];
  for (int i = threadIdx.x; i < L; i += blockDim.x) s_ptr[i] = static_cast<const T*>(layers.p[i]);
  for (int i = threadIdx.x; i < E * L; i += blockDim.x) s_acc[i] = 0.0;
  __syncthreads();
  // per-thread fp32 partials over a bounded number of vectors, then fp64
  for (int j0 = 0; j0 < L; j0 += 16) {
    const int jn = (L - j0) < 16 ? (L - j0) : 16;
    float acc[E][16];
#pragma unroll
    for (int i = 0; i < E; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec;
         v += (int64_t)gridDim.x * blockDim.x) {
      const int64_t e0 = v * N, sb = e0 / per_batch;
      const int64_t src_off = sb * batch_stride + (e0 - sb * per_batch);
      float gv[E][N];
#pragma unroll
      for (int i = 0; i < E; ++i) {
        const float* gp = g + (int64_t)i * elems + v * N;
#pragma unroll
        for (int k = 0; k < N; k += 4) {
          const float4 t = *reinterpret_cast<const float4*>(gp + k);
          gv[i][k] = t.x; gv[i][k + 1] = t.y; gv[i][k + 2] = t.z; gv[i][k + 3] = t.w;
        }
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < jn) {
          float x[N];
          Vec<T>::load(s_ptr[j0 + j] + src_off, x);
#pragma unroll
          for (int i = 0; i < E; ++i) {
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < N; ++k) d = fmaf(gv[i][k], x[k], d);
            acc[i][j] += d;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < E; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (j < jn) {
          const double s = wave_sum_d((double)acc[i][j]);
          if ((threadIdx.x & 63) == 0) atomicAdd(&s_acc[i * L + j0 + j], s);
        }
      }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < E * L; i += blockDim.x) atomicAdd(&dots[i], s_acc[i]);
}

// Vectorised version for D_s, D_t multiples of 4 (default): one workgroup of 1024 threads per
// sample = 4 token groups x 256 column quads; every thread streams 16-byte quads of its columns
// for every 4th token (the kernel above walks one column per thread, a dependent scalar load per
// token: 1.3 TB/s).  Same arithmetic per element; the weighted means are combined over the four
// token groups through LDS.
template <typename TS>
__device__ __forceinline__ float4 load_quad(const TS* p) {
  if constexpr (sizeof(TS) == 2) {
    const uint2 v = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                       __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
  } else {
    return *reinterpret_cast<const float4*>(p);
  }
}

template <typename TS>
__global__ __launch_bounds__(1024) void procrustes_prep_v4_kernel(
    const TS* __restrict__ s_all, const float* __restrict__ t_all, const float* __restrict__ imp_all,
    int N_s, int N_t, int D_s, int D_t, int64_t s_batch_stride, float* __restrict__ sw_all,
    float* __restrict__ tw_all, float* __restrict__ a_all, float* __restrict__ tr_all) {
  extern __shared__ __align__(16) float sm[];
  const int D = D_s + D_t;
  float* s_a = sm;                                   // [N_s] normalised importance
  float* s_ra = s_a + N_s;                           // [N_s] its square root
  int* s_lo = reinterpret_cast<int*>(s_ra + N_s);    // [N_s]
  float* s_fr = reinterpret_cast<float*>(s_lo + N_s);  // [N_s]
  float* s_part = s_fr + N_s + ((4 - (4 * N_s) % 4) % 4);   // [4][D] partial weighted sums (16-byte aligned: N_s*4 floats)
  float* s_mu = s_part + 4 * D;                      // [D]
  float* s_red = s_mu + D;                           // [64]
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const TS* s = s_all + (size_t)b * s_batch_stride;
  const float* t = t_all + (size_t)b * N_t * D_t;
  const float* imp = imp_all + (size_t)b * N_t;
  float* sw = sw_all + (size_t)b * N_s * D_s;
  float* tw = tw_all + (size_t)b * N_s * D_t;
  const bool same_n = (N_t == N_s);

  const float ratio = (float)N_t / (float)N_s;
  float part = 0.f;
  for (int n = tid; n < N_s; n += nt) {
    float pos = ((float)n + 0.5f) * ratio - 0.5f;
    pos = pos < 0.f ? 0.f : pos;
    int lo = (int)pos;
    if (lo > N_t - 1) lo = N_t - 1;
    const int hi = lo + 1 < N_t ? lo + 1 : N_t - 1;
    const float fr = same_n ? 0.f : pos - (float)lo;
    const float v = same_n ? imp[n] : imp[lo] * (1.f - fr) + imp[hi] * fr;
    s_a[n] = v; s_lo[n] = lo; s_fr[n] = fr;
    part += v;
  }
  part = wave_sum(part);
  if ((tid & 63) == 0) s_red[tid >> 6] = part;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < (nt >> 6); ++w) tot += s_red[w];
  __syncthreads();
  for (int n = tid; n < N_s; n += nt) {
    const float an = s_a[n] / tot;
    s_a[n] = an;
    s_ra[n] = sqrtf(an);
    a_all[(size_t)b * N_s + n] = an;
  }
  __syncthreads();
  const int tg = tid >> 8, cq = tid & 255;
  const int Q = D >> 2, Qs = D_s >> 2;
  // ---- weighted column sums, four token groups
  for (int q = cq; q < Q; q += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < Qs) {
      for (int n = tg; n < N_s; n += 4) {
        const float4 x = load_quad<TS>(s + (size_t)n * D_s + 4 * q);
        const float an = s_a[n];
        acc.x = fmaf(an, x.x, acc.x); acc.y = fmaf(an, x.y, acc.y); acc.z = fmaf(an, x.z, acc.z); acc.w = fmaf(an, x.w, acc.w);
      }
    } else {
      const int dd = 4 * (q - Qs);
      for (int n = tg; n < N_s; n += 4) {
        const int lo = s_lo[n];
        float4 x = *reinterpret_cast<const float4*>(t + (size_t)lo * D_t + dd);
        const float fr = s_fr[n];
        if (fr != 0.f) {
          const int hi = lo + 1 < N_t ? lo + 1 : N_t - 1;
          const float4 y = *reinterpret_cast<const float4*>(t + (size_t)hi * D_t + dd);
          x.x = x.x * (1.f - fr) + y.x * fr; x.y = x.y * (1.f - fr) + y.y * fr;
          x.z = x.z * (1.f - fr) + y.z * fr; x.w = x.w * (1.f - fr) + y.w * fr;
        }
        const float an = s_a[n];
        acc.x = fmaf(an, x.x, acc.x); acc.y = fmaf(an, x.y, acc.y); acc.z = fmaf(an, x.z, acc.z); acc.w = fmaf(an, x.w, acc.w);
      }
    }
    *reinterpret_cast<float4*>(s_part + (size_t)tg * D + 4 * q) = acc;
  }
  __syncthreads();
  for (int d = tid; d < D; d += nt) s_mu[d] = (s_part[d] + s_part[D + d]) + (s_part[2 * D + d] + s_part[3 * D + d]);
  __syncthreads();
  // ---- centre, weight, write; traces
  double trs = 0.0, trt = 0.0;
  for (int q = cq; q < Q; q += 256) {
    const float4 mu = *reinterpret_cast<const float4*>(s_mu + 4 * q);
    if (q < Qs) {
      float acc = 0.f;
      for (int n = tg; n < N_s; n += 4) {
        const float4 x = load_quad<TS>(s + (size_t)n * D_s + 4 * q);
        const float ra = s_ra[n];
        const float4 v = make_float4(ra * (x.x - mu.x), ra * (x.y - mu.y), ra * (x.z - mu.z), ra * (x.w - mu.w));
        *reinterpret_cast<float4*>(sw + (size_t)n * D_s + 4 * q) = v;
        acc = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, acc))));
        if ((n & 31) == 31) { trs += (double)acc; acc = 0.f; }     // short fp32 runs, fp64 across them
      }
      trs += (double)acc;
    } else {
      const int dd = 4 * (q - Qs);
      float acc = 0.f;
      for (int n = tg; n < N_s; n += 4) {
        const int lo = s_lo[n];
        float4 x = *reinterpret_cast<const float4*>(t + (size_t)lo * D_t + dd);
        const float fr = s_fr[n];
        if (fr != 0.f) {
          const int hi = lo + 1 < N_t ? lo + 1 : N_t - 1;
          const float4 y = *reinterpret_cast<const float4*>(t + (size_t)hi * D_t + dd);
          x.x = x.x * (1.f - fr) + y.x * fr; x.y = x.y * (1.f - fr) + y.y * fr;
          x.z = x.z * (1.f - fr) + y.z * fr; x.w = x.w * (1.f - fr) + y.w * fr;
        }
        const float ra = s_ra[n];
        const float4 v = make_float4(ra * (x.x - mu.x), ra * (x.y - mu.y), ra * (x.z - mu.z), ra * (x.w - mu.w));
        *reinterpret_cast<float4*>(tw + (size_t)n * D_t + dd) = v;
        acc = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, acc))));
        if ((n & 31) == 31) { trt += (double)acc; acc = 0.f; }
      }
      trt += (double)acc;
    }
  }
  trs = wave_sum_d(trs); trt = wave_sum_d(trt);
  if ((tid & 63) == 0) { s_red[tid >> 6] = (float)trs; s_red[32 + (tid >> 6)] = (float)trt; }
  __syncthreads();
  if (tid == 0) {
    float x = 0.f, y = 0.f;
    for (int w = 0; w < (nt >> 6); ++w) { x += s_red[w]; y += s_red[32 + w]; }
    tr_all[(size_t)b * 2] = x; tr_all[(size_t)b * 2 + 1] = y;
  }
}

// Single-pass version (round 4; D_s and D_t multiples of 64, N_s <= 32 TPT): the kernel above reads every token twice
// (weighted column means, then centring) -- 540 MB per launch at c2 for 366 algorithmic, 0.30 of the HBM roof.  The means
// are per COLUMN, so a slice of 64 columns can be finished on its own: the 1024 threads of a sample are two sub-blocks
// of 16 column quads x 32 token groups; a sub-block loads its slice ONCE into registers (TPT quads per thread), combines
// the weighted partial sums of the 32 token groups through LDS in a fixed order, centres / weights / stores from the
// registers, and moves on to the slice two further (7 quads per thread: 13 per thread with four
// sub-blocks of 16 token groups spilled 43 registers at the 128 a 1024-thread workgroup has).  Traces are accumulated per thread as before (fp64 across slices).
template <typename TS, int TPT>
__global__ __launch_bounds__(1024) void procrustes_prep_v5_kernel(
    const TS* __restrict__ s_all, const float* __restrict__ t_all, const float* __restrict__ imp_all,
    int N_s, int N_t, int D_s, int D_t, int64_t s_batch_stride, float* __restrict__ sw_all,
    float* __restrict__ tw_all, float* __restrict__ a_all, float* __restrict__ tr_all) {
  extern __shared__ __align__(16) float sm[];
  const int n4 = (N_s + 3) & ~3;
  float* s_a = sm;                                   // [N_s] normalised importance
  float* s_ra = s_a + n4;                            // [N_s] its square root
  int* s_lo = reinterpret_cast<int*>(s_ra + n4);     // [N_s]
  float* s_fr = reinterpret_cast<float*>(s_lo + n4); // [N_s]
  float* s_part = s_fr + n4;                         // [2 sub-blocks][32 token groups][64] partial weighted sums
  float* s_mu = s_part + 2 * 32 * 64;                // [2][64]
  float* s_red = s_mu + 2 * 64;                      // [64]
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const TS* s = s_all + (size_t)b * s_batch_stride;
  const float* t = t_all + (size_t)b * N_t * D_t;
  const float* imp = imp_all + (size_t)b * N_t;
  float* sw = sw_all + (size_t)b * N_s * D_s;
  float* tw = tw_all + (size_t)b * N_s * D_t;
  const bool same_n = (N_t == N_s);

  const float ratio = (float)N_t / (float)N_s;
  float part = 0.f;
  for (int n = tid; n < N_s; n += nt) {
    float pos = ((float)n + 0.5f) * ratio - 0.5f;
    pos = pos < 0.f ? 0.f : pos;
    int lo = (int)pos;
    if (lo > N_t - 1) lo = N_t - 1;
    const int hi = lo + 1 < N_t ? lo + 1 : N_t - 1;
    const float fr = same_n ? 0.f : pos - (float)lo;
    const float v = same_n ? imp[n] : imp[lo] * (1.f - fr) + imp[hi] * fr;
    s_a[n] = v; s_lo[n] = lo; s_fr[n] = fr;
    part += v;
  }
  part = wave_sum(part);
  if ((tid & 63) == 0) s_red[tid >> 6] = part;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < (nt >> 6); ++w) tot += s_red[w];
  __syncthreads();
  for (int n = tid; n < N_s; n += nt) {
    const float an = s_a[n] / tot;
    s_a[n] = an;
    s_ra[n] = sqrtf(an);
    a_all[(size_t)b * N_s + n] = an;
  }
  __syncthreads();
  const int sb = tid >> 9, tg = (tid >> 4) & 31, cq = tid & 15;
  const int G_s = D_s >> 6, G = G_s + (D_t >> 6);
  float* my_part = s_part + ((sb * 32 + tg) * 64 + 4 * cq);
  double trs = 0.0, trt = 0.0;
  for (int g0 = 0; g0 < G; g0 += 2) {
    const int g = g0 + sb;
    const bool act = g < G, is_s = g < G_s;
    const int col = is_s ? 64 * g + 4 * cq : 64 * (g - G_s) + 4 * cq;
    float4 x[TPT];
    if (act) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < TPT; ++i) {
        const int n = tg + 32 * i;
        x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < N_s) {
          if (is_s) {
            x[i] = load_quad<TS>(s + (size_t)n * D_s + col);
          } else {
            const int lo = s_lo[n];
            float4 v = *reinterpret_cast<const float4*>(t + (size_t)lo * D_t + col);
            const float fr = s_fr[n];
            if (fr != 0.f) {
              const int hi = lo + 1 < N_t ? lo + 1 : N_t - 1;
              const float4 y = *reinterpret_cast<const float4*>(t + (size_t)hi * D_t + col);
              v.x = v.x * (1.f - fr) + y.x * fr; v.y = v.y * (1.f - fr) + y.y * fr;
              v.z = v.z * (1.f - fr) + y.z * fr; v.w = v.w * (1.f - fr) + y.w * fr;
            }
            x[i] = v;
          }
          const float an = s_a[n];
          acc.x = fmaf(an, x[i].x, acc.x); acc.y = fmaf(an, x[i].y, acc.y);
          acc.z = fmaf(an, x[i].z, acc.z); acc.w = fmaf(an, x[i].w, acc.w);
        }
      }
      *reinterpret_cast<float4*>(my_part) = acc;
    }
    __syncthreads();
    if (act && tg == 0) {                            // the 32 token groups in a fixed order
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
      for (int k = 0; k < 32; ++k) {
        const float4 p = *reinterpret_cast<const float4*>(s_part + ((sb * 32 + k) * 64 + 4 * cq));
        m.x += p.x; m.y += p.y; m.z += p.z; m.w += p.w;
      }
      *reinterpret_cast<float4*>(s_mu + sb * 64 + 4 * cq) = m;
    }
    __syncthreads();
    if (act) {
      const float4 mu = *reinterpret_cast<const float4*>(s_mu + sb * 64 + 4 * cq);
      float* dst = is_s ? sw : tw;
      const int D = is_s ? D_s : D_t;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < TPT; ++i) {
        const int n = tg + 32 * i;
        if (n < N_s) {
          const float ra = s_ra[n];
          const float4 v = make_float4(ra * (x[i].x - mu.x), ra * (x[i].y - mu.y), ra * (x[i].z - mu.z), ra * (x[i].w - mu.w));
          *reinterpret_cast<float4*>(dst + (size_t)n * D + col) = v;
          acc = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, acc))));
        }
      }
      if (is_s) trs += (double)acc; else trt += (double)acc;      // short fp32 runs, fp64 across them
    }
  }
  trs = wave_sum_d(trs); trt = wave_sum_d(trt);
  __syncthreads();
  if ((tid & 63) == 0) { s_red[tid >> 6] = (float)trs; s_red[32 + (tid >> 6)] = (float)trt; }
  __syncthreads();
  if (tid == 0) {
    float x = 0.f, y = 0.f;
    for (int w = 0; w < (nt >> 6); ++w) { x += s_red[w]; y += s_red[32 + w]; }
    tr_all[(size_t)b * 2] = x; tr_all[(size_t)b * 2 + 1] = y;
  }
}

// Procrustes backward, row epilogue.  p = (other side) G^T comes from a plain GEMM; with R = W - p
//   out[row, :] = 2 gl sqrt(a[row]) R[row, :]   (the gradient w.r.t. the raw tokens) and
//   rowdot[row] = 2 gl sum_d R[row, d] W[row, d]  (the part of d loss / d a this side owns)
// in ONE pass over p and W (torch needed five elementwise passes over the [E*B, N, D_t] tensors).
// One wave per row; out may alias p (fp32) or be a bf16 buffer.
template <typename TO>
__global__ __launch_bounds__(256) void procrustes_bwd_rows_kernel(const float* __restrict__ p, const float* __restrict__ w,
                                                                  const float* __restrict__ a, const float* __restrict__ gl,
                                                                  int64_t rows, int rows_per_batch, int D, TO* out,
                                                                  float* __restrict__ rowdot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nq = D >> 2;
  for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < rows; row += (int64_t)gridDim.x * 4) {
    const float c2 = 2.f * gl[row / rows_per_batch];
    const float c = c2 * sqrtf(a[row]);
    const float4* p4 = reinterpret_cast<const float4*>(p + row * D);
    const float4* w4 = reinterpret_cast<const float4*>(w + row * D);
    float dot = 0.f;
    for (int q = lane; q < nq; q += 64) {
      const float4 pv = p4[q], wv = w4[q];
      const float4 rv = make_float4(wv.x - pv.x, wv.y - pv.y, wv.z - pv.z, wv.w - pv.w);
      dot = fmaf(rv.x, wv.x, fmaf(rv.y, wv.y, fmaf(rv.z, wv.z, fmaf(rv.w, wv.w, dot))));
      if constexpr (sizeof(TO) == 4) {
        reinterpret_cast<float4*>(out + row * D)[q] = make_float4(c * rv.x, c * rv.y, c * rv.z, c * rv.w);
      } else {
        __hip_bfloat16 b0 = __float2bfloat16(c * rv.x), b1 = __float2bfloat16(c * rv.y);
        __hip_bfloat16 b2 = __float2bfloat16(c * rv.z), b3 = __float2bfloat16(c * rv.w);
        uint2 o;
        o.x = (unsigned int)(*reinterpret_cast<unsigned short*>(&b0)) | ((unsigned int)(*reinterpret_cast<unsigned short*>(&b1)) << 16);
        o.y = (unsigned int)(*reinterpret_cast<unsigned short*>(&b2)) | ((unsigned int)(*reinterpret_cast<unsigned short*>(&b3)) << 16);
        reinterpret_cast<uint2*>(out + row * D)[q] = o;
      }
    }
    dot = wave_sum(dot);
    if (lane == 0) rowdot[row] = c2 * dot;
  }
}

// one workgroup per sample
template <typename TS>
__global__ __launch_bounds__(512) void procrustes_prep_kernel(
    const TS* __restrict__ s_all, const float* __restrict__ t_all, const float* __restrict__ imp_all,
    int N_s, int N_t, int D_s, int D_t, int64_t s_batch_stride, float* __restrict__ sw_all,
    float* __restrict__ tw_all, float* __restrict__ a_all, float* __restrict__ tr_all) {
  extern __shared__ __align__(16) float sm[];
  float* s_a = sm;                 // [N_s] normalised importance
  int* s_lo = reinterpret_cast<int*>(s_a + N_s);   // [N_s]
  float* s_fr = reinterpret_cast<float*>(s_lo + N_s);  // [N_s]
  float* s_mu = s_fr + N_s;        // [D_s + D_t]
  float* s_red = s_mu + D_s + D_t; // [32]
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const TS* s = s_all + (size_t)b * s_batch_stride;
  const float* t = t_all + (size_t)b * N_t * D_t;
  const float* imp = imp_all + (size_t)b * N_t;
  float* sw = sw_all + (size_t)b * N_s * D_s;
  float* tw = tw_all + (size_t)b * N_s * D_t;

  // resample taps (F.interpolate linear, align_corners=False) and importance
  const float ratio = (float)N_t / (float)N_s;
  float part = 0.f;
  for (int n = tid; n < N_s; n += nt) {
    float pos = ((float)n + 0.5f) * ratio - 0.5f;
    pos = pos < 0.f ? 0.f : pos;
    int lo = (int)pos;
    if (lo > N_t - 1) lo = N_t - 1;
    const int hi = lo + 1 < N_t ? lo + 1 : N_t - 1;
    const float fr = (N_t == N_s) ? 0.f : pos - (float)lo;
    const float v = (N_t == N_s) ? imp[n] : imp[lo] * (1.f - fr) + imp[hi] * fr;
    s_a[n] = v; s_lo[n] = lo; s_fr[n] = fr;
    part += v;
  }
  part = wave_sum(part);
  if ((tid & 63) == 0) s_red[tid >> 6] = part;
  __syncthreads();
  float tot = 0.f;
  for (int w = 0; w < (nt >> 6); ++w) tot += s_red[w];
  __syncthreads();
  for (int n = tid; n < N_s; n += nt) {
    const float an = s_a[n] / tot;
    s_a[n] = an;
    a_all[(size_t)b * N_s + n] = an;
  }
  __syncthreads();
  // weighted means: one thread per feature column (coalesced along d)
  for (int d = tid; d < D_s + D_t; d += nt) {
    float mu = 0.f;
    if (d < D_s) {
      for (int n = 0; n < N_s; ++n) {
        float x;
        if constexpr (sizeof(TS) == 2) x = bf16_bits_to_f32(s[(size_t)n * D_s + d]);
        else x = s[(size_t)n * D_s + d];
        mu = fmaf(s_a[n], x, mu);
      }
    } else {
      const int dd = d - D_s;
      for (int n = 0; n < N_s; ++n) {
        const int lo = s_lo[n];
        const float fr = s_fr[n];
        float x = t[(size_t)lo * D_t + dd];
        if (fr != 0.f) {
          const int hi = lo + 1 < N_t ? lo + 1 : N_t - 1;
          x = x * (1.f - fr) + t[(size_t)hi * D_t + dd] * fr;
        }
        mu = fmaf(s_a[n], x, mu);
      }
    }
    s_mu[d] = mu;
  }
  __syncthreads();
  double trs = 0.0, trt = 0.0;
  for (int d = tid; d < D_s + D_t; d += nt) {
    const float mu = s_mu[d];
    if (d < D_s) {
      for (int n = 0; n < N_s; ++n) {
        float x;
        if constexpr (sizeof(TS) == 2) x = bf16_bits_to_f32(s[(size_t)n * D_s + d]);
        else x = s[(size_t)n * D_s + d];
        const float v = sqrtf(s_a[n]) * (x - mu);
        sw[(size_t)n * D_s + d] = v;
        trs += (double)v * (double)v;
      }
    } else {
      const int dd = d - D_s;
      for (int n = 0; n < N_s; ++n) {
        const int lo = s_lo[n];
        const float fr = s_fr[n];
        float x = t[(size_t)lo * D_t + dd];
        if (fr != 0.f) {
          const int hi = lo + 1 < N_t ? lo + 1 : N_t - 1;
          x = x * (1.f - fr) + t[(size_t)hi * D_t + dd] * fr;
        }
        const float v = sqrtf(s_a[n]) * (x - mu);
        tw[(size_t)n * D_t + dd] = v;
        trt += (double)v * (double)v;
      }
    }
  }
  trs = wave_sum_d(trs); trt = wave_sum_d(trt);
  if ((tid & 63) == 0) { s_red[tid >> 6] = (float)trs; s_red[16 + (tid >> 6)] = (float)trt; }
  __syncthreads();
  if (tid == 0) {
    float x = 0.f, y = 0.f;
    for (int w = 0; w < (nt >> 6); ++w) { x += s_red[w]; y += s_red[16 + w]; }
    tr_all[(size_t)b * 2] = x; tr_all[(size_t)b * 2 + 1] = y;
  }
}

static int grid_for(int64_t nvec) {
  int64_t g = (nvec + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace basd

namespace basd {

template <typename T, int E>
static void launch_mix(const void* const* layers, int L, const float* w, int64_t nvec, float* out,
                       int64_t elems, int64_t pb, int64_t bs, hipStream_t st) {
  LayerPtrs lp;
  for (int i = 0; i < kMaxL; ++i) lp.p[i] = i < L ? layers[i] : nullptr;
  hipLaunchKernelGGL((mix_tokens_kernel<T, E>), dim3(grid_for(nvec)), dim3(256), 0, st, lp, L, w, nvec, out, elems,
                     pb, bs);
}
template <typename T, int E>
static void launch_dots(const void* const* layers, int L, const float* g, int64_t nvec, int64_t elems,
                        double* dots, int64_t pb, int64_t bs, hipStream_t st) {
  LayerPtrs lp;
  for (int i = 0; i < kMaxL; ++i) lp.p[i] = i < L ? layers[i] : nullptr;
  hipLaunchKernelGGL((mix_grad_dots_kernel<T, E>), dim3(grid_for(nvec)), dim3(256), 0, st, lp, L, g, nvec, elems,
                     dots, pb, bs);
}

template <typename T>
static int dispatch_mix(int E, const void* const* layers, int L, const float* w, int64_t nvec, float* out,
                        int64_t elems, int64_t pb, int64_t bs, hipStream_t st) {
  switch (E) {
    case 1: launch_mix<T, 1>(layers, L, w, nvec, out, elems, pb, bs, st); break;
    case 2: launch_mix<T, 2>(layers, L, w, nvec, out, elems, pb, bs, st); break;
    case 3: launch_mix<T, 3>(layers, L, w, nvec, out, elems, pb, bs, st); break;
    case 4: launch_mix<T, 4>(layers, L, w, nvec, out, elems, pb, bs, st); break;
    case 5: launch_mix<T, 5>(layers, L, w, nvec, out, elems, pb, bs, st); break;
    case 6: launch_mix<T, 6>(layers, L, w, nvec, out, elems, pb, bs, st); break;
    case 7: launch_mix<T, 7>(layers, L, w, nvec, out, elems, pb, bs, st); break;
    case 8: launch_mix<T, 8>(layers, L, w, nvec, out, elems, pb, bs, st); break;
    default: return fail(BASD_ERR_SHAPE, "E=%d out of 1..8", E);
  }
  return BASD_OK;
}
template <typename T>
static int dispatch_dots(int E, const void* const* layers, int L, const float* g, int64_t nvec,
                         int64_t elems, double* dots, int64_t pb, int64_t bs, hipStream_t st) {
  switch (E) {
    case 1: launch_dots<T, 1>(layers, L, g, nvec, elems, dots, pb, bs, st); break;
    case 2: launch_dots<T, 2>(layers, L, g, nvec, elems, dots, pb, bs, st); break;
    case 3: launch_dots<T, 3>(layers, L, g, nvec, elems, dots, pb, bs, st); break;
    case 4: launch_dots<T, 4>(layers, L, g, nvec, elems, dots, pb, bs, st); break;
    case 5: launch_dots<T, 5>(layers, L, g, nvec, elems, dots, pb, bs, st); break;
    case 6: launch_dots<T, 6>(layers, L, g, nvec, elems, dots, pb, bs, st); break;
    case 7: launch_dots<T, 7>(layers, L, g, nvec, elems, dots, pb, bs, st); break;
    case 8: launch_dots<T, 8>(layers, L, g, nvec, elems, dots, pb, bs, st); break;
    default: return fail(BASD_ERR_SHAPE, "E=%d out of 1..8", E);
  }
  return BASD_OK;
}

}  // namespace basd

extern "C" int basd_mix_tokens(const void* const* x_layers, int x_dtype, int L, int E, const float* w,
                               int64_t elems, int64_t per_batch, int64_t batch_stride, float* out, void* stream) {
  using namespace basd;
  if (L < 1 || L > kMaxL) return fail(BASD_ERR_SHAPE, "mix_tokens: L=%d out of 1..%d", L, kMaxL);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (x_dtype == BASD_DTYPE_F32) {
    if (elems % 4) return fail(BASD_ERR_SHAPE, "mix_tokens: elems %% 4 != 0");
    if (per_batch % 4) return fail(BASD_ERR_SHAPE, "mix_tokens: per_batch %% 4 != 0");
    rc = dispatch_mix<float>(E, x_layers, L, w, elems / 4, out, elems, per_batch, batch_stride, st);
  } else if (x_dtype == BASD_DTYPE_BF16) {
    if (elems % 8) return fail(BASD_ERR_SHAPE, "mix_tokens: elems %% 8 != 0");
    if (per_batch % 8 || batch_stride % 8) return fail(BASD_ERR_SHAPE, "mix_tokens: per_batch / batch_stride %% 8 != 0");
    rc = dispatch_mix<unsigned short>(E, x_layers, L, w, elems / 8, out, elems, per_batch, batch_stride, st);
  } else {
    return fail(BASD_ERR_DTYPE, "mix_tokens: dtype %d", x_dtype);
  }
  if (rc) return rc;
  return check_launch("mix_tokens");
}

extern "C" int basd_mix_grad_dots(const void* const* x_layers, int x_dtype, int L, int E, const float* g,
                                  int64_t elems, int64_t per_batch, int64_t batch_stride, double* dots,
                                  void* stream) {
  using namespace basd;
  if (L < 1 || L > kMaxL) return fail(BASD_ERR_SHAPE, "mix_grad_dots: L=%d out of 1..%d", L, kMaxL);
  hipStream_t st = (hipStream_t)stream;
  int rc;
  if (x_dtype == BASD_DTYPE_F32) {
    if (elems % 4) return fail(BASD_ERR_SHAPE, "mix_grad_dots: elems %% 4 != 0");
    if (per_batch % 4) return fail(BASD_ERR_SHAPE, "mix_grad_dots: per_batch %% 4 != 0");
    rc = dispatch_dots<float>(E, x_layers, L, g, elems / 4, elems, dots, per_batch, batch_stride, st);
  } else if (x_dtype == BASD_DTYPE_BF16) {
    if (elems % 8) return fail(BASD_ERR_SHAPE, "mix_grad_dots: elems %% 8 != 0");
    if (per_batch % 8 || batch_stride % 8) return fail(BASD_ERR_SHAPE, "mix_grad_dots: per_batch / batch_stride %% 8 != 0");
    rc = dispatch_dots<unsigned short>(E, x_layers, L, g, elems / 8, elems, dots, per_batch, batch_stride, st);
  } else {
    return fail(BASD_ERR_DTYPE, "mix_grad_dots: dtype %d", x_dtype);
  }
  if (rc) return rc;
  return check_launch("mix_grad_dots");
}

extern "C" int basd_procrustes_prep(const void* s, int s_dtype, int64_t s_batch_stride, const float* t,
                                    const float* imp, int B, int N_s, int N_t, int D_s, int D_t, float* s_w,
                                    float* t_w, float* a, float* tr, void* stream) {
  using namespace basd;
  if (B <= 0) return BASD_OK;
  if (N_s < 1 || N_t < 1 || D_s < 1 || D_t < 1)
    return fail(BASD_ERR_SHAPE, "procrustes_prep: bad shape");
  const size_t lds = (size_t)(3 * N_s + D_s + D_t + 32) * 4;
  if (lds > 160 * 1024) return fail(BASD_ERR_SHAPE, "procrustes_prep: LDS %zu too large", lds);
  hipStream_t st = (hipStream_t)stream;
  const size_t lds4 = ((size_t)4 * N_s + 4 + 5 * (size_t)(D_s + D_t) + 64) * 4;
  const bool vec_ok = D_s % 4 == 0 && D_t % 4 == 0 && s_batch_stride % 4 == 0 && ((uintptr_t)s & 15) == 0 &&
                      ((uintptr_t)t & 15) == 0 && ((uintptr_t)s_w & 15) == 0 && ((uintptr_t)t_w & 15) == 0 && lds4 <= 64 * 1024;
  if (vec_ok && (s_dtype == BASD_DTYPE_F32 || s_dtype == BASD_DTYPE_BF16) && D_s % 64 == 0 && D_t % 64 == 0 && N_s <= 256) {
    // single pass over the tokens (BASD_PREP_V5=0: the two-pass kernel below, A/B timing)
    const char* env = getenv("BASD_PREP_V5");
    if (!(env && env[0] == '0')) {
      const size_t lds5 = ((size_t)4 * ((N_s + 3) & ~3) + 2 * 32 * 64 + 2 * 64 + 64) * 4;
#define BASD_PREP5(TS, TPT)                                                                                  \
  hipLaunchKernelGGL((procrustes_prep_v5_kernel<TS, TPT>), dim3(B), dim3(1024), lds5, st, (const TS*)s, t, imp, N_s, N_t, \
                     D_s, D_t, s_batch_stride, s_w, t_w, a, tr)
      if (s_dtype == BASD_DTYPE_F32) { if (N_s <= 224) BASD_PREP5(float, 7); else BASD_PREP5(float, 8); }
      else { if (N_s <= 224) BASD_PREP5(unsigned short, 7); else BASD_PREP5(unsigned short, 8); }
#undef BASD_PREP5
      return check_launch("procrustes_prep (single pass)");
    }
  }
  if (vec_ok && (s_dtype == BASD_DTYPE_F32 || s_dtype == BASD_DTYPE_BF16)) {
    if (s_dtype == BASD_DTYPE_F32)
      hipLaunchKernelGGL(procrustes_prep_v4_kernel<float>, dim3(B), dim3(1024), lds4, st, (const float*)s, t, imp,
                         N_s, N_t, D_s, D_t, s_batch_stride, s_w, t_w, a, tr);
    else
      hipLaunchKernelGGL(procrustes_prep_v4_kernel<unsigned short>, dim3(B), dim3(1024), lds4, st,
                         (const unsigned short*)s, t, imp, N_s, N_t, D_s, D_t, s_batch_stride, s_w, t_w, a, tr);
    return check_launch("procrustes_prep (vectorised)");
  }
  if (s_dtype == BASD_DTYPE_F32) {
    hipLaunchKernelGGL(procrustes_prep_kernel<float>, dim3(B), dim3(512), lds, st, (const float*)s, t, imp,
                       N_s, N_t, D_s, D_t, s_batch_stride, s_w, t_w, a, tr);
  } else if (s_dtype == BASD_DTYPE_BF16) {
    hipLaunchKernelGGL(procrustes_prep_kernel<unsigned short>, dim3(B), dim3(512), lds, st,
                       (const unsigned short*)s, t, imp, N_s, N_t, D_s, D_t, s_batch_stride, s_w, t_w, a, tr);
  } else {
    return fail(BASD_ERR_DTYPE, "procrustes_prep: dtype %d", s_dtype);
  }
  return check_launch("procrustes_prep");
}

extern "C" int basd_procrustes_bwd_rows(const float* r, const float* w, const float* a, const float* gl, int64_t rows,
                                        int rows_per_batch, int D, void* out, int out_dtype, float* rowdot,
                                        void* stream) {
  using namespace basd;
  if (rows <= 0) return BASD_OK;
  if (D < 4 || D % 4 || rows_per_batch < 1) return fail(BASD_ERR_SHAPE, "procrustes_bwd_rows: D=%d rows_per_batch=%d", D, rows_per_batch);
  int64_t grid = (rows + 3) / 4;
  if (grid > 4096) grid = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == BASD_DTYPE_F32)
    hipLaunchKernelGGL(procrustes_bwd_rows_kernel<float>, dim3((int)grid), dim3(256), 0, st, r, w, a, gl, rows,
                       rows_per_batch, D, (float*)out, rowdot);
  else if (out_dtype == BASD_DTYPE_BF16)
    hipLaunchKernelGGL(procrustes_bwd_rows_kernel<unsigned short>, dim3((int)grid), dim3(256), 0, st, r, w, a, gl, rows,
                       rows_per_batch, D, (unsigned short*)out, rowdot);
  else
    return fail(BASD_ERR_DTYPE, "procrustes_bwd_rows: out dtype %d", out_dtype);
  return check_launch("procrustes_bwd_rows");
}
