// Backward of the attention-weighted Procrustes term as ONE C entry (reference: autograd of
// src/losses/relational.py:47-48, i.e. svd_backward of the nuclear norm = the polar factor U V^T, then the centring /
// weighting of :29-46).  With the factors the forward saved (basd_procrustes_fwd: fac_s, a_t) the gradient w.r.t. the
// weighted tokens is a residual
//     R_t = t_w - (s_w G)   = t_w - a_t t_w        [n, d_t]        (G = U V^T, never formed)
//     R_s = s_w - (t_w G^T) = s_w - a_s s_w (token side)  or  s_w - fac_s (feature side)       [n, d_s]
// scaled per row, g = 2 gl sqrt(a) R, plus the row dots 2 gl <R, W> that make up d loss / d a.
//
// The one big product, a_t t_w (batch x [n, n] x [n, d_t]: 60 GFLOP at BASELINE c2), ran as a library fp32 bmm
// (0.62 ms) followed by basd_procrustes_bwd_rows (another 1.9 GB pass).  Here it runs on the bf16 matrix cores as a
// THREE-PRODUCT SPLIT of both fp32 operands (x = hi + mid + ..., hi = bf16(x), mid = bf16(x - hi);
// A W ~ A_hi W_hi + A_hi W_mid + A_mid W_hi: relative error 2^-16 of |A| |W| -- the gradient tolerance is 5e-4), with
// the residual, the scaling and the row dots in the epilogue of the same kernel: t_w is read once as the B operand and
// once (L2-hot) as the residual's W, g_t is written once, nothing else touches HBM.
//
// One workgroup (6 waves; two workgroups per CU, three waves per SIMD: one's barriers and epilogue overlap the other's
// matrix work) per matrix of the batch.  The A factor ([n, n] fp32) is staged per 32-wide K step into LDS as two bf16 planes (split on the way
// in; 80-byte rows: conflict-free 16-byte fragment reads), double buffered, its global loads issued one step ahead; a
// wave owns one 16-column strip of the current 96-column group and keeps its B fragment (loaded straight from global
// memory, 64 contiguous bytes per 16 lanes, split in registers) for one K step at a time, prefetched one step ahead;
// accumulators: MT tiles of 16 x 16 fp32 (52 VGPRs at 196 rows).  v_mfma_f32_16x16x32_bf16, fp32 accumulation.
//
// Measured (round 4, 1024 x [196, 196] x [196, 768], one MI355X): 1.10 ms for the whole backward against 1.00 ms with the
// library fp32 bmm (0.62) + two row passes; inside the captured step the two are equal within the noise (35.93 vs 35.79
// ms).  PMC of this kernel: 1.70 GB fetched / 0.71 GB written (1.23 + 0.62 algorithmic), 74 % of the wave cycles
// WAITING, 14 % issuing, matrix cores busy 23 %: it is bound by one exposed memory round trip per K step (the B strip
// is a cold HBM read and the 39 MFMAs of a step last 0.3 us), not by arithmetic or bandwidth.  Tried on the way and
// dropped: 8 waves x 2 strips (104 accumulator registers: 85 - 300 spilled VGPRs whatever the fencing), the residual in
// the accumulator layout (4-byte loads / stores + 16-lane reductions: 0.8 of 1.2 ms), fragment reads in two batches
// (no change), the whole B strip of a group loaded in one burst into 56 registers (the right idea for the measured
// bottleneck, but 259 spilled registers at the 168-register budget of three waves per SIMD: 1.78 ms).  Next: a
// 4-wave / 256-register layout that holds the B strip, or the strip staged through LDS with transposing reads.
#include <stdlib.h>
#include "basd_common.h"

namespace basd {

typedef short pb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float pb_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pb_f32x2 __attribute__((ext_vector_type(2)));

constexpr int PB_ROWB = 80;                 // bytes per staged A row (32 bf16 = 64 + 16 pad)
constexpr int PB_WAVES = 6;                 // waves per workgroup: two workgroups per CU at three waves per SIMD (168 VGPRs)
constexpr int PB_THREADS = 64 * PB_WAVES;

// (hi, mid) bf16 split of two floats, packed: hi = bf16(x) (round to nearest even), mid = bf16(x - hi)
__device__ __forceinline__ void pb_split2(float x0, float x1, unsigned int& hi, unsigned int& mid) {
  const pb_bf16x2 h = __builtin_convertvector((pb_f32x2){x0, x1}, pb_bf16x2);
  const pb_f32x2 hf = __builtin_convertvector(h, pb_f32x2);
  const pb_bf16x2 m = __builtin_convertvector((pb_f32x2){x0 - hf.x, x1 - hf.y}, pb_bf16x2);
  hi = __builtin_bit_cast(unsigned int, h);
  mid = __builtin_bit_cast(unsigned int, m);
}

template <int MT, typename TO>      // MT = m tiles (16 rows each) >= ceil(n / 16)
__global__ __launch_bounds__(PB_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void procrustes_bwd_side_kernel(
    const float* __restrict__ fac, const float* __restrict__ w, const float* __restrict__ a,
    const float* __restrict__ gl, int n, int d, TO* __restrict__ out, float* __restrict__ rowdot) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int PLANE = MT * 16 * PB_ROWB;                 // one bf16 plane of one K step
  constexpr int ITEMS = (MT * 16 * 4 + PB_THREADS - 1) / PB_THREADS;         // staging items (row, k octet) per thread and K step
  unsigned char* abuf = smem;                              // [2 buffers][2 planes][MT * 16 rows][PB_ROWB]
  float* s_dotw = reinterpret_cast<float*>(smem + 4 * PLANE);  // [waves][MT * 16] row dots, one slice per wave
  float* s_c = s_dotw + PB_WAVES * MT * 16;                           // [MT * 16] row scales 2 gl sqrt(a)
  float* s_park = reinterpret_cast<float*>(abuf);              // [waves][16 rows][16 columns] epilogue tile: the operand
                                                               // buffers are idle then (80 KiB per workgroup: two per CU)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* A = fac + (size_t)b * n * n;
  const float* W = w + (size_t)b * n * d;
  TO* O = out + (size_t)b * n * d;
  const int ksteps = (n + 31) >> 5;
  const int strips = d >> 4;
  const float c2 = 2.f * gl[b];
  for (int i = tid; i < MT * 16; i += PB_THREADS) {
#pragma unroll
    for (int wv_ = 0; wv_ < PB_WAVES; ++wv_) s_dotw[wv_ * MT * 16 + i] = 0.f;
    s_c[i] = i < n ? c2 * __builtin_amdgcn_sqrtf(a[(size_t)b * n + i]) : 0.f;
  }

  // Staging of one K step of A in two halves: the global loads (branch-free: rows / columns beyond n read a clamped,
  // valid address and are zeroed by a select) are issued BEFORE the MFMAs of the current step, the split into the two
  // bf16 planes and the LDS writes follow them -- the L2 round trip hides under the matrix work.
  // n % 4 == 0 and a 16-byte aligned factor (checked by the C entry): every load is 16 bytes wide.  Offsets are
  // computed once per item = (row, k octet); a K step only adds a uniform base.  Loads that would leave the matrix
  // (rows >= n of the padded m tiles, the columns >= n of the last K step) are clamped to its last 16 bytes and the
  // values zeroed by a mask -- applied in the last K step / for the padded rows only.
  float4 sa[ITEMS][2];
  unsigned sa_off[ITEMS];
  const unsigned a_last = (unsigned)n * (unsigned)n - 4u;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int item = tid + PB_THREADS * it;
    const int row = item >> 2, oct = item & 3;
    sa_off[it] = (unsigned)(row < n ? row : n - 1) * (unsigned)n + (unsigned)(oct * 8);
  }
  auto stage_load = [&](int ks) {
    const unsigned kbase = (unsigned)ks * 32u;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const unsigned o0 = sa_off[it] + kbase, o1 = o0 + 4u;
      sa[it][0] = *reinterpret_cast<const float4*>(A + (o0 < a_last ? o0 : a_last));
      sa[it][1] = *reinterpret_cast<const float4*>(A + (o1 < a_last ? o1 : a_last));
    }
  };
  auto stage_store = [&](int ks, int buf) {
    unsigned char* dst = abuf + (size_t)buf * 2 * PLANE;
    const bool tail = (ks + 1) * 32 > n;                   // uniform
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int item = tid + PB_THREADS * it;
      const int row = item >> 2, oct = item & 3;
      if (item < MT * 16 * 4) {
        float4 v0 = sa[it][0], v1 = sa[it][1];
        const int k0 = ks * 32 + oct * 8;
        const float m0 = (row < n && (!tail || k0 + 4 <= n)) ? 1.f : 0.f;
        const float m1 = (row < n && (!tail || k0 + 8 <= n)) ? 1.f : 0.f;
        uint4 hi, mid;
        pb_split2(v0.x * m0, v0.y * m0, hi.x, mid.x); pb_split2(v0.z * m0, v0.w * m0, hi.y, mid.y);
        pb_split2(v1.x * m1, v1.y * m1, hi.z, mid.z); pb_split2(v1.z * m1, v1.w * m1, hi.w, mid.w);
        const unsigned lo = (unsigned)row * PB_ROWB + (unsigned)oct * 16;
        *reinterpret_cast<uint4*>(dst + lo) = hi;
        *reinterpret_cast<uint4*>(dst + PLANE + lo) = mid;
      }
    }
  };

  for (int g0 = 0; g0 < strips; g0 += PB_WAVES) {          // column groups of PB_WAVES strips (96 columns), one per wave
    const int s0 = g0 + wave;
    const bool has = s0 < strips;
    const int col = (lane & 15);
    const unsigned cs = has ? (unsigned)(s0 * 16 + col) : (unsigned)col;                // 32-bit offsets: n d < 2^31
    // B fragment of one K step: lane holds column 16 s0 + (lane & 15), rows k0 + 8 (lane >> 4) + 0..7 (clamped, select)
    float bn[8];
    const unsigned b_off = (unsigned)((lane >> 4) * 8) * (unsigned)d + cs;    // row 8 (lane >> 4) of a K step, this column
    const unsigned b_last = (unsigned)(n - 1) * (unsigned)d + cs;             // same column, last row
    auto fetch_b = [&](int ks) {
      const unsigned o = b_off + (unsigned)ks * 32u * (unsigned)d;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned oj = o + (unsigned)j * (unsigned)d;
        bn[j] = W[oj < b_last ? oj : b_last];
      }
    };
    __syncthreads();                                       // previous group's fragment reads are done
    stage_load(0);
    fetch_b(0);
    stage_store(0, 0);
    __syncthreads();
    pb_f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (pb_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < ksteps; ++ks) {
      const int buf = ks & 1;
      const int kb = ks * 32 + (lane >> 4) * 8;
      pb_bf16x8 bh, bm;
      {
        float x[8];
        if ((ks + 1) * 32 > n) {                           // uniform: the last K step, rows beyond n
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = (has && kb + j < n) ? bn[j] : 0.f;
        } else {
          const float keep = has ? 1.f : 0.f;
#pragma unroll
          for (int j = 0; j < 8; ++j) x[j] = bn[j] * keep;
        }
        uint4 hi, mid;
        pb_split2(x[0], x[1], hi.x, mid.x); pb_split2(x[2], x[3], hi.y, mid.y);
        pb_split2(x[4], x[5], hi.z, mid.z); pb_split2(x[6], x[7], hi.w, mid.w);
        bh = __builtin_bit_cast(pb_bf16x8, hi);
        bm = __builtin_bit_cast(pb_bf16x8, mid);
      }
      const bool more = ks + 1 < ksteps;
      if (more) {                                          // next step's operands: loads only, consumed after the MFMAs
        fetch_b(ks + 1);
        stage_load(ks + 1);
      }
      // A fragments in two batches of (MT + 1) / 2 tiles: all 16-byte fragment reads of a batch are issued together (one
      // exposed LDS latency per batch; a fence per tile pair exposed it MT / 2 times per K step: 4.8 k cycles per step
      // measured with in-kernel stamps against 0.6 k of matrix work), the fence between the batches keeps the
      // scheduler from hoisting the second batch on top of the first (registers)
      const unsigned char* ap = abuf + (size_t)buf * 2 * PLANE + (size_t)(lane & 15) * PB_ROWB + (lane >> 4) * 16;
      constexpr int HB = (MT + 1) / 2;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        pb_bf16x8 fh[HB], fm[HB];
#pragma unroll
        for (int j = 0; j < HB; ++j) {
          const int i = hb * HB + j < MT ? hb * HB + j : MT - 1;
          fh[j] = *reinterpret_cast<const pb_bf16x8*>(ap + (size_t)i * 16 * PB_ROWB);
          fm[j] = *reinterpret_cast<const pb_bf16x8*>(ap + PLANE + (size_t)i * 16 * PB_ROWB);
        }
#pragma unroll
        for (int j = 0; j < HB; ++j) {
          const int i = hb * HB + j;
          if (i < MT) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[j], bh, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[j], bm, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm[j], bh, acc[i], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (more) stage_store(ks + 1, buf ^ 1);
      __syncthreads();                                     // buffer `buf` may be overwritten, `buf ^ 1` is complete
    }
    // ---- epilogue.  acc[i][r] = P at row 16 i + 4 (lane >> 4) + r, column 16 s0 + (lane & 15): one value per lane
    // and row -- consumed in that layout the residual costs a 4-byte load, a 4-byte store and a 16-lane reduction per
    // element group (measured: 0.8 of 1.2 ms).  Instead the wave parks one 16 x 16 tile at a time in its own 1 KiB of
    // LDS and reads it back row-wise: 16 bytes per lane, 64 contiguous bytes per row segment for the W load and the
    // store, a 4-lane reduction per row.  All W loads of the strip are issued before the first tile is processed.
    float* park = s_park + wave * 256;
    const int g4 = lane >> 4;
    const int prow = lane >> 2, pch = lane & 3;            // row-wise walk: 16 rows x 4 chunks of 4 columns
    const int pcol = s0 * 16 + pch * 4;
    constexpr int H0 = (MT + 1) / 2;                       // tiles per batch: the W loads of a batch are in flight together
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      constexpr int HN = H0;
      float4 wq[HN];
#pragma unroll
      for (int j = 0; j < HN; ++j) {
        const int i = hb * H0 + j;
        const int row = i * 16 + prow;
        const bool ok = has && row < n && i < MT;
        wq[j] = *reinterpret_cast<const float4*>(W + (unsigned)(ok ? row : 0) * (unsigned)d + (unsigned)(ok ? pcol : 0));
      }
#pragma unroll
      for (int j = 0; j < HN; ++j) {
        const int i = hb * H0 + j;
        if (i < MT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) park[(g4 * 4 + r) * 16 + col] = acc[i][r];
          // wave-private tile: the wave's own LDS writes precede its reads (in order), no barrier
          const float4 pv = *reinterpret_cast<const float4*>(park + prow * 16 + pch * 4);
          const int row = i * 16 + prow;
          const bool ok = has && row < n;
          const unsigned off = (unsigned)(ok ? row : 0) * (unsigned)d + (unsigned)(ok ? pcol : 0);
          const float4 wv = wq[j];
          const float cr = s_c[row];
          const float4 rv = make_float4(wv.x - pv.x, wv.y - pv.y, wv.z - pv.z, wv.w - pv.w);
          float dot = ok ? fmaf(rv.x, wv.x, fmaf(rv.y, wv.y, fmaf(rv.z, wv.z, rv.w * wv.w))) : 0.f;
          if (ok) {
            if constexpr (sizeof(TO) == 4) {
              *reinterpret_cast<float4*>(O + off) = make_float4(cr * rv.x, cr * rv.y, cr * rv.z, cr * rv.w);
            } else {
              uint2 o;
              o.x = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)(cr * rv.x)) |
                    ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)(cr * rv.y)) << 16);
              o.y = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)(cr * rv.z)) |
                    ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)(cr * rv.w)) << 16);
              *reinterpret_cast<uint2*>(O + off) = o;
            }
          }
          dot += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dot), 0xB1, 0xF, 0xF, true));     // quad_perm 1,0,3,2
          dot += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dot), 0x4E, 0xF, 0xF, true));     // quad_perm 2,3,0,1
          if (pch == 0) s_dotw[wave * (MT * 16) + row] += c2 * dot;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += PB_THREADS) {
    float t = 0.f;
#pragma unroll
    for (int wv_ = 0; wv_ < PB_WAVES; ++wv_) t += s_dotw[wv_ * MT * 16 + i];
    rowdot[(size_t)b * n + i] = t;
  }
}

typedef short pb_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) pb_v4s pb_lds_v4s;
constexpr int PB_BPITCH = 208;              // bytes between the k rows of the staged W tile (96 bf16 = 192 + 16 pad)

// The same kernel with the B operand (W) staged through LDS -- see the comment at `bload` below.  One operand buffer each
// (52 KiB with the row dots: two workgroups per CU), two barriers per K step.  Measured (same box, 1024 x [196, 196] x
// [196, 768]): 1.04 vs 1.15 ms, the c4 shape 0.91 vs 1.01 -- the 4-byte fragment loads were a tenth of the time, not the
// bulk of it.  What is left is traffic: per matrix W is read twice (operand + residual; 300 MB of W are in flight
// across the chip, the second read is not an L2 hit), the [n, n] factor once per 96-column group (8 x 154 KB), the
// gradient written once -- ~3 GB per launch at 3 - 4 TB/s.  Fewer, wider groups (12 waves, one workgroup per CU) would
// halve the factor's re-reads; keeping it resident needs 154 KiB of bf16 planes.
template <int MT, typename TO>      // MT = m tiles (16 rows each) >= ceil(n / 16)
__global__ __launch_bounds__(PB_THREADS) __attribute__((amdgpu_waves_per_eu(3, 3))) void procrustes_bwd_side_lds_kernel(
    const float* __restrict__ fac, const float* __restrict__ w, const float* __restrict__ a,
    const float* __restrict__ gl, int n, int d, TO* __restrict__ out, float* __restrict__ rowdot) {
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int PLANE = MT * 16 * PB_ROWB;                 // one bf16 plane of one K step
  constexpr int ITEMS = (MT * 16 * 4 + PB_THREADS - 1) / PB_THREADS;         // staging items (row, k octet) per thread and K step
  unsigned char* abuf = smem;                              // [2 planes][MT * 16 rows][PB_ROWB]   (ONE buffer)
  unsigned char* bbuf = smem + 2 * PLANE;                  // [2 planes][32 k rows][PB_BPITCH]: the W tile of a K step
  float* s_dotw = reinterpret_cast<float*>(smem + 2 * PLANE + 2 * 32 * PB_BPITCH);  // [waves][MT * 16] row dots
  float* s_c = s_dotw + PB_WAVES * MT * 16;                           // [MT * 16] row scales 2 gl sqrt(a)
  float* s_park = reinterpret_cast<float*>(abuf);              // [waves][16 rows][16 columns] epilogue tile: the operand
                                                               // buffers are idle then (80 KiB per workgroup: two per CU)
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* A = fac + (size_t)b * n * n;
  const float* W = w + (size_t)b * n * d;
  TO* O = out + (size_t)b * n * d;
  const int ksteps = (n + 31) >> 5;
  const int strips = d >> 4;
  const float c2 = 2.f * gl[b];
  for (int i = tid; i < MT * 16; i += PB_THREADS) {
#pragma unroll
    for (int wv_ = 0; wv_ < PB_WAVES; ++wv_) s_dotw[wv_ * MT * 16 + i] = 0.f;
    s_c[i] = i < n ? c2 * __builtin_amdgcn_sqrtf(a[(size_t)b * n + i]) : 0.f;
  }

  // Staging of one K step of A in two halves: the global loads (branch-free: rows / columns beyond n read a clamped,
  // valid address and are zeroed by a select) are issued BEFORE the MFMAs of the current step, the split into the two
  // bf16 planes and the LDS writes follow them -- the L2 round trip hides under the matrix work.
  // n % 4 == 0 and a 16-byte aligned factor (checked by the C entry): every load is 16 bytes wide.  Offsets are
  // computed once per item = (row, k octet); a K step only adds a uniform base.  Loads that would leave the matrix
  // (rows >= n of the padded m tiles, the columns >= n of the last K step) are clamped to its last 16 bytes and the
  // values zeroed by a mask -- applied in the last K step / for the padded rows only.
  float4 sa[ITEMS][2];
  unsigned sa_off[ITEMS];
  const unsigned a_last = (unsigned)n * (unsigned)n - 4u;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int item = tid + PB_THREADS * it;
    const int row = item >> 2, oct = item & 3;
    sa_off[it] = (unsigned)(row < n ? row : n - 1) * (unsigned)n + (unsigned)(oct * 8);
  }
  auto stage_load = [&](int ks) {
    const unsigned kbase = (unsigned)ks * 32u;
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const unsigned o0 = sa_off[it] + kbase, o1 = o0 + 4u;
      sa[it][0] = *reinterpret_cast<const float4*>(A + (o0 < a_last ? o0 : a_last));
      sa[it][1] = *reinterpret_cast<const float4*>(A + (o1 < a_last ? o1 : a_last));
    }
  };
  auto stage_store = [&](int ks, int buf) {
    unsigned char* dst = abuf;
    const bool tail = (ks + 1) * 32 > n;                   // uniform
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
      const int item = tid + PB_THREADS * it;
      const int row = item >> 2, oct = item & 3;
      if (item < MT * 16 * 4) {
        float4 v0 = sa[it][0], v1 = sa[it][1];
        const int k0 = ks * 32 + oct * 8;
        const float m0 = (row < n && (!tail || k0 + 4 <= n)) ? 1.f : 0.f;
        const float m1 = (row < n && (!tail || k0 + 8 <= n)) ? 1.f : 0.f;
        uint4 hi, mid;
        pb_split2(v0.x * m0, v0.y * m0, hi.x, mid.x); pb_split2(v0.z * m0, v0.w * m0, hi.y, mid.y);
        pb_split2(v1.x * m1, v1.y * m1, hi.z, mid.z); pb_split2(v1.z * m1, v1.w * m1, hi.w, mid.w);
        const unsigned lo = (unsigned)row * PB_ROWB + (unsigned)oct * 16;
        *reinterpret_cast<uint4*>(dst + lo) = hi;
        *reinterpret_cast<uint4*>(dst + PLANE + lo) = mid;
      }
    }
  };

  for (int g0 = 0; g0 < strips; g0 += PB_WAVES) {          // column groups of PB_WAVES strips (96 columns), one per wave
    const int s0 = g0 + wave;
    const bool has = s0 < strips;
    const int col = (lane & 15);
    const unsigned cs = has ? (unsigned)(s0 * 16 + col) : (unsigned)col;                // 32-bit offsets: n d < 2^31
    // The W tile of a K step ([32 k rows][96 columns] fp32, 12 KiB) goes global -> registers (16-byte loads, 384
    // contiguous bytes per row: two per thread) -> bf16 (hi, mid) planes in LDS, row-major with the k rows PB_BPITCH
    // bytes apart; a wave's B fragment -- column 16 wave + (lane & 15), rows 8 (lane >> 4) + 0..7 -- comes out of the
    // transposing read ds_read_b64_tr_b16.  (The version above fetched the fragment with eight 4-byte loads per lane,
    // 64-byte segments: 2.2 TB/s of such requests was all the 1.08 ms it took.)
    const int c0 = g0 * 16;                                // first column of the group
    float4 sb[2];
    auto bload = [&](int ks) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int item = tid + PB_THREADS * it;            // 768 items = 32 rows x 24 quads
        const int row = item / 24, q = item - row * 24;
        const int kr = ks * 32 + row, cc = c0 + 4 * q;
        const bool ok = kr < n && cc < d;
        sb[it] = *reinterpret_cast<const float4*>(W + (size_t)(ok ? kr : 0) * d + (ok ? cc : 0));
        if (!ok) sb[it] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto bstore = [&]() {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int item = tid + PB_THREADS * it;
        const int row = item / 24, q = item - row * 24;
        uint2 hi, mid;
        pb_split2(sb[it].x, sb[it].y, hi.x, mid.x); pb_split2(sb[it].z, sb[it].w, hi.y, mid.y);
        const unsigned lo = (unsigned)row * PB_BPITCH + (unsigned)q * 8;
        *reinterpret_cast<uint2*>(bbuf + lo) = hi;
        *reinterpret_cast<uint2*>(bbuf + 32 * PB_BPITCH + lo) = mid;
      }
    };
    __syncthreads();                                       // previous group's fragment reads are done
    stage_load(0);
    bload(0);
    stage_store(0, 0);
    bstore();
    __syncthreads();
    pb_f32x4 acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = (pb_f32x4){0.f, 0.f, 0.f, 0.f};
    for (int ks = 0; ks < ksteps; ++ks) {
      const bool more = ks + 1 < ksteps;
      if (more) {                                          // next step's operands: loads only, stored after the MFMAs
        bload(ks + 1);
        stage_load(ks + 1);
      }
      pb_bf16x8 bh, bm;
      {
        const int li = lane & 15, qq = li >> 2, pp = li & 3;
        const unsigned char* b0 = bbuf + (size_t)((lane >> 4) * 8 + qq) * PB_BPITCH + (size_t)(wave * 16 + 4 * pp) * 2;
        const pb_v4s h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pb_lds_v4s*)b0);
        const pb_v4s h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pb_lds_v4s*)(b0 + 4 * PB_BPITCH));
        const pb_v4s m0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pb_lds_v4s*)(b0 + 32 * PB_BPITCH));
        const pb_v4s m1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((pb_lds_v4s*)(b0 + 32 * PB_BPITCH + 4 * PB_BPITCH));
        bh = (pb_bf16x8){h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
        bm = (pb_bf16x8){m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2], m1[3]};
      }
      // A fragments in two batches of (MT + 1) / 2 tiles: all 16-byte fragment reads of a batch are issued together (one
      // exposed LDS latency per batch; a fence per tile pair exposed it MT / 2 times per K step: 4.8 k cycles per step
      // measured with in-kernel stamps against 0.6 k of matrix work), the fence between the batches keeps the
      // scheduler from hoisting the second batch on top of the first (registers)
      const unsigned char* ap = abuf + (size_t)(lane & 15) * PB_ROWB + (lane >> 4) * 16;
      constexpr int HB = (MT + 1) / 2;
#pragma unroll
      for (int hb = 0; hb < 2; ++hb) {
        pb_bf16x8 fh[HB], fm[HB];
#pragma unroll
        for (int j = 0; j < HB; ++j) {
          const int i = hb * HB + j < MT ? hb * HB + j : MT - 1;
          fh[j] = *reinterpret_cast<const pb_bf16x8*>(ap + (size_t)i * 16 * PB_ROWB);
          fm[j] = *reinterpret_cast<const pb_bf16x8*>(ap + PLANE + (size_t)i * 16 * PB_ROWB);
        }
#pragma unroll
        for (int j = 0; j < HB; ++j) {
          const int i = hb * HB + j;
          if (i < MT) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[j], bh, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[j], bm, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fm[j], bh, acc[i], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();                                     // every wave has read the operands of step ks
      if (more) { stage_store(ks + 1, 0); bstore(); }
      __syncthreads();                                     // step ks + 1 is complete
    }
    // ---- epilogue.  acc[i][r] = P at row 16 i + 4 (lane >> 4) + r, column 16 s0 + (lane & 15): one value per lane
    // and row -- consumed in that layout the residual costs a 4-byte load, a 4-byte store and a 16-lane reduction per
    // element group (measured: 0.8 of 1.2 ms).  Instead the wave parks one 16 x 16 tile at a time in its own 1 KiB of
    // LDS and reads it back row-wise: 16 bytes per lane, 64 contiguous bytes per row segment for the W load and the
    // store, a 4-lane reduction per row.  All W loads of the strip are issued before the first tile is processed.
    float* park = s_park + wave * 256;
    const int g4 = lane >> 4;
    const int prow = lane >> 2, pch = lane & 3;            // row-wise walk: 16 rows x 4 chunks of 4 columns
    const int pcol = s0 * 16 + pch * 4;
    constexpr int H0 = (MT + 1) / 2;                       // tiles per batch: the W loads of a batch are in flight together
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      constexpr int HN = H0;
      float4 wq[HN];
#pragma unroll
      for (int j = 0; j < HN; ++j) {
        const int i = hb * H0 + j;
        const int row = i * 16 + prow;
        const bool ok = has && row < n && i < MT;
        wq[j] = *reinterpret_cast<const float4*>(W + (unsigned)(ok ? row : 0) * (unsigned)d + (unsigned)(ok ? pcol : 0));
      }
#pragma unroll
      for (int j = 0; j < HN; ++j) {
        const int i = hb * H0 + j;
        if (i < MT) {
#pragma unroll
          for (int r = 0; r < 4; ++r) park[(g4 * 4 + r) * 16 + col] = acc[i][r];
          // wave-private tile: the wave's own LDS writes precede its reads (in order), no barrier
          const float4 pv = *reinterpret_cast<const float4*>(park + prow * 16 + pch * 4);
          const int row = i * 16 + prow;
          const bool ok = has && row < n;
          const unsigned off = (unsigned)(ok ? row : 0) * (unsigned)d + (unsigned)(ok ? pcol : 0);
          const float4 wv = wq[j];
          const float cr = s_c[row];
          const float4 rv = make_float4(wv.x - pv.x, wv.y - pv.y, wv.z - pv.z, wv.w - pv.w);
          float dot = ok ? fmaf(rv.x, wv.x, fmaf(rv.y, wv.y, fmaf(rv.z, wv.z, rv.w * wv.w))) : 0.f;
          if (ok) {
            if constexpr (sizeof(TO) == 4) {
              *reinterpret_cast<float4*>(O + off) = make_float4(cr * rv.x, cr * rv.y, cr * rv.z, cr * rv.w);
            } else {
              uint2 o;
              o.x = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)(cr * rv.x)) |
                    ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)(cr * rv.y)) << 16);
              o.y = (unsigned)__builtin_bit_cast(unsigned short, (__bf16)(cr * rv.z)) |
                    ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)(cr * rv.w)) << 16);
              *reinterpret_cast<uint2*>(O + off) = o;
            }
          }
          dot += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dot), 0xB1, 0xF, 0xF, true));     // quad_perm 1,0,3,2
          dot += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(dot), 0x4E, 0xF, 0xF, true));     // quad_perm 2,3,0,1
          if (pch == 0) s_dotw[wave * (MT * 16) + row] += c2 * dot;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += PB_THREADS) {
    float t = 0.f;
#pragma unroll
    for (int wv_ = 0; wv_ < PB_WAVES; ++wv_) t += s_dotw[wv_ * MT * 16 + i];
    rowdot[(size_t)b * n + i] = t;
  }
}

// g_a = (dot_s + dot_t) / (2 a)
__global__ __launch_bounds__(256) void procrustes_ga_kernel(const float* __restrict__ dot_s, const float* __restrict__ dot_t,
                                                           const float* __restrict__ a, int64_t total,
                                                           float* __restrict__ g_a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < total) g_a[i] = (dot_s[i] + dot_t[i]) / (2.0f * a[i]);
}

template <typename TO>
static int launch_side(const float* fac, const float* w, const float* a, const float* gl, int batch, int n, int d,
                       TO* out, float* rowdot, hipStream_t st) {
  const int mt = (n + 15) / 16;
  // W through LDS (BASD_PBWD_LDSB=0: straight from global memory as in the first version, A/B timing); the parked
  // epilogue tiles (6 KiB) alias the A buffer: MT >= 4 rows of tiles
  const char* env = getenv("BASD_PBWD_LDSB");
  const bool ldsb = !(env && env[0] == '0') && mt >= 4;
#define BASD_PB_LAUNCH(MT)                                                                                   \
  do {                                                                                                       \
    const size_t lds = (size_t)4 * MT * 16 * PB_ROWB + (size_t)MT * 16 * 4 * (PB_WAVES + 1);       /* MT >= 2: the parked tiles fit */                                  \
    const size_t lds2 = (size_t)2 * MT * 16 * PB_ROWB + (size_t)2 * 32 * PB_BPITCH + (size_t)MT * 16 * 4 * (PB_WAVES + 1);  \
    if (ldsb) {                                                                                              \
      allow_full_lds((const void*)procrustes_bwd_side_lds_kernel<MT, TO>);                                   \
      hipLaunchKernelGGL((procrustes_bwd_side_lds_kernel<MT, TO>), dim3(batch), dim3(PB_THREADS), lds2, st, fac, w, a, gl, n, \
                         d, out, rowdot);                                                                    \
    } else {                                                                                                 \
      allow_full_lds((const void*)procrustes_bwd_side_kernel<MT, TO>);                                       \
      hipLaunchKernelGGL((procrustes_bwd_side_kernel<MT, TO>), dim3(batch), dim3(PB_THREADS), lds, st, fac, w, a, gl, n, d, \
                         out, rowdot);                                                                       \
    }                                                                                                        \
  } while (0)
  if (mt <= 2) BASD_PB_LAUNCH(2);
  else if (mt <= 4) BASD_PB_LAUNCH(4);
  else if (mt <= 8) BASD_PB_LAUNCH(8);
  else if (mt <= 13) BASD_PB_LAUNCH(13);
  else BASD_PB_LAUNCH(16);
#undef BASD_PB_LAUNCH
  return check_launch("procrustes_bwd (fused residual product)");
}

}  // namespace basd

extern "C" int basd_procrustes_bwd_side(const float* fac, const float* w, const float* a, const float* gl, int batch,
                                        int n, int d, void* out, int out_dtype, float* rowdot, void* stream) {
  using namespace basd;
  if (batch <= 0) return BASD_OK;
  if (n < 4 || n > 256 || n % 4 || d < 16 || d % 16 || (((uintptr_t)fac) & 15) || (((uintptr_t)w) & 15) || (((uintptr_t)out) & 15))
    return fail(BASD_ERR_SHAPE, "procrustes_bwd_side: need 4 <= n <= 256, n %% 4 == 0, d %% 16 == 0, 16-byte aligned "
                                "buffers (n=%d d=%d)", n, d);
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == BASD_DTYPE_F32) return launch_side<float>(fac, w, a, gl, batch, n, d, (float*)out, rowdot, st);
  if (out_dtype == BASD_DTYPE_BF16)
    return launch_side<unsigned short>(fac, w, a, gl, batch, n, d, (unsigned short*)out, rowdot, st);
  return fail(BASD_ERR_DTYPE, "procrustes_bwd_side: out dtype %d", out_dtype);
}

extern "C" int64_t basd_procrustes_bwd_workspace_bytes(int batch, int n) { return (int64_t)2 * batch * n * 4; }

extern "C" int basd_procrustes_bwd(const float* s_w, const float* t_w, const float* a, const float* gl,
                                   const float* fac_s, const float* a_t, int batch, int n, int d_s, int d_t,
                                   void* g_s, int g_s_dtype, float* g_t, float* g_a, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
  using namespace basd;
  if (batch <= 0) return BASD_OK;
  if (workspace == nullptr || workspace_bytes < basd_procrustes_bwd_workspace_bytes(batch, n))
    return fail(BASD_ERR_SHAPE, "procrustes_bwd: workspace of %lld bytes, need %lld", (long long)workspace_bytes,
                (long long)basd_procrustes_bwd_workspace_bytes(batch, n));
  float* dot_s = (float*)workspace;
  float* dot_t = dot_s + (size_t)batch * n;
  int rc = basd_procrustes_bwd_side(a_t, t_w, a, gl, batch, n, d_t, g_t, BASD_DTYPE_F32, dot_t, stream);
  if (rc) return rc;
  if (n <= d_s) {                          // token side: fac_s = a_s [n, n], t_w G^T = a_s s_w
    rc = basd_procrustes_bwd_side(fac_s, s_w, a, gl, batch, n, d_s, g_s, g_s_dtype, dot_s, stream);
  } else {                                 // feature side: fac_s = t_w G^T [n, d_s] was formed in the forward
    rc = basd_procrustes_bwd_rows(fac_s, s_w, a, gl, (int64_t)batch * n, n, d_s, g_s, g_s_dtype, dot_s, stream);
  }
  if (rc) return rc;
  const int64_t total = (int64_t)batch * n;
  hipLaunchKernelGGL(procrustes_ga_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dot_s, dot_t, a, total, g_a);
  return check_launch("procrustes_bwd (g_a)");
}
