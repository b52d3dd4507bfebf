// ViT weight-gradient GEMM on the bf16 matrix cores:
//     dW[n][k] += sum_m dY[m][n] * X[m][k]        db[n] += sum_m dY[m][n]
// dY [M, N], X [M, K] bf16 row-major (M = B * tokens ~ 50k, N, K <= 768), dW / db fp32.
// This is the backward of every student nn.Linear (timm ViT blocks; reference
// src/training/trainer.py:157 via autograd).  The library path picks a 12-workgroup
// launch without split-K for these shapes (68 TFLOP/s, 0.22 ms each, 48 per step).
//
// The reduction dimension M is split over workgroups (grid.z slices); both operands are
// read ROW-major exactly once per tile column/row (coalesced 16-byte loads) and the MFMA
// fragments, which need 8 consecutive m for a fixed n (or k), come out of LDS through the
// gfx950 transposing read ds_read_b64_tr_b16 -- no transposed copy of the activations.
// Workgroup = 256 threads = 4 waves; output tile 64 (n) x TK (k, up to 192), wave w owns a
// quarter of the k tiles and all four n tiles; per 64-row chunk: 2 k-steps of
// v_mfma_f32_16x16x32_bf16 per (n, k) tile.  Partial results are added with fp32 atomics (dW / db
// zero-initialised by the caller); the atomics are 27 % of the kernel (ablation), 512 workgroups
// is the measured optimum between that tail and load parallelism.
#include "basd_common.h"

namespace basd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4s lds_v4s;

constexpr int WG_TN = 64;        // output rows (n) per workgroup
constexpr int WG_MC = 64;        // m rows per chunk
constexpr int DY_LD = 72;        // LDS row stride of the dY chunk, in bf16 (144 B)

__device__ __forceinline__ bf16x8 tr_frag(const unsigned short* tile, int ld, int row0, int col0, int lane) {
  // 8 consecutive rows (row0 .. row0+7) of column col0 + (lane & 15): two 4-row transposing reads
  const int i = lane & 15, q = i >> 2, p = i & 3;
  const unsigned short* a0 = tile + (row0 + q) * ld + col0 + 4 * p;
  const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)a0);
  const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s*)(a0 + 4 * ld));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <int KT>   // k tiles of 16 per workgroup: TK = 16 * KT
__global__ __launch_bounds__(256) void wgrad_bf16_kernel(const unsigned short* __restrict__ dy,
                                                         const unsigned short* __restrict__ x, int64_t M, int N,
                                                         int K, float* __restrict__ dw, float* __restrict__ db,
                                                         int64_t rows_per_slice, int tn, int tk, int slices) {
  constexpr int TK = 16 * KT;
  constexpr int X_LD = TK + 8;                       // LDS row stride of the X chunk, in bf16
  constexpr int DY_VEC = WG_MC * WG_TN / 8 / 256;    // uint4 loads per thread (2)
  constexpr int X_VEC = (WG_MC * TK / 8 + 255) / 256;
  __shared__ __align__(16) unsigned short dYs[WG_MC * DY_LD];
  __shared__ __align__(16) unsigned short Xs[WG_MC * X_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware 1-D grid: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so the tn * tk tiles
  // that re-read ONE M-slab of X / dY must share blockIdx.x % 8 to hit the same L2.  Linear id = ((slice / 8) * tiles
  // + tile) * 8 + slice % 8: the tiles of a slab are also adjacent in dispatch order.  (With the natural (n, k, slice)
  // grid the 12 n-tiles of a slab sat on different XCDs and every one fetched it again: 3x the algorithmic HBM bytes.)
  const int tiles = tn * tk;
  const int lin = blockIdx.x;
  const int slice = (lin / (8 * tiles)) * 8 + (lin & 7);
  if (slice >= slices) return;
  const int tile = (lin >> 3) % tiles;
  const int bx = tile % tn, by = tile / tn;
  const int n0 = bx * WG_TN, k0 = by * TK;
  const int64_t m_begin = (int64_t)slice * rows_per_slice;
  int64_t m_end = m_begin + rows_per_slice;
  if (m_end > M) m_end = M;

  // wave w owns the k tiles [KW w, KW w + KW) and ALL four n tiles: 4 + KW fragment reads per 32-row step for
  // 4 KW MFMAs.  (Owning one n tile and all KT k tiles needs 1 + KT reads for the same KT MFMAs: every wave then
  // re-reads the whole X chunk and the kernel is LDS-read bound, 830 cycles of ds_read_b64_tr per chunk against
  // 384 cycles of MFMA.)
  constexpr int KW = KT / 4;
  f32x4 acc[4][KW];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < KW; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  const bool do_bias = (db != nullptr) && (by == 0);

  uint4 rdy[DY_VEC], rx[X_VEC];
  auto load_chunk = [&](int64_t m0) {
#pragma unroll
    for (int v = 0; v < DY_VEC; ++v) {
      const int idx = tid + 256 * v;
      const int row = idx >> 3, c8 = idx & 7;
      const int64_t m = m0 + row;
      rdy[v] = (m < m_end) ? *reinterpret_cast<const uint4*>(dy + m * N + n0 + c8 * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < X_VEC; ++v) {
      const int idx = tid + 256 * v;
      const int row = idx / (TK / 8), c8 = idx - row * (TK / 8);
      const int64_t m = m0 + row;
      rx[v] = (row < WG_MC && m < m_end) ? *reinterpret_cast<const uint4*>(x + m * K + k0 + c8 * 8)
                                         : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int v = 0; v < DY_VEC; ++v) {
      const int idx = tid + 256 * v;
      const int row = idx >> 3, c8 = idx & 7;
      *reinterpret_cast<uint4*>(dYs + row * DY_LD + c8 * 8) = rdy[v];
    }
#pragma unroll
    for (int v = 0; v < X_VEC; ++v) {
      const int idx = tid + 256 * v;
      const int row = idx / (TK / 8), c8 = idx - row * (TK / 8);
      if (row < WG_MC) *reinterpret_cast<uint4*>(Xs + row * X_LD + c8 * 8) = rx[v];
    }
  };

  if (m_begin < m_end) load_chunk(m_begin);
  for (int64_t m0 = m_begin; m0 < m_end; m0 += WG_MC) {
    __syncthreads();                 // previous chunk's fragment reads are done
    store_chunk();
    __syncthreads();
    if (m0 + WG_MC < m_end) load_chunk(m0 + WG_MC);     // prefetch the next chunk under the MFMAs
#pragma unroll
    for (int ks = 0; ks < WG_MC / 32; ++ks) {
      const int row0 = ks * 32 + 8 * (lane >> 4);
      bf16x8 a[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = tr_frag(dYs, DY_LD, row0, i * 16, lane);
#pragma unroll
      for (int t = 0; t < KW; ++t) {
        const bf16x8 b = tr_frag(Xs, X_LD, row0, (wave * KW + t) * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i][t], 0, 0, 0);
      }
    }
    if (do_bias && tid < WG_TN) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < WG_MC; ++r) s += bf16_bits_to_f32(dYs[r * DY_LD + tid]);
      bsum += s;
    }
  }
  // C/D layout: col = lane & 15 (k), row = (lane >> 4) * 4 + reg (n)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < KW; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int n = n0 + i * 16 + (lane >> 4) * 4 + reg;
        const int k = k0 + (wave * KW + t) * 16 + (lane & 15);
        atomicAdd(&dw[(size_t)n * K + k], acc[i][t][reg]);
      }
  if (do_bias && tid < WG_TN) atomicAdd(&db[n0 + tid], bsum);
}

template <int KT>
static void launch_wgrad(const void* dy, const void* x, int64_t M, int N, int K, float* dw, float* db,
                         hipStream_t st) {
  const int tn = N / WG_TN, tk = K / (16 * KT);
  int64_t slices = 512 / (tn * tk);
  const int64_t max_slices = (M + WG_MC - 1) / WG_MC;
  if (slices < 1) slices = 1;
  if (slices > max_slices) slices = max_slices;
  int64_t rps = (M + slices - 1) / slices;
  rps = (rps + WG_MC - 1) / WG_MC * WG_MC;
  slices = (M + rps - 1) / rps;
  const int groups = (int)((slices + 7) / 8);
  hipLaunchKernelGGL((wgrad_bf16_kernel<KT>), dim3(groups * tn * tk * 8), dim3(256), 0, st,
                     (const unsigned short*)dy, (const unsigned short*)x, M, N, K, dw, db, rps, tn, tk, (int)slices);
}

}  // namespace basd

extern "C" int basd_wgrad_bf16(const void* dy, const void* x, int64_t M, int N, int K, float* dw, float* db,
                               void* stream) {
  using namespace basd;
  if (M <= 0) return BASD_OK;
  if (N % WG_TN || K % 64 || N < WG_TN || K < 64)
    return fail(BASD_ERR_SHAPE, "wgrad_bf16: need N %% 64 == 0 and K %% 64 == 0 (got N=%d K=%d)", N, K);
  hipStream_t st = (hipStream_t)stream;
  if (K % 192 == 0) launch_wgrad<12>(dy, x, M, N, K, dw, db, st);
  else if (K % 128 == 0) launch_wgrad<8>(dy, x, M, N, K, dw, db, st);
  else launch_wgrad<4>(dy, x, M, N, K, dw, db, st);
  return check_launch("wgrad_bf16");
}
