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


// ---------------------------------------------------------------------------------------------
// 192 x 192 output tile, LDS-DMA ring (every width of the DeiT / ViT students is a multiple of 192).
// The kernel above keeps ONE 64-row chunk in flight per workgroup through registers: with two workgroups per CU its
// iteration time is the loaded-machine load latency (~1.5 us per chunk against 0.3 us of matrix work; MFMA busy 22 %),
// and the 64-wide n tile re-reads X once per n tile.  Here a workgroup (512 threads, 8 waves = 4 (n) x 2 (k), 18
// accumulator tiles per wave) owns a whole 192 x 192 tile of dW for its M-slab; the 64-row chunks of dY and X
// (24 KiB each) go global -> LDS with global_load_lds_dwordx4 into a ring of THREE 48 KiB stages, two chunks (96 KiB)
// in flight under the MFMAs of the third, counted s_waitcnt vmcnt(6) + one raw s_barrier per chunk.  The LDS image is
// row-major [64 m][192] (384-byte rows: what ds_read_b64_tr_b16 needs to deliver 8 consecutive m per lane) with the
// 32-byte column groups XOR-swizzled by ((m >> 1) & 1) | (((m >> 3) & 1) << 1): the 16 rows x 32 B of one transposing
// read then cover all 64 banks twice (unswizzled: 16 banks, 8 cycles instead of 2).  M must be a multiple of 64 here
// (LDS-DMA cannot predicate a lane to zero, and a select between the operand and a zero page costs the source
// pointers their noalias scope -- the compiler then drains vmcnt before EVERY LDS read); the launcher hands the
// M % 64 tail rows to the kernel above.  db: one more MFMA per n tile against an all-ones B fragment
// (column sums on the matrix core; no scalar LDS walk).  fp32 atomics into the flat gradient buffer as before.
constexpr int W2_T = 192, W2_MC = 64, W2_ROWB = W2_T * 2, W2_OP = W2_MC * W2_ROWB, W2_STAGE = 2 * W2_OP, W2_NSTAGE = 3;
constexpr int W2_PART4 = W2_T * W2_T / 4;      // float4 per partial tile (9216)
constexpr int W2_SG = 8;                       // slice groups of the reduction (8-way atomics at its end)

__global__ __launch_bounds__(512) void wgrad192_bf16_kernel(const unsigned short* __restrict__ dy,
                                                            const unsigned short* __restrict__ x, int64_t M, int N,
                                                            int K, float* __restrict__ dw, float* __restrict__ db,
                                                            int64_t rows_per_slice, int tn, int tk, int slices,
                                                            float* __restrict__ ws) {
  extern __shared__ __align__(16) unsigned char w_lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles = tn * tk;
  const int lin = blockIdx.x;
  const int slice = (lin / (8 * tiles)) * 8 + (lin & 7);      // XCD-aware: the tiles of one M-slab share blockIdx % 8
  if (slice >= slices) return;
  const int tile = (lin >> 3) % tiles;
  const int n0 = (tile % tn) * W2_T, k0 = (tile / tn) * W2_T;
  const int64_t m_begin = (int64_t)slice * rows_per_slice;
  int64_t m_end = m_begin + rows_per_slice;
  if (m_end > M) m_end = M;
  const int nchunks = (int)((m_end - m_begin + W2_MC - 1) / W2_MC);
  if (nchunks <= 0) return;
  const int wn = wave >> 1, wk = wave & 1;                    // wave tile: n [48 wn, +48), k [96 wk, +96)

  // ---- LDS-DMA pieces: piece j of this wave is 1 KiB number wave + 8 j of the stage (0..23 dY, 24..47 X); lane
  //      writes physical byte o = q * 1024 + lane * 16 of the operand image and fetches the logical (unswizzled) bytes
  const unsigned short* src[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int q = (wave + 8 * j) % 24;
    const int o = q * 1024 + lane * 16;
    const int row = o / W2_ROWB, pb = o - row * W2_ROWB;
    const int f = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    const int lb = pb ^ (f << 5);
    src[j] = (j < 3) ? dy + (m_begin + row) * N + n0 + (lb >> 1) : x + (m_begin + row) * K + k0 + (lb >> 1);
  }
  auto issue_piece = [&](int j, int chunk) {
    const unsigned short* g = src[j] + (int64_t)chunk * W2_MC * (j < 3 ? N : K);
    unsigned char* dst = w_lds + (chunk % W2_NSTAGE) * W2_STAGE + (wave + 8 * j) * 1024;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                     (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
  };

  f32x4 acc[3][6], accb[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 6; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bool do_bias = (db != nullptr) && (k0 == 0) && (wk == 0);
  const bf16x8 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

  // fragment addressing (lane constants): rows 8 (lane >> 4) + q (+ 4), 8-byte column chunk p, swizzle f
  const int fq = (lane & 15) >> 2, fp = lane & 3;
  const int ff = (((fq >> 1) & 1) | (((lane >> 4) & 1) << 1)) << 5;
  const int frow = (8 * (lane >> 4) + fq) * W2_ROWB;
  int a_off[3], b_off[6];
#pragma unroll
  for (int i = 0; i < 3; ++i) a_off[i] = frow + ((((wn * 48 + i * 16) * 2) + 8 * fp) ^ ff);
#pragma unroll
  for (int t = 0; t < 6; ++t) b_off[t] = W2_OP + frow + ((((wk * 96 + t * 16) * 2) + 8 * fp) ^ ff);
  // The transposing reads are issued as inline asm: behind the builtin the compiler drains vmcnt before EVERY such
  // read once an LDS-DMA is in flight (it cannot tell the ring stages apart), which serialises the whole pipeline.
  // The asm has no memory operand; ordering is by program order (volatile) and the waits below are explicit.
  const unsigned lds0 = (unsigned)(uintptr_t)(lds_v4s*)w_lds;
  auto tr_lo = [](unsigned addr) {
    v4s v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
  };
  auto tr_hi = [](unsigned addr) {            // + 4 rows
    v4s v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1536" : "=v"(v) : "v"(addr) : "memory");
    return v;
  };
  static_assert(4 * W2_ROWB == 1536, "offset of the second half of a fragment");

  // prologue: chunks 0 and 1
#pragma unroll
  for (int j = 0; j < 6; ++j) issue_piece(j, 0);
  if (nchunks > 1) {
#pragma unroll
    for (int j = 0; j < 6; ++j) issue_piece(j, 1);
  }
  for (int c = 0; c < nchunks; ++c) {
    if (c + 1 < nchunks) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // chunk c landed; c + 1 may be in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // everyone's pieces of chunk c; everyone is done reading chunk c - 1
    const bool more = c + 2 < nchunks;       // chunk c + 2 goes into the stage chunk c - 1 used
    const unsigned base = lds0 + (c % W2_NSTAGE) * W2_STAGE;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned kb = base + ks * 32 * W2_ROWB;
      v4s al[3], ah[3], bl[6], bh[6];
#pragma unroll
      for (int i = 0; i < 3; ++i) { al[i] = tr_lo(kb + a_off[i]); ah[i] = tr_hi(kb + a_off[i]); }
#pragma unroll
      for (int t = 0; t < 6; ++t) { bl[t] = tr_lo(kb + b_off[t]); bh[t] = tr_hi(kb + b_off[t]); }
      // 18 reads outstanding, returned in order: the A fragments + B fragment t are there when at most 10 - 2 t remain
      bf16x8 a[3];
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        if (t == 0) {
          asm volatile("s_waitcnt lgkmcnt(10)"
                       : "+v"(al[0]), "+v"(ah[0]), "+v"(al[1]), "+v"(ah[1]), "+v"(al[2]), "+v"(ah[2]), "+v"(bl[0]), "+v"(bh[0]));
#pragma unroll
          for (int i = 0; i < 3; ++i)
            a[i] = (bf16x8){al[i][0], al[i][1], al[i][2], al[i][3], ah[i][0], ah[i][1], ah[i][2], ah[i][3]};
        } else if (t == 1) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(bl[1]), "+v"(bh[1]));
        else if (t == 2) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bl[2]), "+v"(bh[2]));
        else if (t == 3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bl[3]), "+v"(bh[3]));
        else if (t == 4) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bl[4]), "+v"(bh[4]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bl[5]), "+v"(bh[5]));
        const bf16x8 b = {bl[t][0], bl[t][1], bl[t][2], bl[t][3], bh[t][0], bh[t][1], bh[t][2], bh[t][3]};
        __builtin_amdgcn_sched_barrier(0);
        if (more && (t & 1) == 0) issue_piece(ks * 3 + (t >> 1), c + 2);   // 3 pieces per k-step, one per 6 MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b, acc[i][t], 0, 0, 0);
      }
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 3; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], ones, accb[i], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // C/D layout: col = lane & 15 (k), row = (lane >> 4) * 4 + reg (n).  With a workspace the partial tile goes out as
  // plain 16-byte stores in REGISTER order (1 KiB per wave instruction) and wgrad192_reduce_kernel adds the slices up:
  // 256 workgroups x 36 864 fp32 atomics execute at the memory side at ~13 ns per 64-byte line and channel, 31 - 48 us
  // per launch whatever the scope (scripts/atomic_scope_bench.hip) -- four times the matrix work of the kernel.
  if (ws != nullptr) {
    float4* part = reinterpret_cast<float4*>(ws) + ((size_t)tile * slices + slice) * W2_PART4;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int t = 0; t < 6; ++t)
        part[(wave * 18 + i * 6 + t) * 64 + lane] = make_float4(acc[i][t][0], acc[i][t][1], acc[i][t][2], acc[i][t][3]);
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    if (ws == nullptr) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int n = n0 + wn * 48 + i * 16 + (lane >> 4) * 4 + reg;
          const int k = k0 + wk * 96 + t * 16 + (lane & 15);
          atomicAdd(&dw[(size_t)n * K + k], acc[i][t][reg]);
        }
    }
    if (do_bias && (lane & 15) == 0) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
        atomicAdd(&db[n0 + wn * 48 + i * 16 + (lane >> 4) * 4 + reg], accb[i][reg]);
    }
  }
}

// dw += sum over the slices of the partial tiles (register order: float4 f = (wave * 18 + i * 6 + t) * 64 + lane).
// Block = 256 consecutive float4 of one tile for one of W2_SG slice groups; the groups meet with 8-way atomics.
__global__ __launch_bounds__(256) void wgrad192_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                              int K, int tn, int slices) {
  const int g = blockIdx.x % W2_SG;
  const int fb = (blockIdx.x / W2_SG) % (W2_PART4 / 256);
  const int tile = blockIdx.x / (W2_SG * (W2_PART4 / 256));
  const int f = fb * 256 + threadIdx.x;
  const float4* src = reinterpret_cast<const float4*>(ws) + (size_t)tile * slices * W2_PART4 + f;
  float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
  int sl = g;
  for (; sl + W2_SG < slices; sl += 2 * W2_SG) {
    const float4 a = src[(size_t)sl * W2_PART4], b = src[(size_t)(sl + W2_SG) * W2_PART4];
    s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
    s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
  }
  if (sl < slices) {
    const float4 a = src[(size_t)sl * W2_PART4];
    s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
  }
  const int lane = f & 63, frag = f >> 6;            // frag = wave * 18 + i * 6 + t
  const int wave = frag / 18, it = frag - wave * 18, i = it / 6, t = it - i * 6;
  const int n = (tile % tn) * W2_T + (wave >> 1) * 48 + i * 16 + (lane >> 4) * 4;
  const int k = (tile / tn) * W2_T + (wave & 1) * 96 + t * 16 + (lane & 15);
  float* o = dw + (size_t)n * K + k;
  atomicAdd(o, s0.x + s1.x);
  atomicAdd(o + K, s0.y + s1.y);
  atomicAdd(o + 2 * (size_t)K, s0.z + s1.z);
  atomicAdd(o + 3 * (size_t)K, s0.w + s1.w);
}

template <int KT>
static void launch_wgrad(const void* dy, const void* x, int64_t M, int N, int K, float* dw, float* db, hipStream_t st);

static int64_t wgrad192_slices(int64_t M, int tiles, int64_t* rows_per_slice) {
  const int64_t max_slices = (M + W2_MC - 1) / W2_MC;
  int64_t slices = 256 / tiles;                      // one workgroup (144 KiB of LDS) per CU
  if (slices < 1) slices = 1;
  if (slices > max_slices) slices = max_slices;
  int64_t rps = (M + slices - 1) / slices;
  rps = (rps + W2_MC - 1) / W2_MC * W2_MC;
  *rows_per_slice = rps;
  return (M + rps - 1) / rps;
}

// a single 192 x 192 tile (the attention output projection of DeiT-T) has too little work per slab to pay for the
// ring's fill and the second launch: measured 36 + 6.5 us against 32 us on the register kernel
static bool wgrad192_applies(int64_t M, int N, int K) {
  return N % W2_T == 0 && K % W2_T == 0 && M >= 4096 && (N / W2_T) * (K / W2_T) >= 2;
}

static size_t wgrad192_workspace_bytes(int64_t M, int N, int K) {
  if (!wgrad192_applies(M, N, K)) return 0;
  int64_t rps;
  const int tiles = (N / W2_T) * (K / W2_T);
  const int64_t slices = wgrad192_slices(M / W2_MC * W2_MC, tiles, &rps);
  return (size_t)tiles * (size_t)slices * W2_T * W2_T * sizeof(float);
}

static void launch_wgrad192(const void* dy, const void* x, int64_t M_all, int N, int K, float* dw, float* db,
                            float* ws, hipStream_t st) {
  const int64_t M = M_all / W2_MC * W2_MC;           // whole chunks here, the M % 64 tail rows on the register kernel
  if (M_all > M)
    launch_wgrad<12>((const unsigned short*)dy + M * N, (const unsigned short*)x + M * K, M_all - M, N, K, dw, db, st);
  const int tn = N / W2_T, tk = K / W2_T, tiles = tn * tk;
  int64_t rps;
  const int64_t slices = wgrad192_slices(M, tiles, &rps);
  const int groups = (int)((slices + 7) / 8);
  allow_full_lds((const void*)wgrad192_bf16_kernel);
  hipLaunchKernelGGL(wgrad192_bf16_kernel, dim3(groups * tiles * 8), dim3(512), W2_NSTAGE * W2_STAGE, st,
                     (const unsigned short*)dy, (const unsigned short*)x, M, N, K, dw, db, rps, tn, tk, (int)slices, ws);
  if (ws != nullptr)
    hipLaunchKernelGGL(wgrad192_reduce_kernel, dim3(tiles * (W2_PART4 / 256) * W2_SG), dim3(256), 0, st, ws, dw, K, tn,
                       (int)slices);
}

}  // namespace basd

static int wgrad_dispatch(const void* dy, const void* x, int64_t M, int N, int K, float* dw, float* db, float* ws,
                          void* stream) {
  using namespace basd;
  if (M <= 0) return BASD_OK;
  if (N % WG_TN || K % 64 || N < WG_TN || K < 64)
    return fail(BASD_ERR_SHAPE, "wgrad_bf16: need N %% 64 == 0 and K %% 64 == 0 (got N=%d K=%d)", N, K);
  hipStream_t st = (hipStream_t)stream;
  if (wgrad192_applies(M, N, K)) launch_wgrad192(dy, x, M, N, K, dw, db, ws, st);
  else if (K % 192 == 0) launch_wgrad<12>(dy, x, M, N, K, dw, db, st);
  else if (K % 128 == 0) launch_wgrad<8>(dy, x, M, N, K, dw, db, st);
  else launch_wgrad<4>(dy, x, M, N, K, dw, db, st);
  return check_launch("wgrad_bf16");
}

extern "C" int basd_wgrad_bf16(const void* dy, const void* x, int64_t M, int N, int K, float* dw, float* db,
                               void* stream) {
  return wgrad_dispatch(dy, x, M, N, K, dw, db, nullptr, stream);
}

extern "C" int64_t basd_wgrad_workspace_bytes(int64_t M, int N, int K) {
  return (int64_t)basd::wgrad192_workspace_bytes(M, N, K);
}

extern "C" int basd_wgrad_bf16_ws(const void* dy, const void* x, int64_t M, int N, int K, float* dw, float* db,
                                  void* workspace, int64_t workspace_bytes, void* stream) {
  using namespace basd;
  const int64_t need = (int64_t)wgrad192_workspace_bytes(M, N, K);
  if (need > 0 && (workspace == nullptr || workspace_bytes < need || ((uintptr_t)workspace & 15)))
    return fail(BASD_ERR_SHAPE, "wgrad_bf16_ws: workspace of %lld bytes (16-byte aligned) required, got %lld",
                (long long)need, (long long)workspace_bytes);
  return wgrad_dispatch(dy, x, M, N, K, dw, db, need > 0 ? (float*)workspace : nullptr, stream);
}
