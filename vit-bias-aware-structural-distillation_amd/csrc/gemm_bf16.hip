// bf16 GEMM of the ViT blocks on the bf16 matrix cores, with the bias / GELU epilogue fused:
//     Y[m][n] = epi( sum_k X[m][k] * W[n][k] + bias[n] )          X [M, K], W [N, K], Y [M, N] bf16 row-major
// i.e. nn.Linear.forward of the timm blocks the reference runs (teacher: src/models/teacher.py:212; student:
// src/training/trainer.py:33) and, with W := W^T, their input gradient dX = dY W (trainer.py:157 via autograd).
// Replaces the hipBLASLt launches + the separate GELU pass of the frozen teacher.
//
// Workgroup = 512 threads = 8 waves (2 along M x 4 along N), output tile 256 (m) x BN (n; 256 or 192), K step 64.
//   * both operand tiles go global -> LDS with global_load_lds_dwordx4 (16 B per lane, no VGPR round trip); the LDS
//     image is made of [16 rows][32 k] bf16 sub-tiles of 1 KiB = one wave instruction, XOR-swizzled
//     (byte ^= ((byte >> 9) & 1) << 5) on the SOURCE address and on the fragment read so that ds_read_b128 of 16
//     rows x 16 B spreads over the bank row; two stages: the loads of K step t + 1 fly under the MFMAs of step t;
//   * the MFMA is v_mfma_f32_16x16x32_bf16 with the WEIGHT tile as the A operand and the activation tile as B:
//     the accumulator tile is then [n][m] with 4 consecutive n per lane, so the epilogue packs 4 results into one
//     8-byte store along the output row (with X as A every lane would own 4 different rows);
//   * wave (wm, wn) owns 128 m x BN/4 n = 8 x NT accumulator tiles (128 fp32 VGPRs at BN = 256);
//   * XCD-aware 1-D grid: the BN-tiles of one 256-row block are adjacent workgroups of ONE XCD (they re-read the same
//     activation rows from that XCD's L2); the weight matrix (<= 4.7 MB) streams through every L2.
// GELU is the exact erf form evaluated with Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7 on erf: invisible after the
// bf16 rounding of the output); fp32 accumulation, one rounding to bf16 at the store.
#include <type_traits>
#include "basd_common.h"

namespace basd {

typedef float gf32x4 __attribute__((ext_vector_type(4)));
typedef short gbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short gu16x4 __attribute__((ext_vector_type(4)));

constexpr int GBM = 256;   // tile rows (m)
constexpr int GBK = 64;    // K step

__device__ __forceinline__ int gemm_swz(int byte) { return byte ^ (((byte >> 9) & 1) << 5); }

// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (lin & 7), each with its own 4 MB L2; XCD x owns
// the 256-row blocks mb = x (mod 8).  Inside an XCD the dispatch order is: for every GROUP of `ngroup` column tiles, for
// every row block of the XCD, the ngroup tiles of that block -- so the ~32 workgroups an XCD runs at a time share ONE
// group's weight rows (ngroup x BN x K x 2 bytes, chosen <= 1.6 MB by the launcher) and that slice stays in L2 while
// the row blocks stream past it.  With all column tiles in one group (the round-1/2 order) the whole weight matrix is
// re-streamed for every row block once it exceeds the L2: 789 MB counted on the teacher's fc1 against 392 MB
// algorithmic.  The price is one more read of the activation rows per extra group.
__device__ __forceinline__ void gemm_tile_of(int lin, int tiles_n, int ngroup, int groups, int& mb, int& nb) {
  const int xcd = lin & 7, idx = lin >> 3;
  const int nbi = idx % ngroup;
  const int rest = idx / ngroup;
  const int mbl = rest % groups, ng = rest / groups;
  mb = mbl * 8 + xcd;
  nb = ng * ngroup + nbi;
}

__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f),
                              0.254829592f);
  const float erf_abs = fmaf(-poly, __expf(-z * z), 1.0f);
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// d/dx of the exact-erf GELU: Phi(x) + x phi(x); exp(-x^2 / 2) is the exponential the erf evaluation already needs
__device__ __forceinline__ float gelu_erf_grad(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f),
                              0.254829592f);
  const float e = __expf(-z * z);
  const float erf_abs = fmaf(-poly, e, 1.0f);
  return fmaf(x * 0.3989422804014327f, e, 0.5f * (1.0f + copysignf(erf_abs, x)));
}

__device__ __forceinline__ unsigned short f32_to_bf16_bits(float v) {
  return __builtin_bit_cast(unsigned short, (__bf16)v);
}


// Epilogue shared by both kernels.  The accumulator tile of a wave is [n][m] with 4 consecutive n per lane: stored
// straight from registers that is an 8-byte store per (lane, tile), 16 rows x 32 B per wave instruction -- partial
// lines, 18 us per 256 x 256 tile (as much as 13 K steps).  Instead every wave parks its 128 x (BN / 4) bf16 results
// in a private LDS region (row stride + 16 B against bank conflicts; the operand ring is dead by now) and writes
// them out as 16 bytes per lane, whole 128-byte (BN = 256) row segments per 8 lanes.
template <int BN, int NT, int EPI>
__device__ __forceinline__ void gemm_epilogue(gf32x4 (&acc)[NT][8], unsigned char* lds, const unsigned short* bias,
                                              unsigned short* aux, unsigned short* Y, int M, int N, int m0, int n0,
                                              int wm, int wn, int lane, int wave) {
  constexpr int WN = BN / 4;                       // columns of a wave
  constexpr int ROWB = WN * 2 + 16;                // LDS row stride in bytes
  constexpr int CH = WN * 2 / 16;                  // 16-byte chunks per row
  unsigned char* region = lds + wave * (128 * ROWB);
  const size_t tile_off = (size_t)(m0 + wm * 128) * N + n0 + wn * WN;
  auto flush = [&](unsigned short* dst) {          // region -> global, 16 bytes per lane along the rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 2 * CH; ++it) {
      const int item = it * 64 + lane;
      const int row = item / CH, ch = item - row * CH;
      const uint4 v = *reinterpret_cast<const uint4*>(region + row * ROWB + ch * 16);
      if (m0 + wm * 128 + row < M) *reinterpret_cast<uint4*>(dst + tile_off + (size_t)row * N + ch * 8) = v;
    }
  };
  __builtin_amdgcn_s_barrier();                    // every wave is done with the operand ring
  if (EPI == 4) {                                  // the saved pre-activation tile comes in the way the result goes out
#pragma unroll
    for (int it = 0; it < 2 * CH; ++it) {
      const int item = it * 64 + lane;
      const int row = item / CH, ch = item - row * CH;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (m0 + wm * 128 + row < M) v = *reinterpret_cast<const uint4*>(aux + tile_off + (size_t)row * N + ch * 8);
      *reinterpret_cast<uint4*>(region + row * ROWB + ch * 16) = v;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // wave-private region: no barrier
  }
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    const int nl = i * 16 + 4 * (lane >> 4);       // local column of this lane's 4 values
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if ((EPI == 1 || EPI == 2 || EPI == 3) && bias != nullptr) {
      const gu16x4 b4 = *reinterpret_cast<const gu16x4*>(bias + n0 + wn * WN + nl);
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = bf16_bits_to_f32(b4[r]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      gu16x4* cell = reinterpret_cast<gu16x4*>(region + (j * 16 + (lane & 15)) * ROWB + nl * 2);
      gu16x4 o, pre;
      if (EPI == 4) pre = *cell;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[i][j][r] + bv[r];
        if (EPI == 2) v = gelu_erf(v);
        if (EPI == 4) v *= gelu_erf_grad(bf16_bits_to_f32(pre[r]));
        o[r] = f32_to_bf16_bits(v);
      }
      *cell = o;
    }
  }
  if (EPI == 3) {                                  // pre-activation out (saved for backward), then GELU of the ROUNDED
    flush(aux);                                    // value in place (what nn.GELU sees after a bf16 nn.Linear)
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      const int nl = i * 16 + 4 * (lane >> 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        gu16x4* cell = reinterpret_cast<gu16x4*>(region + (j * 16 + (lane & 15)) * ROWB + nl * 2);
        gu16x4 p = *cell;
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = f32_to_bf16_bits(gelu_erf(bf16_bits_to_f32(p[r])));
        *cell = p;
      }
    }
  }
  flush(Y);
}

// EPI: 0 = none, 1 = + bias, 2 = gelu(+ bias), 3 = aux <- (+ bias), y <- gelu(aux), 4 = y <- acc * gelu'(aux)
template <int BN, int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_nt_kernel(const unsigned short* __restrict__ X,
                                                           const unsigned short* __restrict__ W,
                                                           const unsigned short* __restrict__ bias,
                                                           unsigned short* aux, unsigned short* __restrict__ Y,
                                                           int M, int N, int K, int tiles_n, int mblocks, int ngroup) {
  extern __shared__ __align__(16) unsigned char g_lds[];
  constexpr int A_BYTES = GBM * GBK * 2;          // 32 KiB
  constexpr int B_BYTES = BN * GBK * 2;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int NT = BN / 64;                     // 16-wide n tiles per wave
  constexpr int NSUB_A = (GBM / 16) * 2, NSUB_B = (BN / 16) * 2;
  constexpr int NSUB = NSUB_A + NSUB_B;           // 1 KiB sub-tiles per stage: 64 (BN 256) or 56 (BN 192)
  static_assert(NSUB % 8 == 0, "every wave issues the same number of LDS-DMA pieces");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int mb, nb;
  gemm_tile_of(blockIdx.x, tiles_n, ngroup, (mblocks + 7) >> 3, mb, nb);
  if (mb >= mblocks) return;
  const int m0 = mb * GBM, n0 = nb * BN;
  const int wm = wave >> 2, wn = wave & 3;

  // ---- per-lane source coordinates of an LDS-DMA piece: LDS byte lane * 16 of the sub-tile holds the (swizzled)
  //      logical byte q = row * 64 + c16 * 16
  const int q_dma = gemm_swz(lane * 16);
  const int dma_row = q_dma >> 6, dma_c16 = (q_dma >> 4) & 3;
  const unsigned short* src[NSUB / 8];
#pragma unroll
  for (int i = 0; i < NSUB / 8; ++i) {
    const int s = wave + 8 * i;
    const bool is_a = s < NSUB_A;
    const int sl = is_a ? s : s - NSUB_A;
    const int rb = sl >> 1, kb = sl & 1;
    if (is_a) {
      int m = m0 + rb * 16 + dma_row;
      m = m < M ? m : M - 1;                       // ragged last block: rows beyond M re-read row M - 1 (never stored)
      src[i] = X + (size_t)m * K + kb * 32 + dma_c16 * 8;
    } else {
      const int n = n0 + rb * 16 + dma_row;        // N % BN == 0
      src[i] = W + (size_t)n * K + kb * 32 + dma_c16 * 8;
    }
  }
  auto issue_piece = [&](int i, int stage, int k0) {
    const int s = wave + 8 * i;
    unsigned char* dst = g_lds + stage * STAGE + s * 1024;        // A sub-tiles first, then B: s * 1024 covers both
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src[i] + k0),
                                     (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
  };

  gf32x4 acc[NT][8];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (gf32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offset inside a sub-tile: row lane & 15, 16-byte k chunk lane >> 4
  const int frag_off = gemm_swz((lane & 15) * 64 + (lane >> 4) * 16);
  const int a_sub0 = (wm * 8) * 2;                 // first sub-tile of this wave's activation rows
  const int b_sub0 = NSUB_A + (wn * NT) * 2;       // first sub-tile of this wave's weight rows
  constexpr int NPIECE = NSUB / 8;                 // LDS-DMA pieces per wave and K step (8, 7 or 6)

  const int nk = K / GBK;
#pragma unroll
  for (int i = 0; i < NPIECE; ++i) issue_piece(i, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // the last K step issues nothing: peeled at compile time (a runtime "more" test became a branch per piece)
  auto k_step = [&](int t, auto more_tag) {
    constexpr bool more = decltype(more_tag)::value;
    const int cur = t & 1;
    const int k_next = (t + 1) * GBK;
    const unsigned char* base = g_lds + cur * STAGE + frag_off;
    // One K step = 2 k-blocks of 32; per k-block: the fragment reads, then the NT x 8 MFMAs in groups of 8 with ONE
    // LDS-DMA piece of the next stage issued in front of each group.  An LDS-DMA issue costs the wave 60-180 cycles:
    // issued as one burst of 8 ahead of the MFMAs (first version of this kernel) they added ~1000 cycles to a step
    // whose matrix work is 1024; spread out they hide behind the MFMA groups of the SIMD's other wave.
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      gbf16x8 wf[NT], xf[8];
#pragma unroll
      for (int i = 0; i < NT; ++i)
        wf[i] = *reinterpret_cast<const gbf16x8*>(base + (b_sub0 + i * 2 + kb) * 1024);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        xf[j] = *reinterpret_cast<const gbf16x8*>(base + (a_sub0 + j * 2 + kb) * 1024);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        // group g of the step's 2 NT groups issues the pieces [g NPIECE / G, (g + 1) NPIECE / G): one each at
        // BN = 256, one or two at the narrower tiles
        constexpr int G = 2 * NT;
        const int g = kb * NT + i;
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
          for (int piece = 0; piece < NPIECE; ++piece)
            if (piece >= g * NPIECE / G && piece < (g + 1) * NPIECE / G) issue_piece(piece, cur ^ 1, k_next);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);                  // all MFMAs of this step are issued before the wait below
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next stage has landed (this wave's pieces) ...
    __syncthreads();                                    // ... everybody's, and everybody is done reading this one
  };
  for (int t = 0; t + 1 < nk; ++t) k_step(t, std::true_type{});
  k_step(nk - 1, std::false_type{});

  gemm_epilogue<BN, NT, EPI>(acc, g_lds, bias, aux, Y, M, N, m0, n0, wm, wn, lane, wave);
}

// ---------------------------------------------------------------------------------------------
// 256 x 256 tile, deep pipeline.  With two whole-K-step stages (kernel above) only ONE step of loads (64 KiB) can be
// in flight while the other stage is being multiplied: 0.85 us of matrix work per step against a loaded-machine L2
// latency of 1.5 - 2 us -- the step time measured was 2 us at every K.  Here the 160 KiB of LDS are a ring of FIVE
// 32 KiB half-units (A_0, B_0, A_1, B_1, ...; the A / B tile of one K step each): two are being multiplied, THREE are
// in flight (1.5 K steps, 96 KiB), waits are counted (s_waitcnt vmcnt(4): everything but the youngest unit has
// landed), the barrier is a raw s_barrier (a __syncthreads() would drain the LDS-DMA queue), one barrier per K step.
// Unit u lives in slot u % 5; during step t >= 1 the units 2t+3 and 2t+4 are issued into the slots step t-1 used,
// one 1 KiB piece in front of each group of 8 MFMAs.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_ring_kernel(const unsigned short* __restrict__ X,
                                                             const unsigned short* __restrict__ W,
                                                             const unsigned short* __restrict__ bias,
                                                             unsigned short* aux, unsigned short* __restrict__ Y,
                                                             int M, int N, int K, int tiles_n, int mblocks, int ngroup) {
  extern __shared__ __align__(16) unsigned char g_lds[];
  constexpr int BN = 256, NT = 4, UNIT = 32768, NSLOT = 5;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int mb, nb;
  gemm_tile_of(blockIdx.x, tiles_n, ngroup, (mblocks + 7) >> 3, mb, nb);
  if (mb >= mblocks) return;
  const int m0 = mb * GBM, n0 = nb * BN;
  const int wm = wave >> 2, wn = wave & 3;

  const int q_dma = gemm_swz(lane * 16);
  const int dma_row = q_dma >> 6, dma_c16 = (q_dma >> 4) & 3;
  const unsigned short* src_a[4];
  const unsigned short* src_b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int s = wave + 8 * j;                    // sub-tile of the unit: row block s >> 1, k block s & 1
    int m = m0 + (s >> 1) * 16 + dma_row;
    m = m < M ? m : M - 1;
    src_a[j] = X + (size_t)m * K + (s & 1) * 32 + dma_c16 * 8;
    src_b[j] = W + (size_t)(n0 + (s >> 1) * 16 + dma_row) * K + (s & 1) * 32 + dma_c16 * 8;
  }
  auto issue_piece = [&](int unit, int j) {        // piece j (0..3) of unit `unit` (wave-uniform)
    const unsigned short* g = ((unit & 1) ? src_b[j] : src_a[j]) + (unit >> 1) * GBK;
    unsigned char* dst = g_lds + (unit % NSLOT) * UNIT + (wave + 8 * j) * 1024;
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)g,
                                     (void __attribute__((address_space(3)))*)dst, 16, 0, 0);
  };

  gf32x4 acc[NT][8];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (gf32x4){0.f, 0.f, 0.f, 0.f};
  const int frag_off = gemm_swz((lane & 15) * 64 + (lane >> 4) * 16);
  const int a_sub0 = (wm * 8) * 2, b_sub0 = (wn * NT) * 2;

  const int nk = K / GBK;
  const int total_units = 2 * nk;
  const int pro = total_units < NSLOT ? total_units : NSLOT;
  for (int u = 0; u < pro; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_piece(u, j);

  // multiply K step t; ISSUE: units 2t+3 / 2t+4 go out in front of the MFMA groups (t >= 1, while units remain)
  auto k_step = [&](int t, auto issue_tag) {
    constexpr bool issue = decltype(issue_tag)::value;
    const unsigned char* abase = g_lds + ((2 * t) % NSLOT) * UNIT + frag_off;
    const unsigned char* bbase = g_lds + ((2 * t + 1) % NSLOT) * UNIT + frag_off;
    const int u0 = 2 * t + 3;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      gbf16x8 wf[NT], xf[8];
#pragma unroll
      for (int i = 0; i < NT; ++i) wf[i] = *reinterpret_cast<const gbf16x8*>(bbase + (b_sub0 + i * 2 + kb) * 1024);
#pragma unroll
      for (int j = 0; j < 8; ++j) xf[j] = *reinterpret_cast<const gbf16x8*>(abase + (a_sub0 + j * 2 + kb) * 1024);
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        if (issue) issue_piece(u0 + kb, i);        // k-block 0: the four pieces of unit 2t+3, k-block 1: of 2t+4
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- step 0: the units 0 and 1 have landed when at most (pro - 2) * 4 younger pieces are outstanding
  if (pro >= 5) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if (pro == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  k_step(0, std::false_type{});
  // ---- steady state: issued so far 2t+3 units, needed 2t+2: one younger unit (4 pieces) may still be in flight
  int t = 1;
  for (; 2 * t + 4 < total_units; ++t) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // everyone's pieces of step t landed; everyone left step t - 1
    k_step(t, std::true_type{});
  }
  // ---- tail: nothing (or only part of a pair) left to issue; drain completely
  for (; t < nk; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (2 * t + 3 < total_units) {                 // exactly one unit left (2t+3 = last): issue it alone
#pragma unroll
      for (int j = 0; j < 4; ++j) issue_piece(2 * t + 3, j);
    }
    k_step(t, std::false_type{});
  }

  gemm_epilogue<BN, NT, EPI>(acc, g_lds, bias, aux, Y, M, N, m0, n0, wm, wn, lane, wave);
}

// column tiles per group (see gemm_tile_of): all of them if the weight matrix is small enough to live in an XCD's L2
// next to the streaming activation rows, else the largest divisor of tiles_n whose weight slice is <= 1.6 MB -- unless
// the activation rows are the big operand (K > N: fc2), where splitting the column tiles of a row block would re-read
// the rows instead
static int gemm_column_group(int tiles_n, int bn, int K) {
  const double tile_bytes = (double)bn * K * 2.0;
  if (tiles_n * tile_bytes <= 1.6e6 || K > tiles_n * bn) return tiles_n;
  int best = 1;
  for (int g = 1; g <= tiles_n; ++g)
    if (tiles_n % g == 0 && g * tile_bytes <= 1.6e6) best = g;
  return best;
}

template <int EPI>
static void launch_gemm_ring(const void* x, const void* w, const void* bias, void* aux, void* y, int M, int N, int K,
                             hipStream_t st) {
  const int tiles_n = N / 256, mblocks = (M + GBM - 1) / GBM;
  const int groups = (mblocks + 7) / 8;
  const int ngroup = gemm_column_group(tiles_n, 256, K);
  allow_full_lds((const void*)gemm_bf16_ring_kernel<EPI>);
  hipLaunchKernelGGL((gemm_bf16_ring_kernel<EPI>), dim3(groups * tiles_n * 8), dim3(512), 5 * 32768, st,
                     (const unsigned short*)x, (const unsigned short*)w, (const unsigned short*)bias,
                     (unsigned short*)aux, (unsigned short*)y, M, N, K, tiles_n, mblocks, ngroup);
}

template <int BN, int EPI>
static void launch_gemm(const void* x, const void* w, const void* bias, void* aux, void* y, int M, int N, int K,
                        hipStream_t st) {
  const int tiles_n = N / BN, mblocks = (M + GBM - 1) / GBM;
  const int groups = (mblocks + 7) / 8;
  const int ngroup = gemm_column_group(tiles_n, BN, K);
  const size_t lds = 2 * (size_t)(GBM * GBK * 2 + BN * GBK * 2);
  allow_full_lds((const void*)gemm_bf16_nt_kernel<BN, EPI>);
  hipLaunchKernelGGL((gemm_bf16_nt_kernel<BN, EPI>), dim3(groups * tiles_n * 8), dim3(512), lds, st,
                     (const unsigned short*)x, (const unsigned short*)w, (const unsigned short*)bias,
                     (unsigned short*)aux, (unsigned short*)y, M, N, K, tiles_n, mblocks, ngroup);
}

}  // namespace basd

static int gemm_dispatch(const void* x, const void* w, const void* bias, void* aux, void* y, int64_t M, int N, int K,
                         int epilogue, void* stream, const char* what) {
  using namespace basd;
  if (M <= 0) return BASD_OK;
  if (M > 0x7fffff00LL) return fail(BASD_ERR_SHAPE, "%s: M = %lld too large", what, (long long)M);
  if (K % GBK || K < GBK || (N % 256 && N % 192 && N % 128) || N < 128)
    return fail(BASD_ERR_SHAPE, "%s: need K %% 64 == 0 and N a multiple of 256, 192 or 128 (got N=%d K=%d)", what, N, K);
  hipStream_t st = (hipStream_t)stream;
  const int m = (int)M;
#define BASD_GEMM_EPI(LAUNCH)                                          \
  do {                                                                 \
    switch (epilogue) {                                                \
      case 0: LAUNCH(0); break;                                        \
      case 1: LAUNCH(1); break;                                        \
      case 2: LAUNCH(2); break;                                        \
      case 3: LAUNCH(3); break;                                        \
      default: LAUNCH(4); break;                                       \
    }                                                                  \
  } while (0)
  // tile width: among the widths that divide N, the one whose tile count fills the 256 CUs best in whole rounds
  // (fc2 of ViT-B: 197 x 3 tiles of 256 columns = 2.3 rounds -> 3; 197 x 6 tiles of 128 = 4.6 -> 5: 77 % -> 92 %);
  // ties go to the wider tile (fewer re-reads of the activation rows)
  const int mblocks = (m + GBM - 1) / GBM;
  int best_bn = 0;
  double best_eff = -1.0;
  for (int bn : {256, 192, 128}) {
    if (N % bn) continue;
    const double tiles = (double)mblocks * (N / bn);
    const double rounds = (double)(((long long)tiles + 255) / 256);
    const double eff = tiles / (rounds * 256.0) * (bn == 256 ? 1.0 : bn == 192 ? 0.8 : 0.6);   // measured: the narrow
                                                       // two-stage tiles run at 0.6 - 0.8 of the ring kernel's rate
    if (eff > best_eff + 1e-9) { best_eff = eff; best_bn = bn; }
  }
#define BASD_RING(E) launch_gemm_ring<E>(x, w, bias, aux, y, m, N, K, st)
#define BASD_NT192(E) launch_gemm<192, E>(x, w, bias, aux, y, m, N, K, st)
#define BASD_NT128(E) launch_gemm<128, E>(x, w, bias, aux, y, m, N, K, st)
  if (best_bn == 256) BASD_GEMM_EPI(BASD_RING);
  else if (best_bn == 192) BASD_GEMM_EPI(BASD_NT192);
  else BASD_GEMM_EPI(BASD_NT128);
#undef BASD_RING
#undef BASD_NT192
#undef BASD_NT128
#undef BASD_GEMM_EPI
  return check_launch(what);
}

extern "C" int basd_gemm_bf16(const void* x, const void* w, const void* bias, void* y, int64_t M, int N, int K,
                              int epilogue, void* stream) {
  using namespace basd;
  if (epilogue < 0 || epilogue > 2) return fail(BASD_ERR_SHAPE, "gemm_bf16: epilogue %d not in {0, 1, 2}", epilogue);
  if (epilogue == 1 && bias == nullptr) epilogue = 0;
  return gemm_dispatch(x, w, bias, nullptr, y, M, N, K, epilogue, stream, "gemm_bf16");
}

extern "C" int basd_gemm_bf16_gelu_fwd(const void* x, const void* w, const void* bias, void* pre, void* y, int64_t M,
                                       int N, int K, void* stream) {
  using namespace basd;
  if (M > 0 && (pre == nullptr || y == nullptr)) return fail(BASD_ERR_SHAPE, "gemm_bf16_gelu_fwd: pre and y are required");
  return gemm_dispatch(x, w, bias, pre, y, M, N, K, 3, stream, "gemm_bf16_gelu_fwd");
}

extern "C" int basd_gemm_bf16_gelu_bwd(const void* dy, const void* wt, const void* pre, void* dpre, int64_t M, int N,
                                       int K, void* stream) {
  using namespace basd;
  if (M > 0 && (pre == nullptr || dpre == nullptr))
    return fail(BASD_ERR_SHAPE, "gemm_bf16_gelu_bwd: pre and dpre are required");
  return gemm_dispatch(dy, wt, nullptr, const_cast<void*>(pre), dpre, M, N, K, 4, stream, "gemm_bf16_gelu_bwd");
}
