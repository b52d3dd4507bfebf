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
// GELU forward: a 9-instruction fit of the exact erf form (gelu_fwd, |error| <= 2.6e-5 absolute: invisible after the
// bf16 rounding of the output); its derivative in the backward epilogue: the exact form via Abramowitz-Stegun 7.1.26;
// fp32 accumulation, one rounding to bf16 at the store.
#include <type_traits>
#include "basd_common.h"

namespace basd {

typedef float gf32x4 __attribute__((ext_vector_type(4)));
typedef short gbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short gu16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int gu32x4 __attribute__((ext_vector_type(4)));

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

// Forward GELU of the epilogues: x Phi(x) with Phi(x) ~ 1 / (1 + 2^(-x p(x^2))), p an even quartic fitted (minimax over
// |x| <= 12) to the exact erf form: |error| <= 2.6e-5 ABSOLUTE everywhere -- 1/150 of a bf16 ulp at |y| = 1, and the
// bf16 output is what every consumer sees; the exact form below costs ~16 VALU instructions per element (two of them
// transcendental), this one 9 (x^2, clamp, two FMAs, product, v_exp_f32, add, v_rcp_f32, product).  x^2 is clamped at
// 50: the quartic's leading coefficient is negative, and beyond |x| = 7 the quotient is 0 or 1 to fp32 precision
// anyway.  Saturates correctly: x -> -inf gives -0, x -> +inf gives x; NaN stays NaN.  The backward keeps the exact
// derivative (gelu_erf_grad): it differs from the derivative of this fit by < 1e-4.
#ifndef BASD_GELU_EXACT
__device__ __forceinline__ float gelu_fwd(float x) {
  const float x2 = fminf(x * x, 50.0f);
  const float p = fmaf(x2, fmaf(x2, -0.0010142630552058877f, 0.10677572400244155f), 2.3011213394571612f);
  const float e = __builtin_amdgcn_exp2f(-x * p);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
#else
__device__ __forceinline__ float gelu_erf(float x);
__device__ __forceinline__ float gelu_fwd(float x) { return gelu_erf(x); }
#endif

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
        if (EPI == 2) v = gelu_fwd(v);
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
        for (int r = 0; r < 4; ++r) p[r] = f32_to_bf16_bits(gelu_fwd(bf16_bits_to_f32(p[r])));
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

// ---------------------------------------------------------------------------------------------
// Register -> global epilogue (no LDS).  The accumulator tile [n][m] of a wave holds, per 16-row block j and lane
// (row r16 = lane & 15, q = lane >> 4), the columns i * 16 + 4 q + r of the wave's 64 (i = 0..3 n tiles, r = 0..3): four
// 8-byte chunks 32 bytes apart.  Two register <-> lane butterflies (v_permlane32_swap on the pairs (i, i ^ 1), then
// v_permlane16_swap on the same pairs) exchange the bits (i0, q1, q0) cyclically: afterwards lane q'' = 2 i0 + q1 owns
// the columns h * 32 + q'' * 8 + [0, 8) for h = 0, 1, i.e. two 16-byte pieces, and one dwordx4 store instruction of the
// wave writes 16 rows x 64 contiguous bytes.  16 swaps per row block; bias / GELU / the saved pre-activation are applied
// in that final layout (16-byte loads).  Leaves the LDS ring alone, so the operands of the NEXT tile keep streaming in.
// EPI 5: fp32 output of pitch N, columns >= N are not stored (the weight matrix is padded to the tile width, the output
// is not): two 16-byte stores per 8 columns.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_direct(gf32x4 (&acc)[4][8], const unsigned short* bias,
                                                     unsigned short* aux, unsigned short* Y, int M, int N, int row0,
                                                     int col0, int lane) {
  typedef unsigned int eu32x2 __attribute__((ext_vector_type(2)));
  const int q = lane >> 4, r16 = lane & 15;
  if constexpr (EPI == 5) {
    float* Y32 = reinterpret_cast<float*>(Y);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v[4][4];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          eu32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[2 * p][j][r]),
                                                       __float_as_uint(acc[2 * p + 1][j][r]), false, false);
          eu32x2 s2 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
          v[2 * p][r] = __uint_as_float(s2[0]);
          v[2 * p + 1][r] = __uint_as_float(s2[1]);
        }
      const int row = row0 + j * 16 + r16;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int col = col0 + h * 32 + q * 8;
        if (row < M && col < N) {
          float* dst = Y32 + (size_t)row * N + col;
#pragma unroll
          for (int k = 0; k < 2; ++k)
            __builtin_nontemporal_store((gu32x4){__float_as_uint(v[2 * h + k][0]), __float_as_uint(v[2 * h + k][1]),
                                                 __float_as_uint(v[2 * h + k][2]), __float_as_uint(v[2 * h + k][3])},
                                        reinterpret_cast<gu32x4*>(dst + 4 * k));
        }
      }
    }
    return;
  }
  float bv[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int k = 0; k < 8; ++k) bv[h][k] = 0.f;
  if ((EPI == 1 || EPI == 2 || EPI == 3) && bias != nullptr) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint4 b8 = *reinterpret_cast<const uint4*>(bias + col0 + h * 32 + q * 8);
      const unsigned int w4[4] = {b8.x, b8.y, b8.z, b8.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bv[h][2 * k] = __uint_as_float(w4[k] << 16);
        bv[h][2 * k + 1] = __uint_as_float(w4[k] & 0xffff0000u);
      }
    }
  }
  auto pack2 = [](float lo, float hi) -> unsigned int {
    return (unsigned int)f32_to_bf16_bits(lo) | ((unsigned int)f32_to_bf16_bits(hi) << 16);
  };
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float v[4][4];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        eu32x2 s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[2 * p][j][r]),
                                                     __float_as_uint(acc[2 * p + 1][j][r]), false, false);
        eu32x2 s2 = __builtin_amdgcn_permlane16_swap(s1[0], s1[1], false, false);
        v[2 * p][r] = __uint_as_float(s2[0]);
        v[2 * p + 1][r] = __uint_as_float(s2[1]);
      }
    const int row = row0 + j * 16 + r16;
    if (row < M) {
      const size_t off = (size_t)row * N + col0 + q * 8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float f[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) f[k] = v[2 * h + (k >> 2)][k & 3] + bv[h][k];
        if (EPI == 2) {
#pragma unroll
          for (int k = 0; k < 8; ++k) f[k] = gelu_fwd(f[k]);
        }
        if (EPI == 4) {
          const uint4 p8 = *reinterpret_cast<const uint4*>(aux + off + h * 32);
          const unsigned int w4[4] = {p8.x, p8.y, p8.z, p8.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            f[2 * k] *= gelu_erf_grad(__uint_as_float(w4[k] << 16));
            f[2 * k + 1] *= gelu_erf_grad(__uint_as_float(w4[k] & 0xffff0000u));
          }
        }
        uint4 o = make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
        if (EPI == 3) {
          __builtin_nontemporal_store((gu32x4){o.x, o.y, o.z, o.w}, reinterpret_cast<gu32x4*>(aux + off + h * 32));   // pre-activation, rounded
          const unsigned int w4[4] = {o.x, o.y, o.z, o.w};
          unsigned int g4[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            g4[k] = pack2(gelu_fwd(__uint_as_float(w4[k] << 16)), gelu_fwd(__uint_as_float(w4[k] & 0xffff0000u)));
          o = make_uint4(g4[0], g4[1], g4[2], g4[3]);
        }
        __builtin_nontemporal_store((gu32x4){o.x, o.y, o.z, o.w}, reinterpret_cast<gu32x4*>(Y + off + h * 32));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// PERSISTENT ring kernel.  One workgroup per CU; the LDS ring of five 32 KiB units never drains: the last two K steps
// of a tile already issue the first units of the NEXT one, the epilogue goes register -> global
// (gemm_epilogue_direct), so the next tile's operands land while the epilogue runs.  The non-persistent ring kernel
// pays, per tile, a cold prologue (the first operands: an L2 / HBM round trip with nothing to multiply), the LDS-staged
// epilogue, the store drain before the workgroup can retire and the launch of its successor: 10.9 us against 17.5 us of
// K loop at K = 768.
//
// Work split (per XCD x = blockIdx.x & 7: the P = gridDim.x / 8 workgroups with that residue share one L2 and own the
// row blocks mb = x (mod 8): rb of them, n = rb * tiles_n tiles in the locality order q of gemm_tile_of -- q walks the
// column tiles of a group fastest, then the row blocks).  Workgroup i of the XCD takes block i = the tiles {r P + i} plus
// at most one of the n mod P remainder tiles: at any time the P workgroups multiply P NEIGHBOURING tiles at the SAME K
// step, so a weight slice / activation row block read by several of them comes from HBM once and from L2 otherwise.
// (Measured and rejected: stream-K ranges of equal length per workgroup -- 2.31 tiles per CU instead of 3 for a third of
// the CUs on ViT-B's fc2 -- put the workgroups of an XCD at different K steps; what the balance won, the lost L2
// sharing took back: 221 vs 226 us.)
//
// In lock-step all 256 CUs also reach their epilogues together (32 MB of stores at once); the first wait on an operand
// issued behind those stores (two K steps into the next tile: vmcnt retires in order) then costs 3 800 - 9 100 stamped
// cycles per tile.  Also measured and rejected: running XCD x a fraction x / 8 of a tile out of phase (every workgroup
// parks the tail part of its first tile in a workspace and finishes that tile last): the parking round trip cost more
// than the smoother stores saved (fc1 305 vs 293 us, proj 90 vs 68).  Non-temporal stores of the output keep it out of
// the L2 the operands live in (1 - 2 %).  Measured and rejected as well: cutting the tiles of the last, partial round
// in two along K across pairs of workgroups (2.5 rounds instead of 3 on fc2 / proj): parking both fp32 halves, the drain,
// the flag and the 512 KB reload cost the finishing workgroup ~20 us -- more than the half tile saves at K = 768 and
// about what it saves at K = 3072 (fc2 229 vs 224 us, proj 79 vs 69, qkv 194 vs 176).
// Where the time goes (ablations of this kernel, timing only, M = 50 432 on a box where the full kernel runs fc1 + GELU /
// fc2 / qkv / proj in 336 / 238 / 184 / 72 us): without the MFMAs 206 / 197 / 137 / 51 -- the machinery around them is
// 61 - 83 % of the time; without the LDS fragment reads 264 / 191 / 148 / 60; without the LDS-DMA 295 / 195 / 161 / 63;
// without the barrier: no change; without the epilogue 208 / 231 / 149 / 56: it costs 38 % of fc1 + GELU (13.9 us per
// tile: ~5 us for the 128 KB of stores that all 256 CUs issue in the same microseconds -- HBM-write-bound bursts -- and
// ~9 us of exact-erf GELU, ~21 VALU issue slots per element with its rcp and exp), 19 - 22 % of qkv / proj.  All eight
// waves reach the epilogue together, so none of it hides under MFMAs; a second accumulator set (to interleave the
// epilogue of tile t with the K loop of tile t + 1) does not fit 2 waves per SIMD, and two half-size workgroups per CU
// halve the prefetch depth of the ring (the duo experiment above).
// (Starting XCD x of the first chunk x * 1 000 .. 10 000 cycles late, so that the XCDs reach their epilogues at different
// times without parking anything: no gain -- fc1 + GELU 331 - 351 vs 334 - 342 us, qkv 185 vs 185, fc2 / proj 5 - 30 %
// slower -- so the epilogue's cost is per CU, not contention for HBM writes.)
// Operand addresses are an SGPR base (tile, K step) + a 32-bit per-lane offset that does not depend on the tile: eight
// VGPRs instead of sixteen 64-bit pointers, and the next tile costs scalar arithmetic only.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_bf16_pring_kernel(const unsigned short* __restrict__ X,
                                                              const unsigned short* __restrict__ W,
                                                              const unsigned short* __restrict__ bias,
                                                              unsigned short* aux, unsigned short* __restrict__ Y,
                                                              int M, int N, int K, int tiles_n, int mblocks, int ngroup,
                                                              int chunk_tiles, int ka, int P) {
  // EPI 5 (bf16x3 product, fp32 out): the activation panel has row pitch ka and is walked K / ka times (once per
  // split of the weights, whose rows hold the splits side by side: pitch K = 3 ka); N = columns stored = output pitch.
  extern __shared__ __align__(16) unsigned char g_lds[];
  constexpr int NT = 4, UNIT = 32768, NSLOT = 5;
  constexpr int NST = (EPI == 3) ? 32 : 16;                 // global stores of a wave per epilogue
  const int KA = EPI == 5 ? ka : K;                         // row pitch of the activation panel
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int nk = K / GBK;                                   // >= 3 (launcher)

  // ---- this workgroup's range of the XCD's stream
  // grid = chunks x 8 P: workgroup (chunk c, XCD x, rank i) walks the tiles [c T, (c + 1) T) of block i (T = chunk_tiles;
  // one chunk = fully persistent).  Chunks are dispatched in order as CUs come free.
  // P = workgroups per XCD (launcher: 32 = every CU; fewer leave CUs to another stream for the whole launch)
  const int chunk = blockIdx.x / (8 * P), slot_id = blockIdx.x - chunk * (8 * P);
  const int xcd = slot_id & 7, ci = slot_id >> 3;
  const int rb = (mblocks - xcd + 7) >> 3;                  // >= 1 (launcher: mblocks >= 8)
  const int n_x = rb * tiles_n, R0 = n_x / P, rem = n_x - R0 * P;
  auto block_pos = [&](int b) { return b * R0 + (b * rem + P - 1) / P; };      // position of block b's first tile
  auto tile_at = [&](int o, int& m0_, int& n0_) {           // o-th tile of this workgroup's block -> tile origin
    const int q = o < R0 ? o * P + ci : R0 * P + (ci * rem + P - 1) / P;
    const int nbi = q % ngroup, rest = q / ngroup;
    const int mbl = rest % rb, ng = rest / rb;
    m0_ = (mbl * 8 + xcd) * GBM;
    n0_ = (ng * ngroup + nbi) * 256;
  };
  const int block_tiles = block_pos(ci + 1) - block_pos(ci);   // R0 or R0 + 1 (>= 1: launcher)
  int o = chunk * chunk_tiles;                              // tile of the block being multiplied
  const int ntiles = o + chunk_tiles < block_tiles ? o + chunk_tiles : block_tiles;
  if (o >= ntiles) return;

  // ---- per-lane byte offsets of the LDS-DMA pieces inside a tile's operand panel (row pitch K): piece j of a unit
  //      is sub-tile s = wave + 8 j: row block s >> 1, k block s & 1
  const int q_dma = gemm_swz(lane * 16);
  const int dma_row = q_dma >> 6, dma_c16 = (q_dma >> 4) & 3;
  unsigned int off_a[4], off_b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int s = wave + 8 * j;
    off_b[j] = (unsigned int)((((s >> 1) * 16 + dma_row) * K + (s & 1) * 32 + dma_c16 * 8) * 2);
  }
  auto set_off_a = [&](int m0_) {                           // ragged last row block: rows beyond M re-read row M - 1
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = wave + 8 * j;
      int m = m0_ + (s >> 1) * 16 + dma_row;
      m = m < M ? m : M - 1;
      off_a[j] = (unsigned int)(((m - m0_) * KA + (s & 1) * 32 + dma_c16 * 8) * 2);
    }
  };
  // SGPR base + 32-bit lane offset (the saddr form: hipcc widens the offsets to 64-bit VGPR pairs and adds them on the
  // VALU when this is written with the builtin).  Inline asm: M0 saved / restored around the statement, the nops cover
  // SALU-written base -> VMEM read (5 states) and M0 write -> LDS-DMA; the compiler does not count these loads (wanted:
  // every wait on them below is a hand-counted vmcnt).
  const unsigned int lds_base = (unsigned int)(size_t)g_lds;
  auto issue_piece = [&](const unsigned char* base_u, unsigned int off, int slot, int j) {
    const unsigned int dst = lds_base + slot * UNIT + (wave + 8 * j) * 1024;
    unsigned int keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_nop 3\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(off), "s"(base_u), "s"(dst) : "memory");
  };
  auto wrap = [](int s) { return s >= NSLOT ? s - NSLOT : s; };

  gf32x4 acc[NT][8];
  const int frag_off = gemm_swz((lane & 15) * 64 + (lane >> 4) * 16);
  const int a_sub0 = (wm * 8) * 2, b_sub0 = (wn * NT) * 2;

  // ---- one K step (64 deep = two k-blocks of 32), software-pipelined through registers so that no MFMA group waits
  // for an LDS read issued just before it (with the reads in front of each k-block all eight waves stall together on
  // a 96 KiB read burst right behind the barrier).  With W0 / W1 = the weight fragments of the two k-blocks, X0 / X1
  // [0..7] the activation fragments, unit slots s (A), s+1 (B), and the next step's units in s+2, s+3:
  //   entry:  wa = W0, x[0..3] = X0[0..3]            (read during the previous step, behind its barrier)
  //   kb0:    x[4..7] <- X0[4..7] | 16 MFMAs wa x[0..3] | wb <- W1, x[0..3] <- X1[0..3] | 16 MFMAs wa x[4..7] + the four
  //           LDS-DMA pieces of the A unit two steps ahead (slot s+4: the previous step's B slot)
  //   kb1:    x[4..7] <- X1[4..7] | 16 MFMAs wb x[0..3] | WAIT: lgkmcnt(0), vmcnt: the next step's units have landed;
  //           BARRIER: everybody's have, and everybody has read this step completely | wa <- W0', x[0..3] <- X0'[0..3]
  //           of the NEXT step | 16 MFMAs wb x[4..7] + the four pieces of the B unit two steps ahead (slot s: this
  //           step's A slot, free since the barrier)
  // ONE instance of this code in the kernel (a single loop over segments and K steps, every step issues): with peeled
  // variants the accumulators went through 128-register phi webs and spilled.
  gbf16x8 xf[8], wa[NT], wb[NT];
  auto read_w = [&](gbf16x8 (&w)[NT], int slot, int kb) {
    const unsigned char* base = g_lds + slot * UNIT + frag_off;
#pragma unroll
    for (int i = 0; i < NT; ++i) w[i] = *reinterpret_cast<const gbf16x8*>(base + (b_sub0 + i * 2 + kb) * 1024);
  };
  auto read_x = [&](int j0, int slot, int kb) {
    const unsigned char* base = g_lds + slot * UNIT + frag_off;
#pragma unroll
    for (int j = j0; j < j0 + 4; ++j) xf[j] = *reinterpret_cast<const gbf16x8*>(base + (a_sub0 + j * 2 + kb) * 1024);
  };
  // 16 MFMAs w[0..3] x xf[j0..j0+3]; before_group(i) runs in front of MFMA group i, after_first() behind group 0
  auto half = [&](gbf16x8 (&w)[NT], int j0, auto&& after_first, auto&& before_group) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      before_group(i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = j0; j < j0 + 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i], xf[j], acc[i][j], 0, 0, 0);
      if (i == 0) {
        __builtin_amdgcn_sched_barrier(0);
        after_first();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto nothing = []() {};
  auto nothing_i = [](int) {};

  // ---- first segment
  int m0, n0;
  tile_at(o, m0, n0);
  const unsigned char* a_cur = reinterpret_cast<const unsigned char*>(X) + (size_t)m0 * KA * 2;
  const unsigned char* b_cur = reinterpret_cast<const unsigned char*>(W) + (size_t)n0 * K * 2;
  set_off_a(m0);
  // ---- prologue: the units A, B of the first two steps into slots 0..3; the first step's fragments into registers
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < 4; ++j) issue_piece(((u & 1) ? b_cur : a_cur) + (u >> 1) * (GBK * 2), (u & 1) ? off_b[j] : off_a[j], u, j);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  read_w(wa, 1, 0);
  read_x(0, 0, 0);
  int slot = 0;
  int st_pending = 0;                                       // stores issued behind the youngest LDS-DMA unit (counted waits)
#pragma nounroll
  for (;;) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = (gf32x4){0.f, 0.f, 0.f, 0.f};
    const bool has_next = o + 1 < ntiles;
    // the tile whose first units the last two K steps issue; behind the last tile they re-read this tile's first K steps
    // into ring slots nobody multiplies (keeps every step identical; drained before the kernel ends)
    int m0n = m0, n0n = n0;
    if (has_next) tile_at(o + 1, m0n, n0n);
    const unsigned char* a_nxt = reinterpret_cast<const unsigned char*>(X) + (size_t)m0n * KA * 2;
    const unsigned char* b_nxt = reinterpret_cast<const unsigned char*>(W) + (size_t)n0n * K * 2;
#pragma nounroll
    for (int t = 0; t < nk; ++t) {
      if (t == nk - 2) set_off_a(m0n);                      // every A unit of this tile has been issued
      int ta = t + 2;                                       // K step of the activation panel (EPI 5: modulo its depth)
      if (EPI == 5 && ta < nk) {
        const int nka = KA / GBK;
        ta -= ta >= nka ? nka : 0;
        ta -= ta >= nka ? nka : 0;
      }
      const unsigned char* a_src = t + 2 < nk ? a_cur + ta * (GBK * 2) : a_nxt + (t + 2 - nk) * (GBK * 2);
      const unsigned char* b_src = t + 2 < nk ? b_cur + (t + 2) * (GBK * 2) : b_nxt + (t + 2 - nk) * (GBK * 2);
      const int s_b = wrap(slot + 1), s_a1 = wrap(slot + 2), s_b1 = wrap(slot + 3), s_a2 = wrap(slot + 4);
      // ---- k-block 0
      half(wa, 0, [&]() { read_x(4, slot, 0); }, nothing_i);
      read_w(wb, s_b, 1);
      read_x(0, slot, 1);
      half(wa, 4, nothing, [&](int i) { issue_piece(a_src, off_a[i], s_a2, i); });
      // ---- k-block 1
      half(wb, 0, [&]() { read_x(4, slot, 1); }, nothing_i);
      // units of step t + 1: everything but the youngest unit (the A unit just issued) -- and, on the first step behind
      // an epilogue, its st_pending stores, which were issued before that unit
      if (t == 0 && st_pending != 0) {
        if (NST == 16) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(36) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      read_w(wa, s_b1, 0);
      read_x(0, s_a1, 0);
      half(wb, 4, nothing, [&](int i) { issue_piece(b_src, off_b[i], slot, i); });
      slot = s_a1;
    }
    const bool full_rows = m0 + GBM <= M;                   // every store of the epilogue is issued by every wave
    gemm_epilogue_direct<EPI>(acc, bias, aux, Y, M, N, m0 + wm * 128, n0 + wn * 64, lane);
    st_pending = (full_rows && EPI != 5) ? NST : 0;
    if (!full_rows || EPI == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ragged rows (EPI 5: columns): not every wave issues every store
    if (!has_next) break;
    ++o;
    m0 = m0n;
    n0 = n0n;
    a_cur = a_nxt;
    b_cur = b_nxt;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the trailing LDS-DMA pieces must not outlive the workgroup
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

// grid of the persistent kernel: 8 P workgroups, P <= 32 per XCD, every XCD's stream at least P tiles long
static int pring_cus_per_xcd(int M, int N) {
  const int tiles_n = N / 256, mblocks = (M + GBM - 1) / GBM;
  if (N % 256 || mblocks < 8) return 0;
  const int n_min = (mblocks / 8) * tiles_n;                 // tiles of the XCD with the fewest row blocks
  return n_min < 32 ? n_min : 32;
}

template <int EPI>
static void launch_gemm_pring(const void* x, const void* w, const void* bias, void* aux, void* y, int M, int N, int K,
                              int tile_run, hipStream_t st, int ka = 0) {
  const int tiles_n = (N + 255) / 256, mblocks = (M + GBM - 1) / GBM;
  const int ngroup = gemm_column_group(tiles_n, 256, K);
  // tile_run < 0: fully persistent on -tile_run (8 .. 32) workgroups per XCD -- the launch leaves the other CUs alone
  int P = 32;
  if (tile_run < 0) { P = -tile_run < 8 ? 8 : (-tile_run > 32 ? 32 : -tile_run); tile_run = 0; }
  const int chunk_tiles = tile_run > 0 ? tile_run : 1 << 20;
  const int max_block = ((mblocks + 7) / 8 * tiles_n + P - 1) / P;
  const int chunks = (max_block + chunk_tiles - 1) / chunk_tiles;
  const int grid = 8 * P * chunks;
  allow_full_lds((const void*)gemm_bf16_pring_kernel<EPI>);
  hipLaunchKernelGGL((gemm_bf16_pring_kernel<EPI>), dim3(grid), dim3(512), 5 * 32768, st,
                     (const unsigned short*)x, (const unsigned short*)w, (const unsigned short*)bias,
                     (unsigned short*)aux, (unsigned short*)y, M, N, K, tiles_n, mblocks, ngroup, chunk_tiles, ka, P);
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
                         int epilogue, int tile_run, void* stream, const char* what) {
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
  // the persistent kernel where every XCD has at least one tile per CU of a full grid and K >= 3 steps (measured on one
  // MI355X, M = 50 432: fc1 + GELU 288 vs 327 us, fc2 220 vs 237, qkv 181 vs 207, proj 68 vs 76, student fc1 51 vs 66)
  if (pring_cus_per_xcd(m, N) == 32 && K >= 192) {
#define BASD_PRING(E) launch_gemm_pring<E>(x, w, bias, aux, y, m, N, K, tile_run, st)
    BASD_GEMM_EPI(BASD_PRING);
#undef BASD_PRING
    return check_launch(what);
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
                              int epilogue, int tile_run, void* stream) {
  using namespace basd;
  if (epilogue < 0 || epilogue > 2) return fail(BASD_ERR_SHAPE, "gemm_bf16: epilogue %d not in {0, 1, 2}", epilogue);
  if (epilogue == 1 && bias == nullptr) epilogue = 0;
  return gemm_dispatch(x, w, bias, nullptr, y, M, N, K, epilogue, tile_run, stream, "gemm_bf16");
}

extern "C" int basd_gemm_bf16x3_f32(const void* x, const void* w3, float* y, int64_t M, int N, int K, void* stream) {
  using namespace basd;
  if (M <= 0) return BASD_OK;
  if (M > 0x7fffff00LL) return fail(BASD_ERR_SHAPE, "gemm_bf16x3_f32: M = %lld too large", (long long)M);
  if (K % GBK || K < GBK || N < 1 || N % 8)
    return fail(BASD_ERR_SHAPE, "gemm_bf16x3_f32: need K %% 64 == 0 and N %% 8 == 0 (got N=%d K=%d)", N, K);
  if ((M + GBM - 1) / GBM < 8) return fail(BASD_ERR_SHAPE, "gemm_bf16x3_f32: needs M > 1792 rows (got %lld)", (long long)M);
  if (K < 2 * GBK) return fail(BASD_ERR_SHAPE, "gemm_bf16x3_f32: K >= 128 required (got %d)", K);
  launch_gemm_pring<5>(x, w3, nullptr, nullptr, y, (int)M, N, 3 * K, 0, (hipStream_t)stream, K);
  return check_launch("gemm_bf16x3_f32");
}

extern "C" int basd_gemm_bf16_gelu_fwd(const void* x, const void* w, const void* bias, void* pre, void* y, int64_t M,
                                       int N, int K, int tile_run, void* stream) {
  using namespace basd;
  if (M > 0 && (pre == nullptr || y == nullptr)) return fail(BASD_ERR_SHAPE, "gemm_bf16_gelu_fwd: pre and y are required");
  return gemm_dispatch(x, w, bias, pre, y, M, N, K, 3, tile_run, stream, "gemm_bf16_gelu_fwd");
}

extern "C" int basd_gemm_bf16_gelu_bwd(const void* dy, const void* wt, const void* pre, void* dpre, int64_t M, int N,
                                       int K, int tile_run, void* stream) {
  using namespace basd;
  if (M > 0 && (pre == nullptr || dpre == nullptr))
    return fail(BASD_ERR_SHAPE, "gemm_bf16_gelu_bwd: pre and dpre are required");
  return gemm_dispatch(dy, wt, nullptr, const_cast<void*>(pre), dpre, M, N, K, 4, tile_run, stream, "gemm_bf16_gelu_bwd");
}
