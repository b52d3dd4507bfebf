// Fused attention forward (frozen teacher blocks, and the student with the LSE output for basd_attention_bwd_bf16):
// out = softmax(Q K^T / sqrt(hd)) V
// straight from the packed qkv projection [B, T, 3, H, hd] (bf16) into [B, T, H * hd] (bf16) -- the
// layout the output projection reads -- plus, optionally, the attention tap the distillation loss
// consumes, per head: importance[b, h, t-1] = softmax_t(bf16(q_cls . k_t) * scale) / H  (summed over h by the
// caller: atomics into a shared [B, T-1] row cost 45 us per call; reference
// src/models/teacher.py:27-39 builds the full [B, H, T, T] map from a second qkv GEMM;
// src/losses/relational.py:22-27 keeps the head-averaged CLS row).
//
// One workgroup (4 waves) per (batch, head).  T <= 272 tokens, hd = 64 or 80 (ViT-H/14: the contraction of Q K^T runs
// over three 32-deep steps with d = 80..95 zero, the output over five 16-column tiles): K and V of the head live in LDS
// (row-major, rows of hd (padded to a multiple of 32) + 8 elements), every wave owns 16-query tiles.
//   S^T = K Q^T   v_mfma_f32_16x16x32_bf16, A = K rows from LDS (ds_read_b128), B = Q rows from global:
//                 the accumulator of key tile kt holds, for query column lane & 15, the keys
//                 16 kt + 4 (lane >> 4) + {0..3};
//   softmax       per query over all its keys = over registers and over the four lane groups (xor 16, 32);
//   O   = P V     A = P: the two S^T accumulators of a 32-key step ARE the A fragment (same query per lane,
//                 eight keys; the contraction only needs A and B to agree on the key order), B = V through the
//                 transposing LDS read ds_read_b64_tr_b16 on the same eight keys -- no transpose of P or V;
//   the 16 x 64 output tile is staged through LDS so that every lane stores 16 contiguous bytes.
// Arithmetic: fp32 logits and probabilities like a flash kernel (P rounded to bf16 for the second MFMA).
#include "basd_common.h"

namespace basd {

typedef float at_f32x4 __attribute__((ext_vector_type(4)));
typedef short at_bf16x8 __attribute__((ext_vector_type(8)));
typedef short at_v4s __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) at_v4s at_lds_v4s;


typedef float at_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 at_bf16x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 (round to nearest even): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned int pack_bf16(float a, float b) {
  at_bf16x2 r = __builtin_convertvector((at_f32x2){a, b}, at_bf16x2);
  return *reinterpret_cast<unsigned int*>(&r);
}

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f) {
  unsigned int u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

template <int CTRL> __device__ __forceinline__ float at_dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// QMEAN: the tap of a teacher WITHOUT a CLS token (reference src/losses/relational.py:25-27: the attention map averaged
// over heads AND queries): importance[b, h, key] = sum_q P[q][key] / (H T) from the probabilities of the main softmax
// (fp32 logits, as the torch form of this tap used before), [B, H, T]; the caller sums over h as for the CLS tap.
template <int NKT, int AT_HD, bool QMEAN = false>   // key tiles of 16: T <= 16 * NKT; head dim 64 or 80
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_fwd_kernel(const unsigned short* __restrict__ qkv, int T, int H,
                                                            float scale, unsigned short* __restrict__ out,
                                                            float* __restrict__ importance, float* __restrict__ lse) {
  constexpr int NKS = (NKT + 1) / 2;                 // 32-key steps of the P V product
  constexpr int KROWS = 32 * NKS;                    // LDS rows (zero padded)
  constexpr int NDS = (AT_HD + 31) / 32;             // 32-deep steps of the Q K^T contraction
  constexpr int NDT = AT_HD / 16;                    // 16-column tiles of the output
  constexpr int NCH = AT_HD / 8;                     // 16-byte chunks of a row that hold data
  constexpr int NCHP = NDS * 4;                      // ... of a padded row
  constexpr int AT_LD = NDS * 32 + 8;                // LDS row stride in bf16 (16-byte aligned, bank rotation)
  extern __shared__ __align__(16) unsigned short sm[];
  unsigned short* Ks = sm;                           // [KROWS][AT_LD]
  unsigned short* Vs = Ks + KROWS * AT_LD;           // [KROWS][AT_LD]
  unsigned short* Os = Vs + KROWS * AT_LD;           // [4 waves][16][AT_LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const size_t row = (size_t)3 * H * AT_HD;          // elements per token
  const unsigned short* base = qkv + (size_t)b * T * row + (size_t)h * AT_HD;

  // ---- stage K and V of this head (16-byte chunks; rows >= T are zero).  All loads are issued before the
  //      first LDS store: a load-store-per-iteration loop exposes one HBM round trip per iteration (7 of
  //      them: 14 us of a 24 us workgroup in the first version of this kernel).
  constexpr int NLD = (KROWS * NCHP + 255) / 256;
  uint4 kreg[NLD], vreg[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx / NCHP, c8 = idx - r * NCHP;
    kreg[i] = make_uint4(0, 0, 0, 0);
    vreg[i] = make_uint4(0, 0, 0, 0);
    if (r < T && c8 < NCH) {
      const unsigned short* p = base + (size_t)r * row + c8 * 8;
      kreg[i] = *reinterpret_cast<const uint4*>(p + (size_t)H * AT_HD);
      vreg[i] = *reinterpret_cast<const uint4*>(p + (size_t)2 * H * AT_HD);
    }
  }
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = tid + 256 * i;
    const int r = idx / NCHP, c8 = idx - r * NCHP;
    if (idx < KROWS * NCHP) {
      *reinterpret_cast<uint4*>(Ks + r * AT_LD + c8 * 8) = kreg[i];
      *reinterpret_cast<uint4*>(Vs + r * AT_LD + c8 * 8) = vreg[i];
    }
  }
  __syncthreads();

  const float sl2 = scale * 1.4426950408889634f;     // exp(x * scale) = exp2(x * sl2)
  const int nqt = (T + 15) >> 4;
  const int full_tiles = T >> 4;                     // key tiles without padding
  unsigned short* Ow = Os + wave * 16 * AT_LD;
  auto load_q = [&](int qt_, uint4 (&dst)[NDS]) {
#pragma unroll
    for (int ks = 0; ks < NDS; ++ks) {
      dst[ks] = make_uint4(0, 0, 0, 0);
      if (qt_ < nqt && qt_ * 16 + li < T && 32 * ks + 8 * g < AT_HD)
        dst[ks] = *reinterpret_cast<const uint4*>(base + (size_t)(qt_ * 16 + li) * row + 32 * ks + 8 * g);
    }
  };
  uint4 qnext[NDS];
  load_q(wave, qnext);
  float csum[QMEAN ? NKT : 1][4];                    // QMEAN: this lane's query column summed over the wave's tiles
#pragma unroll
  for (int kt = 0; kt < (QMEAN ? NKT : 1); ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) csum[kt][r] = 0.f;
  for (int qt = wave; qt < nqt; qt += 4) {
    const int q0 = qt * 16;
    // Q fragments of this tile: query q0 + li, d = 32 ks + 8 g .. + 7 (fetched one tile ahead)
    at_bf16x8 qf[NDS];
#pragma unroll
    for (int ks = 0; ks < NDS; ++ks) qf[ks] = *reinterpret_cast<at_bf16x8*>(&qnext[ks]);
    load_q(qt + 4, qnext);
    // ---- S^T tiles
    at_f32x4 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      at_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < NDS; ++ks) {
        const uint4 kk = *reinterpret_cast<const uint4*>(Ks + (16 * kt + li) * AT_LD + 32 * ks + 8 * g);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const at_bf16x8*>(&kk), qf[ks], acc, 0, 0, 0);
      }
      s[kt] = acc;
      if ((kt & 3) == 3) __builtin_amdgcn_sched_barrier(0);     // keep at most four tiles of K fragments in flight
    }
    // ---- the attention tap: CLS query (column 0 of tile 0) with bf16-rounded logits, head-averaged
    if (!QMEAN && importance != nullptr && qt == 0) {
      float lr[NKT][4];                              // transient: dead before the P V product needs registers
      float tmx = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const unsigned int w = pack_bf16(s[kt][r], s[kt][r + 1]);          // the autocast matmul's bf16 output
          lr[kt][r] = __uint_as_float(w << 16) * scale;
          lr[kt][r + 1] = __uint_as_float(w & 0xffff0000u) * scale;
          if (kt >= full_tiles) {
            if (16 * kt + 4 * g + r >= T) lr[kt][r] = -3.0e38f;
            if (16 * kt + 4 * g + r + 1 >= T) lr[kt][r + 1] = -3.0e38f;
          }
          tmx = fmaxf(tmx, fmaxf(lr[kt][r], lr[kt][r + 1]));
        }
      tmx = fmaxf(tmx, __shfl_xor(tmx, 16, 64));
      tmx = fmaxf(tmx, __shfl_xor(tmx, 32, 64));
      float tsum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lr[kt][r] = __builtin_amdgcn_exp2f((lr[kt][r] - tmx) * 1.4426950408889634f);
          tsum += lr[kt][r];
        }
      tsum += __shfl_xor(tsum, 16, 64);
      tsum += __shfl_xor(tsum, 32, 64);
      if (li == 0) {
        const float tinv = 1.f / (tsum * (float)H);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 16 * kt + 4 * g + r;
            if (key >= 1 && (kt < full_tiles || key < T))
              importance[((size_t)b * H + h) * (T - 1) + key - 1] = lr[kt][r] * tinv;
          }
      }
    }
    // ---- softmax over the keys of every query column
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (kt >= full_tiles && 16 * kt + 4 * g + r >= T) s[kt][r] = -3.0e38f;   // only the last tiles can hold padding keys
        mx = fmaxf(mx, s[kt][r]);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f((s[kt][r] - mx) * sl2);   // v_exp_f32; padding keys: exp2(-huge) = 0
        s[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    if constexpr (QMEAN) {
      const float qinv = (q0 + li < T) ? inv : 0.f;  // padding queries (zero rows of Q) are not part of the mean
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) csum[kt][r] = fmaf(s[kt][r], qinv, csum[kt][r]);
    }
    // log-sum-exp of the scaled logits (natural log): what the backward kernel recomputes P from
    if (lse != nullptr && g == 0 && q0 + li < T) lse[((size_t)b * H + h) * T + q0 + li] = fmaf(mx, scale, __logf(sum));
    // ---- O = P V
    at_f32x4 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) o[dt] = (at_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const int K1 = (2 * ks + 1 < NKT) ? 2 * ks + 1 : 0;      // constant after unrolling
      uint4 pw;
      pw.x = pack_bf16(s[2 * ks][0] * inv, s[2 * ks][1] * inv);
      pw.y = pack_bf16(s[2 * ks][2] * inv, s[2 * ks][3] * inv);
      pw.z = (2 * ks + 1 < NKT) ? pack_bf16(s[K1][0] * inv, s[K1][1] * inv) : 0u;
      pw.w = (2 * ks + 1 < NKT) ? pack_bf16(s[K1][2] * inv, s[K1][3] * inv) : 0u;
      const at_bf16x8 pa = *reinterpret_cast<const at_bf16x8*>(&pw);
      // B fragment: column d = 16 dt + li, keys {32 ks + 4 g + e} and {32 ks + 16 + 4 g + e}, e = 0..3
      const int qq = li >> 2, pp = li & 3;
#pragma unroll
      for (int dt = 0; dt < NDT; ++dt) {
        const unsigned short* a0 = Vs + (32 * ks + 4 * g + qq) * AT_LD + 16 * dt + 4 * pp;
        const at_v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_v4s*)a0);
        const at_v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((at_lds_v4s*)(a0 + 16 * AT_LD));
        const at_bf16x8 vb = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, vb, o[dt], 0, 0, 0);
      }
    }
    // ---- o[dt][r] = O[query q0 + 4 g + r][d = 16 dt + li]: through LDS, then 16-byte row stores
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const unsigned int w = pack_bf16(o[dt][r], o[dt][r + 1]);
        Ow[(4 * g + r) * AT_LD + 16 * dt + li] = (unsigned short)(w & 0xffffu);
        Ow[(4 * g + r + 1) * AT_LD + 16 * dt + li] = (unsigned short)(w >> 16);
      }
    // the tile is private to this wave: LDS operations of a wave complete in order
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0)
#pragma unroll
    for (int c = 0; c < (16 * NCH + 63) / 64; ++c) {
      const int idx = lane + 64 * c, r = idx / NCH, c8 = idx - r * NCH;
      if (idx < 16 * NCH && q0 + r < T) {
        const uint4 v = *reinterpret_cast<const uint4*>(Ow + r * AT_LD + c8 * 8);
        *reinterpret_cast<uint4*>(out + ((size_t)(b * T + q0 + r) * H + h) * AT_HD + c8 * 8) = v;
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
  }
  if constexpr (QMEAN) {
    // sum over the 16 query lanes of a DPP row, then over the waves through the (now free) output staging tiles
    float* cw = reinterpret_cast<float*>(Os) + wave * (16 * AT_LD / 2);      // 16 * AT_LD bf16 = 8 AT_LD floats >= KROWS
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = csum[kt][r];
        v = at_dpp_add<0xB1>(v);                     // quad_perm [1,0,3,2]
        v = at_dpp_add<0x4E>(v);                     // quad_perm [2,3,0,1]
        v = at_dpp_add<0x124>(v);                    // row_ror:4
        v = at_dpp_add<0x128>(v);                    // row_ror:8
        if (li == 0) cw[16 * kt + 4 * g + r] = v;
      }
    __syncthreads();
    const float* c0 = reinterpret_cast<const float*>(Os);
    const float norm = 1.f / ((float)H * (float)T);
    for (int key = tid; key < T; key += 256) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += c0[w * (16 * AT_LD / 2) + key];
      importance[((size_t)b * H + h) * T + key] = v * norm;
    }
  }
}

template <int NKT, int HD, bool QMEAN = false>
static void launch_attention(const void* qkv, int B, int T, int H, float scale, void* out, float* importance,
                             float* lse, hipStream_t st) {
  constexpr int KROWS = 32 * ((NKT + 1) / 2);
  constexpr int LD = (HD + 31) / 32 * 32 + 8;
  const size_t lds = ((size_t)2 * KROWS * LD + 4 * 16 * LD) * sizeof(unsigned short);
  static_assert(8 * LD >= KROWS, "the output staging tile of a wave holds its query-mean column sums");
  allow_full_lds((const void*)attention_fwd_kernel<NKT, HD, QMEAN>);
  hipLaunchKernelGGL((attention_fwd_kernel<NKT, HD, QMEAN>), dim3(B * H), dim3(256), lds, st, (const unsigned short*)qkv, T, H,
                     scale, (unsigned short*)out, importance, lse);
}

}  // namespace basd

extern "C" int basd_attention_fwd_bf16(const void* qkv, int B, int T, int H, int hd, float scale, void* out,
                                       float* importance, float* lse, void* stream) {
  using namespace basd;
  if (B <= 0) return BASD_OK;
  if ((hd != 64 && hd != 80) || T < 1 || T > 272 || H < 1)
    return fail(BASD_ERR_SHAPE, "attention_fwd: T=%d H=%d hd=%d unsupported (hd 64 | 80, T <= 272)", T, H, hd);
  if (importance != nullptr && T < 2) return fail(BASD_ERR_SHAPE, "attention_fwd: importance needs T >= 2");
  hipStream_t st = (hipStream_t)stream;
  if (hd == 64) {
    if (T <= 64) launch_attention<4, 64>(qkv, B, T, H, scale, out, importance, lse, st);
    else if (T <= 208) launch_attention<13, 64>(qkv, B, T, H, scale, out, importance, lse, st);
    else launch_attention<17, 64>(qkv, B, T, H, scale, out, importance, lse, st);
  } else {
    if (T <= 208) launch_attention<13, 80>(qkv, B, T, H, scale, out, importance, lse, st);
    else launch_attention<17, 80>(qkv, B, T, H, scale, out, importance, lse, st);
  }
  return check_launch("attention_fwd");
}

extern "C" int basd_attention_fwd_qmean_bf16(const void* qkv, int B, int T, int H, int hd, float scale, void* out,
                                             float* importance, void* stream) {
  using namespace basd;
  if (B <= 0) return BASD_OK;
  if ((hd != 64 && hd != 80) || T < 1 || T > 272 || H < 1)
    return fail(BASD_ERR_SHAPE, "attention_fwd_qmean: T=%d H=%d hd=%d unsupported (hd 64 | 80, T <= 272)", T, H, hd);
  if (importance == nullptr) return fail(BASD_ERR_SHAPE, "attention_fwd_qmean: importance is required");
  hipStream_t st = (hipStream_t)stream;
  if (hd == 64) {
    if (T <= 64) launch_attention<4, 64, true>(qkv, B, T, H, scale, out, importance, nullptr, st);
    else if (T <= 208) launch_attention<13, 64, true>(qkv, B, T, H, scale, out, importance, nullptr, st);
    else launch_attention<17, 64, true>(qkv, B, T, H, scale, out, importance, nullptr, st);
  } else {
    if (T <= 208) launch_attention<13, 80, true>(qkv, B, T, H, scale, out, importance, nullptr, st);
    else launch_attention<17, 80, true>(qkv, B, T, H, scale, out, importance, nullptr, st);
  }
  return check_launch("attention_fwd_qmean");
}
