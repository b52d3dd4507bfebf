// Shared helpers for the gfx950 BASD kernels (C-ABI in include/basd_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/basd_hip.h"

namespace basd {

char* err_buf();                       // thread-local, 256 bytes
int fail(int code, const char* fmt, ...);

// Raise a kernel's dynamic-LDS limit to the full 160 KiB ONCE per kernel (not per launch): the call is a
// host-side attribute change and must not be issued while a stream is being captured into a hipGraph.
inline void allow_full_lds(const void* fn) {
  static thread_local const void* done[64];
  static thread_local int ndone = 0;
  for (int i = 0; i < ndone; ++i)
    if (done[i] == fn) return;
  hipFuncAttributes attr;
  int static_lds = 0;
  if (hipFuncGetAttributes(&attr, fn) == hipSuccess) static_lds = (int)attr.sharedSizeBytes;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - static_lds);
  if (ndone < 64) done[ndone++] = fn;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(BASD_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return BASD_OK;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the wave's
// outstanding global stores (s_waitcnt vmcnt(0)), which puts an L2/HBM round trip into every step
// of the latency-bound factorisation loops (28 us per step on an otherwise idle GPU).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) {
  return __uint_as_float(((unsigned int)b) << 16);
}

// sum over the 8 lanes of an aligned 8-lane group (wave64, no LDS traffic: DPP)
__device__ __forceinline__ float group8_sum(float v) {
  int x = __float_as_int(v);
  // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true));
  x = __float_as_int(v);
  // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true));
  x = __float_as_int(v);
  // row_half_mirror: lane l <-> 7-l inside each 8-lane half row
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true));
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

}  // namespace basd
