// micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs DPP mov (wave64, gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float s = 1.0f + threadIdx.x * 1e-6f, t = threadIdx.x * 1e-7f;
  if (MODE == 0) {
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = threadIdx.x + j;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = fmaf(a[j], s, t);
    }
    float r = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) r += a[j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
  } else {
    v2f a[16];
    const v2f s2 = {s, s}, t2 = {t, t};
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = (v2f){(float)threadIdx.x + j, (float)j};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = __builtin_elementwise_fma(a[j], s2, t2);
    }
    float r = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) r += a[j].x + a[j].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
  }
}
template <int MODE>
void run(float* d, const char* name, double fma_per_instr) {
  const int iters = 20000;
  for (int wg : {256, 512, 1024, 2048}) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<MODE><<<wg, 256>>>(d, 100);
    hipEventRecord(s); k<MODE><<<wg, 256>>>(d, iters); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    const double instr = (double)wg * 4 * iters * 16;   // wave-instructions
    printf("%s wg=%d (%d waves/SIMD): %.3f ms, %.1f TFLOP/s, %.2f cycles/instr/SIMD at 2.4 GHz\n", name, wg, wg / 256, ms,
           instr * 64 * fma_per_instr * 2 / ms / 1e9, ms * 1e-3 * 2.4e9 / (instr / 1024.0));
  }
}
int main() {
  float* d; hipMalloc(&d, 2048 * 256 * 4);
  run<0>(d, "v_fma_f32   ", 1);
  run<1>(d, "v_pk_fma_f32", 2);
  return 0;
}
