// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 (is the fp64 matrix peak 78.6 TF?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double f64x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters) {
  f64x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
int main() {
  double* d; hipMalloc(&d, 1024 * 256 * 8);
  const int iters = 20000;
  for (int wg : {256, 512, 1024}) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<<<wg, 256>>>(d, 100);
    hipEventRecord(s); k<<<wg, 256>>>(d, iters); hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    double flops = (double)wg * 4 /*waves*/ * iters * 4 * 2048.0;
    printf("wg=%d: %.3f ms, %.1f TFLOP/s fp64 MFMA\n", wg, ms, flops / ms / 1e9);
  }
  return 0;
}
