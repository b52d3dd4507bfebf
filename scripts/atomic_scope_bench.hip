// How fast are fp32 atomic adds of many workgroups into one 192 x 192 fp32 tile (the tail of the weight-gradient
// kernel), by memory scope?  agent scope executes at the memory side (coherent across the 8 XCDs' L2s), workgroup
// scope in the issuing XCD's L2.  Also prints which XCC each blockIdx % 8 ran on.
//   hipcc -O3 --offload-arch=gfx950 scripts/atomic_scope_bench.hip -o /tmp/asb && /tmp/asb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int SCOPE>
__global__ __launch_bounds__(512) void tile_atomics(float* out, int tile_elems, int tiles, unsigned* xcc_of_block) {
  const int tile = (blockIdx.x >> 3) % tiles;
  float* dst = out + (size_t)tile * tile_elems;
  if (threadIdx.x == 0 && xcc_of_block) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc_of_block[blockIdx.x] = xcc & 0xf;
  }
  for (int i = threadIdx.x; i < tile_elems; i += 512) {
    if (SCOPE == 0) __hip_atomic_fetch_add(dst + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_fetch_add(dst + i, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}

int main() {
  const int tile_elems = 192 * 192;
  for (int tiles : {1, 4, 16}) {
    float* out;
    unsigned* xcc;
    hipMalloc(&out, sizeof(float) * tile_elems * tiles);
    hipMalloc(&xcc, sizeof(unsigned) * 4096);
    for (int scope = 0; scope < 2; ++scope) {
      hipMemset(out, 0, sizeof(float) * tile_elems * tiles);
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      const int grid = 256;
      auto launch = [&]() {
        if (scope == 0) hipLaunchKernelGGL(tile_atomics<0>, dim3(grid), dim3(512), 0, 0, out, tile_elems, tiles, xcc);
        else hipLaunchKernelGGL(tile_atomics<1>, dim3(grid), dim3(512), 0, 0, out, tile_elems, tiles, xcc);
      };
      launch();
      hipDeviceSynchronize();
      hipMemset(out, 0, sizeof(float) * tile_elems * tiles);
      hipEventRecord(e0);
      for (int r = 0; r < 10; ++r) launch();
      hipEventRecord(e1);
      hipDeviceSynchronize();
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<float> h(tile_elems * tiles);
      hipMemcpy(h.data(), out, sizeof(float) * h.size(), hipMemcpyDeviceToHost);
      double mn = 1e30, mx = 0;
      for (float v : h) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
      printf("tiles %2d scope %s: %7.1f us per launch (256 WGs x %d atomics); sums min %.0f max %.0f (expected %d)\n", tiles,
             scope == 0 ? "agent    " : "workgroup", ms * 100.0, tile_elems, mn, mx, 10 * grid / tiles);
    }
    std::vector<unsigned> hx(256);
    hipMemcpy(hx.data(), xcc, sizeof(unsigned) * 256, hipMemcpyDeviceToHost);
    int ok = 0;
    for (int b = 0; b < 256; ++b) ok += (hx[b] == hx[b & 7]);
    printf("  XCC of blocks 0..7:");
    for (int b = 0; b < 8; ++b) printf(" %u", hx[b]);
    printf("; blocks with XCC(b) == XCC(b %% 8): %d / 256\n", ok);
    hipFree(out);
    hipFree(xcc);
  }
  return 0;
}
