// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate (no memory traffic), 1 or 2 independent
// accumulator chains per wave, 1/2 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0;
  for (int c = 0; c < CHAINS; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS>
void run(int blocks, const char* name) {
  float* d;
  hipMalloc(&d, blocks * 256 * 4);
  int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<CHAINS><<<blocks, 256>>>(d, 10, 1.f, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<CHAINS><<<blocks, 256>>>(d, iters, 1.f, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * 16 * CHAINS * 4096.0;
  printf("%s blocks=%d chains=%d: %.3f ms  %.1f TFLOP/s\n", name, blocks, CHAINS, ms, flops / ms / 1e9);
  hipFree(d);
}
int main() {
  run<1>(256, "1 wave/SIMD");
  run<2>(256, "1 wave/SIMD");
  run<4>(256, "1 wave/SIMD");
  run<1>(512, "2 waves/SIMD");
  run<2>(512, "2 waves/SIMD");
  run<1>(1024, "4 waves/SIMD");
  return 0;
}
