// Micro-benchmark: one wave per SIMD (or two) issuing  [1 MFMA f16 32x32x16 + NV VALU]  repeatedly:
// does the VALU work hide under the 32-cycle MFMA when both come from the SAME wave?
// hipcc --offload-arch=gfx950 -O3 -o interleave
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NV, int WAVES>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES / 4, WAVES / 4))) void k(
    float* out, int iters, float seed) {
  f32x16 acc[2];
  for (int c = 0; c < 2; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(seed + i);
    b[i] = (_Float16)(seed - i);
  }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[u & 1]) : "v"(a), "v"(b));
#pragma unroll
      for (int j = 0; j < NV; ++j)
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(1.0000001f), "v"(1e-7f));
    }
  }
  float s = 0;
  for (int c = 0; c < 2; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int NV, int WAVES>
void run() {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<NV, WAVES><<<256, 64 * WAVES>>>(d, 10, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<NV, WAVES><<<256, 64 * WAVES>>>(d, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%d waves/SIMD, %2d VALU per MFMA: %.3f ms  = %.1f ns per MFMA+VALU group per wave\n", WAVES / 4, NV, ms,
         ms * 1e6 / (iters * 16.0));
  hipFree(d);
}

int main() {
  run<0, 4>();
  run<2, 4>();
  run<4, 4>();
  run<6, 4>();
  run<8, 4>();
  run<12, 4>();
  run<16, 4>();
  run<0, 8>();
  run<4, 8>();
  run<8, 8>();
  run<16, 8>();
  return 0;
}
