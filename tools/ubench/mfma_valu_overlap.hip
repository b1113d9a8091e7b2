// Micro-benchmark: do v_mfma_f32_32x32x16_f16 and fp32 VALU work of DIFFERENT waves on one SIMD
// overlap?  Workgroups of 512 threads (two waves per SIMD); waves 0-3 and waves 4-7 each run
// either an MFMA loop or a VALU fma loop.  hipcc --offload-arch=gfx950 -O3 -o overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float mfma_loop(int iters, float seed) {
#ifdef MFMA_PRIO
  __builtin_amdgcn_s_setprio(MFMA_PRIO);
#endif
  f32x16 acc[2];
  for (int c = 0; c < 2; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)(seed + i);
    b[i] = (_Float16)(seed - i);
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[1], 0, 0, 0);
    }
  }
  float s = 0;
  for (int c = 0; c < 2; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  return s;
}

// 16 MFMAs of 8 passes (32 cycles) = 512 cycles per iteration; VALU: 128 instructions per iteration
// KIND 0: v_fma_f32 (asm, not packable)  1: v_pk_fma_f32  2: v_add_u32/v_xor  3: v_exp_f32
// 4: v_cvt_pk_f16_f32 (+ v_cvt_f32_f16)
template <int KIND>
__device__ __forceinline__ float valu_loop(int iters, float seed) {
#ifdef VALU_PRIO
  __builtin_amdgcn_s_setprio(VALU_PRIO);
#endif
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = seed + i;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(1.0000001f), "v"(1e-7f));
        if (KIND == 1 && (j & 1) == 0)
          asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double*)&v[j]) : "v"(1.0000001), "v"(1e-7));
        if (KIND == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j]) : "v"(u));
        if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[j]));
        if (KIND == 4) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[j]));
      }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += v[i];
  return s;
}

// mode bit 0: what waves 0-3 do (0 MFMA, 1 VALU); bit 1: waves 4-7; bit 2: waves 4-7 idle
template <int KIND>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(float* out, int iters, int mode,
                                                                                    float seed) {
  const int wave = threadIdx.x >> 6;
  float s = 0.f;
  if (wave < 4) {
    s = (mode & 1) ? valu_loop<KIND>(iters, seed) : mfma_loop(iters, seed);
  } else if (!(mode & 4)) {
    s = (mode & 2) ? valu_loop<KIND>(iters, seed) : mfma_loop(iters, seed);
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int KIND>
void run(const char* kind) {
  float* d;
  hipMalloc(&d, 256 * 512 * 4);
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const char* names[] = {"MFMA | MFMA", "MFMA | VALU", "VALU | VALU", "MFMA | idle", "VALU | idle"};
  const int modes[] = {0, 2, 3, 4, 5};
  float t[5];
  for (int m = 0; m < 5; ++m) {
    k<KIND><<<256, 512>>>(d, 10, modes[m], 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<256, 512>>>(d, iters, modes[m], 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&t[m], e0, e1);
  }
  printf("%-18s", kind);
  for (int m = 0; m < 5; ++m) printf("  %s %.3f", names[m], t[m]);
  printf("  ms | overlap of MFMA|VALU: %.0f %% of the shorter one hidden\n",
         100.0 * (t[3] + t[4] - t[1]) / (t[3] < t[4] ? t[3] : t[4]));
  hipFree(d);
}

int main() {
  run<0>("v_fma_f32");
  run<1>("v_pk_fma_f32");
  run<2>("v_add_u32");
  run<3>("v_exp_f32");
  run<4>("v_cvt_f16_f32");
  return 0;
}
