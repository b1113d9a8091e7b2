// Shared helpers for the gfx950 kernels of the ParSeNet hot path.
// Everything here is device/host plumbing: error reporting for the C ABI,
// order-preserving float<->uint keys, and wave64 primitives.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define PN_WAVE 64

// ---- error reporting (thread-local, see include/parsenet_hip.h) ------------
extern "C" const char* pn_last_error(void);
void pn_set_error(const char* fmt, ...);

#define PN_OK 0
#define PN_ERR_ARG (-1)
#define PN_ERR_HIP (-2)
#define PN_ERR_WORKSPACE (-3)
#define PN_ERR_UNSUPPORTED (-4)

#define PN_CHECK_ARG(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      pn_set_error(__VA_ARGS__);                \
      return PN_ERR_ARG;                        \
    }                                           \
  } while (0)

#define PN_CHECK_HIP(expr)                                                   \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      pn_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                   __FILE__, __LINE__);                                      \
      return PN_ERR_HIP;                                                     \
    }                                                                        \
  } while (0)

#define PN_CHECK_LAUNCH()                                                    \
  do {                                                                       \
    hipError_t _e = hipGetLastError();                                       \
    if (_e != hipSuccess) {                                                  \
      pn_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),\
                   __FILE__, __LINE__);                                      \
      return PN_ERR_HIP;                                                     \
    }                                                                        \
  } while (0)

// ---- optional per-kernel HIP-event timing (prof.hip) -------------------------------
void pn_prof_begin(const char* name, hipStream_t s, int* token);
void pn_prof_end(hipStream_t s, int token);
struct PnProfScope {
  hipStream_t s;
  int token;
  PnProfScope(const char* name, hipStream_t st) : s(st) { pn_prof_begin(name, st, &token); }
  ~PnProfScope() { pn_prof_end(s, token); }
};
#define PN_CAT2(a, b) a##b
#define PN_CAT(a, b) PN_CAT2(a, b)
#define PN_PROF(name, stream) PnProfScope PN_CAT(_pn_prof_scope_, __LINE__)(name, stream)

static inline size_t pn_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int pn_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// hipFuncSetAttribute is PER DEVICE and the library takes tensors on any device of the process: a call site keeps
// one word of "done" bits (a function-local static), one bit per device.  True the first time the current device
// passes the site (a race sets the attribute twice: idempotent); always true beyond 32 devices.
static inline bool pn_first_on_device(unsigned* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return true;
  const unsigned bit = 1u << dev;
  if (__atomic_load_n(done, __ATOMIC_RELAXED) & bit) return false;
  __atomic_fetch_or(done, bit, __ATOMIC_RELAXED);
  return true;
}

// ---- order-preserving float keys ------------------------------------------
// ord(f) is monotone increasing in f over all non-NaN floats.
__host__ __device__ static inline uint32_t pn_f2ord(float f) {
  uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = __float_as_uint(f);
#else
  memcpy(&u, &f, 4);
#endif
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ static inline float pn_ord2f(uint32_t o) {
  uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float f;
  memcpy(&f, &u, 4);
  return f;
#endif
}

// ---- wave64 helpers --------------------------------------------------------
__device__ static inline int pn_lane() { return threadIdx.x & 63; }

__device__ static inline float pn_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ static inline double pn_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ static inline float pn_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ static inline unsigned long long pn_wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long w = __shfl_xor(v, o, 64);
    v = w > v ? w : v;
  }
  return v;
}
__device__ static inline unsigned long long pn_wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long w = __shfl_xor(v, o, 64);
    v = w < v ? w : v;
  }
  return v;
}
__device__ static inline int pn_wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// number of set bits of m strictly below this lane
__device__ static inline int pn_mbcnt(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32),
                                   __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}
