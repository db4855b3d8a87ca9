// Shared device/host helpers for libqatiger_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define QT_WAVE 64
// ABI of the extern "C" entry points; qa-tiger_amd/_lib.py (ABI_VERSION) refuses a library that reports another value.
// Bump on ANY change of an exported signature.
#define QT_ABI_VERSION 200

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---- error reporting (thread-local, read through qt_last_error()) ----------
extern thread_local char qt_err_buf[512];
#define QT_FAIL(code, ...)                                   \
    do {                                                     \
        snprintf(qt_err_buf, sizeof(qt_err_buf), __VA_ARGS__); \
        return (code);                                       \
    } while (0)
#define QT_CHECK_ARG(cond, ...) \
    do { if (!(cond)) QT_FAIL(-2, __VA_ARGS__); } while (0)
#define QT_CHECK_LAUNCH(name)                                                  \
    do {                                                                       \
        hipError_t e_ = hipGetLastError();                                     \
        if (e_ != hipSuccess) QT_FAIL(-3, "%s launch: %s", name, hipGetErrorString(e_)); \
    } while (0)

enum { QT_OK = 0, QT_ERR_RUNTIME = -1, QT_ERR_ARG = -2, QT_ERR_LAUNCH = -3, QT_ERR_UNSUPPORTED = -4 };

// ---- counter-based dropout RNG ---------------------------------------------
// Stateless counter RNG over (seed, element index): the backward kernels regenerate the forward mask from
// the same pair, nothing is stored.  32-bit mixer (two multiply-xorshift rounds, the "lowbias32" constants)
// keyed by both seed halves: the first version used the splitmix64 finaliser, whose 64-bit multiplies run
// at quarter rate on the vector ALU and made the dropout-carrying LayerNorm backward VALU-bound.
__host__ __device__ inline uint32_t qt_rand_u32(uint64_t seed, uint64_t idx) {
    uint32_t x = (uint32_t)idx + (uint32_t)seed;
    const uint32_t hi = (uint32_t)(idx >> 32) + (uint32_t)(seed >> 32);     // index bits 32.. (zero on this path's tensors)
    x ^= (hi << 13) | (hi >> 19);
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x + (uint32_t)(seed >> 32);
}
// keep-probability 1-p; returns the inverted-dropout multiplier (0 or 1/(1-p))
__host__ __device__ inline float qt_drop_mult(uint64_t seed, uint64_t idx, float p, float inv_keep) {
    const float u = (float)(qt_rand_u32(seed, idx) >> 8) * (1.0f / 16777216.0f);
    return u >= p ? inv_keep : 0.0f;
}

// ---- wave reductions (64 lanes) ---------------------------------------------
__device__ inline float qt_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float qt_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline bool qt_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// ---- zero a [M, N] region with row pitch ldc -----------------------------------
// Contiguous regions use a 1-D memset; strided ones a small kernel.  (hipMemset2DAsync is avoided: its graph-capture
// node faulted on replay at the primary shapes, and every entry point of this library is meant to be capturable.)
static __global__ __launch_bounds__(256) void qt_zero2d_kernel(float* __restrict__ C, long ldc, int M, int N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long)M * N) { const long r = i / N; C[r * ldc + (i - r * N)] = 0.f; }
}
static inline hipError_t qt_zero_2d(float* C, long ldc, int M, int N, hipStream_t st) {
    if (ldc == N) return hipMemsetAsync(C, 0, sizeof(float) * (size_t)M * (size_t)N, st);
    const long n = (long)M * N;
    hipLaunchKernelGGL(qt_zero2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, C, ldc, M, N);
    return hipGetLastError();
}
