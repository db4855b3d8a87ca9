// Shared device/host helpers for libqatiger_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define QT_WAVE 64
// ABI of the extern "C" entry points; qa-tiger_amd/_lib.py (ABI_VERSION) refuses a library that reports another value.
// Bump on ANY change of an exported signature.
#define QT_ABI_VERSION 212

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

// ---- per-step dropout salt ---------------------------------------------------------------------------------------
// Dropout seeds are kernel ARGUMENTS (one fresh value per site per call from the host counter) — inside a captured HIP graph they are
// frozen, and every replay would draw the same masks.  A device word set with qt_set_dropout_salt() is therefore added to every seed
// by the kernels themselves; bumping it (qt_bump_dropout_salt, a one-thread kernel that is part of the captured step) gives every
// replay fresh masks while forward and backward of one step still agree.  Null (the default): seeds are used as passed.
extern const unsigned long long* qt_salt_ptr;          // host-side variable of the library (misc.hip); launchers copy it into their kernel arguments

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

// ReLU that lets NaN through, as torch.relu does (fmaxf(NaN, 0) is 0: an overflow upstream would turn into plausible finite numbers)
__device__ inline float qt_relu(float v) { return v < 0.f ? 0.f : v; }

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

// ---- device max|x| records ---------------------------------------------------------------------------------------
// A tensor's max |x| lives in QT_AMAX_SLOTS floats (256 B): producers fold their block maximum into slot (block index & 63) with
// one atomicMax on the bit pattern (non-negative floats order like unsigned integers), so a launch of 100k workgroups does not
// serialise on one address; consumers take the maximum of the 64 slots (one load per lane + a wave reduction).
#define QT_AMAX_SLOTS 64
__device__ inline float qt_amax_read(const float* amax) {             // call with all 64 lanes of the wave active
    return qt_wave_max(amax[threadIdx.x & (QT_AMAX_SLOTS - 1)]);
}
// block-level: every thread's value -> one atomicMax per workgroup.  `red` = at least (blockDim.x / 64) floats of LDS.
__device__ inline void qt_amax_fold(float v, float* red, float* amax) {
    v = qt_wave_max(v);
    const int wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = red[0];
        for (int i = 1; i < nw; ++i) m = fmaxf(m, red[i]);
        if (m > 0.f || m != m) atomicMax(reinterpret_cast<unsigned*>(amax) + (blockIdx.x & (QT_AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, m));
    }
}

// ---- per-tensor power-of-two scales of 16-bit / 8-bit operand parts -------------------------------------------
// fp16 (hi, lo) parts: 2^e * amax lands in [2^13, 2^14) — a factor 4 under fp16's largest finite value (65504) for the rounding
// of hi, 27 binades above the fp16 subnormal spacing for lo: the parts keep 11 + 11 significand bits whatever the tensor's
// magnitude (gradients of 1e-9, weights of 1e-4, activations of 1e5).
__host__ __device__ inline float qt_plane_scale(float amax) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
    const int ex = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127;     // floor(log2(amax)) for normal floats
    int e = 13 - ex;
    e = e > 126 ? 126 : (e < -126 ? -126 : e);
    return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}
// Operands converted INSIDE a GEMM (gemm_split.hip) pay one multiply per element pair for a scale, so there it is applied only when
// it buys precision: an operand whose maximum already lies in [2^-6, 2^14] keeps >= 19 of the 22 bits unscaled (weights and
// activations of ordinary magnitude: scale 1, no multiply); gradients (1e-3 .. 1e-9), tiny weights and huge activations are scaled.
__host__ __device__ inline float qt_plane_scale_if_needed(float amax) {
    return (amax >= 0.015625f && amax <= 16384.0f) ? 1.0f : qt_plane_scale(amax);
}
// e4m3 images: 2^e * amax lands in [2^7, 2^8) (the format's largest finite value is 448)
__host__ __device__ inline float qt_fp8_scale(float amax) {
    if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
    const int ex = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu) - 127;
    int e = 7 - ex;
    e = e > 126 ? 126 : (e < -126 ? -126 : e);
    return __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}

// ---- zero a [M, N] region with row pitch ldc -----------------------------------
// Contiguous regions use the 1-D zeroing kernel; strided ones a small 2-D kernel.  (hipMemset2DAsync is avoided: its graph-capture
// node faulted on replay at the primary shapes, and every entry point of this library is meant to be capturable.)
static __global__ __launch_bounds__(256) void qt_zero2d_kernel(float* __restrict__ C, long ldc, int M, int N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long)M * N) { const long r = i / N; C[r * ldc + (i - r * N)] = 0.f; }
}
// (no hipMemsetAsync / hipMemset2DAsync anywhere in the library: runtime blit nodes inside a captured graph are the class of
// operation the ROCm 7.2 replay fault of round 1 was narrowed down to; every zeroing is an ordinary kernel of this library.)
static __global__ __launch_bounds__(256) void qt_zero1d_kernel(float4* __restrict__ p4, long n4, float* __restrict__ tail, int ntail) {
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0.f;
}
static inline hipError_t qt_zero_1d(float* p, long n, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    if ((((uintptr_t)p) & 15u) != 0) {               // unaligned start: the 2-D kernel handles any pointer
        hipLaunchKernelGGL(qt_zero2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, n, 1, (int)n);
        return hipGetLastError();
    }
    const long n4 = n >> 2;
    long blocks = (n4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(qt_zero1d_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<float4*>(p), n4, p + (n4 << 2), (int)(n & 3));
    return hipGetLastError();
}
static inline hipError_t qt_zero_2d(float* C, long ldc, int M, int N, hipStream_t st) {
    if (ldc == N && (long)M * N < (1L << 31)) return qt_zero_1d(C, (long)M * N, st);
    const long n = (long)M * N;
    hipLaunchKernelGGL(qt_zero2d_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, C, ldc, M, N);
    return hipGetLastError();
}
