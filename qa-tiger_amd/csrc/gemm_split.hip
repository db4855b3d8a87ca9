// Split-precision GEMM: fp32 in, fp32 out, three half-precision MFMAs per product.
//
// Same role and epilogue as gemm.hip (the nn.Linear / MHA projection GEMMs of
// reference src/models/modules.py and their backward), for the large problems
// that dominate the path (99 % of its FLOPs are projection GEMMs).
// gfx950 has no TF32/xf32 matrix path and its fp32 MFMA runs at 1/16 of the
// 16-bit rate, so each fp32 operand x is split on the fly into two 16-bit parts
//     x = hi + lo,   hi = round16(x),   lo = round16(x - hi)
// and a·b is evaluated as  hi_a·hi_b + hi_a·lo_b + lo_a·hi_b  with fp32
// accumulation in v_mfma_f32_32x32x16_{f16,bf16} (the lo·lo term is below the
// fp32 rounding of the sum).
//   HT = 0  fp16 parts: 11+11 significand bits -> ~fp32-accurate products.  Used
//           for forward GEMMs (activations/weights are O(1e-3..1e1), inside
//           fp16 range; fp16 subnormals are honoured by the MFMA).
//   HT = 1  bf16 parts: 8+8 bits (~1.5e-5 relative) with the full fp32 exponent
//           range.  Used for backward GEMMs whose operands are gradients
//           (down to 1e-9) that fp16 cannot hold.
//
// Tiling: 128x128 block tile, BK = 32, 256 threads = 2x2 waves, each wave a
// 64x64 tile = 2x2 MFMA tiles of 32x32 (64 accumulator VGPRs).  Per BK step a
// wave issues 24 MFMAs (2 k16-steps x 4 tiles x 3 products).  Tiles are
// register-staged: global fp32 -> registers (in flight during the MFMAs of
// the current tile) -> split -> LDS as [row][k] 16-bit images, hi and lo, rows
// padded to 80 B so the 16-byte fragment reads of a 16-lane group fall on 16
// distinct bank slots.  Double-buffered (80 KiB LDS -> 2 workgroups per CU).
// An operand whose k index is the slow memory dimension ("K-slow": B given as
// [K,N], A given as [K,M]) is loaded as coalesced dwords, 8 consecutive k per
// thread, so that it can be written to the same [row][k] image without a
// scatter.
#include "qt_common.h"

#define SBK 32
#define SLD 40            // 16-bit elements per LDS row (32 + 8 pad = 80 B)
#define STILE (128 * SLD) // one 128-row image

struct GemmArgs {          // identical to gemm.hip
    int M, N, K;
    const float* A; long lda;
    const float* B; long ldb;
    float* C; long ldc;
    float alpha;
    const float* bias;
    const float* add_src; long ld_add;
    int relu;
    const float* mask_src; long ld_mask; float mask_scale;
    float drop_p; float inv_keep; unsigned long long drop_seed;
    int accumulate;
    int split_k;
    int k_per_split;
    int tiles_n;
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

template <int HT> __device__ inline void split2(float x, u16& hi, u16& lo) {
    if (HT == 0) {
        const _Float16 h = (_Float16)x;
        const _Float16 l = (_Float16)(x - (float)h);
        hi = __builtin_bit_cast(u16, h); lo = __builtin_bit_cast(u16, l);
    } else {
        const __bf16 h = (__bf16)x;
        const __bf16 l = (__bf16)(x - (float)h);
        hi = __builtin_bit_cast(u16, h); lo = __builtin_bit_cast(u16, l);
    }
}

template <int HT> __device__ inline floatx16 mma(const uint4& a, const uint4& b, floatx16 c) {
    if (HT == 0) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else         return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <bool TA, bool TB, int HT>
__global__ __launch_bounds__(256, 2) void qt_gemm_x3_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) u16 smem[];     // 2 buffers x {A_hi, A_lo, B_hi, B_lo}
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    const int nwg = gridDim.x, orig = blockIdx.x;                  // XCD-aware tile order (see gemm.hip)
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tile_m = wgid / p.tiles_n, tile_n = wgid - tile_m * p.tiles_n;
    const int m0 = tile_m * 128, n0 = tile_n * 128;
    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float ra[16], rb[16];                                          // one tile's worth of fp32 per thread, per operand

    // K-contiguous operand: 4 x float4, thread -> (row = idx>>3, k-quad = idx&7)
    auto load_kc = [&](const float* base, long ld, int row0, int rows_total, int k0, float* dst) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256;
            const int row = idx >> 3, kq = idx & 7;
            const int gr = min(row0 + row, rows_total - 1);
            const int gk = k0 + kq * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < kend) v = *reinterpret_cast<const float4*>(base + (long)gr * ld + gk);   // K % 4 == 0 (host check)
            dst[it * 4 + 0] = v.x; dst[it * 4 + 1] = v.y; dst[it * 4 + 2] = v.z; dst[it * 4 + 3] = v.w;
        }
    };
    // K-slow operand: 16 coalesced dwords, thread -> (col = tid&127, k-octets 2*(tid>>7) + {0,1})
    auto load_ks = [&](const float* base, long ld, int col0, int cols_total, int k0, float* dst) {
        const int col = tid & 127, oh = tid >> 7;
        const int gc = min(col0 + col, cols_total - 1);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int gk = k0 + (oh * 2 + e) * 8 + j;
                dst[e * 8 + j] = gk < kend ? base[(long)gk * ld + gc] : 0.f;
            }
    };
    auto store_kc = [&](u16* hi_img, u16* lo_img, const float* src) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256;
            const int row = idx >> 3, kq = idx & 7;
            u16 hh[4], ll[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) split2<HT>(src[it * 4 + e], hh[e], ll[e]);
            uint2 vh, vl;
            vh.x = hh[0] | ((unsigned)hh[1] << 16); vh.y = hh[2] | ((unsigned)hh[3] << 16);
            vl.x = ll[0] | ((unsigned)ll[1] << 16); vl.y = ll[2] | ((unsigned)ll[3] << 16);
            *reinterpret_cast<uint2*>(hi_img + row * SLD + kq * 4) = vh;
            *reinterpret_cast<uint2*>(lo_img + row * SLD + kq * 4) = vl;
        }
    };
    auto store_ks = [&](u16* hi_img, u16* lo_img, const float* src) {
        const int col = tid & 127, oh = tid >> 7;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            u16 hh[8], ll[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) split2<HT>(src[e * 8 + j], hh[j], ll[j]);
            uint4 vh, vl;
            vh.x = hh[0] | ((unsigned)hh[1] << 16); vh.y = hh[2] | ((unsigned)hh[3] << 16);
            vh.z = hh[4] | ((unsigned)hh[5] << 16); vh.w = hh[6] | ((unsigned)hh[7] << 16);
            vl.x = ll[0] | ((unsigned)ll[1] << 16); vl.y = ll[2] | ((unsigned)ll[3] << 16);
            vl.z = ll[4] | ((unsigned)ll[5] << 16); vl.w = ll[6] | ((unsigned)ll[7] << 16);
            *reinterpret_cast<uint4*>(hi_img + col * SLD + (oh * 2 + e) * 8) = vh;
            *reinterpret_cast<uint4*>(lo_img + col * SLD + (oh * 2 + e) * 8) = vl;
        }
    };
    auto gload = [&](int k0) {
        if (TA) load_ks(p.A, p.lda, m0, p.M, k0, ra); else load_kc(p.A, p.lda, m0, p.M, k0, ra);
        if (TB) load_kc(p.B, p.ldb, n0, p.N, k0, rb); else load_ks(p.B, p.ldb, n0, p.N, k0, rb);
    };
    auto sstore = [&](int buf) {
        u16* b0 = smem + buf * (4 * STILE);
        if (TA) store_ks(b0, b0 + STILE, ra); else store_kc(b0, b0 + STILE, ra);
        if (TB) store_kc(b0 + 2 * STILE, b0 + 3 * STILE, rb); else store_ks(b0 + 2 * STILE, b0 + 3 * STILE, rb);
    };

    const int nk = (kend - kbeg + SBK - 1) / SBK;
    if (nk > 0) { gload(kbeg); sstore(0); }
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) gload(kbeg + (t + 1) * SBK);
        const u16* b0 = smem + cur * (4 * STILE);
        const u16 *sAh = b0, *sAl = b0 + STILE, *sBh = b0 + 2 * STILE, *sBl = b0 + 3 * STILE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                               // two k16 steps per BK
            uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int off = (wm * 64 + i * 32 + r) * SLD + s * 16 + h * 8;
                ah[i] = *reinterpret_cast<const uint4*>(sAh + off);
                al[i] = *reinterpret_cast<const uint4*>(sAl + off);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int off = (wn * 64 + j * 32 + r) * SLD + s * 16 + h * 8;
                bh[j] = *reinterpret_cast<const uint4*>(sBh + off);
                bl[j] = *reinterpret_cast<const uint4*>(sBl + off);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[i][j] = mma<HT>(al[i], bh[j], acc[i][j]);
                    acc[i][j] = mma<HT>(ah[i], bl[j], acc[i][j]);
                    acc[i][j] = mma<HT>(ah[i], bh[j], acc[i][j]);
                }
        }
        if (t + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
    }

    const bool first_split = (blockIdx.y == 0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + r;
            if (n >= p.N) continue;
            const float bias_v = (p.bias && first_split) ? p.bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m >= p.M) continue;
                float v = p.alpha * acc[i][j][e] + bias_v;
                float* dst = p.C + (long)m * p.ldc + n;
                if (p.split_k > 1) {
                    if (p.add_src && first_split) v += p.add_src[(long)m * p.ld_add + n];
                    atomicAdd(dst, v);
                    continue;
                }
                if (p.add_src) v += p.add_src[(long)m * p.ld_add + n];
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.mask_src) v = (p.mask_src[(long)m * p.ld_mask + n] > 0.f) ? v * p.mask_scale : 0.f;
                if (p.drop_p > 0.f) v *= qt_drop_mult(p.drop_seed, (unsigned long long)m * (unsigned long long)p.N + n, p.drop_p, p.inv_keep);
                if (p.accumulate) v += *dst;
                *dst = v;
            }
        }
    }
}

template <int HT>
static void launch_x3(const GemmArgs& a, int transA, int transB, dim3 grid, size_t lds, hipStream_t st) {
#define QT_X3(TA_, TB_)                                                                                           \
    do {                                                                                                          \
        auto kern = qt_gemm_x3_kernel<TA_, TB_, HT>;                                                              \
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, a);                                                    \
    } while (0)
    if (!transA && transB) QT_X3(false, true);
    else if (!transA && !transB) QT_X3(false, false);
    else if (transA && !transB) QT_X3(true, false);
    else QT_X3(true, true);
#undef QT_X3
}

// Returns QT_ERR_UNSUPPORTED (without touching C) when the shape/alignment does not fit
// this kernel; the caller then uses qt_gemm_f32.
extern "C" int qt_gemm_x3(void* stream, int half_type, int transA, int transB, int M, int N, int K, float alpha,
                          const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                          const float* bias, const float* add_src, long ld_add, int relu,
                          const float* mask_src, long ld_mask, float mask_scale,
                          float drop_p, unsigned long long drop_seed, int accumulate, int split_k) {
    QT_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C, "qt_gemm_x3: bad problem");
    QT_CHECK_ARG(half_type == 0 || half_type == 1, "qt_gemm_x3: half_type must be 0 (fp16) or 1 (bf16)");
    QT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "qt_gemm_x3: drop_p=%f", drop_p);
    QT_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "qt_gemm_x3: leading dimension too small");
    // K-contiguous operands are staged with 16-byte loads
    if (!transA && !(qt_aligned16(A) && lda % 4 == 0 && K % 4 == 0)) QT_FAIL(QT_ERR_UNSUPPORTED, "qt_gemm_x3: A not 16-byte aligned");
    if (transB && !(qt_aligned16(B) && ldb % 4 == 0 && K % 4 == 0)) QT_FAIL(QT_ERR_UNSUPPORTED, "qt_gemm_x3: B not 16-byte aligned");
    if ((long)M * N < 128L * 128L || M < 64 || N < 64) QT_FAIL(QT_ERR_UNSUPPORTED, "qt_gemm_x3: problem too small for the 128x128 tile");
    hipStream_t st = (hipStream_t)stream;
    GemmArgs a;
    a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
    a.alpha = alpha; a.bias = bias; a.add_src = add_src; a.ld_add = ld_add; a.relu = relu;
    a.mask_src = mask_src; a.ld_mask = ld_mask; a.mask_scale = mask_scale;
    a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.drop_seed = drop_seed; a.accumulate = accumulate;
    const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
    const long tiles = (long)tiles_m * tiles_n;
    a.tiles_n = tiles_n;
    const bool plain_epilogue = !relu && !mask_src && drop_p == 0.f;
    if (split_k <= 0) {
        split_k = 1;
        if (plain_epilogue && tiles < 384 && K >= 16 * SBK) {      // 2 workgroups/CU x 256 CUs: fill the chip
            const long want = (512 + tiles - 1) / tiles;
            const long maxs = K / (8 * SBK);
            split_k = (int)(want < maxs ? want : maxs);
            if (split_k < 1) split_k = 1;
        }
    }
    QT_CHECK_ARG(split_k == 1 || plain_epilogue, "qt_gemm_x3: split_k>1 needs a linear epilogue");
    int kps = (K + split_k - 1) / split_k;
    kps = ((kps + SBK - 1) / SBK) * SBK;
    split_k = (K + kps - 1) / kps;
    a.split_k = split_k; a.k_per_split = kps;
    if (split_k > 1 && !accumulate) {
        hipError_t e = hipMemset2DAsync(C, (size_t)ldc * sizeof(float), 0, (size_t)N * sizeof(float), (size_t)M, st);
        if (e != hipSuccess) QT_FAIL(QT_ERR_RUNTIME, "qt_gemm_x3: memset: %s", hipGetErrorString(e));
    }
    QT_CHECK_ARG(tiles < (1L << 31), "qt_gemm_x3: grid too large");
    const size_t lds = sizeof(u16) * 2 * 4 * STILE;               // 81920 B
    dim3 grid((unsigned)tiles, (unsigned)split_k);
    if (half_type == 0) launch_x3<0>(a, transA, transB, grid, lds, st);
    else launch_x3<1>(a, transA, transB, grid, lds, st);
    QT_CHECK_LAUNCH("qt_gemm_x3");
    return QT_OK;
}
