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
#include <type_traits>

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
    int tiles_m;
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
    // Inside an XCD's contiguous chunk, walk groups of GM M-tiles: M fastest, then N.  The 64
    // workgroups resident on an XCD then cover an ~8x8 window of tiles, so each A panel and each
    // B panel fetched into that XCD's 4 MiB L2 is shared by 8 workgroups (a plain N-fastest walk
    // needs every B panel at once: 7 MB at N=2304, 49 % L2 hits measured).
    constexpr int GM = 8;
    const int group = wgid / (GM * p.tiles_n);
    const int gm_rows = min(GM, p.tiles_m - group * GM);
    const int within = wgid - group * (GM * p.tiles_n);
    const int tile_n = within / gm_rows, tile_m = group * GM + (within - tile_n * gm_rows);
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

    // two register sets: tile t+1 (loaded last iteration) and tile t+2 (issued this iteration)
    float ra0[16], rb0[16], ra1[16], rb1[16];

    // Addressing is hoisted: a wave-uniform base (kernel argument + k0) plus a per-thread 32-bit
    // byte offset computed once, so a load in the loop is one instruction and no 64-bit VALU math.
    // K-contiguous operand: 4 x float4, thread -> (row = idx>>3, k-quad = idx&7)
    // K-slow operand: 16 coalesced dwords, thread -> (col = tid&127, k-octets 2*(tid>>7) + {0,1})
    unsigned offA[4], offB[4];
    {
        const int colA = min(m0 + (tid & 127), p.M - 1), colB = min(n0 + (tid & 127), p.N - 1);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int idx = tid + it * 256;
            const int row = idx >> 3, kq = idx & 7;
            offA[it] = TA ? 0u : (unsigned)(((long)min(m0 + row, p.M - 1) * p.lda + kq * 4) * 4);
            offB[it] = TB ? (unsigned)(((long)min(n0 + row, p.N - 1) * p.ldb + kq * 4) * 4) : 0u;
        }
        if (TA) offA[0] = (unsigned)(((long)((tid >> 7) * 16) * p.lda + colA) * 4);
        if (!TB) offB[0] = (unsigned)(((long)((tid >> 7) * 16) * p.ldb + colB) * 4);
    }
    // `full` is a compile-time tag: the hot path (whole BK inside [kbeg,kend)) is straight-line
    // loads; only the last, partial tile of a K range takes the predicated build.
    auto load_kc = [&](const float* base, const unsigned (&off)[4], int k0, auto full, float (&dst)[16]) {
        const char* b = reinterpret_cast<const char*>(base + k0);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            float4 v;
            if constexpr (decltype(full)::value) {
                v = *reinterpret_cast<const float4*>(b + off[it]);
            } else {
                v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + (int)(((tid + it * 256) & 7) * 4) < kend) v = *reinterpret_cast<const float4*>(b + off[it]);
            }
            dst[it * 4 + 0] = v.x; dst[it * 4 + 1] = v.y; dst[it * 4 + 2] = v.z; dst[it * 4 + 3] = v.w;
        }
    };
    auto load_ks = [&](const float* base, long ld, const unsigned (&off)[4], int k0, auto full, float (&dst)[16]) {
        const char* b = reinterpret_cast<const char*>(base + (long)k0 * ld) + off[0];
        const int kt = k0 + (tid >> 7) * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if constexpr (decltype(full)::value) dst[j] = *reinterpret_cast<const float*>(b + (long)j * ld * 4);
            else dst[j] = (kt + j < kend) ? *reinterpret_cast<const float*>(b + (long)j * ld * 4) : 0.f;
        }
    };
    auto store_kc = [&](u16* hi_img, u16* lo_img, const float (&src)[16]) {
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
    auto store_ks = [&](u16* hi_img, u16* lo_img, const float (&src)[16]) {
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
    auto gload = [&](int k0, float (&xa)[16], float (&xb)[16]) {
        if (k0 + SBK <= kend) {                                     // wave-uniform
            if (TA) load_ks(p.A, p.lda, offA, k0, std::true_type{}, xa); else load_kc(p.A, offA, k0, std::true_type{}, xa);
            if (TB) load_kc(p.B, offB, k0, std::true_type{}, xb); else load_ks(p.B, p.ldb, offB, k0, std::true_type{}, xb);
        } else {
            if (TA) load_ks(p.A, p.lda, offA, k0, std::false_type{}, xa); else load_kc(p.A, offA, k0, std::false_type{}, xa);
            if (TB) load_kc(p.B, offB, k0, std::false_type{}, xb); else load_ks(p.B, p.ldb, offB, k0, std::false_type{}, xb);
        }
    };
    auto sstore = [&](int buf, const float (&xa)[16], const float (&xb)[16]) {
        u16* b0 = smem + buf * (4 * STILE);
        if (TA) store_ks(b0, b0 + STILE, xa); else store_kc(b0, b0 + STILE, xa);
        if (TB) store_kc(b0 + 2 * STILE, b0 + 3 * STILE, xb); else store_ks(b0 + 2 * STILE, b0 + 3 * STILE, xb);
    };
    auto compute = [&](int cur) {
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
    };

    // Pipeline: while tile t feeds the MFMAs from LDS, tile t+1 sits in one register set and
    // the loads of tile t+2 are in flight into the other, so every load has ~2 iterations to land.
    const int nk = (kend - kbeg + SBK - 1) / SBK;
    if (nk > 0) { gload(kbeg, ra0, rb0); sstore(0, ra0, rb0); }
    if (nk > 1) gload(kbeg + SBK, ra0, rb0);                        // tile 1 -> set 0
    __syncthreads();
    for (int t = 0; t < nk; t += 2) {
        // even step: compute tile t (buf 0); set 0 holds tile t+1; fetch tile t+2 into set 1
        if (t + 2 < nk) gload(kbeg + (t + 2) * SBK, ra1, rb1);
        compute(0);
        if (t + 1 < nk) sstore(1, ra0, rb0);
        __syncthreads();
        if (t + 1 >= nk) break;
        // odd step: compute tile t+1 (buf 1); set 1 holds tile t+2; fetch tile t+3 into set 0
        if (t + 3 < nk) gload(kbeg + (t + 3) * SBK, ra0, rb0);
        compute(1);
        if (t + 2 < nk) sstore(0, ra1, rb1);
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
    {   // per-thread offsets are 32-bit byte offsets from a wave-uniform base
        const long spanA = transA ? ((long)32 * lda + M) : ((long)M * lda + 32);
        const long spanB = transB ? ((long)N * ldb + 32) : ((long)32 * ldb + N);
        if (spanA * 4 >= (1L << 32) || spanB * 4 >= (1L << 32)) QT_FAIL(QT_ERR_UNSUPPORTED, "qt_gemm_x3: operand spans more than 4 GiB");
    }
    if ((long)M * N < 128L * 128L || M < 64 || N < 64) QT_FAIL(QT_ERR_UNSUPPORTED, "qt_gemm_x3: problem too small for the 128x128 tile");
    hipStream_t st = (hipStream_t)stream;
    GemmArgs a;
    a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
    a.alpha = alpha; a.bias = bias; a.add_src = add_src; a.ld_add = ld_add; a.relu = relu;
    a.mask_src = mask_src; a.ld_mask = ld_mask; a.mask_scale = mask_scale;
    a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.drop_seed = drop_seed; a.accumulate = accumulate;
    const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
    const long tiles = (long)tiles_m * tiles_n;
    a.tiles_n = tiles_n; a.tiles_m = tiles_m;
    const bool plain_epilogue = !relu && !mask_src && drop_p == 0.f;
    if (split_k <= 0) {
        // 512 workgroups are resident at once (2 per CU).  Choose the split whose grid wastes the
        // least of its last round; ties go to the larger split (shorter tail per workgroup).
        split_k = 1;
        if (plain_epilogue && tiles < 1024 && K >= 16 * SBK) {
            const long slots = 512;
            long maxs = K / (8 * SBK);
            if (maxs > 32) maxs = 32;
            double best = -1.0;
            for (long s = 1; s <= maxs; ++s) {
                const long blocks = tiles * s;
                const long rounds = (blocks + slots - 1) / slots;
                double eff = (double)blocks / (double)(rounds * slots);
                if (s > 1) eff *= 0.97;                             // atomics + memset are not free
                if (eff >= best - 1e-9) { best = eff; split_k = (int)s; }
            }
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
