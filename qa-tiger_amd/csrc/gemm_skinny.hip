// Skinny GEMM: the same contract, arithmetic and epilogue as gemm_split.hip (fp32 in / out, three 16-bit MFMAs per product on
// (hi, lo) parts split in flight), for the SMALL problems of the path — the per-sample layers whose M is the batch size
// (quest_proj, the TempMoE query / out-projection / second expert layer, QstGrounding: reference modules.py:304-347, 536-612;
// net.py:118-142), their dX / dW products and the head-batched products of the few-query attentions at small batches.
//
// Those products are 0.1-2 GFLOP each: nothing for the matrix pipes, but ~75 launches per train step at B = 256 and the bulk
// of the 8-GPU shard's step.  On the 128x128 / 256x256 tiles they are a handful of workgroups that each walk the whole K
// behind a barrier per 32 k, or a split-K grid that needs a zeroing launch first and finishes with fp32 atomics (22-38 us per
// launch measured; run-to-run different last bits).  Here:
//   * one workgroup = one 64x64 output tile, 4 waves, and the K range is split over the WAVES: wave w multiplies the w-th
//     quarter of K for the whole tile (2x2 MFMA tiles of 32x32, v_mfma_f32_32x32x16_{f16,bf16}), the four partial tiles meet
//     in LDS and are summed in a fixed order — no atomics, no zeroing launch, bit-reproducible;
//   * no LDS staging and no barrier in the K loop: a lane loads the 8 consecutive k of its own fragment row straight from
//     global memory (L2-resident operands: at most a few MB), four 16-k steps of loads in flight per wave (register ring);
//   * 4x the workgroups of the 128x128 tile for the same output, so a 256x768 output is 48 workgroups of 4 independent waves.
// A grid-level K split (atomics, caller-visible as in gemm_split.hip) remains for a long K over very few tiles.
#include "qt_common.h"
#include <type_traits>
#include <utility>

struct SkArgs {
    int M, N, K;
    const float* A; long lda;
    const float* B; long ldb;
    float* C; long ldc;
    float alpha;
    const float* bias;
    const float* add_src; long ld_add;
    int relu;
    const float* mask_src; long ld_mask; float mask_scale;
    float drop_p; float inv_keep; unsigned long long drop_seed; const unsigned long long* salt;
    int accumulate;
    int split_k, k_per_split;
    int tiles_n;
    long a_bs, b_bs, c_bs;
    float* a_colsum;
    const float* a_amax; const float* b_amax; float* c_amax;
};

typedef _Float16 sk_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sk_bf16x8 __attribute__((ext_vector_type(8)));
typedef float sk_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 sk_f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 sk_bf16x2 __attribute__((ext_vector_type(2)));

template <int HT> __device__ inline void sk_split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    const sk_f32x2 x = {x0, x1};
    if ((HT & 1) == 0) {
        const sk_f16x2 hh = __builtin_convertvector(x, sk_f16x2);
        const sk_f32x2 d = x - __builtin_convertvector(hh, sk_f32x2);
        hi = __builtin_bit_cast(unsigned, hh);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector(d, sk_f16x2));
    } else {
        const sk_bf16x2 hh = __builtin_convertvector(x, sk_bf16x2);
        hi = __builtin_bit_cast(unsigned, hh);
        const sk_f32x2 hf = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
        const sk_f32x2 d = x - hf;
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector(d, sk_bf16x2));
    }
}
template <int HT> __device__ inline floatx16 sk_mma(const uint4& a, const uint4& b, floatx16 c) {
    if ((HT & 1) == 0) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sk_f16x8, a), __builtin_bit_cast(sk_f16x8, b), c, 0, 0, 0);
    else               return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8, a), __builtin_bit_cast(sk_bf16x8, b), c, 0, 0, 0);
}
template <class F, int... Is> __device__ inline void sk_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F> __device__ inline void sk_static_for(F&& f) { sk_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

#define SK_PF 4          // 16-k steps of loads in flight per wave

// HT: 0 fp16 parts x3, 1 bf16 parts x3, 3 bf16 hi parts only.  TA: A given as [K, M] (K-slow), TB: B given as [N, K] (K-contiguous).
template <int HT, bool TA, bool TB>
__global__ __launch_bounds__(256) void qt_gemm_sk_kernel(SkArgs p) {
    if (p.salt && p.drop_p > 0.f) p.drop_seed += *p.salt;
    extern __shared__ __attribute__((aligned(16))) float sk_smem[];
    float (*red)[4][16][64] = reinterpret_cast<float (*)[4][16][64]>(sk_smem);          // [wave][sub-tile][accumulator element][lane]: 64 KiB
    float (*cs)[64] = reinterpret_cast<float (*)[64]>(sk_smem + 4 * 4 * 16 * 64);       // [wave][row]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int tn = (int)(blockIdx.x % (unsigned)p.tiles_n), tm = (int)(blockIdx.x / (unsigned)p.tiles_n);
    const int m0 = tm * 64, n0 = tn * 64;
    const int ksplit = blockIdx.y;
    const float* const A = p.A + (long)blockIdx.z * p.a_bs;
    const float* const B = p.B + (long)blockIdx.z * p.b_bs;
    float* const C = p.C + (long)blockIdx.z * p.c_bs;
    const int kbeg = ksplit * p.k_per_split, kend = min(p.K, kbeg + p.k_per_split);
    const int kq = ((kend - kbeg + 63) / 64) * 16;                  // k per wave, a multiple of 16
    const int kw0 = kbeg + wave * kq, kw1 = min(kend, kw0 + kq);
    const int ns = kw1 > kw0 ? (kw1 - kw0 + 15) / 16 : 0;

    const float sA = p.a_amax ? qt_plane_scale_if_needed(qt_amax_read(p.a_amax)) : 1.0f;     // (all lanes active here)
    const float sB = p.b_amax ? qt_plane_scale_if_needed(qt_amax_read(p.b_amax)) : 1.0f;

    // per-lane fragment rows (clamped: out-of-range rows compute garbage that is never stored)
    long rowA[2], rowB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = min(m0 + i * 32 + r, p.M - 1), n = min(n0 + i * 32 + r, p.N - 1);
        rowA[i] = TA ? (long)m : (long)m * p.lda;
        rowB[i] = TB ? (long)n * p.ldb : (long)n;
    }
    const bool do_csum = TA && p.a_colsum != nullptr && tn == 0;
    float csum[2] = {0.f, 0.f};

    // one 16-k step of one operand fragment: the lane's 8 consecutive k (k = step base + 8 h ..) of its row
    auto load8 = [&](const float* base, long ld, long row, bool kslow, int k, float* dst) {
        const int kk = k + h * 8;
        if (!kslow) {
            const float* ptr = base + row + kk;
            float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
            if (kk + 4 <= kw1) v0 = *reinterpret_cast<const float4*>(ptr);          // K % 4 == 0 for K-contiguous operands (launcher)
            if (kk + 8 <= kw1) v1 = *reinterpret_cast<const float4*>(ptr + 4);
            dst[0] = v0.x; dst[1] = v0.y; dst[2] = v0.z; dst[3] = v0.w; dst[4] = v1.x; dst[5] = v1.y; dst[6] = v1.z; dst[7] = v1.w;
        } else {
            const float* ptr = base + (long)kk * ld + row;
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[j] = (kk + j < kw1) ? ptr[(long)j * ld] : 0.f;
        }
    };
    float ra[SK_PF][2][8], rb[SK_PF][2][8];
    auto gload = [&](int s, auto BUF) {
        constexpr int b = decltype(BUF)::value;
        const int k = kw0 + s * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) load8(A, p.lda, rowA[i], TA, k, ra[b][i]);
#pragma unroll
        for (int j = 0; j < 2; ++j) load8(B, p.ldb, rowB[j], !TB, k, rb[b][j]);
    };
    auto pack = [&](const float* x, float sc, uint4& hi, uint4& lo) {
        sk_split_pair<HT>(x[0] * sc, x[1] * sc, hi.x, lo.x);
        sk_split_pair<HT>(x[2] * sc, x[3] * sc, hi.y, lo.y);
        sk_split_pair<HT>(x[4] * sc, x[5] * sc, hi.z, lo.z);
        sk_split_pair<HT>(x[6] * sc, x[7] * sc, hi.w, lo.w);
    };
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    auto compute = [&](auto BUF) {
        constexpr int b = decltype(BUF)::value;
        uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (do_csum) {
#pragma unroll
                for (int j = 0; j < 8; ++j) csum[i] += ra[b][i][j];
            }
            pack(ra[b][i], sA, ah[i], al[i]);
            pack(rb[b][i], sB, bh[i], bl[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if constexpr (HT < 2) {
                    acc[i][j] = sk_mma<HT>(al[i], bh[j], acc[i][j]);
                    acc[i][j] = sk_mma<HT>(ah[i], bl[j], acc[i][j]);
                }
                acc[i][j] = sk_mma<HT>(ah[i], bh[j], acc[i][j]);
            }
    };

    sk_static_for<SK_PF>([&](auto J) { if (decltype(J)::value < ns) gload(decltype(J)::value, J); });
    for (int s0 = 0; s0 < ns; s0 += SK_PF) {
        sk_static_for<SK_PF>([&](auto J) {
            const int s = s0 + decltype(J)::value;
            if (s < ns) {                                           // wave-uniform
                compute(J);
                if (s + SK_PF < ns) gload(s + SK_PF, J);
            }
        });
    }

    // ---- the four K-quarters meet in LDS; wave w finishes sub-tile w (rows 32 (w >> 1) .., columns 32 (w & 1) ..)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) red[wave][i * 2 + j][e][lane] = acc[i][j][e];
    if (do_csum) {                                                  // (the two k-halves of a row live in lanes r and r + 32)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float t = csum[i] + __shfl_xor(csum[i], 32, 64);
            if (h == 0 && i == 0) cs[wave][r] = t;
            if (h == 0 && i == 1) cs[wave][32 + r] = t;
        }
    }
    __syncthreads();
    if (do_csum && tid < 64 && m0 + tid < p.M) {
        const float t = ((cs[0][tid] + cs[1][tid]) + cs[2][tid]) + cs[3][tid];
        float* dst = p.a_colsum + m0 + tid;
        if (p.split_k > 1) atomicAdd(dst, t);
        else *dst = p.accumulate ? *dst + t : t;
    }
    const int si = wave >> 1, sj = wave & 1;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = ((red[0][wave][e][lane] + red[1][wave][e][lane]) + red[2][wave][e][lane]) + red[3][wave][e][lane];

    const bool first_split = ksplit == 0;
    const bool atomic = p.split_k > 1;
    const int col = n0 + sj * 32 + r;
    const int colc = min(col, p.N - 1);
    const int mrow0 = m0 + si * 32 + 4 * h;
    const float alpha_s = p.alpha / (sA * sB);
    const float bias_v = (p.bias && first_split) ? p.bias[colc] : 0.f;
    const bool use_add = p.add_src && (!atomic || first_split);
    float addv[16], mskv[16], oldv[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const long mc = min(mrow0 + (e & 3) + 8 * (e >> 2), p.M - 1);
        addv[e] = use_add ? p.add_src[mc * p.ld_add + colc] : 0.f;
        mskv[e] = p.mask_src ? p.mask_src[mc * p.ld_mask + colc] : 1.f;
        oldv[e] = (p.accumulate && !atomic) ? C[mc * p.ldc + colc] : 0.f;
    }
    float cmax = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = mrow0 + (e & 3) + 8 * (e >> 2);
        if (m < p.M && col < p.N) {
            float x = alpha_s * v[e] + bias_v + addv[e];
            float* dst = C + (long)m * p.ldc + col;
            if (atomic) { atomicAdd(dst, x); continue; }
            if (p.relu == 1) x = qt_relu(x);
            else if (p.relu == 2) x = x / (1.0f + __expf(-1.702f * x));
            if (p.mask_src) x = (mskv[e] > 0.f) ? x * p.mask_scale : 0.f;
            if (p.drop_p > 0.f) x *= qt_drop_mult(p.drop_seed, (unsigned long long)m * (unsigned long long)p.N + col, p.drop_p, p.inv_keep);
            x += oldv[e];
            cmax = fmaxf(cmax, fabsf(x));
            *dst = x;
        }
    }
    if (p.c_amax) {
        cmax = qt_wave_max(cmax);
        if (lane == 0 && (cmax > 0.f || cmax != cmax))
            atomicMax(reinterpret_cast<unsigned*>(p.c_amax) + ((blockIdx.x + blockIdx.y + blockIdx.z) & (QT_AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, cmax));
    }
}

#define SK_LDS ((4 * 4 * 16 * 64 + 4 * 64) * sizeof(float))
template <int HT> static void sk_launch(const SkArgs& a, int transA, int transB, dim3 grid, hipStream_t st) {
#define QT_SK(TA_, TB_)                                                                                           \
    do {                                                                                                          \
        auto kern = qt_gemm_sk_kernel<HT, TA_, TB_>;                                                              \
        static bool attr_set = false;   /* once per instance, not per launch (and never inside a stream capture) */ \
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SK_LDS); attr_set = true; } \
        hipLaunchKernelGGL(kern, grid, dim3(256), SK_LDS, st, a);                                                 \
    } while (0)
    if (!transA && transB) QT_SK(false, true);
    else if (!transA && !transB) QT_SK(false, false);
    else if (transA && !transB) QT_SK(true, false);
    else QT_SK(true, true);
#undef QT_SK
}

// Which problems go here (called by qt_gemm_x3 after its own argument checks).  Measured on MI355X, GPU time per launch inside a HIP
// graph (tools/bench_small_gemm.py, profiles/r03_small_gemm.txt): a launch costs ~8 us + ~0.9 us per 16 k of a wave's K quarter (the
// fragment loads touch 32 cache lines per instruction), one round of <= 512 resident workgroups.  Against the 128x128 tile (which pays a
// zeroing launch and atomics whenever it must split K to fill the chip) that wins for short K — the dW products of the per-sample layers
// (768x768x256: 23.0 -> 11.7 us), the head-batched products of the few-query attentions (96x768x256 x8: 22.7 -> 11.6) — and for
// K <= 768 when there are enough 64x64 tiles to fill the chip without a split (960x768x768: 26.6 -> 19.9; 512x3072x768: 36.5 -> 27.8),
// and loses for few tiles over a long K (256x768x768: 14.0 vs 18.6; 256x768x3072: 27.9 vs 52.9), which stay where they were.
bool qt_gemm_sk_wants(int M, int N, int K, int batch) {
    const long tiles = (long)((M + 63) / 64) * ((N + 63) / 64) * batch;
    if (M < 16 || N < 32 || tiles > 512) return false;
    if (batch > 1 && K <= 768) return true;             // head-batched products of a small batch (256x96x768 x8: 24.9 -> 18.2 us; the tiled kernel would split K with atomics)
    return K <= 384 || (K <= 768 && tiles >= 48);        // (with the tiled kernel's K splits meeting in a workspace + summing pass: 256x768x768 18.4 us tiled vs 18.6 here, 512x768x768 21.1 vs 18.3 — one launch instead of two)
}

// Same contract as qt_gemm_x3 (include/qatiger.h); returns 2 when a grid-level K split kept it from measuring max|C|.
int qt_gemm_sk(void* stream, int half_type, int transA, int transB, int M, int N, int K, float alpha,
               const float* A, long lda, const float* B, long ldb, float* C, long ldc,
               const float* bias, const float* add_src, long ld_add, int relu,
               const float* mask_src, long ld_mask, float mask_scale,
               float drop_p, unsigned long long drop_seed, int accumulate, int split_k, float* a_colsum,
               int batch, long a_bs, long b_bs, long c_bs, const float* a_amax, const float* b_amax, float* c_amax) {
    hipStream_t st = (hipStream_t)stream;
    SkArgs a;
    a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
    a.alpha = alpha; a.bias = bias; a.add_src = add_src; a.ld_add = ld_add; a.relu = relu;
    a.mask_src = mask_src; a.ld_mask = ld_mask; a.mask_scale = mask_scale;
    a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.drop_seed = drop_seed; a.salt = qt_salt_ptr; a.accumulate = accumulate;
    a.a_colsum = a_colsum; a.a_amax = a_amax; a.b_amax = b_amax; a.c_amax = c_amax; a.a_bs = a_bs; a.b_bs = b_bs; a.c_bs = c_bs;
    const int tiles_m = (M + 63) / 64, tiles_n = (N + 63) / 64;
    const long tiles = (long)tiles_m * tiles_n;
    a.tiles_n = tiles_n;
    const bool plain = !relu && !mask_src && drop_p == 0.f;
    QT_CHECK_ARG(split_k <= 1 || plain, "qt_gemm_sk: split_k>1 needs a linear epilogue");
    if (split_k <= 0) {
        // a grid-level split only for a long K over a handful of tiles: each workgroup keeps >= 1024 k (16 steps per wave)
        split_k = 1;
        if (plain && tiles * batch <= 32 && K >= 4096) {
            long s = K / 1024, cap = 128 / (tiles * batch);
            if (s > cap) s = cap;
            if (s > 1) split_k = (int)s;
        }
    }
    int kps = (K + split_k - 1) / split_k;
    kps = ((kps + 31) / 32) * 32;
    split_k = (K + kps - 1) / kps;
    a.split_k = split_k; a.k_per_split = kps;
    if (split_k > 1 && !accumulate) {
        for (int z = 0; z < batch; ++z) {
            hipError_t e = qt_zero_2d(C + (long)z * c_bs, ldc, M, N, st);
            if (e != hipSuccess) QT_FAIL(QT_ERR_RUNTIME, "qt_gemm_sk: zeroing: %s", hipGetErrorString(e));
        }
        if (a_colsum) {
            hipError_t e = qt_zero_1d(a_colsum, M, st);
            if (e != hipSuccess) QT_FAIL(QT_ERR_RUNTIME, "qt_gemm_sk: zeroing: %s", hipGetErrorString(e));
        }
    }
    bool cmax_skipped = false;
    if (c_amax && split_k > 1) { a.c_amax = nullptr; cmax_skipped = true; }
    QT_CHECK_ARG(tiles < (1L << 31) && batch <= 65535 && split_k <= 65535, "qt_gemm_sk: grid too large");
    dim3 grid((unsigned)tiles, (unsigned)split_k, (unsigned)batch);
    if (half_type == 0) sk_launch<0>(a, transA, transB, grid, st);
    else if (half_type == 1) sk_launch<1>(a, transA, transB, grid, st);
    else sk_launch<3>(a, transA, transB, grid, st);
    QT_CHECK_LAUNCH("qt_gemm_sk");
    return cmax_skipped ? 2 : QT_OK;
}
