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
//   HT = 3  bf16 hi parts only (one MFMA per product): the reduced-precision mode, not used by the parity path.
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
#include <utility>

#define SBK 32
#ifndef QT_MIN_SPLIT_STEPS
#define QT_MIN_SPLIT_STEPS 4   // fewest BK steps a K split may get (tools/bench_small_gemm.py: 256x768x768 30 -> 22 us against 8; 2 is mixed)
#endif
#define SLD 40            // 16-bit elements per LDS row (32 + 8 pad = 80 B)

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
    float drop_p; float inv_keep; unsigned long long drop_seed; const unsigned long long* salt;   // salt: qt_common.h "per-step dropout salt"
    int accumulate;
    int split_k;
    int k_per_split;
    int tiles_n;
    int tiles_m;
    int batch; long a_bs, b_bs, c_bs;   // batched launch (one problem per attention head): count and element strides
    float* a_colsum;          // transA only: a_colsum[m] += sum_k A[k,m]  (bias gradient riding on the dW GEMM)
    const float* a_amax;      // nullable: device max|A|; A is multiplied by the power of two that puts it in fp16's normal range before its split
    const float* b_amax;      // the same for B
    float* c_amax;            // nullable: max|C| record (64 floats) of the values this launch writes (non-split-K launches): the next layer's dY scale (products whose gradient operand is B: dV = P^T dO, dW_k = q^T dU)
    float* ws; long ws_stride; // split-K through a workspace (batch == 1): K split z stores its partial tile at ws + z * ws_stride (row pitch N) with plain
                              // stores and qt_splitk_reduce sums the slabs into C in a fixed order — no atomics, no zeroing launch, reproducible bits
};

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// Two elements at a time: one packed convert for the hi parts, the hi parts widened back with
// shifts/masks (bf16) or v_cvt_f32_f16 (fp16), one packed subtract, one packed convert for the lo
// parts: 5 VALU instructions per pair.
template <int HT> __device__ inline void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
    const f32x2 x = {x0, x1};
    if ((HT & 1) == 0) {
        const f16x2 h = __builtin_convertvector(x, f16x2);
        const f32x2 d = x - __builtin_convertvector(h, f32x2);
        hi = __builtin_bit_cast(unsigned, h);
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector(d, f16x2));
    } else {
        const bf16x2 h = __builtin_convertvector(x, bf16x2);
        hi = __builtin_bit_cast(unsigned, h);
        const f32x2 hf = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
        const f32x2 d = x - hf;
        lo = __builtin_bit_cast(unsigned, __builtin_convertvector(d, bf16x2));
    }
}

// K-contiguous staging: thread idx -> (row, k-quad).  Eight lanes cover one 128-B global row segment.
// With 80-B LDS rows, the two rows written by one 16-lane ds_write_b64 group must sit 4 rows apart
// (80*4 = 320 = 64 mod 128) to land on disjoint banks, so the 8 row slots of a wave are walked in the
// order 0,4,1,5,2,6,3,7 (measured before: 33 % of LDS cycles were bank-conflict cycles).
__device__ inline int kc_row(int idx) {
    const int j = (idx >> 3) & 7;
    return ((idx >> 6) << 3) | ((j & 1) << 2) | (j >> 1);
}

// compile-time loop: guarantees static indices into register arrays however large the body is
template <class F, int... Is> __device__ inline void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ inline void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

template <int HT> __device__ inline floatx16 mma(const uint4& a, const uint4& b, floatx16 c) {
    if ((HT & 1) == 0) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else         return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Tile configuration: NWM x NWN waves, each owning WMT x WNT MFMA tiles of 32x32.
//   <2,2,2,2, PREF=2>  128x128 block, 256 threads, 80 KiB LDS (2 workgroups/CU), two tiles of loads in flight
//   <4,2,2,4, PREF=1>  256x256 block, 512 threads, the whole 160 KiB LDS, one tile in flight: HALF the
//                      L2->CU bytes per FLOP.  The kernel is supply-bound (an A/B against a wave-specialised
//                      build showed 4 MFMA-issuing waves per CU reach the same rate), so this is the lever
//                      for the M = B*T*P GEMMs whose grids are large enough to fill 256 single-slot CUs.
template <int WMT, int WNT, int NWM, int NWN, int PREF, bool TA, bool TB, int HT>
__global__ __launch_bounds__(NWM * NWN * 64, (NWM * NWN >= 8) ? 2 : 2) void qt_gemm_x3_kernel(GemmArgs p) {
    if (p.salt && p.drop_p > 0.f) p.drop_seed += *p.salt;
    constexpr int NT = NWM * NWN * 64;
    constexpr int BM = NWM * WMT * 32, BN = NWN * WNT * 32;
    constexpr int A_IMG = BM * SLD, B_IMG = BN * SLD;               // one 16-bit image
    constexpr int STAGE = 2 * A_IMG + 2 * B_IMG;                    // {A_hi, A_lo, B_hi, B_lo}
    constexpr int LA = BM * 8 / NT, LB = BN * 8 / NT;               // float4 loads per thread per tile (K-contiguous)
    static_assert(LA * 4 == BM * SBK / NT && LB * 4 == BN * SBK / NT, "tile/threads mismatch");
    static_assert(NT % BM == 0 && NT % BN == 0, "K-slow staging needs NT multiple of the tile edge");
    extern __shared__ __attribute__((aligned(16))) u16 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NWN, wn = wave - wm * NWN;
    const int r = lane & 31, h = lane >> 5;
    const float* const A0 = p.A; const float* const B0 = p.B; float* const C0 = p.C;

    // Persistent workgroups: the grid is one round of resident workgroups and each walks the work
    // items  w = blockIdx.x, blockIdx.x + gridDim.x, ...  (work item = (tile, K split, batch)).
    // The C write-out of a tile (22 % of the kernel at K = 768 when it was exposed) then drains
    // behind the next tile's loads and MFMAs.  Inside a round, the workgroups of one XCD
    // (blockIdx % 8, round-robin dispatch) take a contiguous run of tiles, walked M-fastest inside
    // groups of GM M-tiles, so an XCD's resident workgroups cover a compact window of tiles and
    // share A and B panels through its 4 MiB L2 (N-fastest order: 49 % L2 hits measured).
    constexpr int GM = (BM == 128) ? 8 : 4;
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, q8 = G >> 3, r8 = G & 7;
    const int slot = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);   // bijective on [0,G)
    const int tiles = p.tiles_m * p.tiles_n;
    const long total = (long)tiles * p.split_k * p.batch;
    int m0 = 0, n0 = 0, kbeg = 0, kend = 0, tile_n = 0, ksplit = 0;
    auto decode3 = [&](int t, int ks, int z) {                      // sets m0, n0, kbeg, kend, operand bases
        ksplit = ks;
        const int group = t / (GM * p.tiles_n);
        const int gm_rows = min(GM, p.tiles_m - group * GM);
        const int within = t - group * (GM * p.tiles_n);
        tile_n = within / gm_rows;
        m0 = (group * GM + (within - tile_n * gm_rows)) * BM;
        n0 = tile_n * BN;
        kbeg = ksplit * p.k_per_split;
        kend = min(p.K, kbeg + p.k_per_split);
        p.A = A0 + (long)z * p.a_bs; p.B = B0 + (long)z * p.b_bs; p.C = C0 + (long)z * p.c_bs;
    };
    auto decode = [&](long w) {
        const long rest = w / tiles;
        decode3((int)(w - rest * tiles), (int)(rest % p.split_k), (int)(rest / p.split_k));
    };

    floatx16 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; ++i)
#pragma unroll
        for (int j = 0; j < WNT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float ra0[LA * 4], rb0[LB * 4], ra1[PREF == 2 ? LA * 4 : 1], rb1[PREF == 2 ? LB * 4 : 1];

    // Addressing is hoisted: a wave-uniform base (kernel argument + k0) plus a per-thread 32-bit
    // byte offset computed once, so a load in the loop is one instruction and no 64-bit VALU math.
    // K-contiguous operand: float4s, thread -> (row = kc_row(idx), k-quad = idx&7)
    // K-slow operand: coalesced dwords, thread -> (col = tid % edge, 32*edge/NT consecutive k)
    unsigned offA[LA], offB[LB];
    auto set_offsets = [&]() {
#pragma unroll
        for (int it = 0; it < LA; ++it) {
            const int idx = tid + it * NT;
            offA[it] = TA ? 0u : (unsigned)(((long)min(m0 + kc_row(idx), p.M - 1) * p.lda + (idx & 7) * 4) * 4);
        }
#pragma unroll
        for (int it = 0; it < LB; ++it) {
            const int idx = tid + it * NT;
            offB[it] = TB ? (unsigned)(((long)min(n0 + kc_row(idx), p.N - 1) * p.ldb + (idx & 7) * 4) * 4) : 0u;
        }
        if (TA) offA[0] = (unsigned)(((long)((tid / BM) * (LA * 4)) * p.lda + min(m0 + (tid % BM), p.M - 1)) * 4);
        if (!TB) offB[0] = (unsigned)(((long)((tid / BN) * (LB * 4)) * p.ldb + min(n0 + (tid % BN), p.N - 1)) * 4);
    };
    // `full` is a compile-time tag: the hot path (whole BK inside [kbeg,kend)) is straight-line
    // loads; only the last, partial tile of a K range takes the predicated build.
    auto load_kc = [&](const float* base, const auto& off, int k0, auto full, auto& dst) {
        constexpr int L = sizeof(off) / sizeof(off[0]);
        const char* b = reinterpret_cast<const char*>(base + k0);
#pragma unroll
        for (int it = 0; it < L; ++it) {
            float4 v;
            if constexpr (decltype(full)::value) {
                v = *reinterpret_cast<const float4*>(b + off[it]);
            } else {
                v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + (int)(((tid + it * NT) & 7) * 4) < kend) v = *reinterpret_cast<const float4*>(b + off[it]);
            }
            dst[it * 4 + 0] = v.x; dst[it * 4 + 1] = v.y; dst[it * 4 + 2] = v.z; dst[it * 4 + 3] = v.w;
        }
    };
    auto load_ks = [&](const float* base, long ld, const auto& off, int edge, int k0, auto full, auto& dst) {
        constexpr int KN = sizeof(dst) / sizeof(dst[0]);            // consecutive k per thread
        const char* b = reinterpret_cast<const char*>(base + (long)k0 * ld) + off[0];
        const int kt = k0 + (tid / edge) * KN;
#pragma unroll
        for (int j = 0; j < KN; ++j) {
            if constexpr (decltype(full)::value) dst[j] = *reinterpret_cast<const float*>(b + (long)j * ld * 4);
            else dst[j] = (kt + j < kend) ? *reinterpret_cast<const float*>(b + (long)j * ld * 4) : 0.f;
        }
    };
    // sA: the gradient operand's power-of-two scale (1 when the operand is taken as it is).  Backward GEMMs hand in max|dY|
    // (a_amax): dY * sA sits in fp16's normal range whatever its magnitude, so the fp16 parts keep 11 + 11 bits there too.
    const float sA = p.a_amax ? qt_plane_scale_if_needed(qt_amax_read(p.a_amax)) : 1.0f;     // (all lanes active here)
    const float sB = p.b_amax ? qt_plane_scale_if_needed(qt_amax_read(p.b_amax)) : 1.0f;
    const bool scA = sA != 1.0f, scB = sB != 1.0f;              // wave-uniform: the unscaled stores carry no multiply
    // `sc`: a float scale, or std::false_type for "none" (no multiply in the code at all)
    auto mulsc = [](float x, auto sc) { if constexpr (std::is_same<decltype(sc), float>::value) return x * sc; else return x; };
    auto store_kc = [&](u16* hi_img, u16* lo_img, const auto& src, auto sc) {
        constexpr int L = sizeof(src) / sizeof(src[0]) / 4;
#pragma unroll
        for (int it = 0; it < L; ++it) {
            const int idx = tid + it * NT;
            const int row = kc_row(idx), kq = idx & 7;
            uint2 vh, vl;
            split_pair<HT>(mulsc(src[it * 4 + 0], sc), mulsc(src[it * 4 + 1], sc), vh.x, vl.x);
            split_pair<HT>(mulsc(src[it * 4 + 2], sc), mulsc(src[it * 4 + 3], sc), vh.y, vl.y);
            *reinterpret_cast<uint2*>(hi_img + row * SLD + kq * 4) = vh;
            if constexpr (HT < 2) *reinterpret_cast<uint2*>(lo_img + row * SLD + kq * 4) = vl;
        }
    };
    float csum = 0.f;                                               // this thread's share of a_colsum (TA only)
    float cmax = 0.f;                                               // max |C| written by this thread over all its tiles (c_amax)
    bool do_csum = false;
    auto store_ks = [&](u16* hi_img, u16* lo_img, int edge, const auto& src, auto sc) {
        constexpr int KN = sizeof(src) / sizeof(src[0]);
        const int col = tid % edge, kb = (tid / edge) * KN;
#pragma unroll
        for (int e = 0; e < KN / 8; ++e) {
            uint4 vh, vl;
            split_pair<HT>(mulsc(src[e * 8 + 0], sc), mulsc(src[e * 8 + 1], sc), vh.x, vl.x);
            split_pair<HT>(mulsc(src[e * 8 + 2], sc), mulsc(src[e * 8 + 3], sc), vh.y, vl.y);
            split_pair<HT>(mulsc(src[e * 8 + 4], sc), mulsc(src[e * 8 + 5], sc), vh.z, vl.z);
            split_pair<HT>(mulsc(src[e * 8 + 6], sc), mulsc(src[e * 8 + 7], sc), vh.w, vl.w);
            *reinterpret_cast<uint4*>(hi_img + col * SLD + kb + e * 8) = vh;
            if constexpr (HT < 2) *reinterpret_cast<uint4*>(lo_img + col * SLD + kb + e * 8) = vl;
        }
    };
    auto gload = [&](int k0, auto& xa, auto& xb) {
        if (k0 + SBK <= kend) {                                     // wave-uniform
            if (TA) load_ks(p.A, p.lda, offA, BM, k0, std::true_type{}, xa); else load_kc(p.A, offA, k0, std::true_type{}, xa);
            if (TB) load_kc(p.B, offB, k0, std::true_type{}, xb); else load_ks(p.B, p.ldb, offB, BN, k0, std::true_type{}, xb);
        } else {
            if (TA) load_ks(p.A, p.lda, offA, BM, k0, std::false_type{}, xa); else load_kc(p.A, offA, k0, std::false_type{}, xa);
            if (TB) load_kc(p.B, offB, k0, std::false_type{}, xb); else load_ks(p.B, p.ldb, offB, BN, k0, std::false_type{}, xb);
        }
    };
    auto sstore = [&](int buf, const auto& xa, const auto& xb) {
        u16* b0 = smem + buf * STAGE;
        if (TA) {
            if (scA) store_ks(b0, b0 + A_IMG, BM, xa, sA); else store_ks(b0, b0 + A_IMG, BM, xa, std::false_type{});
            if (do_csum) {                                          // K-tail and out-of-range rows were loaded as zeros
                constexpr int KN = sizeof(xa) / sizeof(xa[0]);
#pragma unroll
                for (int j = 0; j < KN; ++j) csum += xa[j];
            }
        } else { if (scA) store_kc(b0, b0 + A_IMG, xa, sA); else store_kc(b0, b0 + A_IMG, xa, std::false_type{}); }
        if (TB) { if (scB) store_kc(b0 + 2 * A_IMG, b0 + 2 * A_IMG + B_IMG, xb, sB); else store_kc(b0 + 2 * A_IMG, b0 + 2 * A_IMG + B_IMG, xb, std::false_type{}); }
        else { if (scB) store_ks(b0 + 2 * A_IMG, b0 + 2 * A_IMG + B_IMG, BN, xb, sB); else store_ks(b0 + 2 * A_IMG, b0 + 2 * A_IMG + B_IMG, BN, xb, std::false_type{}); }
    };
    auto compute = [&](int cur) {
        const u16* b0 = smem + cur * STAGE;
        const u16 *sAh = b0, *sAl = b0 + A_IMG, *sBh = b0 + 2 * A_IMG, *sBl = b0 + 2 * A_IMG + B_IMG;
#pragma unroll
        for (int s = 0; s < 2; ++s) {                               // two k16 steps per BK
            uint4 ah[WMT], al[WMT], bh[WNT], bl[WNT];
#pragma unroll
            for (int i = 0; i < WMT; ++i) {
                const int off = (wm * (WMT * 32) + i * 32 + r) * SLD + s * 16 + h * 8;
                ah[i] = *reinterpret_cast<const uint4*>(sAh + off);
                if constexpr (HT < 2) al[i] = *reinterpret_cast<const uint4*>(sAl + off);
            }
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                const int off = (wn * (WNT * 32) + j * 32 + r) * SLD + s * 16 + h * 8;
                bh[j] = *reinterpret_cast<const uint4*>(sBh + off);
                if constexpr (HT < 2) bl[j] = *reinterpret_cast<const uint4*>(sBl + off);
            }
#pragma unroll
            for (int i = 0; i < WMT; ++i)
#pragma unroll
                for (int j = 0; j < WNT; ++j) {
                    if constexpr (HT < 2) {
                        acc[i][j] = mma<HT>(al[i], bh[j], acc[i][j]);
                        acc[i][j] = mma<HT>(ah[i], bl[j], acc[i][j]);
                    }
                    acc[i][j] = mma<HT>(ah[i], bh[j], acc[i][j]);
                }
        }
    };

    const bool late = (PREF == 1) && __builtin_amdgcn_readfirstlane(wave) >= (NWM * NWN) / 2;
    constexpr bool PERSIST = (PREF == 1);   // big tile only: the small tile (2 workgroups/CU) is launched one item per workgroup
    long w = slot;
    if constexpr (PERSIST) {
        if (w >= total) return;
        decode(w);
    } else {
        decode3(slot, blockIdx.y, blockIdx.z);                      // grid = (tiles, split_k, batch)
    }
    set_offsets();
    if (kend > kbeg) gload(kbeg, ra0, rb0);                         // first tile of the first work item
    for (;;) {
        do_csum = TA && p.a_colsum != nullptr && tile_n == 0;
        csum = 0.f;
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int nk = (kend - kbeg + SBK - 1) / SBK;
        // registers already hold tile 0 of this work item (loaded before the previous epilogue)
        if constexpr (PREF == 2) {
            // while tile t feeds the MFMAs from LDS, tile t+1 sits in one register set and the loads of
            // tile t+2 are in flight into the other: every load has ~2 iterations to land
            if (nk > 0) sstore(0, ra0, rb0);
            if (nk > 1) gload(kbeg + SBK, ra0, rb0);
            __syncthreads();
            for (int t = 0; t < nk; t += 2) {
                if (t + 2 < nk) gload(kbeg + (t + 2) * SBK, ra1, rb1);
                compute(0);
                if (t + 1 < nk) sstore(1, ra0, rb0);
                __syncthreads();
                if (t + 1 >= nk) break;
                if (t + 3 < nk) gload(kbeg + (t + 3) * SBK, ra0, rb0);
                compute(1);
                if (t + 2 < nk) sstore(0, ra1, rb1);
                __syncthreads();
            }
        } else {
            // 8 waves = 2 per SIMD, one barrier per K tile.  Run in lock-step, both waves of a SIMD convert
            // together (matrix pipe idle) and then compete for the matrix pipe.  Waves 4-7 therefore run
            // half an iteration out of phase: they convert + store tile t+1 FIRST (its loads were issued an
            // iteration earlier), then multiply tile t, so a SIMD's matrix pipe sees one wave's MFMAs beside
            // the other wave's VALU/LDS-store work.  Same LDS buffers, same barrier count, same results.
            if (nk > 0) sstore(0, ra0, rb0);
            if (late && nk > 1) gload(kbeg + SBK, ra0, rb0);
            __syncthreads();
            // The late half's loads of tile t+2 go out right after tile t+1 has left its registers; the early
            // half issues tile t+1 at the top of step t.  (Issuing the early half's loads one step sooner too,
            // before its barrier wait, measured 6-18 % SLOWER on the 215040-row shapes; the 16x16x32 MFMA form with
            // 16-byte C stores measured within +-3 % and 2x slower split-K atomics; cutting the tiles of the last,
            // partial round along K so that idle CUs share them (360 tiles on 256 CUs) measured no gain: the memset
            // and atomics cost what the shorter tail saves.)
            const int ahead = late ? 2 : 1;
            for (int t = 0; t < nk; ++t) {
                if (late && t + 1 < nk) sstore((t + 1) & 1, ra0, rb0);
                if (t + ahead < nk) gload(kbeg + (t + ahead) * SBK, ra0, rb0);
                compute(t & 1);
                if (!late && t + 1 < nk) sstore((t + 1) & 1, ra0, rb0);
                __syncthreads();
            }
        }
        // this work item's output coordinates, kept for the epilogue while the NEXT item's loads go out
        const int em0 = m0, en0 = n0;
        float* const eC = p.C;
        const bool first_split = (ksplit == 0);
        float* const ws_slab = p.ws ? p.ws + (long)ksplit * p.ws_stride : nullptr;
        if (do_csum && m0 + (tid % BM) < p.M) atomicAdd(p.a_colsum + m0 + (tid % BM), csum);
        w += G;
        const bool more = w < total;
        // big tile (1 workgroup/CU): nothing else on the CU hides the write-out, so the next item's first
        // loads go out before it.  The small tile runs 2 workgroups/CU and has no registers to spare for this.
        constexpr bool EARLY = PERSIST;
        if (EARLY && more) {
            decode(w);
            set_offsets();
            if (kend > kbeg) gload(kbeg, ra0, rb0);                 // older than the stores below in the vmcnt queue
        }
    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5).
    // At K = 768 a tile has only 24 BK steps, so the epilogue's instruction count matters (a first
    // version with per-element 64-bit addressing and six runtime flags ran the residual out-projection
    // at 148 TFLOP/s against 305 for the plain store).  One row pointer per output row; the feature
    // set is a compile-time tag: plain (bias [+ReLU] store or split-K atomics), +residual, or full
    // (mask / dropout / accumulate).
    const int mrow0 = em0 + wm * (WMT * 32) + 4 * h;
    const int ncol0 = en0 + wn * (WNT * 32) + r;
    float bias_v[WNT];
#pragma unroll
    for (int j = 0; j < WNT; ++j) {
        const int n = ncol0 + j * 32;
        bias_v[j] = (p.bias && first_split && n < p.N) ? p.bias[n] : 0.f;
    }
    const float alpha_s = p.alpha / (sA * sB);
    auto epilogue = [&](auto ADD, auto FULL) {
        constexpr bool has_add = decltype(ADD)::value, full = decltype(FULL)::value;
        const bool atomic = p.split_k > 1;
        const int act = p.relu;                                     // 0 none, 1 ReLU, 2 QuickGELU x*sigmoid(1.702x)
        const bool use_add = has_add && (!atomic || first_split);
        int ncl[WNT];                                               // clamped columns: loads below need no branches
#pragma unroll
        for (int j = 0; j < WNT; ++j) ncl[j] = min(ncol0 + j * 32, p.N - 1);
        static_for<WMT>([&](auto I) {
            constexpr int i = decltype(I)::value;
#pragma unroll
            for (int eh = 0; eh < 2; ++eh) {                        // 8 rows x WNT columns at a time
                // phase 1: every epilogue operand of this slab goes in flight together (a first version
                // loaded them one by one behind per-element conditionals: +0.8 ms on a 215040x768 residual)
                float addv[8][WNT], mskv[8][WNT], oldv[8][WNT];
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = eh * 8 + e8;
                    const long mc = min(mrow0 + i * 32 + (e & 3) + 8 * (e >> 2), p.M - 1);
#pragma unroll
                    for (int j = 0; j < WNT; ++j) {
                        if (has_add) addv[e8][j] = use_add ? p.add_src[mc * p.ld_add + ncl[j]] : 0.f;
                        if (full) {
                            mskv[e8][j] = p.mask_src ? p.mask_src[mc * p.ld_mask + ncl[j]] : 1.f;
                            oldv[e8][j] = p.accumulate ? eC[mc * p.ldc + ncl[j]] : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int e8 = 0; e8 < 8; ++e8) {
                    const int e = eh * 8 + e8;
                    const int m = mrow0 + i * 32 + (e & 3) + 8 * (e >> 2);
                    if (m < p.M) {
                        float* crow = eC + (long)m * p.ldc + ncol0;
                        static_for<WNT>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            if (ncol0 + j * 32 < p.N) {
                                float v = alpha_s * acc[i][j][e] + bias_v[j];
                                if (has_add) v += addv[e8][j];
                                if (atomic) {
                                    if (ws_slab) ws_slab[(long)m * p.N + ncol0 + j * 32] = v;
                                    else atomicAdd(crow + j * 32, v);
                                    return;
                                }
                                if (act == 1) v = qt_relu(v);
                                else if (act == 2) v = v / (1.0f + __expf(-1.702f * v));
                                if constexpr (full) {
                                    if (p.mask_src) v = (mskv[e8][j] > 0.f) ? v * p.mask_scale : 0.f;
                                    if (p.drop_p > 0.f)
                                        v *= qt_drop_mult(p.drop_seed, (unsigned long long)m * (unsigned long long)p.N + (ncol0 + j * 32), p.drop_p, p.inv_keep);
                                    v += oldv[e8][j];
                                }
                                cmax = fmaxf(cmax, fabsf(v));
                                crow[j * 32] = v;
                            }
                        });
                    }
                }
            }
        });
    };
    if (p.mask_src || p.drop_p > 0.f || p.accumulate) {
        if (p.add_src) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::false_type{}, std::true_type{});
    } else if (p.add_src) epilogue(std::true_type{}, std::false_type{});
    else epilogue(std::false_type{}, std::false_type{});
        if (!PERSIST || !more) break;
    }
    if (p.c_amax) {                                                 // one atomic per wave, spread over the record's 64 slots
        cmax = qt_wave_max(cmax);
        if (lane == 0 && (cmax > 0.f || cmax != cmax))
            atomicMax(reinterpret_cast<unsigned*>(p.c_amax) + ((blockIdx.x + blockIdx.y + blockIdx.z) & (QT_AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, cmax));
    }
}

template <int WMT, int WNT, int NWM, int NWN, int PREF, int HT>
static void launch_x3(const GemmArgs& a, int transA, int transB, dim3 grid, hipStream_t st) {
    constexpr int NT = NWM * NWN * 64;
    constexpr size_t lds = sizeof(u16) * 2 * (2 * (NWM * WMT * 32) + 2 * (NWN * WNT * 32)) * SLD;
#define QT_X3(TA_, TB_)                                                                                           \
    do {                                                                                                          \
        auto kern = qt_gemm_x3_kernel<WMT, WNT, NWM, NWN, PREF, TA_, TB_, HT>;                                    \
        static bool attr_set = false;   /* once per instance, not per launch (and never inside a stream capture) */ \
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = true; } \
        hipLaunchKernelGGL(kern, grid, dim3(NT), lds, st, a);                                                     \
    } while (0)
    if (!transA && transB) QT_X3(false, true);
    else if (!transA && !transB) QT_X3(false, false);
    else if (transA && !transB) QT_X3(true, false);
    else QT_X3(true, true);
#undef QT_X3
}

// This file is compiled seven times (qa-tiger_amd/_lib.py): -DQT_X3_PART=1..6 each instantiate the four
// transpose variants of ONE (tile, half type) pair, part 0 holds the C entry point.  The parts build in
// parallel; one translation unit with all 24 kernels took six minutes.
#ifndef QT_X3_PART
#define QT_X3_PART 0
#endif
#define QT_X3_DECL(NAME) void NAME(const GemmArgs& a, int transA, int transB, dim3 grid, hipStream_t st)
QT_X3_DECL(qt_x3_big_0); QT_X3_DECL(qt_x3_big_1); QT_X3_DECL(qt_x3_big_3);
QT_X3_DECL(qt_x3_small_0); QT_X3_DECL(qt_x3_small_1); QT_X3_DECL(qt_x3_small_3);
#if QT_X3_PART == 1
QT_X3_DECL(qt_x3_big_0) { launch_x3<4, 2, 2, 4, 1, 0>(a, transA, transB, grid, st); }
#elif QT_X3_PART == 2
QT_X3_DECL(qt_x3_big_1) { launch_x3<4, 2, 2, 4, 1, 1>(a, transA, transB, grid, st); }
#elif QT_X3_PART == 3
QT_X3_DECL(qt_x3_big_3) { launch_x3<4, 2, 2, 4, 1, 3>(a, transA, transB, grid, st); }
#elif QT_X3_PART == 4
QT_X3_DECL(qt_x3_small_0) { launch_x3<2, 2, 2, 2, 2, 0>(a, transA, transB, grid, st); }
#elif QT_X3_PART == 5
QT_X3_DECL(qt_x3_small_1) { launch_x3<2, 2, 2, 2, 2, 1>(a, transA, transB, grid, st); }
#elif QT_X3_PART == 6
QT_X3_DECL(qt_x3_small_3) { launch_x3<2, 2, 2, 2, 2, 3>(a, transA, transB, grid, st); }
#else

// Returns QT_ERR_UNSUPPORTED (without touching C) when the shape/alignment does not fit
// this kernel; the caller then uses qt_gemm_f32.
// half_type: 0 fp16 parts x3, 1 bf16 parts x3, 3 = ONE bf16 product per term (reduced precision, BASELINE config 5's
// 16-bit mode: operands rounded to bf16, fp32 accumulate); +16 forces the 128x128 tile, +32 forces the 256x256 tile (A/B testing).
bool qt_gemm_sk_wants(int M, int N, int K, int batch);                                    // gemm_skinny.hip
int qt_gemm_sk(void* stream, int half_type, int transA, int transB, int M, int N, int K, float alpha,
               const float* A, long lda, const float* B, long ldb, float* C, long ldc,
               const float* bias, const float* add_src, long ld_add, int relu,
               const float* mask_src, long ld_mask, float mask_scale,
               float drop_p, unsigned long long drop_seed, int accumulate, int split_k, float* a_colsum,
               int batch, long a_bs, long b_bs, long c_bs, const float* a_amax, const float* b_amax, float* c_amax);

int qt_splitk_reduce(void* stream, const float* ws, int split, int M, int N, float* C, long ldc, int accumulate, float* c_amax);   // gemm_plane.hip

extern "C" int qt_gemm_x3(void* stream, int half_type, int transA, int transB, int M, int N, int K, float alpha,
                          const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                          const float* bias, const float* add_src, long ld_add, int relu,
                          const float* mask_src, long ld_mask, float mask_scale,
                          float drop_p, unsigned long long drop_seed, int accumulate, int split_k, float* a_colsum,
                          int batch, long a_bs, long b_bs, long c_bs, const float* a_amax, const float* b_amax, float* c_amax,
                          float* ws, long ws_floats) {
    QT_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C, "qt_gemm_x3: bad problem");
    QT_CHECK_ARG(!a_colsum || transA, "qt_gemm_x3: a_colsum needs transA (A given as [K,M])");
    if (batch < 1) batch = 1;
    QT_CHECK_ARG(batch == 1 || (!bias && !add_src && !mask_src && !a_colsum), "qt_gemm_x3: batched launches take no bias/residual/mask/colsum");
    QT_CHECK_ARG(batch <= 65535, "qt_gemm_x3: batch too large");
    const int force = half_type >> 4;                               // 1 / 2: force the 128 / 256 tile; 4: force the skinny kernel; 8: never the skinny kernel
    half_type &= 15;
    QT_CHECK_ARG(half_type == 0 || half_type == 1 || half_type == 3, "qt_gemm_x3: half_type must be 0 (3x fp16), 1 (3x bf16) or 3 (1x bf16)");
    QT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "qt_gemm_x3: drop_p=%f", drop_p);
    QT_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N, "qt_gemm_x3: leading dimension too small");
    // K-contiguous operands are staged with 16-byte loads
    if (batch > 1 && ((!transA && a_bs % 4) || (transB && b_bs % 4))) QT_FAIL(QT_ERR_UNSUPPORTED, "qt_gemm_x3: batch stride breaks 16-byte alignment");
    if (!transA && !(qt_aligned16(A) && lda % 4 == 0 && K % 4 == 0)) QT_FAIL(QT_ERR_UNSUPPORTED, "qt_gemm_x3: A not 16-byte aligned");
    if (transB && !(qt_aligned16(B) && ldb % 4 == 0 && K % 4 == 0)) QT_FAIL(QT_ERR_UNSUPPORTED, "qt_gemm_x3: B not 16-byte aligned");
    {   // per-thread offsets are 32-bit byte offsets from a wave-uniform base
        const long spanA = transA ? ((long)32 * lda + M) : ((long)M * lda + 32);
        const long spanB = transB ? ((long)N * ldb + 32) : ((long)32 * ldb + N);
        if (spanA * 4 >= (1L << 32) || spanB * 4 >= (1L << 32)) QT_FAIL(QT_ERR_UNSUPPORTED, "qt_gemm_x3: operand spans more than 4 GiB");
    }
    // small problems (the per-sample layers, M = batch size): one 64x64 tile per workgroup, K split over its waves (gemm_skinny.hip)
    if (!(force & 11) && ((force & 4) || qt_gemm_sk_wants(M, N, K, batch)))
        return qt_gemm_sk(stream, half_type, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, add_src, ld_add, relu, mask_src, ld_mask,
                          mask_scale, drop_p, drop_seed, accumulate, split_k, a_colsum, batch, a_bs, b_bs, c_bs, a_amax, b_amax, c_amax);
    // one problem: small ones belong to the exact-fp32 kernel's 64x64 tiles.  Many small problems in one launch (TSPM's
    // 60x60x512 score / gradient products, one per item) still fill the chip with one 128x128 tile each.
    if (batch == 1 ? ((long)M * N < 128L * 128L || M < 64 || N < 64) : (M < 16 || N < 16))
        QT_FAIL(QT_ERR_UNSUPPORTED, "qt_gemm_x3: problem too small for the 128x128 tile");
    hipStream_t st = (hipStream_t)stream;
    GemmArgs a;
    a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
    a.alpha = alpha; a.bias = bias; a.add_src = add_src; a.ld_add = ld_add; a.relu = relu;
    a.mask_src = mask_src; a.ld_mask = ld_mask; a.mask_scale = mask_scale;
    a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.drop_seed = drop_seed; a.salt = qt_salt_ptr; a.accumulate = accumulate;
    a.a_colsum = a_colsum; a.a_amax = a_amax; a.b_amax = b_amax; a.c_amax = c_amax; a.batch = batch; a.a_bs = a_bs; a.b_bs = b_bs; a.c_bs = c_bs;
    const bool plain_epilogue = !relu && !mask_src && drop_p == 0.f;
    QT_CHECK_ARG(split_k <= 1 || plain_epilogue, "qt_gemm_x3: split_k>1 needs a linear epilogue");

    // Tile choice (measured, tools/bench_gemm.py, profiles/r01_v10_gemm_microbench.txt): the 256x256 tile
    // needs half the operand bytes per FLOP but owns a whole CU.  Since it became persistent (write-out
    // hidden behind the next tile) and its wave halves run staggered it wins on every shape of the path
    // that has 256 rows and columns, +2..30 % over the 128x128 tile.
    bool big = (M >= 256 && N >= 256);
    if (big && K <= 4096) {
        // Few 256-tiles (small batches, the M = B rows layers): a round of 512 resident 128-tiles takes ~0.68 of a round of
        // 256 resident 256-tiles (tools/bench_mid_gemm.py: 3840x768x768 = 45 big tiles: 39 us against 62 us; 135 tiles:
        // 102 against 68), so compare rounds.  Long-K problems (dW) keep the big tile: split-K fills the chip there.
        const long tb = (long)((M + 255) / 256) * ((N + 255) / 256) * batch;
        const long rounds_small = (4 * tb + 511) / 512, rounds_big = (tb + 255) / 256;
        if (tb <= 1024 && 0.68 * (double)rounds_small < (double)rounds_big) big = false;
    }
    if (force & 1) big = false;
    if (force & 2) big = true;
    const int BT = big ? 256 : 128;
    const long slots = big ? 256 : 512;                             // resident workgroups
    const int tiles_m = (M + BT - 1) / BT, tiles_n = (N + BT - 1) / BT;
    const long tiles = (long)tiles_m * tiles_n;
    a.tiles_n = tiles_n; a.tiles_m = tiles_m;
    if (split_k <= 0) {
        // Split K over s workgroups when the tile grid alone cannot fill the chip.  Cost model:
        // rounds of resident workgroups x (BK steps per workgroup + a fixed prologue/epilogue/atomics
        // overhead worth `ovh` steps); minimise it.
        split_k = 1;
        if (plain_epilogue && tiles * batch < 2 * slots && K >= 16 * SBK && batch <= 16) {   // (a split needs one memset per problem)
            const long ovh = big ? 16 : 8;
            long maxs = K / (QT_MIN_SPLIT_STEPS * SBK);
            if (maxs > 64) maxs = 64;
            long best = -1;
            for (long s = 1; s <= maxs; ++s) {
                const long rounds = (tiles * s * batch + slots - 1) / slots;
                const long ksteps = ((K + s - 1) / s + SBK - 1) / SBK;
                const long cost = rounds * (ksteps + ovh);
                if (best < 0 || cost < best) { best = cost; split_k = (int)s; }
            }
        }
    }
    int kps = (K + split_k - 1) / split_k;
    kps = ((kps + SBK - 1) / SBK) * SBK;
    split_k = (K + kps - 1) / kps;
    a.split_k = split_k; a.k_per_split = kps;
    // K splits meet in the caller's workspace when it holds the slabs (one problem, rows of C 16-byte aligned): plain stores + one
    // summing pass instead of a zeroing launch + fp32 atomics (768x768x3840 at split 14: 8.3 M atomics = 30 of the launch's 56 us)
    const bool use_ws = split_k > 1 && batch == 1 && ws && qt_aligned16(ws) && (long)split_k * M * N <= ws_floats &&
                        N % 4 == 0 && ldc % 4 == 0 && qt_aligned16(C) && !getenv("QT_PS_ATOMIC");
    a.ws = use_ws ? ws : nullptr; a.ws_stride = (long)M * N;
    if (split_k > 1 && !accumulate && !use_ws) {
        for (int z = 0; z < batch; ++z) {
            hipError_t e = qt_zero_2d(C + (long)z * c_bs, ldc, M, N, st);
            if (e != hipSuccess) QT_FAIL(QT_ERR_RUNTIME, "qt_gemm_x3: memset: %s", hipGetErrorString(e));
        }
    }
    bool cmax_skipped = false;
    if (c_amax) {
        if (split_k > 1) { a.c_amax = nullptr; cmax_skipped = true; }     // partial sums are not the values of C: the caller measures it (return 2)
    }
    if (a_colsum && !accumulate) {                  // accumulate: C and a_colsum both hold earlier contributions (a pre-zeroed gradient buffer)
        hipError_t e = qt_zero_1d(a_colsum, M, st);
        if (e != hipSuccess) QT_FAIL(QT_ERR_RUNTIME, "qt_gemm_x3: memset: %s", hipGetErrorString(e));
    }
    QT_CHECK_ARG(tiles < (1L << 31), "qt_gemm_x3: grid too large");
    const long total_work = tiles * split_k * batch;
    // big tile: persistent, one resident round of workgroups walks all work items; small tile: one item each
    dim3 grid((unsigned)tiles, (unsigned)split_k, (unsigned)batch);
    if (big) grid = dim3((unsigned)(total_work > slots ? slots : total_work));
    if (big) {
        if (half_type == 0) qt_x3_big_0(a, transA, transB, grid, st);
        else if (half_type == 1) qt_x3_big_1(a, transA, transB, grid, st);
        else qt_x3_big_3(a, transA, transB, grid, st);
    } else {
        if (half_type == 0) qt_x3_small_0(a, transA, transB, grid, st);
        else if (half_type == 1) qt_x3_small_1(a, transA, transB, grid, st);
        else qt_x3_small_3(a, transA, transB, grid, st);
    }
    QT_CHECK_LAUNCH("qt_gemm_x3");
    if (use_ws) { const int rc = qt_splitk_reduce(stream, ws, split_k, M, N, C, ldc, accumulate, c_amax); if (rc) return rc; return QT_OK; }   // (the summing pass measured max|C|)
    return cmax_skipped ? 2 : QT_OK;
}
#endif  // QT_X3_PART == 0
