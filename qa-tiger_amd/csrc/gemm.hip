// fp32 GEMM with fused epilogue on the gfx950 matrix cores.
//
// Replaces every `nn.Linear` / packed MHA in-/out-projection on the QA-TIGER
// path (reference src/models/modules.py:33,43 `Projection`; 174-182 MHA +
// linear1/2; 284-290; 382-398; 637-645; net.py:60 head) and, through the two
// transposed operand modes, their backward (dX = dY·W, dW = dY^T·X).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 — exact fp32 products, fp32 accumulate
// (bitwise an fmaf chain in k order), so parity with the CPU fp32 path is
// limited only by summation order.
//
// Tiling (64-wide wavefronts): 256 threads = 4 waves in a 2x2 grid; each wave
// owns TM x TN MFMA tiles of 32x32 (TM=TN=2: 128x128 block tile, 64 accumulator
// VGPRs/lane; TM=TN=1: 64x64 for skinny problems).  K is consumed in steps of
// BK=16 through double-buffered LDS tiles, register-staged (global -> VGPR ->
// LDS) so that tile t+1 is in flight while tile t feeds the MFMAs.
//
// Operand layouts in LDS:
//   "K-contiguous" operand (A [M,K] row-major, or B given as [N,K] = Linear
//   weight): tile [rows][BK] padded to 20 floats/row; a lane reads two 16-byte
//   vectors = its 8 k-values of this BK step.
//   "K-slow" operand (A given as [K,M], or B given as [K,N]): tile [BK][cols];
//   a lane reads 8 dwords, consecutive lanes consecutive columns.
// MFMA k-order inside a BK step is the permutation k = 8g + 4h + s (g: 16-byte
// group, h: lane half, s: element) for BOTH operands, which is all that a dot
// product needs.
#include "qt_common.h"
#include <stdlib.h>

#define BK 16
#define LDK (BK + 4)   // padded row of a K-contiguous tile (floats); 80 B keeps 16-B alignment

struct GemmArgs {
    int M, N, K;
    const float* A; long lda;
    const float* B; long ldb;
    float* C; long ldc;
    float alpha;
    const float* bias;                 // [N], added before the activation
    const float* add_src; long ld_add; // residual added before the activation
    int relu;
    const float* mask_src; long ld_mask; float mask_scale;  // out = mask_src>0 ? out*mask_scale : 0
    float drop_p; float inv_keep; unsigned long long drop_seed; const unsigned long long* salt;   // salt: qt_common.h "per-step dropout salt"
    int accumulate;                    // C += out
    int split_k;                       // >1: K split over gridDim.y, atomically added into C
    int k_per_split;                   // multiple of BK
    int tiles_n;
    long a_bs, b_bs, c_bs;             // strides between the gridDim.z problems of a batched launch
    float* ws; long ws_stride;         // split-K through a workspace (one problem): split z stores its partial tile at ws + z * ws_stride (row pitch N)
};

template <int TM, int TN, bool TA, bool TB, int VEC>
__global__ __launch_bounds__(256) void qt_gemm_f32_kernel(GemmArgs p) {
    if (p.salt && p.drop_p > 0.f) p.drop_seed += *p.salt;
    constexpr int BM = 64 * TM, BN = 64 * TN;
    constexpr int A_TILE = TA ? BK * BM : BM * LDK;
    constexpr int B_TILE = TB ? BN * LDK : BK * BN;
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + B_TILE)];
    p.A += (long)blockIdx.z * p.a_bs; p.B += (long)blockIdx.z * p.b_bs; p.C += (long)blockIdx.z * p.c_bs;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware tile order: workgroups that share an XCD (same blockIdx % 8,
    // round-robin dispatch) walk neighbouring N-tiles of one M-panel, so the
    // A panel is fetched into that XCD's L2 once.  Bijective for any grid size.
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (orig >> 3);
    const int tile_m = wgid / p.tiles_n, tile_n = wgid - tile_m * p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int kbeg = blockIdx.y * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    constexpr int A_LD = TM;   // float4 loads per thread per tile: BM*BK/4/256 = TM
    constexpr int B_LD = TN;
    float4 ra[A_LD], rb[B_LD];

    auto load_kcontig = [&](const float* base, long ld, int row0, int rows_total, int k0, int idx) -> float4 {
        const int row = idx >> 2, c4 = idx & 3;
        const int gr = min(row0 + row, rows_total - 1);      // clamped rows are masked at the store
        const int gk = k0 + c4 * 4;
        const float* src = base + (long)gr * ld + gk;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (VEC == 4) {
            if (gk < kend) v = *reinterpret_cast<const float4*>(src);
        } else {
            if (gk + 0 < kend) v.x = src[0];
            if (gk + 1 < kend) v.y = src[1];
            if (gk + 2 < kend) v.z = src[2];
            if (gk + 3 < kend) v.w = src[3];
        }
        return v;
    };
    auto load_kslow = [&](const float* base, long ld, int col0, int cols_total, int cols_tile, int k0, int idx) -> float4 {
        const int per_row = cols_tile >> 2;
        const int krow = idx / per_row, c4 = idx - krow * per_row;
        const int gk = k0 + krow, gc = col0 + c4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gk < kend) {
            const float* src = base + (long)gk * ld + gc;
            if (VEC == 4) {
                if (gc < cols_total) v = *reinterpret_cast<const float4*>(src);
            } else {
                if (gc + 0 < cols_total) v.x = src[0];
                if (gc + 1 < cols_total) v.y = src[1];
                if (gc + 2 < cols_total) v.z = src[2];
                if (gc + 3 < cols_total) v.w = src[3];
            }
        }
        return v;
    };
    auto gload = [&](int k0) {
#pragma unroll
        for (int it = 0; it < A_LD; ++it) {
            const int idx = tid + it * 256;
            ra[it] = TA ? load_kslow(p.A, p.lda, m0, p.M, BM, k0, idx)
                        : load_kcontig(p.A, p.lda, m0, p.M, k0, idx);
        }
#pragma unroll
        for (int it = 0; it < B_LD; ++it) {
            const int idx = tid + it * 256;
            rb[it] = TB ? load_kcontig(p.B, p.ldb, n0, p.N, k0, idx)
                        : load_kslow(p.B, p.ldb, n0, p.N, BN, k0, idx);
        }
    };
    auto sstore = [&](int buf) {
        float* sA = smem + buf * (A_TILE + B_TILE);
        float* sB = sA + A_TILE;
#pragma unroll
        for (int it = 0; it < A_LD; ++it) {
            const int idx = tid + it * 256;
            if (TA) { const int per_row = BM >> 2; const int krow = idx / per_row, c4 = idx - krow * per_row;
                      *reinterpret_cast<float4*>(sA + krow * BM + c4 * 4) = ra[it]; }
            else    { *reinterpret_cast<float4*>(sA + (idx >> 2) * LDK + (idx & 3) * 4) = ra[it]; }
        }
#pragma unroll
        for (int it = 0; it < B_LD; ++it) {
            const int idx = tid + it * 256;
            if (TB) { *reinterpret_cast<float4*>(sB + (idx >> 2) * LDK + (idx & 3) * 4) = rb[it]; }
            else    { const int per_row = BN >> 2; const int krow = idx / per_row, c4 = idx - krow * per_row;
                      *reinterpret_cast<float4*>(sB + krow * BN + c4 * 4) = rb[it]; }
        }
    };

    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        gload(kbeg);
        sstore(0);
    }
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        if (t + 1 < nk) gload(kbeg + (t + 1) * BK);          // in flight during the MFMAs below

        const float* sA = smem + cur * (A_TILE + B_TILE);
        const float* sB = sA + A_TILE;
        float af[TM][8], bf[TN][8];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm * (32 * TM) + i * 32 + r;
            if (TA) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int s = 0; s < 4; ++s) af[i][g * 4 + s] = sA[(g * 8 + h * 4 + s) * BM + row];
            } else {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 v = *reinterpret_cast<const float4*>(sA + row * LDK + g * 8 + h * 4);
                    af[i][g * 4 + 0] = v.x; af[i][g * 4 + 1] = v.y; af[i][g * 4 + 2] = v.z; af[i][g * 4 + 3] = v.w;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = wn * (32 * TN) + j * 32 + r;
            if (TB) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 v = *reinterpret_cast<const float4*>(sB + col * LDK + g * 8 + h * 4);
                    bf[j][g * 4 + 0] = v.x; bf[j][g * 4 + 1] = v.y; bf[j][g * 4 + 2] = v.z; bf[j][g * 4 + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int s = 0; s < 4; ++s) bf[j][g * 4 + s] = sB[(g * 8 + h * 4 + s) * BN + col];
            }
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);

        if (t + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)
    const bool first_split = (blockIdx.y == 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (32 * TN) + j * 32 + r;
            if (n >= p.N) continue;
            const float bias_v = (p.bias && first_split) ? p.bias[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (32 * TM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m >= p.M) continue;
                float v = p.alpha * acc[i][j][e] + bias_v;
                float* dst = p.C + (long)m * p.ldc + n;
                if (p.split_k > 1) {
                    if (p.add_src && first_split) v += p.add_src[(long)m * p.ld_add + n];
                    if (p.ws) p.ws[(long)blockIdx.y * p.ws_stride + (long)m * p.N + n] = v;     // summed by qt_splitk_reduce in a fixed order
                    else atomicAdd(dst, v);
                    continue;
                }
                if (p.add_src) v += p.add_src[(long)m * p.ld_add + n];
                if (p.relu == 1) v = qt_relu(v);
                else if (p.relu == 2) v = v / (1.0f + __expf(-1.702f * v));   // QuickGELU (CLIP text tower MLP)
                if (p.mask_src) v = (p.mask_src[(long)m * p.ld_mask + n] > 0.f) ? v * p.mask_scale : 0.f;
                if (p.drop_p > 0.f) v *= qt_drop_mult(p.drop_seed, (unsigned long long)m * (unsigned long long)p.N + n, p.drop_p, p.inv_keep);
                if (p.accumulate) v += *dst;
                *dst = v;
            }
        }
    }
}

template <int TM, int TN, int VEC>
static void launch_cfg(const GemmArgs& a, int transA, int transB, dim3 grid, hipStream_t st) {
    if (!transA && transB)      hipLaunchKernelGGL((qt_gemm_f32_kernel<TM, TN, false, true, VEC>), grid, dim3(256), 0, st, a);
    else if (!transA && !transB) hipLaunchKernelGGL((qt_gemm_f32_kernel<TM, TN, false, false, VEC>), grid, dim3(256), 0, st, a);
    else if (transA && !transB)  hipLaunchKernelGGL((qt_gemm_f32_kernel<TM, TN, true, false, VEC>), grid, dim3(256), 0, st, a);
    else                         hipLaunchKernelGGL((qt_gemm_f32_kernel<TM, TN, true, true, VEC>), grid, dim3(256), 0, st, a);
}

int qt_splitk_reduce(void* stream, const float* ws, int split, int M, int N, float* C, long ldc, int accumulate, float* c_amax);   // gemm_plane.hip

extern "C" int qt_gemm_f32_ws(void* stream, int transA, int transB, int M, int N, int K, float alpha,
                           const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                           const float* bias, const float* add_src, long ld_add, int relu,
                           const float* mask_src, long ld_mask, float mask_scale,
                           float drop_p, unsigned long long drop_seed, int accumulate, int split_k,
                           int batch, long a_bs, long b_bs, long c_bs, float* ws, long ws_floats) {
    QT_CHECK_ARG(M > 0 && N > 0 && K > 0, "qt_gemm_f32: empty problem M=%d N=%d K=%d", M, N, K);
    QT_CHECK_ARG(A && B && C, "qt_gemm_f32: null operand");
    QT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "qt_gemm_f32: drop_p=%f", drop_p);
    QT_CHECK_ARG(lda >= (transA ? M : K) && ldb >= (transB ? K : N) && ldc >= N,
                 "qt_gemm_f32: leading dimension smaller than row (lda=%ld ldb=%ld ldc=%ld)", lda, ldb, ldc);
    if (batch < 1) batch = 1;
    QT_CHECK_ARG(batch == 1 || (!bias && !add_src && !mask_src), "qt_gemm_f32: batched launches take no bias/residual/mask");
    QT_CHECK_ARG(batch <= 65535, "qt_gemm_f32: batch too large");
    hipStream_t st = (hipStream_t)stream;
    GemmArgs a;
    a.a_bs = a_bs; a.b_bs = b_bs; a.c_bs = c_bs;
    a.M = M; a.N = N; a.K = K; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
    a.alpha = alpha; a.bias = bias; a.add_src = add_src; a.ld_add = ld_add; a.relu = relu;
    a.mask_src = mask_src; a.ld_mask = ld_mask; a.mask_scale = mask_scale;
    a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.drop_seed = drop_seed; a.salt = qt_salt_ptr; a.accumulate = accumulate;

    // 16-byte vector staging needs aligned rows; otherwise the scalar-load build of the same kernel
    const bool vecA = qt_aligned16(A) && (lda % 4 == 0) && ((transA ? M : K) % 4 == 0) && (batch == 1 || a_bs % 4 == 0);
    const bool vecB = qt_aligned16(B) && (ldb % 4 == 0) && ((transB ? K : N) % 4 == 0) && (batch == 1 || b_bs % 4 == 0);
    const int vec = (vecA && vecB) ? 4 : 1;

    const bool small = ((long)M * N <= 512L * 768L) || M <= 64 || N <= 64;
    const int BMt = small ? 64 : 128, BNt = small ? 64 : 128;
    const int tiles_m = (M + BMt - 1) / BMt, tiles_n = (N + BNt - 1) / BNt;
    const long tiles = (long)tiles_m * tiles_n;
    a.tiles_n = tiles_n;

    const bool plain_epilogue = !relu && !mask_src && drop_p == 0.f;
    if (split_k <= 0) {                 // auto: fill >= ~3 workgroups per CU when the tile grid is small
        split_k = 1;
        if (plain_epilogue && tiles < 512 && K >= 8 * BK) {
            const long want = (768 + tiles - 1) / tiles;
            const long maxs = K / (4 * BK);
            split_k = (int)(want < maxs ? want : maxs);
            if (split_k < 1) split_k = 1;
        }
    }
    QT_CHECK_ARG(split_k == 1 || plain_epilogue, "qt_gemm_f32: split_k>1 needs a linear epilogue");
    int kps = (K + split_k - 1) / split_k;
    kps = ((kps + BK - 1) / BK) * BK;
    split_k = (K + kps - 1) / kps;
    a.split_k = split_k; a.k_per_split = kps;
    // K splits of one problem meet in the caller's workspace when it holds the slabs (plain stores + one summing pass, fixed order);
    // otherwise through atomics into a zeroed C
    const bool use_ws = split_k > 1 && batch == 1 && ws && (long)split_k * M * N <= ws_floats && !getenv("QT_PS_ATOMIC");
    a.ws = use_ws ? ws : nullptr; a.ws_stride = (long)M * N;
    if (split_k > 1 && !accumulate && !use_ws) {   // partial sums are atomically added into a zeroed C
        for (int z = 0; z < batch; ++z) {
            hipError_t e = qt_zero_2d(C + (long)z * c_bs, ldc, M, N, st);
            if (e != hipSuccess) QT_FAIL(QT_ERR_RUNTIME, "qt_gemm_f32: memset: %s", hipGetErrorString(e));
        }
    }
    QT_CHECK_ARG(tiles < (1L << 31), "qt_gemm_f32: grid too large");
    dim3 grid((unsigned)tiles, (unsigned)split_k, (unsigned)batch);
    if (small) { if (vec == 4) launch_cfg<1, 1, 4>(a, transA, transB, grid, st); else launch_cfg<1, 1, 1>(a, transA, transB, grid, st); }
    else       { if (vec == 4) launch_cfg<2, 2, 4>(a, transA, transB, grid, st); else launch_cfg<2, 2, 1>(a, transA, transB, grid, st); }
    QT_CHECK_LAUNCH("qt_gemm_f32");
    if (use_ws) return qt_splitk_reduce(stream, ws, split_k, M, N, C, ldc, accumulate, nullptr);
    return QT_OK;
}

extern "C" int qt_gemm_f32(void* stream, int transA, int transB, int M, int N, int K, float alpha,
                           const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                           const float* bias, const float* add_src, long ld_add, int relu,
                           const float* mask_src, long ld_mask, float mask_scale,
                           float drop_p, unsigned long long drop_seed, int accumulate, int split_k,
                           int batch, long a_bs, long b_bs, long c_bs) {
    return qt_gemm_f32_ws(stream, transA, transB, M, N, K, alpha, A, lda, B, ldb, C, ldc, bias, add_src, ld_add, relu, mask_src, ld_mask, mask_scale,
                          drop_p, drop_seed, accumulate, split_k, batch, a_bs, b_bs, c_bs, nullptr, 0);
}
