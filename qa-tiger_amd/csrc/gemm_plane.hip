// Pre-split GEMM: the fp32-equivalent projection GEMM on operands that were split ONCE, by their producer, into
// 16-bit hi/lo images, streamed global -> LDS by LDS-DMA (global_load_lds_dwordx4) with no conversion, no VALU
// staging and no ds_write in the consumer.
//
// Same role as gemm_split.hip (the nn.Linear / MHA projection GEMMs of reference src/models/modules.py and their
// backward).  gemm_split.hip converts fp32 -> (hi, lo) inside the GEMM: every activation panel is re-converted by
// every N tile and every weight panel by every M tile (840 times at M = 215040), and the convert -> ds_write ->
// barrier -> ds_read chain keeps the matrix pipes at ~49 % busy.  Here:
//
//   * SPLIT IMAGE of a row-major fp32 matrix X[R, C] (qt_split_f16 below; C a multiple of 32): for every row and
//     every block of 32 consecutive columns one 128-byte line  [ 32 x fp16 hi | 32 x fp16 lo ],
//         hi = fp16(s*x),  lo = fp16(s*x - hi),   s = 2^e a per-tensor power of two (optional)
//     -> the same bytes as the fp32 tensor, and a K step of 32 reads whole 128-byte lines.
//     With s chosen from the tensor's max |x| (qt_amax_f32) the parts sit in fp16's normal range whatever the
//     tensor's magnitude (gradients down to 1e-9, weights of 1e-4, activations of 1e5): hi + lo carries 22
//     significand bits of every element that matters for the sum, products are hi*hi + hi*lo + lo*hi with fp32
//     accumulation -> fp32-equivalent in the FORWARD AND THE BACKWARD.  The epilogue multiplies by 1/(s_a*s_b).
//   * an operand is consumed either K-contiguous ("row form": its contiguous dimension is the reduction dimension;
//     fragments by ds_read_b128) or K-slow ("column form": dW = g^T x reads both g and x this way; fragments by the
//     transposing ds_read_b64_tr_b16) FROM THE SAME IMAGE: no transposed copies.
//   * 256x256 tile, BK = 32, 8 waves of 128x64, 64 KiB of LDS per stage, two stages.  Per K step a wave issues
//     8 LDS-DMA instructions (1 KiB each), 24 ds_read_b128 (or 48 transposing reads) and 48 MFMAs
//     (v_mfma_f32_32x32x16_f16): the matrix pipe is the only unit that is busy for the whole step.
//     The DMA of step t+1 is issued right after the barrier that opens step t and lands while step t multiplies.
//     LDS images are lane-linear (DMA destination = base + 16*lane), so the bank-conflict swizzle is applied to the
//     per-lane SOURCE address and again on the fragment read (row form: 16-byte slot ^= (row>>1)&7; column form:
//     64-byte unit ^= k&3): every fragment read is conflict-free.
//   * the DMA is issued from inline asm: with the builtin, hipcc (ROCm 7.2) puts an `s_waitcnt vmcnt(0)` in front of
//     the first ds_read after every DMA (it cannot tell the two LDS stages apart), which serialises load and
//     multiply.  The waits are placed by hand: vmcnt(0) + s_barrier once per K step, one step after issue.
//   * persistent workgroups with the XCD-aware tile walk of gemm_split.hip; the first DMA of the next tile goes out
//     before the write-out of the current one.
//
// Epilogue contract = qt_gemm_x3 (bias, residual, ReLU/QuickGELU, mask, dropout, accumulate, split-K atomics).
#include "qt_common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define PBK 32                       // K step
#define P_OPER 32768                 // one operand tile of a K step: 256 rows (or 32 k-rows x 256 columns) x 128 B
#define P_A_STAGES 3                 // A (the activation / gradient operand, streamed from HBM) is prefetched TWO steps ahead,
#define P_B_BASE (P_A_STAGES * P_OPER)   // B (weights: L2 / Infinity-Cache resident) one step ahead: 3 + 2 tiles = all 160 KiB of LDS
#define P_LDS_BYTES (P_B_BASE + 2 * P_OPER)

struct PlaneArgs {
    int M, N, K;
    const char* A; long lda;         // split image of the A operand; lda = BYTES per image row
    const char* B; long ldb;
    float* C; long ldc;
    float alpha;
    const float* a_amax; const float* b_amax;   // nullable: max|x| the operand's image was scaled by (qt_split_f16)
    float* c_amax;                              // nullable: record that receives max|C| as written (never with split-K atomics)
    char* c_img; long ld_img;                   // nullable: ALSO write C as a split image (unscaled (hi | lo) fp16 lines, row pitch ld_img bytes): the operand of the
                                                // Linear layer that consumes C, without a split pass (16x16x32 form, C^T orientation, no K split)
    const float* bias;
    const float* add_src; long ld_add;
    int relu;
    const float* mask_src; long ld_mask; float mask_scale;
    float drop_p; float inv_keep; unsigned long long drop_seed; const unsigned long long* salt;   // salt: qt_common.h "per-step dropout salt"
    int accumulate;
    int split_k, k_per_split, tiles_n, tiles_m;
    long ws_stride;                  // > 0: split-K through a WORKSPACE — K split z stores its partial tile (plain stores, C^T orientation) at C + z * ws_stride
                                     // (C = the workspace, ldc = N) and qt_splitk_reduce sums the slabs: no atomics, no zeroing, a fixed summation order
    int debug;                       // QT_PS_DEBUG (profiling probes only): 1 skip the write-out, 2 DMA only for the first step of a tile, 4 no MFMAs
    int batch;                       // independent problems of one shape in one launch (the H head slabs of a fold GEMM): problem z reads A + z a_bs,
    long a_bs, b_bs, c_bs;           // B + z b_bs (bytes) and writes C + z c_bs (floats); no K split, no epilogue operands
};

// one LDS-DMA wave instruction: 64 lanes x 16 B from (sbase + voff[lane]) to LDS bytes [lds_addr, lds_addr + 1024).
// lds_addr and sbase are wave-uniform (SGPRs).  M0 carries the LDS address; nothing else in this kernel uses M0.
__device__ __forceinline__ void qt_dma16(unsigned lds_addr, const void* sbase, unsigned voff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_addr), "v"(voff), "s"(sbase) : "memory");
}

template <class F, int... Is> __device__ inline void p_static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F> __device__ inline void p_static_for(F&& f) { p_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// AKS / BKS: the operand is K-slow (its image rows are k, its contiguous dimension is m / n).
// SWAP: the MFMAs take (B fragment, A fragment), i.e. each 32x32 accumulator holds C^T: a lane then owns FOUR CONSECUTIVE COLUMNS
// of one row of C per register quad, and the write-out (and every epilogue operand) moves as 16-byte vectors — 32 stores per lane
// instead of 128.  Split-K launches keep the plain orientation: their fp32 atomics run at full rate only on 128-byte row segments.
// FMT 0: fp16 (hi | lo) lines, 32 k per 128-byte line, three v_mfma_f32_32x32x16_f16 per product (fp32-equivalent).
// FMT 1: e4m3 fp8 lines, 128 k per 128-byte line, ONE v_mfma_scale_f32_32x32x64_f8f6f4 per 64 k (unit block scales): BASELINE
//        configs[4] ("fp8 MFMA on projection GEMMs"); row-form operands only (forward GEMMs).  Same tile, same LDS map, same DMA
//        schedule: a K step is always one 128-byte line per row.
// WMT: 32-row MFMA tiles per wave along M.  4 = the 256-row tile; 3 = a 192-row tile for problems whose 256-row tiling leaves the last
//      round of the persistent walk mostly empty (M = 30720, N = 768: 360 tiles on 256 CUs = 2 rounds at 70 %; 480 tiles of 192 rows = 94 %).
//      Row-form fp16 A operands only; the B tile, the LDS map and the stage pitch are unchanged (an A stage just uses 24 of its 32 KiB).
// NWM: wave rows.  2 = 8 waves (2 x 4) of (32 WMT) x 64; 4 = SIXTEEN waves (4 x 4) of 64 x 64 on the same 256 x 256 tile (WMT = 2): four
//      waves per SIMD instead of two to cover the per-K-step barrier + first-fragment bubble.  Measured and NOT instantiated: at the
//      128 registers a wave then gets, the epilogue spills (214-327 registers) and the loop reads 1.33x more LDS per MFMA:
//      215040x2304x768 2.06 -> 2.69 ms, the dX orientation 1.93 -> 5.76 ms (tools/ab_waves16.py at commit time).
// MF: MFMA shape of the fp16 path.  0 = v_mfma_f32_32x32x16_f16 (two per 32-k line and 32x32 tile); 1 = v_mfma_f32_16x16x32_f16 (one per
//     line and 16x16 tile: the same cycles per FLOP, the same LDS bytes per wave tile, but the chip holds a higher clock on this shape
//     under a matrix-dense load, MI355X_MICROARCH.md "DVFS give-back" item 7).  Row-form operands, SWAP orientation.
// NWN: wave columns.  4 = waves of (32 WMT) x 64; 2 = FOUR waves (2 x 2) of 128 x 128 on the same 256 x 256 tile (WMT = 4, the 16x16x32 form):
//      one wave per SIMD with all 256 accumulators in AGPRs; per K step a wave reads 32 fragments for 192 MFMAs instead of 24 for 96,
//      i.e. a third less LDS traffic per MFMA, and four waves meet at the per-step barrier instead of eight.
//      Measured and NOT shipped (profiles/r03_pipe_ab.txt): bit-identical results; with the barrier-first loop 5-13 % slower than eight waves
//      (one wave per SIMD: nothing covers its LDS-DMA issue and its waits); with the late-barrier pipeline and two B fragment sets (PIPE 2)
//      2-5 % FASTER on the long-K small-output products (2304x768x215040: 1.73 vs 1.77 ms) but 15-25 % slower wherever the tile write-out
//      matters (each wave issues 64 store instructions instead of 32, and the stores are issue-bound).
// PIPE (16x16x32 form): the per-K-step barrier sits in front of the LAST segment of a step instead of in front of the first, and the last
//      segment re-loads the B fragments (each pair right after the three MFMAs that were its last readers) and the first A fragment from
//      the NEXT step's stages: when a step begins its first MFMA has its operands in registers, so the matrix pipe no longer idles through
//      barrier + ~20 KiB of fragment reads per wave at every step.  Transfers run one step further ahead on the same stages (B(t+2) and
//      A(t+3) go out right after the barrier of step t, which is also the point where every wave has finished reading B(t) and A(t)).
//      PIPE = 3 (eight waves): the barrier in front of the last TWO segments (compute16p); the default where B is K-slow.
//      PIPE = 2 (four-wave layout: 256 of its 512 registers hold the accumulators, ~130 are free): TWO B fragment sets — the next step's
//      B fragments are all requested right after the barrier, in front of the last segment's MFMAs, instead of pair by pair behind them.
template <bool AKS, bool BKS, bool SWAP, int FMT, int WMT = 4, int NWM = 2, int MF = 0, int NWN = 4, int PIPE = 0>
__global__ __launch_bounds__(NWM * NWN * 64) void qt_gemm_ps_kernel(PlaneArgs p) {
    if (p.salt && p.drop_p > 0.f) p.drop_seed += *p.salt;
    static_assert(FMT == 0 || (!AKS && !BKS), "fp8 lines are consumed K-contiguous only");
    static_assert(MF == 0 || (NWM == 2 && (FMT == 0 || SWAP)), "16x16 forms: 8 waves; the fp8 one in the C^T orientation");
    constexpr int KS = FMT == 1 ? 128 : 32;            // k per step
    static_assert((NWM == 2 && (WMT == 4 || ((WMT == 3 || WMT == 2) && !AKS && FMT == 0))) || (NWM == 4 && WMT == 2 && FMT == 0),
                  "tile shapes: 8 waves x (128|96|64) x 64, or 16 waves x 64 x 64");
    static_assert(PIPE == 0 || (MF == 1 && FMT == 0), "the late-barrier pipeline: the fp16 16x16x32 form");
    static_assert(PIPE != 2 || NWN == 2, "two B fragment sets: only the four-wave layout has the registers");
    static_assert(PIPE != 3 || NWN == 4, "the two-segment tail: the eight-wave layout");
    static_assert(NWN == 4 || (NWN == 2 && NWM == 2 && WMT == 4 && MF == 1 && FMT == 0), "four waves of 128 x 128: the fp16 16x16x32 form only");
    constexpr int WNT = 8 / NWN;                      // wave tile (32 WMT) x (32 WNT): WMT x WNT MFMA tiles of 32 x 32
    constexpr int BM = NWM * WMT * 32, BN = 256, GM = 4;
    constexpr int NBP = 32 / (NWM * NWN);              // 1 KiB DMA pieces of the B tile per wave
    constexpr int NAP = 4 * WMT / NWN;                 // ... and of the A tile (BM / 8 pieces over NWM x NWN waves)
    // A stages.  The 192- and 128-row tiles leave a quarter (half) of every 32 KiB A stage unused: at a pitch of 24 KiB the same 96 KiB
    // hold FOUR stages (B still starts at 96 KiB) and A runs one more step ahead.  Built and measured (QT_PS_NAS4, profiles/r03_pipe_ab.txt
    // section 9): bit-identical, no gain (30720x2304x768 0.297 ms against 0.288-0.298 with three stages) — the 11-19 % that the transfers
    // cost a step (QT_PS_DEBUG=2 switches them off) are not latency that a deeper prefetch hides.  Off by default.
#ifdef QT_PS_NAS4
    constexpr int NAS = ((PIPE == 1 || PIPE == 3) && WMT < 4) ? 4 : P_A_STAGES;
#else
    constexpr int NAS = P_A_STAGES;
#endif
    constexpr unsigned A_PITCH = NAS == 4 ? 24576u : (unsigned)P_OPER;
    static_assert(NAS * A_PITCH == P_B_BASE, "the A stages fill the LDS below the B stages");
    constexpr int NS = 2 * WMT;                       // compute segments per K step
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave - wm * NWN;
    const int r = lane & 31, h = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)smem;     // LDS byte address of the dynamic array (0 here)

    // ---- persistent tile walk (as gemm_split.hip): XCD-contiguous runs of tiles, M-fastest inside groups of GM M-tiles
    const int G = gridDim.x;
    const int xcd = blockIdx.x & 7, q8 = G >> 3, r8 = G & 7;
    const int slot = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (blockIdx.x >> 3);
    const int tiles = p.tiles_m * p.tiles_n;
    const long per_problem = (long)tiles * p.split_k;
    const long total = per_problem * p.batch;
    int m0 = 0, n0 = 0, kbeg = 0, kend = 0, ksplit = 0;
    const char* gA = p.A; const char* gB = p.B; float* gC = p.C;      // the current problem's operands (batch > 1: they move with the tile walk)
    auto decode = [&](long w) {
        if (p.batch > 1) {
            const int z = __builtin_amdgcn_readfirstlane((int)(w / per_problem));
            w -= (long)z * per_problem;
            gA = p.A + (long)z * p.a_bs; gB = p.B + (long)z * p.b_bs; gC = p.C + (long)z * p.c_bs;
        }
        const int t = (int)(w % tiles);
        ksplit = (int)(w / tiles);
        const int group = t / (GM * p.tiles_n);
        const int gm_rows = min(GM, p.tiles_m - group * GM);
        const int within = t - group * (GM * p.tiles_n);
        const int tile_n = within / gm_rows;
        // (the divisions above run on the vector unit; the tile coordinates feed the wave-uniform DMA base addresses, which must be scalar)
        m0 = __builtin_amdgcn_readfirstlane((group * GM + (within - tile_n * gm_rows)) * BM);
        n0 = __builtin_amdgcn_readfirstlane(tile_n * BN);
        ksplit = __builtin_amdgcn_readfirstlane(ksplit);
        kbeg = ksplit * p.k_per_split;
        kend = min(p.K, kbeg + p.k_per_split);
    };

    // ---- per-lane DMA source offsets (bytes from the operand base), 4 pieces of 1 KiB per operand per wave
    unsigned offA[NAP], offB[NBP];
    auto set_offsets = [&]() {
#pragma unroll
        for (int i = 0; i < NAP; ++i) {
            const int pc = wave * NAP + i;                       // piece 0..BM/8-1 of the A tile
            if (!AKS) {                                          // rows pc*8 .. pc*8+7, one 128-B line each
                const int row = pc * 8 + (lane >> 3);
                const int sl = (lane & 7) ^ ((row >> 1) & 7);
                offA[i] = (unsigned)((long)(min(m0 + row, p.M - 1) - m0) * p.lda + sl * 16);      // relative to the tile's first row (a_tile)
            } else {                                             // k row pc, 1 KiB = 8 blocks of 32 columns
                const int unit = (lane >> 2) ^ (pc & 3);
                const int blk = min((m0 >> 5) + (unit >> 1), (p.M >> 5) - 1);
                offA[i] = (unsigned)((long)pc * p.lda + blk * 128 + (unit & 1) * 64 + (lane & 3) * 16);
            }
        }
#pragma unroll
        for (int i = 0; i < NBP; ++i) {
            const int pc = wave * NBP + i;                       // piece 0..31 of the 32 KiB B tile
            if (!BKS) {
                const int row = pc * 8 + (lane >> 3);
                const int sl = (lane & 7) ^ ((row >> 1) & 7);
                offB[i] = (unsigned)((long)min(n0 + row, p.N - 1) * p.ldb + sl * 16);
            } else {
                const int unit = (lane >> 2) ^ (pc & 3);
                const int blk = min((n0 >> 5) + (unit >> 1), (p.N >> 5) - 1);
                offB[i] = (unsigned)((long)pc * p.ldb + blk * 128 + (unit & 1) * 64 + (lane & 3) * 16);
            }
        }
    };
    // Row-form A: the wave-uniform base moves with the tile (per-lane offsets are 32-bit: a gradient image of 537600 x 3072
    // — BASELINE config 4 at B = 128 — is 6.6 GB), m0 is uniform and current whenever a DMA of this tile is issued.
    auto dma_a = [&](int k0, int stage) {
        const char* sa = AKS ? gA + (long)k0 * p.lda : gA + (long)m0 * p.lda + (long)(k0 / KS) * 128;
        const unsigned la = lds0 + stage * A_PITCH + wave * (NAP * 1024);
#pragma unroll
        for (int i = 0; i < NAP; ++i) qt_dma16(la + i * 1024, sa, offA[i]);
    };
    auto dma_a1 = [&](int k0, int stage, int i) {          // one piece
        const char* sa = AKS ? gA + (long)k0 * p.lda : gA + (long)m0 * p.lda + (long)(k0 / KS) * 128;
        qt_dma16(lds0 + stage * A_PITCH + wave * (NAP * 1024) + i * 1024, sa, offA[i]);
    };
    auto dma_b1 = [&](int k0, int stage, int i) {
        const char* sb = BKS ? gB + (long)k0 * p.ldb : gB + (long)(k0 / KS) * 128;
        qt_dma16(lds0 + P_B_BASE + stage * P_OPER + wave * (NBP * 1024) + i * 1024, sb, offB[i]);
    };
    auto dma_b = [&](int k0, int stage) {
        const char* sb = BKS ? gB + (long)k0 * p.ldb : gB + (long)(k0 / KS) * 128;
        const unsigned lb = lds0 + P_B_BASE + stage * P_OPER + wave * (NBP * 1024);
#pragma unroll
        for (int i = 0; i < NBP; ++i) qt_dma16(lb + i * 1024, sb, offB[i]);
    };

    // ---- per-lane fragment read offsets inside a stage
    // row form: row = base + 32*i + r; 16-byte slot (hl*4 + 2*s + h) ^ ((r>>1)&7)
    // column form: lane 4q+p of 16-lane group g supplies row k = 16*s + 8*(g>>1) + 4*j + q, unit (2*b + hl) ^ q, bytes 32*(g&1) + 8*p
    unsigned fa[4], fb[4];
    {
        const int x = (r >> 1) & 7;
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (!AKS) fa[c] = (unsigned)((wm * (WMT * 32) + r) * 128 + (((c >> 1) * 4 + 2 * (c & 1) + h) ^ x) * 16);   // c = hl*2 + s
            else      fa[c] = (unsigned)(((8 * (g >> 1) + q) * 1024) + 32 * (g & 1) + 8 * pp + (wm * 2 * WMT) * 64 + ((c ^ q) * 64));   // c = unit & 3
            if (!BKS) fb[c] = (unsigned)((wn * (WNT * 32) + r) * 128 + (((c >> 1) * 4 + 2 * (c & 1) + h) ^ x) * 16);
            else      fb[c] = (unsigned)(((8 * (g >> 1) + q) * 1024) + 32 * (g & 1) + 8 * pp + (wn * 2 * WNT) * 64 + ((c ^ q) * 64));
        }
    }

    // fp8 lines: MFMA q (k 64q .. 64q+63) takes bytes [64q + 32h, +32) of a row from every lane: slots 4q + 2h and 4q + 2h + 1.
    // (Both operands use the same lane -> k assignment, so the product is the dot product whatever the instruction's internal k order.)
    unsigned fa8[4], fb8[4];
    {
        const int x = (r >> 1) & 7;
#pragma unroll
        for (int c = 0; c < 4; ++c) {                     // c = q*2 + half
            const int slot = 4 * (c >> 1) + 2 * h + (c & 1);
            fa8[c] = (unsigned)((wm * (WMT * 32) + r) * 128 + ((slot ^ x) * 16));
            fb8[c] = (unsigned)((wn * 64 + r) * 128 + ((slot ^ x) * 16));
        }
    }

    floatx16 acc[WMT][WNT];
    float cmax = 0.f;                                  // max |C| stored by this lane over all its tiles (c_amax)
    auto lds_u4 = [&](unsigned off) -> uint4 { return *reinterpret_cast<const uint4*>(smem + off); };
    auto lds_tr = [&](unsigned off) -> uint2 {
        const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(size_t)(lds0 + off));
        return __builtin_bit_cast(uint2, v);
    };
    // One K step = 8 segments (k16 half s, A row tile i): [fragments of the NEXT segment] -> [one LDS-DMA piece of a later step]
    // -> [6 MFMAs of this segment].  The DMA issue (60-185 cycles of the wave's issue stream each) sits between MFMA groups
    // instead of in one block after the barrier, where both waves of a SIMD would stall the matrix pipe together.
    auto compute = [&](unsigned sbase, unsigned sbase_b, auto&& issue) {
        uint4 ah[2], al[2], bh[2][WNT], bl[2][WNT];
        auto load_a = [&](int s, int i, uint4& xh, uint4& xl) {
            if (!AKS) {
                xh = lds_u4(sbase + fa[0 * 2 + s] + i * 4096);
                xl = lds_u4(sbase + fa[1 * 2 + s] + i * 4096);
            } else {
                // block i of the wave's 4: units 2i (hi), 2i+1 (lo); unit&3 = 2*(i&1) + hl, unit>>2 = i>>1
                const unsigned kb = sbase + (unsigned)(s * 16 * 1024 + (i >> 1) * 256);
                const uint2 h0 = lds_tr(kb + fa[2 * (i & 1) + 0]), h1 = lds_tr(kb + fa[2 * (i & 1) + 0] + 4096);
                const uint2 l0 = lds_tr(kb + fa[2 * (i & 1) + 1]), l1 = lds_tr(kb + fa[2 * (i & 1) + 1] + 4096);
                xh = make_uint4(h0.x, h0.y, h1.x, h1.y);
                xl = make_uint4(l0.x, l0.y, l1.x, l1.y);
            }
        };
        auto load_b = [&](int s, uint4* xh, uint4* xl) {
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                if (!BKS) {
                    xh[j] = lds_u4(sbase_b + fb[0 * 2 + s] + j * 4096);
                    xl[j] = lds_u4(sbase_b + fb[1 * 2 + s] + j * 4096);
                } else {
                    const unsigned kb = sbase_b + (unsigned)(s * 16 * 1024 + (j >> 1) * 256);
                    const uint2 h0 = lds_tr(kb + fb[2 * (j & 1) + 0]), h1 = lds_tr(kb + fb[2 * (j & 1) + 0] + 4096);
                    const uint2 l0 = lds_tr(kb + fb[2 * (j & 1) + 1]), l1 = lds_tr(kb + fb[2 * (j & 1) + 1] + 4096);
                    xh[j] = make_uint4(h0.x, h0.y, h1.x, h1.y);
                    xl[j] = make_uint4(l0.x, l0.y, l1.x, l1.y);
                }
            }
        };
        auto mm = [&](const uint4& x, const uint4& y, floatx16 c) {
            if (SWAP) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, y), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
            else      return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
        };
        load_b(0, bh[0], bl[0]);
        load_a(0, 0, ah[0], al[0]);
        p_static_for<NS>([&](auto SEG) {
            constexpr int seg = decltype(SEG)::value, s = seg / WMT, i = seg % WMT, cur = seg & 1;
            if constexpr (seg + 1 < NS) {
                constexpr int s1 = (seg + 1) / WMT, i1 = (seg + 1) % WMT;
                if constexpr (i1 == 0) load_b(s1, bh[s1], bl[s1]);
                load_a(s1, i1, ah[cur ^ 1], al[cur ^ 1]);
            }
            issue(SEG);
            p_static_for<WNT>([&](auto J) {
                constexpr int j = decltype(J)::value;
                acc[i][j] = mm(al[cur], bh[s][j], acc[i][j]);
                acc[i][j] = mm(ah[cur], bl[s][j], acc[i][j]);
                acc[i][j] = mm(ah[cur], bh[s][j], acc[i][j]);
            });
        });
    };

    // fp8: 8 segments (64-k half q, A row tile i): [32-byte fragments of the next segment] -> [one DMA piece] -> [2 MFMAs]
    auto compute8 = [&](unsigned sbase, unsigned sbase_b, auto&& issue) {
        typedef int intx8 __attribute__((ext_vector_type(8)));
        uint4 a0[2], a1[2], b0[2][WNT], b1[2][WNT];
        auto load_a = [&](int q, int i, uint4& x0, uint4& x1) {
            x0 = lds_u4(sbase + fa8[2 * q + 0] + i * 4096);
            x1 = lds_u4(sbase + fa8[2 * q + 1] + i * 4096);
        };
        auto load_b = [&](int q, uint4* x0, uint4* x1) {
#pragma unroll
            for (int j = 0; j < WNT; ++j) {
                x0[j] = lds_u4(sbase_b + fb8[2 * q + 0] + j * 4096);
                x1[j] = lds_u4(sbase_b + fb8[2 * q + 1] + j * 4096);
            }
        };
        auto pack = [](const uint4& lo, const uint4& hi) {
            intx8 v; v[0] = (int)lo.x; v[1] = (int)lo.y; v[2] = (int)lo.z; v[3] = (int)lo.w; v[4] = (int)hi.x; v[5] = (int)hi.y; v[6] = (int)hi.z; v[7] = (int)hi.w;
            return v;
        };
        load_b(0, b0[0], b1[0]);
        load_a(0, 0, a0[0], a1[0]);
        p_static_for<NS>([&](auto SEG) {
            constexpr int seg = decltype(SEG)::value, q = seg / WMT, i = seg % WMT, cur = seg & 1;
            if constexpr (seg + 1 < NS) {
                constexpr int q1 = (seg + 1) / WMT, i1 = (seg + 1) % WMT;
                if constexpr (i1 == 0) load_b(q1, b0[q1], b1[q1]);
                load_a(q1, i1, a0[cur ^ 1], a1[cur ^ 1]);
            }
            issue(SEG);
            p_static_for<WNT>([&](auto J) {
                constexpr int j = decltype(J)::value;
                const intx8 va = pack(a0[cur], a1[cur]), vb = pack(b0[q][j], b1[q][j]);
                // cbsz / blgp 0 = e4m3 for A / B; block scales 2^0 (E8M0 127): the per-tensor scale is applied in the epilogue
                if (SWAP) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(vb, va, acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                else      acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            });
        });
    };

    // ---- 16x16x32 form: lane (r16 = lane & 15, g4 = lane >> 4) supplies row r16 of a 16-row tile, k = 8 g4 .. 8 g4 + 7.
    // row form: that is 16-byte slot (hl*4 + g4) of the row's line; the swizzle (row >> 1) & 7 depends on r16 only (tiles start at
    //   multiples of 16 rows).  One ds_read_b128 per (tile, part) covers the whole 32-k step: 24 reads and 96 MFMAs per K step for the
    //   128 x 64 wave tile.
    // column form: the 16-lane group g4 reads k rows 8 g4 + 4 j + q (q = (lane >> 2) & 3, j = the two transposing reads) of the tile's
    //   16 columns: lane 4 q + p supplies columns 4 p .. 4 p + 3 of k row q and receives column (lane & 15) of the four k rows; the tile is
    //   half (it & 1) of block it >> 1, units 2 (it >> 1) + hl, unit position ^= k & 3 as the DMA stored it.
    typedef float floatx4_t __attribute__((ext_vector_type(4)));
    floatx4_t acc16[MF ? 2 * WMT : 1][MF ? 2 * WNT : 1];
    unsigned fa16[4], fb16[4];
    {
        const int r16 = lane & 15, g4 = lane >> 4, x16 = (r16 >> 1) & 7;
        const int q = (lane >> 2) & 3, pp = lane & 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) {                  // row form: c = hl (0, 1);  column form: c = unit & 3
            if (!AKS) fa16[c] = (unsigned)((wm * (WMT * 32) + r16) * 128 + ((((c & 1) * 4 + g4) ^ x16) * 16));
            else      fa16[c] = (unsigned)((8 * g4 + q) * 1024 + 8 * pp + (wm * 2 * WMT) * 64 + ((c ^ q) * 64));
            if (!BKS) fb16[c] = (unsigned)((wn * (WNT * 32) + r16) * 128 + ((((c & 1) * 4 + g4) ^ x16) * 16));
            else      fb16[c] = (unsigned)((8 * g4 + q) * 1024 + 8 * pp + (wn * 2 * WNT) * 64 + ((c ^ q) * 64));
        }
    }
    auto compute16 = [&](unsigned sbase, unsigned sbase_b, auto&& issue) {
        uint4 ah[2], al[2], bh[2 * WNT], bl[2 * WNT];
        auto frag = [&](unsigned base, const unsigned* f, bool kslow, int t, int hl) -> uint4 {
            if (!kslow) return lds_u4(base + f[hl] + t * 2048);
            const unsigned kb = base + (unsigned)((t >> 2) * 256 + 32 * (t & 1)) + f[2 * ((t >> 1) & 1) + hl];
            const uint2 v0 = lds_tr(kb), v1 = lds_tr(kb + 4096);
            return make_uint4(v0.x, v0.y, v1.x, v1.y);
        };
        auto mm = [&](const uint4& x, const uint4& y, floatx4_t c) {     // SWAP: C^T tile, first operand = the B (column) fragment
            if (SWAP) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, y), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
            else      return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
        };
#pragma unroll
        for (int jt = 0; jt < 2 * WNT; ++jt) bh[jt] = frag(sbase_b, fb16, BKS, jt, 0);
        al[0] = frag(sbase, fa16, AKS, 0, 1);
        ah[0] = frag(sbase, fa16, AKS, 0, 0);
#pragma unroll
        for (int jt = 0; jt < 2 * WNT; ++jt) bl[jt] = frag(sbase_b, fb16, BKS, jt, 1);
        p_static_for<NS>([&](auto IT) {
            constexpr int it = decltype(IT)::value, cur = it & 1;
            if constexpr (it + 1 < NS) {
                al[cur ^ 1] = frag(sbase, fa16, AKS, it + 1, 1);
                ah[cur ^ 1] = frag(sbase, fa16, AKS, it + 1, 0);
            }
            issue(IT);
            p_static_for<2 * WNT>([&](auto J) {
                constexpr int jt = decltype(J)::value;
                acc16[it][jt] = mm(al[cur], bh[jt], acc16[it][jt]);
                acc16[it][jt] = mm(ah[cur], bl[jt], acc16[it][jt]);
                acc16[it][jt] = mm(ah[cur], bh[jt], acc16[it][jt]);
            });
        });
    };

    // ---- PIPE: fragments that live across K steps
    uint4 pbh[PIPE == 2 ? 2 : 1][PIPE ? 2 * WNT : 1], pbl[PIPE == 2 ? 2 : 1][PIPE ? 2 * WNT : 1], pal[PIPE == 3 ? 3 : 2], pah[PIPE == 3 ? 3 : 2];
    auto frag16 = [&](unsigned base, const unsigned* f, bool kslow, int t, int hl) -> uint4 {
        if (!kslow) return lds_u4(base + f[hl] + t * 2048);
        const unsigned kb = base + (unsigned)((t >> 2) * 256 + 32 * (t & 1)) + f[2 * ((t >> 1) & 1) + hl];
        const uint2 v0 = lds_tr(kb), v1 = lds_tr(kb + 4096);
        return make_uint4(v0.x, v0.y, v1.x, v1.y);
    };
    auto first_frags = [&](unsigned sA, unsigned sB) {       // a tile's first step: nothing to hide them behind
        if constexpr (PIPE) {
#pragma unroll
            for (int jt = 0; jt < 2 * WNT; ++jt) pbh[0][jt] = frag16(sB, fb16, BKS, jt, 0);
            pal[0] = frag16(sA, fa16, AKS, 0, 1);
            pah[0] = frag16(sA, fa16, AKS, 0, 0);
#pragma unroll
            for (int jt = 0; jt < 2 * WNT; ++jt) pbl[0][jt] = frag16(sB, fb16, BKS, jt, 1);
        }
    };
    // one K step; HAS_NEXT: the tile has another step (stages sA1 / sB1), whose barrier `sync` and first fragments belong to this step's last segment
    // PAR (PIPE = 2): the B fragment set this step multiplies with; the next step's go into the other one
    // (Tried on top, no gain: the second wave of every SIMD issuing its LDS-DMA pieces in the MIDDLE of a segment's MFMAs instead of in front
    //  of them, so that one wave multiplies while the other's issue stream is held by a DMA instruction: +-0.5 % on the row-form products,
    //  -9...-12 % on the dX orientation — profiles/r03_pipe_ab.txt section 6.)
    auto compute16p = [&](auto HAS_NEXT, auto PAR, unsigned sA, unsigned sA1, unsigned sB1, auto&& issue, auto&& sync) {
        constexpr bool has_next = decltype(HAS_NEXT)::value;
        constexpr int bs = PIPE == 2 ? decltype(PAR)::value : 0, bn = PIPE == 2 ? (bs ^ 1) : 0;
        auto mm = [&](const uint4& x, const uint4& y, floatx4_t c) {
            if (SWAP) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, y), __builtin_bit_cast(f16x8, x), c, 0, 0, 0);
            else      return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
        };
        if constexpr (PIPE == 3) {
            // The barrier sits in front of the last TWO segments, which run as one: for every B pair the three MFMAs of row tile NS-2 and the
            // three of row tile NS-1, then that pair's re-load from the next step's stage — the 2 WNT re-loads are spread over 12 WNT MFMAs
            // instead of 6 WNT (eight waves re-loading 8 KiB each keep the LDS busy for ~500 cycles).  Both row tiles' A fragments are
            // fetched BEFORE the barrier (a third fragment buffer), so that after it nothing reads this step's stages any more.
            p_static_for<NS - 2>([&](auto IT) {
                constexpr int it = decltype(IT)::value, cur = it & 1;
                pal[cur ^ 1] = frag16(sA, fa16, AKS, it + 1, 1);
                pah[cur ^ 1] = frag16(sA, fa16, AKS, it + 1, 0);
                if constexpr (it == NS - 3) { pal[2] = frag16(sA, fa16, AKS, NS - 1, 1); pah[2] = frag16(sA, fa16, AKS, NS - 1, 0); }
                issue(IT);
                p_static_for<2 * WNT>([&](auto J) {
                    constexpr int jt = decltype(J)::value;
                    acc16[it][jt] = mm(pal[cur], pbh[0][jt], acc16[it][jt]);
                    acc16[it][jt] = mm(pah[cur], pbl[0][jt], acc16[it][jt]);
                    acc16[it][jt] = mm(pah[cur], pbh[0][jt], acc16[it][jt]);
                });
            });
            // (NS is even: row tile NS-2's fragments are in buffer 0, NS-1's in buffer 2; buffer 1 is free)
            if constexpr (has_next) {
                sync();
                pal[1] = frag16(sA1, fa16, AKS, 0, 1);
                pah[1] = frag16(sA1, fa16, AKS, 0, 0);
            }
            issue(std::integral_constant<int, NS - 2>{});
            p_static_for<2 * WNT>([&](auto J) {
                constexpr int jt = decltype(J)::value;
                if constexpr (jt == WNT) issue(std::integral_constant<int, NS - 1>{});
                acc16[NS - 2][jt] = mm(pal[0], pbh[0][jt], acc16[NS - 2][jt]);
                acc16[NS - 2][jt] = mm(pah[0], pbl[0][jt], acc16[NS - 2][jt]);
                acc16[NS - 2][jt] = mm(pah[0], pbh[0][jt], acc16[NS - 2][jt]);
                acc16[NS - 1][jt] = mm(pal[2], pbh[0][jt], acc16[NS - 1][jt]);
                acc16[NS - 1][jt] = mm(pah[2], pbl[0][jt], acc16[NS - 1][jt]);
                acc16[NS - 1][jt] = mm(pah[2], pbh[0][jt], acc16[NS - 1][jt]);
                if constexpr (has_next) {
                    pbh[0][jt] = frag16(sB1, fb16, BKS, jt, 0);
                    pbl[0][jt] = frag16(sB1, fb16, BKS, jt, 1);
                    __builtin_amdgcn_sched_barrier(0);               // keep each pair's re-load right behind its last readers (the scheduler otherwise gathers them in two groups)
                }
            });
            if constexpr (has_next) { pal[0] = pal[1]; pah[0] = pah[1]; }      // the next step's segment 0 multiplies out of buffer 0
        } else if constexpr (PIPE) {
            p_static_for<NS>([&](auto IT) {
                constexpr int it = decltype(IT)::value, cur = it & 1;
                if constexpr (it + 1 < NS) {
                    pal[cur ^ 1] = frag16(sA, fa16, AKS, it + 1, 1);
                    pah[cur ^ 1] = frag16(sA, fa16, AKS, it + 1, 0);
                } else if constexpr (has_next) {
                    sync();                                          // every wave has read all of A(t) (and B(t), one step ago); A(t+1), B(t+1) have landed
                    if constexpr (PIPE == 2) {
#pragma unroll
                        for (int jt = 0; jt < 2 * WNT; ++jt) pbh[bn][jt] = frag16(sB1, fb16, BKS, jt, 0);
                    }
                    pal[cur ^ 1] = frag16(sA1, fa16, AKS, 0, 1);     // (NS is even: the next step's segment 0 reads slot 0)
                    pah[cur ^ 1] = frag16(sA1, fa16, AKS, 0, 0);
                    if constexpr (PIPE == 2) {
#pragma unroll
                        for (int jt = 0; jt < 2 * WNT; ++jt) pbl[bn][jt] = frag16(sB1, fb16, BKS, jt, 1);
                    }
                }
                issue(IT);
                p_static_for<2 * WNT>([&](auto J) {
                    constexpr int jt = decltype(J)::value;
                    acc16[it][jt] = mm(pal[cur], pbh[bs][jt], acc16[it][jt]);
                    acc16[it][jt] = mm(pah[cur], pbl[bs][jt], acc16[it][jt]);
                    acc16[it][jt] = mm(pah[cur], pbh[bs][jt], acc16[it][jt]);
                    if constexpr (PIPE == 1 && it + 1 == NS && has_next) {   // the last readers of this B pair have issued: fetch the next step's pair
                        pbh[0][jt] = frag16(sB1, fb16, BKS, jt, 0);
                        pbl[0][jt] = frag16(sB1, fb16, BKS, jt, 1);
                    }
                });
            });
        }
    };

    // fp8 lines on v_mfma_scale_f32_16x16x128_f8f6f4: lane (r16, g4) supplies row r16 of a 16-row tile, k = 32 g4 .. 32 g4 + 31 = the 16-byte
    // slots 2 g4 and 2 g4 + 1 of the row's line (both operands use the same lane -> k assignment); ONE MFMA per tile and 128-k line:
    // 24 ds_read_b128 and 32 MFMAs per K step.  Same reasoning as the fp16 16x16x32 form: equal cycles per FLOP, higher sustained clock.
    auto compute8_16 = [&](unsigned sbase, unsigned sbase_b, auto&& issue) {
        typedef int intx8 __attribute__((ext_vector_type(8)));
        const int r16 = lane & 15, g4 = lane >> 4, x16 = (r16 >> 1) & 7;
        const unsigned oa0 = (unsigned)((wm * (WMT * 32) + r16) * 128 + (((2 * g4) ^ x16) * 16)), oa1 = (unsigned)((wm * (WMT * 32) + r16) * 128 + (((2 * g4 + 1) ^ x16) * 16));
        const unsigned ob0 = (unsigned)((wn * 64 + r16) * 128 + (((2 * g4) ^ x16) * 16)), ob1 = (unsigned)((wn * 64 + r16) * 128 + (((2 * g4 + 1) ^ x16) * 16));
        auto pack = [](const uint4& lo, const uint4& hi) {
            intx8 v; v[0] = (int)lo.x; v[1] = (int)lo.y; v[2] = (int)lo.z; v[3] = (int)lo.w; v[4] = (int)hi.x; v[5] = (int)hi.y; v[6] = (int)hi.z; v[7] = (int)hi.w;
            return v;
        };
        intx8 vb[2 * WNT], va[2];
#pragma unroll
        for (int jt = 0; jt < 2 * WNT; ++jt) vb[jt] = pack(lds_u4(sbase_b + ob0 + jt * 2048), lds_u4(sbase_b + ob1 + jt * 2048));
        va[0] = pack(lds_u4(sbase + oa0), lds_u4(sbase + oa1));
        p_static_for<NS>([&](auto IT) {
            constexpr int it = decltype(IT)::value, cur = it & 1;
            if constexpr (it + 1 < NS) va[cur ^ 1] = pack(lds_u4(sbase + oa0 + (it + 1) * 2048), lds_u4(sbase + oa1 + (it + 1) * 2048));
            issue(IT);
            p_static_for<2 * WNT>([&](auto J) {
                constexpr int jt = decltype(J)::value;
                acc16[it][jt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(vb[jt], va[cur], acc16[it][jt], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            });
        });
    };

    // 1 / (s_a s_b) from the operands' max|x| records, read once (all 64 lanes active here)
    float scale_ab = 1.0f;
    if (p.a_amax) { const float am = qt_amax_read(p.a_amax); scale_ab *= (FMT == 1 ? qt_fp8_scale(am) : qt_plane_scale(am)); }
    if (p.b_amax) { const float am = qt_amax_read(p.b_amax); scale_ab *= (FMT == 1 ? qt_fp8_scale(am) : qt_plane_scale(am)); }
    long w = slot;
    if (w >= total) return;
    decode(w);
    set_offsets();
    if ((p.debug & 16) && wave >= 4) __builtin_amdgcn_s_setprio(1);     // probe: static priority for the second-dispatched half (MI355X_MICROARCH.md "Two waves per SIMD" item 4)
    if (p.debug & 8) {                                 // probe: de-synchronise the CUs' tile phases (write-out bursts, DMA bursts)
        const int d = (int)((blockIdx.x >> 3) & 7) * ((kend - kbeg) / KS);
        for (int i = 0; i < d; ++i) __builtin_amdgcn_s_sleep(6);       // ~384 cycles x K steps x (0..7) = 0..7/8 of a tile
    }
    // Stages are addressed by running step counters that continue across tiles: A step g lives in stage g % 3, B step g in g & 1.
    int sa_cur = 0, gb = 0;                            // A stage of the current step; B step parity
    auto nxt3 = [](int s) { return s == NAS - 1 ? 0 : s + 1; };            // (next A stage; the name is from the three-stage days)
    auto prevA = [](int s) { return s == 0 ? NAS - 1 : s - 1; };
    auto tile_prologue = [&]() {     // !PIPE: A(0), A(1), B(0);  PIPE: A(0), B(0), A(1), B(1), A(2) [, A(3) with four A stages] — in this order (the counted waits rely on it)
        const int nk0 = (kend - kbeg) / KS;
        if constexpr (PIPE) {
            dma_a(kbeg, sa_cur);
            dma_b(kbeg, gb & 1);
            if (nk0 > 1) { dma_a(kbeg + KS, nxt3(sa_cur)); dma_b(kbeg + KS, (gb + 1) & 1); }
            if (nk0 > 2) dma_a(kbeg + 2 * KS, nxt3(nxt3(sa_cur)));
            if constexpr (NAS == 4) { if (nk0 > 3) dma_a(kbeg + 3 * KS, prevA(sa_cur)); }
        } else {
            dma_a(kbeg, sa_cur);
            if (nk0 > 1) dma_a(kbeg + KS, nxt3(sa_cur));
            dma_b(kbeg, gb & 1);
        }
    };
    tile_prologue();
    for (;;) {
#pragma unroll
        for (int i = 0; i < WMT; ++i)
#pragma unroll
            for (int j = 0; j < WNT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        if constexpr (MF) {
#pragma unroll
            for (int i = 0; i < 2 * WMT; ++i)
#pragma unroll
                for (int j = 0; j < 2 * WNT; ++j) acc16[i][j] = floatx4_t{0.f, 0.f, 0.f, 0.f};
        }
        const int nk = (kend - kbeg) / KS;
        if constexpr (PIPE) {
            // tile start: the prologue's transfers (and the previous tile's write-out: one in-order counter) are complete in every wave
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            first_frags((unsigned)(sa_cur * A_PITCH), (unsigned)(P_B_BASE + (gb & 1) * P_OPER));
            auto step = [&](const int t, auto HAS_NEXT, auto PAR) {
                constexpr int D = NAS - 1;                           // the A tile requested during step t is A(t + D); it takes the stage that held A(t-1)
                const int kb1 = kbeg + (t + 1) * KS, ka2 = kbeg + (t + D) * KS;
                const int stb = (gb + 1) & 1, sta = prevA(sa_cur);
                // Transfers are issued in CYCLES that open right after a step's barrier: cycle(t) = B(t+2) pieces, then A(t+1+D) pieces; its first
                // slot is the last segment of step t, the others are segments 0 .. NS-2 of step t+1.  (cycle(-1) = B(1), A(D) is the prologue's.)
                // At the barrier of step t the youngest transfers are the NAP pieces of A(t+D) (if it exists): everything older has landed.
                auto issue = [&](auto SEG) {
                    constexpr int seg = decltype(SEG)::value;
                    constexpr int NP = NBP + NAP;
                    if (p.debug & 2) return;                         // probe: only the tile prologue's transfers (the steps multiply stale stages)
                    p_static_for<NP>([&](auto PC) {
                        constexpr int pc = decltype(PC)::value;
                        if constexpr (PIPE == 3) {
                            // B(t+2) right behind the barrier (segments NS-2, NS-1) into the stage that held B(t); A(t+2) over segments 0 .. NS-3
                            // into the stage that held A(t-1) (every wave finished reading it before the barrier of step t-1)
                            if constexpr (pc < NBP) { if constexpr (seg == NS - 2 + (pc * 2) / NBP) { if (t + 2 < nk) dma_b1(kbeg + (t + 2) * KS, gb & 1, pc); } }
                            else if constexpr (seg == ((pc - NBP) * (NS - 2)) / NAP) { if (t > 0 && t + D < nk) dma_a1(ka2, sta, pc - NBP); }
                            return;
                        }
                        constexpr int slot = pc < NBP ? (pc * 2) / NBP : 2 + ((pc - NBP) * (NS - 2)) / NAP;     // B first (it has one step to land), A behind it
                        if constexpr (slot == 0) {
                            static_assert(pc < NBP, "slot 0 holds B pieces only");
                            if constexpr (seg == NS - 1) { if (t + 2 < nk) dma_b1(kbeg + (t + 2) * KS, gb & 1, pc); }      // cycle(t): B(t+2) -> the stage that held B(t)
                        } else if constexpr (slot - 1 == seg) {      // cycle(t-1): B(t+1), A(t+D)
                            if (t > 0) {
                                if constexpr (pc < NBP) { if (t + 1 < nk) dma_b1(kb1, stb, pc); }
                                else { if (t + D < nk) dma_a1(ka2, sta, pc - NBP); }
                            }
                        }
                    });
                };
                auto sync = [&]() {
                    if (t + D < nk) {                                // the youngest transfers are the NAP pieces of A(t+D); everything older — A(t+1), B(t+1) among it — has landed
                        if constexpr (NAP == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                        else if constexpr (NAP == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                        else if constexpr (NAP == 3) asm volatile("s_waitcnt vmcnt(3)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(2)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                    } else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                };
                const unsigned sA = (unsigned)(sa_cur * A_PITCH), sA1 = (unsigned)(nxt3(sa_cur) * A_PITCH), sB1 = (unsigned)(P_B_BASE + ((gb + 1) & 1) * P_OPER);
                compute16p(HAS_NEXT, PAR, sA, sA1, sB1, issue, sync);
                sa_cur = nxt3(sa_cur);
                ++gb;
            };
            using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
            if constexpr (PIPE == 2) {           // two steps per trip: the fragment sets alternate
                int t = 0;
                for (; t + 2 < nk; t += 2) { step(t, std::true_type{}, P0{}); step(t + 1, std::true_type{}, P1{}); }
                if (t + 2 == nk) { step(t, std::true_type{}, P0{}); step(t + 1, std::false_type{}, P1{}); }
                else step(t, std::false_type{}, P0{});
            } else {
                for (int t = 0; t + 1 < nk; ++t) step(t, std::true_type{}, P0{});
                step(nk - 1, std::false_type{}, P0{});
            }
            // every wave is done reading the last step's stages before anyone's next-tile prologue overwrites them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else
        for (int t = 0; t < nk; ++t) {
            // Issue order is B(t+1) then A(t+2) at every step, so when step t opens the only younger transfers than the ones it
            // reads (A(t): two steps old, B(t): one step old) are the 4 pieces of A(t+1) — if that was issued at all.  The first
            // step of a tile also drains the previous tile's write-out (in-order counter; a counted wait would need the exact
            // number of stores, which ragged tiles make data-dependent).
            if (t > 0 && t + 1 < nk) {
                if constexpr (NAP == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                else if constexpr (NAP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else if constexpr (NAP == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // (always: the NAP pieces of A(t+1) are the youngest transfers)
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();              // every wave's pieces of this step have landed; the stages of step t-1 are free
            asm volatile("" ::: "memory");
            const bool do_b = (t + 1 < nk) && !(p.debug & 2), do_a = (t + 2 < nk) && !(p.debug & 2);
            const int kb1 = kbeg + (t + 1) * KS, ka2 = kbeg + (t + 2) * KS;
            const int stb = (gb + 1) & 1, sta = nxt3(nxt3(sa_cur));
            auto issue = [&](auto SEG) {                   // pieces 0..NBP-1: B(t+1), then the NAP pieces of A(t+2) — this order is what vmcnt(NAP) counts on
                constexpr int seg = decltype(SEG)::value;
                constexpr int NP = NBP + NAP;                  // pieces per wave per step, spread evenly over the NS segments in this order
                p_static_for<NP>([&](auto PC) {
                    constexpr int pc = decltype(PC)::value;
                    if constexpr ((pc * NS) / NP == seg) {
                        if constexpr (pc < NBP) { if (do_b) dma_b1(kb1, stb, pc); }
                        else { if (do_a) dma_a1(ka2, sta, pc - NBP); }
                    }
                });
            };
            if (!(p.debug & 4)) {
                if constexpr (FMT == 1 && MF == 1) compute8_16((unsigned)(sa_cur * P_OPER), (unsigned)(P_B_BASE + (gb & 1) * P_OPER), issue);
                else if constexpr (FMT == 1) compute8((unsigned)(sa_cur * P_OPER), (unsigned)(P_B_BASE + (gb & 1) * P_OPER), issue);
                else if constexpr (MF == 1) compute16((unsigned)(sa_cur * P_OPER), (unsigned)(P_B_BASE + (gb & 1) * P_OPER), issue);
                else compute((unsigned)(sa_cur * P_OPER), (unsigned)(P_B_BASE + (gb & 1) * P_OPER), issue);
            }
            else p_static_for<NS>([&](auto SEG) { issue(SEG); });
            sa_cur = nxt3(sa_cur);
            ++gb;
        }
        // ---- this tile's output coordinates, kept for the epilogue while the NEXT tile's first DMA goes out
        const int em0 = m0, en0 = n0;
        const bool first_split = (ksplit == 0);
        float* const Cw = gC + (long)ksplit * p.ws_stride;       // (ws_stride = 0: C itself; gC: this tile's problem — decode() below moves it on)
        float* const eC = gC;
        w += G;
        const bool more = w < total;
        if (more) {
            decode(w);
            set_offsets();
            // next tile's A(0), A(1), B(0): their stages were last read at steps g-2, g-1, g-1 of the running count, and every wave
            // has passed the barrier of step g (the last one of this tile), i.e. has finished multiplying step g-1
            tile_prologue();
        }
        if ((p.debug & 1) && (MF ? acc16[0][0][0] : acc[0][0][0]) != 12345.678f) { if (!more) break; continue; }
        // ---- epilogue (C/D map of the 32x32 MFMA: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5))
        float alpha = p.alpha;
        alpha /= scale_ab;
        const int mrow0 = em0 + wm * (WMT * 32) + 4 * h;
        const int ncol0 = en0 + wn * (WNT * 32) + r;
        float bias_v[WNT];
#pragma unroll
        for (int j = 0; j < WNT; ++j) {
            const int n = ncol0 + j * 32;
            bias_v[j] = (p.bias && first_split && n < p.N) ? p.bias[n] : 0.f;
        }
        auto epilogue = [&](auto ADD, auto FULL) {
            constexpr bool has_add = decltype(ADD)::value, full = decltype(FULL)::value;
            const bool atomic = p.split_k > 1;
            const int act = p.relu;
            const bool use_add = has_add && (!atomic || first_split);
            int ncl[WNT];
#pragma unroll
            for (int j = 0; j < WNT; ++j) ncl[j] = min(ncol0 + j * 32, p.N - 1);
            p_static_for<WMT>([&](auto I) {
                constexpr int i = decltype(I)::value;
#pragma unroll
                for (int eh = 0; eh < 2; ++eh) {
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
                            p_static_for<WNT>([&](auto J) {
                                constexpr int j = decltype(J)::value;
                                if (ncol0 + j * 32 < p.N) {
                                    float v = alpha * acc[i][j][e] + bias_v[j];
                                    if (has_add) v += addv[e8][j];
                                    if (atomic) { atomicAdd(crow + j * 32, v); return; }
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
        // SWAP orientation: acc[i][j] holds C^T of the 32x32 tile: lane column r <-> row m of C, register e <-> column
        // n = 8*(e>>2) + 4*h + (e&3): four consecutive columns per register quad -> float4 everywhere (N % 4 == 0, 16-byte aligned rows)
        auto epilogue_sw = [&](auto ADD, auto FULL) {
            constexpr bool has_add = decltype(ADD)::value, full = decltype(FULL)::value;
            const int act = p.relu;
            const int nb = en0 + wn * (WNT * 32) + 4 * h;
            p_static_for<WMT>([&](auto I) {
                constexpr int i = decltype(I)::value;
                const int m = em0 + wm * (WMT * 32) + i * 32 + r;
                const long mc = min(m, p.M - 1);
                p_static_for<WNT>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    // phase 1: the four 16-byte operands of this 32x32 tile's row go in flight together
                    float4 bias4[4], addv[4], mskv[4], oldv[4];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nc = min(nb + j * 32 + 8 * g, p.N - 4);
                        bias4[g] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
                        if (has_add) addv[g] = *reinterpret_cast<const float4*>(p.add_src + mc * p.ld_add + nc);
                        if (full) {
                            mskv[g] = p.mask_src ? *reinterpret_cast<const float4*>(p.mask_src + mc * p.ld_mask + nc) : make_float4(1.f, 1.f, 1.f, 1.f);
                            oldv[g] = p.accumulate ? *reinterpret_cast<const float4*>(Cw + mc * p.ldc + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                    }
                    p_static_for<4>([&](auto Gq) {
                        constexpr int g = decltype(Gq)::value;
                        const int n = nb + j * 32 + 8 * g;
                        float v[4];
                        const float bb[4] = {bias4[g].x, bias4[g].y, bias4[g].z, bias4[g].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = alpha * acc[i][j][4 * g + c] + bb[c];
                        if (has_add) { v[0] += addv[g].x; v[1] += addv[g].y; v[2] += addv[g].z; v[3] += addv[g].w; }
                        if (act == 1) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] = qt_relu(v[c]);
                        } else if (act == 2) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] = v[c] / (1.0f + __expf(-1.702f * v[c]));
                        }
                        if constexpr (full) {
                            if (p.mask_src) {
                                const float mk[4] = {mskv[g].x, mskv[g].y, mskv[g].z, mskv[g].w};
#pragma unroll
                                for (int c = 0; c < 4; ++c) v[c] = (mk[c] > 0.f) ? v[c] * p.mask_scale : 0.f;
                            }
                            if (p.drop_p > 0.f) {
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    v[c] *= qt_drop_mult(p.drop_seed, (unsigned long long)m * (unsigned long long)p.N + (n + c), p.drop_p, p.inv_keep);
                            }
                            v[0] += oldv[g].x; v[1] += oldv[g].y; v[2] += oldv[g].z; v[3] += oldv[g].w;
                        }
                        if (m < p.M && n < p.N) {
                            cmax = fmaxf(fmaxf(cmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                            *reinterpret_cast<float4*>(Cw + (long)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                    });
                });
            });
        };
        // 16x16x32 form, plain orientation (split-K atomics): C/D map of the 16x16 MFMA: col = lane & 15, row = 4 (lane >> 4) + e
        auto epilogue16 = [&](auto ADD, auto FULL) {
            constexpr bool has_add = decltype(ADD)::value, full = decltype(FULL)::value;
            const bool atomic = p.split_k > 1;
            const int act = p.relu;
            const bool use_add = has_add && (!atomic || first_split);
            const int r16 = lane & 15, g4 = lane >> 4;
            const int mr0 = em0 + wm * (WMT * 32) + 4 * g4;
            const int nc0 = en0 + wn * (WNT * 32) + r16;
            int ncl[2 * WNT];
            float bias16[2 * WNT];
#pragma unroll
            for (int jt = 0; jt < 2 * WNT; ++jt) {
                ncl[jt] = min(nc0 + jt * 16, p.N - 1);
                bias16[jt] = (p.bias && first_split) ? p.bias[ncl[jt]] : 0.f;
            }
            p_static_for<(MF ? 2 * WMT : 0)>([&](auto I) {
                constexpr int it = decltype(I)::value;
                float addv[4][2 * WNT], mskv[4][2 * WNT], oldv[4][2 * WNT];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long mc = min(mr0 + it * 16 + e, p.M - 1);
#pragma unroll
                    for (int jt = 0; jt < 2 * WNT; ++jt) {
                        if (has_add) addv[e][jt] = use_add ? p.add_src[mc * p.ld_add + ncl[jt]] : 0.f;
                        if (full) {
                            mskv[e][jt] = p.mask_src ? p.mask_src[mc * p.ld_mask + ncl[jt]] : 1.f;
                            oldv[e][jt] = p.accumulate ? eC[mc * p.ldc + ncl[jt]] : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = mr0 + it * 16 + e;
                    if (m < p.M) {
                        float* crow = eC + (long)m * p.ldc + nc0;
                        p_static_for<2 * WNT>([&](auto J) {
                            constexpr int jt = decltype(J)::value;
                            if (nc0 + jt * 16 < p.N) {
                                float v = alpha * acc16[it][jt][e] + bias16[jt];
                                if (has_add) v += addv[e][jt];
                                if (atomic) { atomicAdd(crow + jt * 16, v); return; }
                                if (act == 1) v = qt_relu(v);
                                else if (act == 2) v = v / (1.0f + __expf(-1.702f * v));
                                if constexpr (full) {
                                    if (p.mask_src) v = (mskv[e][jt] > 0.f) ? v * p.mask_scale : 0.f;
                                    if (p.drop_p > 0.f)
                                        v *= qt_drop_mult(p.drop_seed, (unsigned long long)m * (unsigned long long)p.N + (nc0 + jt * 16), p.drop_p, p.inv_keep);
                                    v += oldv[e][jt];
                                }
                                cmax = fmaxf(cmax, fabsf(v));
                                crow[jt * 16] = v;
                            }
                        });
                    }
                }
            });
        };
        // 16x16x32 form: acc16[it][jt] holds C^T of a 16x16 tile: lane column r16 <-> row m of C, register e <-> column n = 4 g4 + e:
        // one float4 of four consecutive columns per tile
        auto epilogue_sw16 = [&](auto ADD, auto FULL) {
            constexpr bool has_add = decltype(ADD)::value, full = decltype(FULL)::value;
            const int act = p.relu;
            const int r16 = lane & 15, g4 = lane >> 4;
            const int nb = en0 + wn * (WNT * 32) + 4 * g4;
            p_static_for<(MF ? 2 * WMT : 0)>([&](auto I) {
                constexpr int it = decltype(I)::value;
                const int m = em0 + wm * (WMT * 32) + it * 16 + r16;
                const long mc = min(m, p.M - 1);
                float4 bias4[2 * WNT], addv[2 * WNT], mskv[2 * WNT], oldv[2 * WNT];
#pragma unroll
                for (int jt = 0; jt < 2 * WNT; ++jt) {
                    const int nc = min(nb + jt * 16, p.N - 4);
                    bias4[jt] = p.bias ? *reinterpret_cast<const float4*>(p.bias + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (has_add) addv[jt] = *reinterpret_cast<const float4*>(p.add_src + mc * p.ld_add + nc);
                    if (full) {
                        mskv[jt] = p.mask_src ? *reinterpret_cast<const float4*>(p.mask_src + mc * p.ld_mask + nc) : make_float4(1.f, 1.f, 1.f, 1.f);
                        oldv[jt] = p.accumulate ? *reinterpret_cast<const float4*>(Cw + mc * p.ldc + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                p_static_for<2 * WNT>([&](auto J) {
                    constexpr int jt = decltype(J)::value;
                    const int n = nb + jt * 16;
                    float v[4];
                    const float bb[4] = {bias4[jt].x, bias4[jt].y, bias4[jt].z, bias4[jt].w};
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = alpha * acc16[it][jt][c] + bb[c];
                    if (has_add) { v[0] += addv[jt].x; v[1] += addv[jt].y; v[2] += addv[jt].z; v[3] += addv[jt].w; }
                    if (act == 1) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = qt_relu(v[c]);
                    } else if (act == 2) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = v[c] / (1.0f + __expf(-1.702f * v[c]));
                    }
                    if constexpr (full) {
                        if (p.mask_src) {
                            const float mk[4] = {mskv[jt].x, mskv[jt].y, mskv[jt].z, mskv[jt].w};
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[c] = (mk[c] > 0.f) ? v[c] * p.mask_scale : 0.f;
                        }
                        if (p.drop_p > 0.f) {
#pragma unroll
                            for (int c = 0; c < 4; ++c)
                                v[c] *= qt_drop_mult(p.drop_seed, (unsigned long long)m * (unsigned long long)p.N + (n + c), p.drop_p, p.inv_keep);
                        }
                        v[0] += oldv[jt].x; v[1] += oldv[jt].y; v[2] += oldv[jt].z; v[3] += oldv[jt].w;
                    }
                    if (m < p.M && n < p.N) {
                        cmax = fmaxf(fmaxf(cmax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
                        *reinterpret_cast<float4*>(Cw + (long)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                        if (p.c_img) {                                  // the same four columns as fp16 parts: 8 bytes of hi, 8 bytes of lo
                            const f32x2 x01 = {v[0], v[1]}, x23 = {v[2], v[3]};
                            const f16x2 h01 = __builtin_convertvector(x01, f16x2), h23 = __builtin_convertvector(x23, f16x2);
                            const f16x2 l01 = __builtin_convertvector(x01 - __builtin_convertvector(h01, f32x2), f16x2);
                            const f16x2 l23 = __builtin_convertvector(x23 - __builtin_convertvector(h23, f32x2), f16x2);
                            char* line = p.c_img + (long)m * p.ld_img + (long)(n >> 5) * 128 + (n & 31) * 2;
                            *reinterpret_cast<uint2*>(line) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
                            *reinterpret_cast<uint2*>(line + 64) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
                        }
                    }
                });
            });
        };
        if constexpr (MF == 1 && SWAP) {
            if (p.mask_src || p.drop_p > 0.f || p.accumulate) {
                if (p.add_src) epilogue_sw16(std::true_type{}, std::true_type{}); else epilogue_sw16(std::false_type{}, std::true_type{});
            } else if (p.add_src) epilogue_sw16(std::true_type{}, std::false_type{});
            else epilogue_sw16(std::false_type{}, std::false_type{});
        } else if constexpr (MF == 1) {
            if (p.mask_src || p.drop_p > 0.f || p.accumulate) {
                if (p.add_src) epilogue16(std::true_type{}, std::true_type{}); else epilogue16(std::false_type{}, std::true_type{});
            } else if (p.add_src) epilogue16(std::true_type{}, std::false_type{});
            else epilogue16(std::false_type{}, std::false_type{});
        } else if constexpr (SWAP) {
            if (p.mask_src || p.drop_p > 0.f || p.accumulate) {
                if (p.add_src) epilogue_sw(std::true_type{}, std::true_type{}); else epilogue_sw(std::false_type{}, std::true_type{});
            } else if (p.add_src) epilogue_sw(std::true_type{}, std::false_type{});
            else epilogue_sw(std::false_type{}, std::false_type{});
        } else {
            if (p.mask_src || p.drop_p > 0.f || p.accumulate) {
                if (p.add_src) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::false_type{}, std::true_type{});
            } else if (p.add_src) epilogue(std::true_type{}, std::false_type{});
            else epilogue(std::false_type{}, std::false_type{});
        }
        if (!more) break;
    }
    if (p.c_amax) {                                    // one atomic per wave per launch (NaN propagates: it orders above every float)
        bool bad = cmax != cmax;
        cmax = qt_wave_max(bad ? 0.f : cmax);
        bad = __any(bad);
        if (bad) cmax = __builtin_nanf("");
        if (lane == 0 && (cmax > 0.f || bad))
            atomicMax(reinterpret_cast<unsigned*>(p.c_amax) + ((blockIdx.x * 8 + wave) & (QT_AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, cmax));
    }
}

// This file is compiled eight times (qa-tiger_amd/_lib.py): -DQT_PS_PART=3..9 each instantiate one group of the GEMM kernel (tile heights
// x loop form; the fp8 form), part 0 holds the operand passes and the C entry points.  The parts build in parallel.
// Parts 1 and 2 (round 2's 32x32x16 MFMA shape: QT_PS_MF=0, with -DQT_PS_WITH_MF0 on every part) and parts 10 and 11 (the four-wave layout,
// -DQT_PS_WITH_W4) are built only when QT_PS_WITH_MF0 / QT_PS_WITH_W4 is set in the environment: measured alternatives, kept for A/B runs.
#ifndef QT_PS_PART
#define QT_PS_PART 0
#endif
// one launcher per group: (a_kslow, b_kslow, swap) and, for the short tiles, WMT select the instance
#define QT_PS_DECL(NAME) void NAME(const PlaneArgs& a, int a_kslow, int b_kslow, int swap, int wmt, dim3 grid, hipStream_t st)
QT_PS_DECL(qt_ps_launch_mf0_t4); QT_PS_DECL(qt_ps_launch_mf0_t32); QT_PS_DECL(qt_ps_launch_fp8); QT_PS_DECL(qt_ps_launch_mf1_t4); QT_PS_DECL(qt_ps_launch_mf1_t32);
QT_PS_DECL(qt_ps_launch_mf1_w4); QT_PS_DECL(qt_ps_launch_mf1_t4_pipe); QT_PS_DECL(qt_ps_launch_mf1_w4_pipe); QT_PS_DECL(qt_ps_launch_mf1_t32_pipe);
QT_PS_DECL(qt_ps_launch_mf1_t4_pipe3); QT_PS_DECL(qt_ps_launch_mf1_t32_pipe3);
#define QT_PS_GO_T(THREADS, ...)                                                                               \
    do {                                                                                                      \
        auto kern = qt_gemm_ps_kernel<__VA_ARGS__>;                                                           \
        static bool attr_set = false;   /* once per instance, not per launch (and never inside a stream capture) */ \
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P_LDS_BYTES); attr_set = true; } \
        hipLaunchKernelGGL(kern, grid, dim3(THREADS), P_LDS_BYTES, st, a);                                    \
    } while (0)
#define QT_PS_GO(...) QT_PS_GO_T(512, __VA_ARGS__)
#define QT_PS_GO4(...) QT_PS_GO_T(256, __VA_ARGS__, 4, 2, 1, 2)      /* four waves of 128 x 128, the 16x16x32 form */
#define QT_PS_GO8P(...) QT_PS_GO_T(512, __VA_ARGS__, 4, 2, 1, 4, 1)  /* eight waves, late-barrier pipeline */
#define QT_PS_GO8P3(...) QT_PS_GO_T(512, __VA_ARGS__, 4, 2, 1, 4, 3) /* eight waves, late-barrier pipeline with the two-segment tail */
#define QT_PS_GO4P(...) QT_PS_GO_T(256, __VA_ARGS__, 4, 2, 1, 2, 2)  /* four waves, late-barrier pipeline, two B fragment sets */
#define QT_PS_ALL8(GO)                                                                                         \
    do {                                                                                                      \
        if (swap) { if (!a_kslow && !b_kslow) GO(false, false, true, 0); else if (!a_kslow) GO(false, true, true, 0);        \
                    else if (!b_kslow) GO(true, false, true, 0); else GO(true, true, true, 0); }                             \
        else      { if (!a_kslow && !b_kslow) GO(false, false, false, 0); else if (!a_kslow) GO(false, true, false, 0);      \
                    else if (!b_kslow) GO(true, false, false, 0); else GO(true, true, false, 0); }                           \
    } while (0)
#define QT_PS_TALL(MF_)                                                                                        \
    do {                                                                                                      \
        if (swap) { if (!a_kslow && !b_kslow) QT_PS_GO(false, false, true, 0, 4, 2, MF_); else if (!a_kslow) QT_PS_GO(false, true, true, 0, 4, 2, MF_);      \
                    else if (!b_kslow) QT_PS_GO(true, false, true, 0, 4, 2, MF_); else QT_PS_GO(true, true, true, 0, 4, 2, MF_); }                           \
        else      { if (!a_kslow && !b_kslow) QT_PS_GO(false, false, false, 0, 4, 2, MF_); else if (!a_kslow) QT_PS_GO(false, true, false, 0, 4, 2, MF_);    \
                    else if (!b_kslow) QT_PS_GO(true, false, false, 0, 4, 2, MF_); else QT_PS_GO(true, true, false, 0, 4, 2, MF_); }                         \
    } while (0)
#define QT_PS_SHORT(MF_)                                                                                       \
    do {                                                                                                      \
        if (wmt == 3) { if (swap) { if (b_kslow) QT_PS_GO(false, true, true, 0, 3, 2, MF_); else QT_PS_GO(false, false, true, 0, 3, 2, MF_); }               \
                        else      { if (b_kslow) QT_PS_GO(false, true, false, 0, 3, 2, MF_); else QT_PS_GO(false, false, false, 0, 3, 2, MF_); } }           \
        else          { if (swap) { if (b_kslow) QT_PS_GO(false, true, true, 0, 2, 2, MF_); else QT_PS_GO(false, false, true, 0, 2, 2, MF_); }               \
                        else      { if (b_kslow) QT_PS_GO(false, true, false, 0, 2, 2, MF_); else QT_PS_GO(false, false, false, 0, 2, 2, MF_); } }           \
    } while (0)
#if QT_PS_PART == 1
QT_PS_DECL(qt_ps_launch_mf0_t4) { QT_PS_TALL(0); }
#elif QT_PS_PART == 2
QT_PS_DECL(qt_ps_launch_mf0_t32) { QT_PS_SHORT(0); }
#elif QT_PS_PART == 3
QT_PS_DECL(qt_ps_launch_fp8) {          // (wmt carries the MFMA shape here: 1 = the 16x16x128 form, C^T orientation only)
    if (swap && wmt == 1) QT_PS_GO(false, false, true, 1, 4, 2, 1);
    else if (swap) QT_PS_GO(false, false, true, 1);
    else QT_PS_GO(false, false, false, 1);
}
#elif QT_PS_PART == 4
QT_PS_DECL(qt_ps_launch_mf1_t4) { QT_PS_TALL(1); }
#elif QT_PS_PART == 5
QT_PS_DECL(qt_ps_launch_mf1_t32) { QT_PS_SHORT(1); }
#elif QT_PS_PART == 8
QT_PS_DECL(qt_ps_launch_mf1_t4_pipe3) { QT_PS_ALL8(QT_PS_GO8P3); }
#elif QT_PS_PART == 9
QT_PS_DECL(qt_ps_launch_mf1_t32_pipe3) {
    if (wmt == 3) { if (swap) { if (b_kslow) QT_PS_GO(false, true, true, 0, 3, 2, 1, 4, 3); else QT_PS_GO(false, false, true, 0, 3, 2, 1, 4, 3); }
                    else      { if (b_kslow) QT_PS_GO(false, true, false, 0, 3, 2, 1, 4, 3); else QT_PS_GO(false, false, false, 0, 3, 2, 1, 4, 3); } }
    else          { if (swap) { if (b_kslow) QT_PS_GO(false, true, true, 0, 2, 2, 1, 4, 3); else QT_PS_GO(false, false, true, 0, 2, 2, 1, 4, 3); }
                    else      { if (b_kslow) QT_PS_GO(false, true, false, 0, 2, 2, 1, 4, 3); else QT_PS_GO(false, false, false, 0, 2, 2, 1, 4, 3); } }
}
#elif QT_PS_PART == 10
QT_PS_DECL(qt_ps_launch_mf1_w4) { QT_PS_ALL8(QT_PS_GO4); }
#elif QT_PS_PART == 6
QT_PS_DECL(qt_ps_launch_mf1_t4_pipe) { QT_PS_ALL8(QT_PS_GO8P); }
#elif QT_PS_PART == 11
QT_PS_DECL(qt_ps_launch_mf1_w4_pipe) { QT_PS_ALL8(QT_PS_GO4P); }
#elif QT_PS_PART == 7
QT_PS_DECL(qt_ps_launch_mf1_t32_pipe) {      // the 192- and 128-row tiles with the late-barrier pipeline
    if (wmt == 3) { if (swap) { if (b_kslow) QT_PS_GO(false, true, true, 0, 3, 2, 1, 4, 1); else QT_PS_GO(false, false, true, 0, 3, 2, 1, 4, 1); }
                    else      { if (b_kslow) QT_PS_GO(false, true, false, 0, 3, 2, 1, 4, 1); else QT_PS_GO(false, false, false, 0, 3, 2, 1, 4, 1); } }
    else          { if (swap) { if (b_kslow) QT_PS_GO(false, true, true, 0, 2, 2, 1, 4, 1); else QT_PS_GO(false, false, true, 0, 2, 2, 1, 4, 1); }
                    else      { if (b_kslow) QT_PS_GO(false, true, false, 0, 2, 2, 1, 4, 1); else QT_PS_GO(false, false, false, 0, 2, 2, 1, 4, 1); } }
}
#else

// ---------------------------------------------------------------------------------------------------------------
// fp32 [R, C] (row pitch ldx) -> split image [R][C/32][hi 32 | lo 32] (row pitch ld_img BYTES), optional per-tensor scale from
// a device max|x|, optional column sums (the bias gradient of a Linear layer whose dY is being split for the dW / dX GEMMs).
// ---------------------------------------------------------------------------------------------------------------
__device__ inline void qt_split8(const float4& a, const float4& b, float s, uint4& hi, uint4& lo) {
    auto pair = [&](float x0, float x1, unsigned& ho, unsigned& lw) {
        const f32x2 x = {x0 * s, x1 * s};
        const f16x2 hh = __builtin_convertvector(x, f16x2);
        const f32x2 d = x - __builtin_convertvector(hh, f32x2);
        ho = __builtin_bit_cast(unsigned, hh);
        lw = __builtin_bit_cast(unsigned, __builtin_convertvector(d, f16x2));
    };
    pair(a.x, a.y, hi.x, lo.x); pair(a.z, a.w, hi.y, lo.y); pair(b.x, b.y, hi.z, lo.z); pair(b.z, b.w, hi.w, lo.w);
}

// grid-stride over (row, group of 8 columns); consecutive lanes take consecutive groups: 32-byte reads, 16-byte writes, whole lines per 4 lanes
__global__ __launch_bounds__(256) void qt_split_f16_kernel(const float* __restrict__ x, long ldx, long R, int C, const float* __restrict__ amax,
                                                           char* __restrict__ img, long ld_img) {
    const float s = amax ? qt_plane_scale(qt_amax_read(amax)) : 1.0f;
    const int g8 = C >> 3;
    const long total = R * g8;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long row = idx / g8; const int c8 = (int)(idx - row * g8);
        const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c8 * 8);
        const float4 b = *reinterpret_cast<const float4*>(x + row * ldx + c8 * 8 + 4);
        uint4 hi, lo;
        qt_split8(a, b, s, hi, lo);
        char* line = img + row * ld_img + (long)(c8 >> 2) * 128 + (c8 & 3) * 16;
        *reinterpret_cast<uint4*>(line) = hi;
        *reinterpret_cast<uint4*>(line + 64) = lo;
    }
}
// dense rows (ldx == C, image pitch == 4 C): the tensor is one flat run of 8-element groups and the image one flat run of lines,
// so there is no index arithmetic; four independent groups per thread per trip keep 128 B of loads in flight per lane
__global__ __launch_bounds__(256) void qt_split_f16_flat_kernel(const float* __restrict__ x, long n8, const float* __restrict__ amax,
                                                                char* __restrict__ img) {
    const float s = amax ? qt_plane_scale(qt_amax_read(amax)) : 1.0f;
    const long stride = (long)gridDim.x * 256;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    for (; idx + 3 * stride < n8; idx += 4 * stride) {
        float4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const float4*>(x + (idx + u * stride) * 8);
            b[u] = *reinterpret_cast<const float4*>(x + (idx + u * stride) * 8 + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long i = idx + u * stride;
            uint4 hi, lo;
            qt_split8(a[u], b[u], s, hi, lo);
            char* line = img + (i >> 2) * 128 + (i & 3) * 16;
            *reinterpret_cast<uint4*>(line) = hi;
            *reinterpret_cast<uint4*>(line + 64) = lo;
        }
    }
    for (; idx < n8; idx += stride) {
        const float4 a = *reinterpret_cast<const float4*>(x + idx * 8);
        const float4 b = *reinterpret_cast<const float4*>(x + idx * 8 + 4);
        uint4 hi, lo;
        qt_split8(a, b, s, hi, lo);
        char* line = img + (idx >> 2) * 128 + (idx & 3) * 16;
        *reinterpret_cast<uint4*>(line) = hi;
        *reinterpret_cast<uint4*>(line + 64) = lo;
    }
}

// the same with column sums: block = 32 column groups (256 columns, 1 KiB of a row) x 8 rows; grid.x = column strips, grid.y = row
// chunks.  A thread keeps its 8 columns, walks the chunk's rows 8 apart with four rows of loads in flight, and the block's 8 row
// lanes meet in LDS before one atomic per column.
#define QT_SPC_TX 32
#define QT_SPC_TY 8
__global__ __launch_bounds__(256) void qt_split_f16_colsum_kernel(const float* __restrict__ x, long ldx, long R, int C, const float* __restrict__ amax,
                                                                  char* __restrict__ img, long ld_img, int rows_per_chunk, float* __restrict__ colsum) {
    const float s = amax ? qt_plane_scale(qt_amax_read(amax)) : 1.0f;
    const int tx = threadIdx.x & (QT_SPC_TX - 1), ty = threadIdx.x / QT_SPC_TX;
    const int c8 = blockIdx.x * QT_SPC_TX + tx;
    const bool on = c8 * 8 < C;
    const long r_beg = (long)blockIdx.y * rows_per_chunk, r_end = min(R, r_beg + rows_per_chunk);
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (on) {
        const float* xc = x + c8 * 8;
        char* ic = img + (long)(c8 >> 2) * 128 + (c8 & 3) * 16;
        long row = r_beg + ty;
        for (; row + 3 * QT_SPC_TY < r_end; row += 4 * QT_SPC_TY) {
            float4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = *reinterpret_cast<const float4*>(xc + (row + u * QT_SPC_TY) * ldx);
                b[u] = *reinterpret_cast<const float4*>(xc + (row + u * QT_SPC_TY) * ldx + 4);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                cs[0] += a[u].x; cs[1] += a[u].y; cs[2] += a[u].z; cs[3] += a[u].w; cs[4] += b[u].x; cs[5] += b[u].y; cs[6] += b[u].z; cs[7] += b[u].w;
                uint4 hi, lo;
                qt_split8(a[u], b[u], s, hi, lo);
                char* line = ic + (row + u * QT_SPC_TY) * ld_img;
                *reinterpret_cast<uint4*>(line) = hi;
                *reinterpret_cast<uint4*>(line + 64) = lo;
            }
        }
        for (; row < r_end; row += QT_SPC_TY) {
            const float4 a = *reinterpret_cast<const float4*>(xc + row * ldx);
            const float4 b = *reinterpret_cast<const float4*>(xc + row * ldx + 4);
            cs[0] += a.x; cs[1] += a.y; cs[2] += a.z; cs[3] += a.w; cs[4] += b.x; cs[5] += b.y; cs[6] += b.z; cs[7] += b.w;
            uint4 hi, lo;
            qt_split8(a, b, s, hi, lo);
            char* line = ic + row * ld_img;
            *reinterpret_cast<uint4*>(line) = hi;
            *reinterpret_cast<uint4*>(line + 64) = lo;
        }
    }
    __shared__ float red[QT_SPC_TY][QT_SPC_TX][9];
#pragma unroll
    for (int e = 0; e < 8; ++e) red[ty][tx][e] = cs[e];
    __syncthreads();
    // 256 threads = 32 column groups x 8 columns: thread t sums column (t>>3 group, t&7 element) over the 8 row lanes
    {
        const int gx = threadIdx.x >> 3, e = threadIdx.x & 7;
        const int col = (blockIdx.x * QT_SPC_TX + gx) * 8 + e;
        if (col < C) {
            float v = 0.f;
#pragma unroll
            for (int y = 0; y < QT_SPC_TY; ++y) v += red[y][gx][e];
            atomicAdd(colsum + col, v);
        }
    }
}

// max |x| of an fp32 [R, C] tensor into *amax (device float, must be zeroed by the caller or hold an earlier max): non-negative
// floats order like unsigned integers, so one atomicMax on the bit pattern per wave.  NaN / inf propagate as a huge pattern: the
// scale helper then returns 1 and the GEMM's result is non-finite, as it would be in fp32.
__global__ __launch_bounds__(256) void qt_amax_kernel(const float* __restrict__ x, long ldx, long R, int C, float* __restrict__ amax) {
    const int g4 = C >> 2;
    const long total = R * g4;
    float m = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long row = idx / g4; const int c4 = (int)(idx - row * g4);
        const float4 a = *reinterpret_cast<const float4*>(x + row * ldx + c4 * 4);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
    }
    __shared__ float red[4];
    qt_amax_fold(m, red, amax);
}
// dense rows: one flat run of float4s, eight independent loads per thread per trip
__global__ __launch_bounds__(256) void qt_amax_flat_kernel(const float* __restrict__ x, long n4, float* __restrict__ amax) {
    const long stride = (long)gridDim.x * 256;
    long idx = (long)blockIdx.x * 256 + threadIdx.x;
    float m0 = 0.f, m1 = 0.f;
    for (; idx + 7 * stride < n4; idx += 8 * stride) {
        float4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = reinterpret_cast<const float4*>(x)[idx + u * stride];
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            m0 = fmaxf(fmaxf(m0, fmaxf(fabsf(a[u].x), fabsf(a[u].y))), fmaxf(fabsf(a[u].z), fabsf(a[u].w)));
            m1 = fmaxf(fmaxf(m1, fmaxf(fabsf(a[u + 1].x), fabsf(a[u + 1].y))), fmaxf(fabsf(a[u + 1].z), fabsf(a[u + 1].w)));
        }
    }
    for (; idx < n4; idx += stride) {
        const float4 a = reinterpret_cast<const float4*>(x)[idx];
        m0 = fmaxf(fmaxf(m0, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
    }
    __shared__ float red[4];
    qt_amax_fold(fmaxf(m0, m1), red, amax);
}

static __global__ void qt_zero_n_kernel(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = 0.f; }

// max |x| of many tensors living in one flat buffer (train.FlatParams: every parameter of the model) in ONE launch:
// segment s = flat[off[s] .. off[s] + cnt[s]); grid = (blocks per segment, segments); amax[s] must be zero on entry.
__global__ __launch_bounds__(256) void qt_amax_segments_kernel(const float* __restrict__ flat, const long* __restrict__ off, const long* __restrict__ cnt,
                                                               float* __restrict__ amax) {
    const int sgm = blockIdx.y;
    const float* x = flat + off[sgm];
    const long n = cnt[sgm];
    if ((long)blockIdx.x * 256 >= n) return;                          // block-uniform
    float m = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    __shared__ float red[4];
    qt_amax_fold(m, red, amax + (long)sgm * QT_AMAX_SLOTS);
}

extern "C" int qt_amax_segments(void* stream, const float* flat, const long* offsets, const long* counts, int n_segments, float* amax) {
    QT_CHECK_ARG(flat && offsets && counts && amax && n_segments > 0, "qt_amax_segments: bad arguments");
    QT_CHECK_ARG(n_segments <= 65535, "qt_amax_segments: too many segments");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = qt_zero_1d(amax, (long)n_segments * QT_AMAX_SLOTS, st);
    if (e != hipSuccess) QT_FAIL(QT_ERR_RUNTIME, "qt_amax_segments: zeroing: %s", hipGetErrorString(e));
    // 64 workgroups per segment (the large weight matrices are ~2.4 M elements; workgroups past a short segment's end fold a 0)
    hipLaunchKernelGGL(qt_amax_segments_kernel, dim3(64, (unsigned)n_segments), dim3(256), 0, st, flat, offsets, counts, amax);
    QT_CHECK_LAUNCH("qt_amax_segments");
    return QT_OK;
}

// Split images of n_segments tensors of one flat buffer in one launch (every weight matrix after an optimizer step): segment s is
// flat[off[s] .. off[s] + cnt[s]) and its image occupies the SAME element range of `img` (an image has the size of its tensor);
// each is scaled by its own record amax[s].  Segments that cannot be an image (start or length not a multiple of 32 elements:
// biases, LayerNorm vectors) are skipped.
__global__ __launch_bounds__(256) void qt_split_segments_kernel(const float* __restrict__ flat, const long* __restrict__ off, const long* __restrict__ cnt,
                                                                const float* __restrict__ amax, char* __restrict__ img) {
    const int sgm = blockIdx.y;
    const long o = off[sgm], n = cnt[sgm];
    if ((o & 31) || (n & 31)) return;                                 // block-uniform
    const float s = qt_plane_scale(qt_amax_read(amax + (long)sgm * QT_AMAX_SLOTS));
    const float* x = flat + o;
    char* im = img + o * 4;
    const long n8 = n >> 3;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n8; idx += (long)gridDim.x * 256) {
        const float4 a = *reinterpret_cast<const float4*>(x + idx * 8);
        const float4 b = *reinterpret_cast<const float4*>(x + idx * 8 + 4);
        uint4 hi, lo;
        qt_split8(a, b, s, hi, lo);
        char* line = im + (idx >> 2) * 128 + (idx & 3) * 16;
        *reinterpret_cast<uint4*>(line) = hi;
        *reinterpret_cast<uint4*>(line + 64) = lo;
    }
}

extern "C" int qt_split_f16_segments(void* stream, const float* flat, const long* offsets, const long* counts, int n_segments,
                                     const float* amax, void* img) {
    QT_CHECK_ARG(flat && offsets && counts && amax && img && n_segments > 0, "qt_split_f16_segments: bad arguments");
    QT_CHECK_ARG(n_segments <= 65535 && qt_aligned16(flat) && ((size_t)img & 127) == 0, "qt_split_f16_segments: segments / alignment");
    hipLaunchKernelGGL(qt_split_segments_kernel, dim3(64, (unsigned)n_segments), dim3(256), 0, (hipStream_t)stream, flat, offsets, counts, amax, (char*)img);
    QT_CHECK_LAUNCH("qt_split_f16_segments");
    return QT_OK;
}

// any row length / alignment (the head's [B, 42] logit gradients, ...)
__global__ __launch_bounds__(256) void qt_amax_scalar_kernel(const float* __restrict__ x, long ldx, long R, int C, float* __restrict__ amax) {
    const long total = R * C;
    float m = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long row = idx / C;
        m = fmaxf(m, fabsf(x[row * ldx + (idx - row * C)]));
    }
    __shared__ float red[4];
    qt_amax_fold(m, red, amax);
}

extern "C" int qt_amax_f32(void* stream, const float* x, long ldx, long R, int C, float* amax, int zero_first) {
    QT_CHECK_ARG(x && amax && R > 0 && C > 0 && ldx >= C, "qt_amax_f32: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (zero_first) hipLaunchKernelGGL(qt_zero_n_kernel, dim3(1), dim3(64), 0, st, amax, QT_AMAX_SLOTS);
    if (!(C % 4 == 0 && ldx % 4 == 0 && qt_aligned16(x))) {
        long blocks = (R * C + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(qt_amax_scalar_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, ldx, R, C, amax);
        QT_CHECK_LAUNCH("qt_amax_f32");
        return QT_OK;
    }
    const long n4 = R * (C >> 2);
    long blocks = (n4 + 256 * 8 - 1) / (256 * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    if (ldx == C) hipLaunchKernelGGL(qt_amax_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n4, amax);
    else hipLaunchKernelGGL(qt_amax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, ldx, R, C, amax);
    QT_CHECK_LAUNCH("qt_amax_f32");
    return QT_OK;
}

extern "C" int qt_split_f16(void* stream, const float* x, long ldx, long R, int C, const float* amax, void* img, long ld_img_bytes,
                            float* colsum, int colsum_accumulate) {
    QT_CHECK_ARG(x && img && R > 0 && C > 0 && ldx >= C, "qt_split_f16: bad arguments");
    QT_CHECK_ARG(C % 32 == 0, "qt_split_f16: the image is made of 32-column lines (C=%d)", C);
    QT_CHECK_ARG(ldx % 4 == 0 && qt_aligned16(x) && qt_aligned16(img) && ld_img_bytes % 128 == 0 && ld_img_bytes >= 4L * C,
                 "qt_split_f16: alignment (ldx=%ld, ld_img=%ld)", ldx, ld_img_bytes);
    hipStream_t st = (hipStream_t)stream;
    if (!colsum) {
        const long n8 = R * (C >> 3);
        long blocks = (n8 + 256 * 4 - 1) / (256 * 4);
        if (blocks > 8192) blocks = 8192;
        if (blocks < 1) blocks = 1;
        if (ldx == C && ld_img_bytes == 4L * C)
            hipLaunchKernelGGL(qt_split_f16_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n8, amax, (char*)img);
        else
            hipLaunchKernelGGL(qt_split_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, ldx, R, C, amax, (char*)img, ld_img_bytes);
    } else {
        const int strips = (C / 8 + QT_SPC_TX - 1) / QT_SPC_TX;
        long chunks = (4096 + strips - 1) / strips;                 // ~4096 workgroups
        if (chunks > (R + 31) / 32) chunks = (R + 31) / 32;
        if (chunks < 1) chunks = 1;
        const int rpc = (int)((((R + chunks - 1) / chunks) + 31) / 32 * 32);
        chunks = (R + rpc - 1) / rpc;
        if (!colsum_accumulate) hipLaunchKernelGGL(qt_zero_n_kernel, dim3((C + 255) / 256), dim3(256), 0, st, colsum, C);
        hipLaunchKernelGGL(qt_split_f16_colsum_kernel, dim3((unsigned)strips, (unsigned)chunks), dim3(256), 0, st, x, ldx, R, C, amax,
                           (char*)img, ld_img_bytes, rpc, colsum);
    }
    QT_CHECK_LAUNCH("qt_split_f16");
    return QT_OK;
}


// Sum of the K-split slabs of a workspace GEMM: C[m, n] (+)= sum_z ws[z][m][n], slabs in a fixed order (deterministic, unlike atomics);
// c_amax (nullable, zeroed by the caller): max|C| of the values as stored — what a split launch's own epilogue cannot know.
__global__ __launch_bounds__(256) void qt_splitk_reduce_kernel(const float* __restrict__ ws, int split, long mn, int N, float* __restrict__ C, long ldc, int accumulate,
                                                               float* __restrict__ c_amax) {
    __shared__ float red[4];
    const long n4 = mn >> 2;
    float mx = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 acc = *reinterpret_cast<const float4*>(ws + i * 4);
        for (int z = 1; z < split; ++z) {
            const float4 v = *reinterpret_cast<const float4*>(ws + (long)z * mn + i * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const long e = i * 4, m = e / N;
        float4* dst = reinterpret_cast<float4*>(C + m * ldc + (e - m * N));          // N % 4 == 0: the four columns share a row
        if (accumulate) { const float4 o = *dst; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
        *dst = acc;
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
        if (acc.x != acc.x || acc.y != acc.y || acc.z != acc.z || acc.w != acc.w) mx = __builtin_nanf("");
    }
    if (c_amax) qt_amax_fold(mx != mx ? __builtin_inff() : mx, red, c_amax);         // (a NaN in C reads as an infinite maximum: the consumer's scale degenerates loudly)
}
__global__ __launch_bounds__(256) void qt_splitk_reduce1_kernel(const float* __restrict__ ws, int split, long mn, int N, float* __restrict__ C, long ldc, int accumulate,
                                                                float* __restrict__ c_amax) {
    __shared__ float red[4];
    float mx = 0.f;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < mn; e += (long)gridDim.x * 256) {     // any N, any alignment
        float acc = ws[e];
        for (int z = 1; z < split; ++z) acc += ws[(long)z * mn + e];
        const long m = e / N;
        float* dst = C + m * ldc + (e - m * N);
        if (accumulate) acc += *dst;
        *dst = acc;
        mx = acc != acc ? __builtin_inff() : fmaxf(mx, fabsf(acc));
    }
    if (c_amax) qt_amax_fold(mx, red, c_amax);
}
int qt_splitk_reduce(void* stream, const float* ws, int split, int M, int N, float* C, long ldc, int accumulate, float* c_amax) {     // (also used by gemm_split.hip, gemm.hip)
    const bool vec = N % 4 == 0 && ldc % 4 == 0 && qt_aligned16(C) && qt_aligned16(ws);
    long blocks = (((long)M * N >> (vec ? 2 : 0)) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (!vec) hipLaunchKernelGGL(qt_splitk_reduce1_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ws, split, (long)M * N, N, C, ldc, accumulate, c_amax);
    else hipLaunchKernelGGL(qt_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ws, split, (long)M * N, N, C, ldc, accumulate, c_amax);
    QT_CHECK_LAUNCH("qt_splitk_reduce");
    return QT_OK;
}

// Shared launcher.  fmt 0: fp16 (hi | lo) lines, K step 32;  fmt 1: e4m3 lines, K step 128, row-form operands only.
static int plane_gemm(const char* who, int fmt, void* stream, int a_kslow, int b_kslow, int M, int N, int K, float alpha,
                      const void* A, long lda_bytes, const float* a_amax, const void* B, long ldb_bytes, const float* b_amax,
                      float* C, long ldc, const float* bias, const float* add_src, long ld_add, int relu,
                      const float* mask_src, long ld_mask, float mask_scale,
                      float drop_p, unsigned long long drop_seed, int accumulate, int split_k, float* c_amax,
                      float* ws = nullptr, long ws_floats = 0, void* c_img = nullptr, long ld_img = 0,
                      int batch = 1, long a_bs = 0, long b_bs = 0, long c_bs = 0) {
    const int KS = fmt == 1 ? 128 : PBK;
    QT_CHECK_ARG(M > 0 && N > 0 && K > 0 && A && B && C, "%s: bad problem", who);
    QT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "%s: drop_p=%f", who, drop_p);
    QT_CHECK_ARG(ldc >= N && lda_bytes % 128 == 0 && ldb_bytes % 128 == 0 && qt_aligned16(A) && qt_aligned16(B), "%s: image pitch / alignment", who);
    QT_CHECK_ARG(fmt == 0 || (!a_kslow && !b_kslow), "%s: fp8 lines are K-contiguous only", who);
    QT_CHECK_ARG(batch >= 1 && (batch == 1 || (!bias && !add_src && !mask_src && drop_p == 0.f && !accumulate && !c_img && !c_amax &&
                                               a_bs % 128 == 0 && b_bs % 128 == 0 && c_bs % 4 == 0)),
                 "%s: a batched launch takes no epilogue operands and 128-byte aligned operand strides", who);
    if (batch > 1) split_k = 1;
    if (K % KS != 0) QT_FAIL(QT_ERR_UNSUPPORTED, "%s: K=%d is not a multiple of %d", who, K, KS);
    if ((a_kslow && M % 32) || (b_kslow && N % 32)) QT_FAIL(QT_ERR_UNSUPPORTED, "%s: a K-slow operand needs its width to be a multiple of 32", who);
    if (M < 256 || N < 128) QT_FAIL(QT_ERR_UNSUPPORTED, "%s: problem too small for the 256x256 tile", who);
    {   // per-lane source offsets are 32-bit byte offsets from a wave-uniform base
        const long spanA = a_kslow ? 32 * lda_bytes + 4L * M : 256L * lda_bytes;      // row-form A is addressed relative to its tile
        const long spanB = b_kslow ? 32 * ldb_bytes + 4L * N : (long)N * ldb_bytes;
        if (spanA >= (1L << 32) || spanB >= (1L << 32)) QT_FAIL(QT_ERR_UNSUPPORTED, "%s: operand spans more than 4 GiB", who);
    }
    hipStream_t st = (hipStream_t)stream;
    PlaneArgs a;
    a.M = M; a.N = N; a.K = K; a.A = (const char*)A; a.lda = lda_bytes; a.B = (const char*)B; a.ldb = ldb_bytes; a.C = C; a.ldc = ldc;
    a.alpha = alpha; a.a_amax = a_amax; a.b_amax = b_amax; a.bias = bias; a.add_src = add_src; a.ld_add = ld_add; a.relu = relu;
    a.mask_src = mask_src; a.ld_mask = ld_mask; a.mask_scale = mask_scale;
    a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.drop_seed = drop_seed; a.salt = qt_salt_ptr; a.accumulate = accumulate;
    a.batch = batch; a.a_bs = a_bs; a.b_bs = b_bs; a.c_bs = c_bs;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("QT_PS_DEBUG"); dbg = e ? atoi(e) : 0; } a.debug = dbg; }
    const bool plain_epilogue = !relu && !mask_src && drop_p == 0.f;
    QT_CHECK_ARG(split_k <= 1 || plain_epilogue, "%s: split_k>1 needs a linear epilogue", who);
    // Tile height (256 or 192 rows) and split-K by one cost model: rounds of the persistent walk x (K steps + fixed per-tile cost),
    // in units of one K step of a 256-row tile.  The 192-row tile does 3/4 of the work per step at slightly worse operand reuse (+5 %).
    const int tiles_n = (N + 255) / 256;
    const long slots = 256;
    const long ovh = 16;
    int force_wmt = 0;                                   // QT_PS_WMT=3|4 pins the tile height (tests, A/B runs); read per call
    { const char* e = getenv("QT_PS_WMT"); if (e) force_wmt = atoi(e); }
    const bool wmt3_ok = fmt == 0 && !a_kslow && force_wmt != 4;      // the 192- and 128-row tiles: row-form fp16 A operands
    int wmt = 4;
    {
        const bool fixed_split = split_k > 0;
        long best = -1;
        int best_s = fixed_split ? split_k : 1;
        for (int cand = 4; cand >= 2; --cand) {
            if (cand < 4 && !wmt3_ok) break;
            if (force_wmt >= 2 && force_wmt <= 4 && wmt3_ok && cand != force_wmt) continue;
            const long tl = (long)((M + 64 * cand - 1) / (64 * cand)) * tiles_n * batch;
            long maxs = 1;
            if (!fixed_split && plain_epilogue && tl < 2 * slots && K >= 16 * KS) { maxs = K / (4 * KS); if (maxs > 64) maxs = 64; }
            for (long s = fixed_split ? split_k : 1; s <= (fixed_split ? split_k : maxs); ++s) {
                const long rounds = (tl * s + slots - 1) / slots;
                const long ksteps = ((K + s - 1) / s + KS - 1) / KS;
                long cost = rounds * (ksteps + ovh) * (cand == 4 ? 100 : (cand == 3 ? 79 : 58));      // 75 x 1.05, 50 x 1.16: shorter tiles reuse B less
                // split-K pays for its fp32 atomics at the L2's rate (measured: ~270 per ns for the whole chip = 513e3 per K-step time
                // of 1.9 us) plus the zeroing launch and the drain (~20 us): tools/sweep_splitk.py
                if (s > 1) cost += (long)(((double)s * M * N / 513e3 + 10.0) * 100.0);
                if (best < 0 || cost < best) { best = cost; best_s = (int)s; wmt = cand; }
            }
        }
        split_k = best_s;
    }
    const int tiles_m = (M + 64 * wmt - 1) / (64 * wmt);
    const long tiles = (long)tiles_m * tiles_n;
    a.tiles_m = tiles_m; a.tiles_n = tiles_n;
    int kps = (K + split_k - 1) / split_k;
    kps = ((kps + KS - 1) / KS) * KS;
    split_k = (K + kps - 1) / kps;
    a.split_k = split_k; a.k_per_split = kps;
    a.c_amax = split_k > 1 ? nullptr : c_amax;       // partial sums are not the values of C: the caller measures it (return 2)
    // the 16-byte epilogue needs 16-byte aligned rows of C and of every epilogue operand; anything else keeps the dword orientation
    const bool vec_ok = N % 4 == 0 && ldc % 4 == 0 && qt_aligned16(C) && (!bias || qt_aligned16(bias)) &&
                        (!add_src || (qt_aligned16(add_src) && ld_add % 4 == 0)) && (!mask_src || (qt_aligned16(mask_src) && ld_mask % 4 == 0));
    // Split-K through a workspace when the caller gave one that holds the slabs (dW GEMMs: small outputs over a long K): every K split
    // stores its partial tile with plain 16-byte stores and one pass sums the slabs into C — 768x768x30720 at split 28: 16.5 M fp32
    // atomics (61 us at the L2's ~270 per ns) become 66 MB of stores plus a 66 MB read, no zeroing launch, and the sum has a fixed order.
    const bool use_ws = split_k > 1 && vec_ok && fmt == 0 && plain_epilogue && !bias && !add_src && ws && qt_aligned16(ws) &&
                        (long)split_k * M * N <= ws_floats && !(getenv("QT_PS_ATOMIC"));
    a.ws_stride = 0;
    if (use_ws) { a.C = ws; a.ldc = N; a.ws_stride = (long)M * N; a.accumulate = 0; }
    else if (split_k > 1 && !accumulate) {
        hipError_t e = qt_zero_2d(C, ldc, M, N, st);
        if (e != hipSuccess) QT_FAIL(QT_ERR_RUNTIME, "%s: zeroing C: %s", who, hipGetErrorString(e));
    }
    const long total_work = tiles * split_k * batch;
    dim3 grid((unsigned)(total_work > slots ? slots : total_work));
    const bool swap = (split_k == 1 || use_ws) && vec_ok;
    // MFMA shape of the fp16 path: the 16x16x32 form by default (the chip holds a higher clock on it: 215040x2304x768 2.04 -> 1.88 ms on
    // the same images, tools/ab_mfma_shape.py); QT_PS_MF=0 selects the 32x32x16 form (A/B runs; read per call)
    int mf = 1;
#ifdef QT_PS_WITH_MF0                                   // (the 32x32x16 instances, parts 1 and 2, are built only with QT_PS_WITH_MF0 in the environment)
    { const char* e = getenv("QT_PS_MF"); if (e) mf = atoi(e); }
#endif
    // wave layout of the 256-row tile: QT_PS_W4=1 selects four waves of 128 x 128 — only in a library built with -DQT_PS_WITH_W4
    // (measured and not shipped: kernel comment "NWN")
    int w4 = 0;
#ifdef QT_PS_WITH_W4
    { const char* e = getenv("QT_PS_W4"); if (e) w4 = atoi(e); }
#endif
    // the late-barrier pipeline (kernel comment "PIPE") by default: +1...+6 % on every shape of the path, bit-identical results
    // (tools/ab_ps_variants.py, profiles/r03_pipe_ab.txt).  Its two-segment tail (PIPE 3) where B is K-slow (dX, dW: the transposing
    // fragment reads are twice as many instructions, spreading them pays: +2...+4.5 % over PIPE 1), the one-segment tail (PIPE 1) where B
    // is in row form (forward: PIPE 3 measured 1-4 % slower there).  QT_PS_PIPE=0 selects the barrier-first loop, 1 / 3 force a form
    // (A/B runs; read per call)
    int pipe = b_kslow ? 3 : 1;
    { const char* e = getenv("QT_PS_PIPE"); if (e) pipe = atoi(e); }
    // the image of C rides on the 16x16x32 C^T epilogue of an unsplit launch; otherwise it is not written (return 3: the caller splits C)
    const bool img_ok = c_img && fmt == 0 && mf && swap && split_k == 1 && N % 32 == 0 && ld_img % 128 == 0 && qt_aligned16(c_img);
    a.c_img = img_ok ? (char*)c_img : nullptr; a.ld_img = ld_img;
    if (fmt == 1) qt_ps_launch_fp8(a, 0, 0, swap, mf ? 1 : 0, grid, st);
    else if (wmt != 4 && mf && pipe == 3) qt_ps_launch_mf1_t32_pipe3(a, a_kslow, b_kslow, swap, wmt, grid, st);
    else if (wmt != 4 && mf && pipe) qt_ps_launch_mf1_t32_pipe(a, a_kslow, b_kslow, swap, wmt, grid, st);
#ifdef QT_PS_WITH_MF0
    else if (wmt != 4 && !mf) qt_ps_launch_mf0_t32(a, a_kslow, b_kslow, swap, wmt, grid, st);
    else if (!mf) qt_ps_launch_mf0_t4(a, a_kslow, b_kslow, swap, wmt, grid, st);
#endif
    else if (wmt != 4) qt_ps_launch_mf1_t32(a, a_kslow, b_kslow, swap, wmt, grid, st);
#ifdef QT_PS_WITH_W4
    else if (mf && w4) { if (pipe) qt_ps_launch_mf1_w4_pipe(a, a_kslow, b_kslow, swap, wmt, grid, st); else qt_ps_launch_mf1_w4(a, a_kslow, b_kslow, swap, wmt, grid, st); }
#endif
    else if (mf && pipe == 3) qt_ps_launch_mf1_t4_pipe3(a, a_kslow, b_kslow, swap, wmt, grid, st);
    else if (mf && pipe) qt_ps_launch_mf1_t4_pipe(a, a_kslow, b_kslow, swap, wmt, grid, st);
    else qt_ps_launch_mf1_t4(a, a_kslow, b_kslow, swap, wmt, grid, st);
    QT_CHECK_LAUNCH(who);
    if (use_ws) { const int rc = qt_splitk_reduce(stream, ws, split_k, M, N, C, ldc, accumulate, c_amax); if (rc) return rc; }
    if (c_img && !img_ok) return 3;
    return (c_amax && split_k > 1 && !use_ws) ? 2 : QT_OK;          // (the summing pass of a workspace split measures max|C| while it writes C)
}

// C[M,N] = epilogue(alpha/(s_a s_b) * op(A) op(B)) on split images.
//   a_kslow = 0: A image is [M rows][K]  (K contiguous)      a_kslow = 1: A image is [K rows][M]
//   b_kslow = 0: B image is [N rows][K]  (Linear weight)     b_kslow = 1: B image is [K rows][N]
// Returns QT_ERR_UNSUPPORTED (without touching C) when the shape does not fit (callers then use qt_gemm_x3).
extern "C" int qt_gemm_ps(void* stream, int a_kslow, int b_kslow, int M, int N, int K, float alpha,
                          const void* A, long lda_bytes, const float* a_amax, const void* B, long ldb_bytes, const float* b_amax,
                          float* C, long ldc, const float* bias, const float* add_src, long ld_add, int relu,
                          const float* mask_src, long ld_mask, float mask_scale,
                          float drop_p, unsigned long long drop_seed, int accumulate, int split_k, float* c_amax,
                          float* ws, long ws_floats) {
    return plane_gemm("qt_gemm_ps", 0, stream, a_kslow, b_kslow, M, N, K, alpha, A, lda_bytes, a_amax, B, ldb_bytes, b_amax, C, ldc,
                      bias, add_src, ld_add, relu, mask_src, ld_mask, mask_scale, drop_p, drop_seed, accumulate, split_k, c_amax, ws, ws_floats);
}

// `batch` independent products of ONE shape in one launch: problem z multiplies the images at A + z a_bs_bytes and B + z b_bs_bytes and
// writes C + z c_bs_floats (plain epilogue: alpha only; no K split).  The H head slabs of a fold GEMM (u_h = q_h W_k,h: M = rows, N = D,
// K = d_h = three K steps) are bound by writing their slabs: as one launch their 1-2 rounds of tiles each become one walk of H times as
// many tiles, without H - 1 launch ramps and half-empty last rounds.
extern "C" int qt_gemm_ps_batched(void* stream, int a_kslow, int b_kslow, int M, int N, int K, float alpha,
                                  const void* A, long lda_bytes, const float* a_amax, const void* B, long ldb_bytes, const float* b_amax,
                                  float* C, long ldc, int batch, long a_bs_bytes, long b_bs_bytes, long c_bs_floats) {
    return plane_gemm("qt_gemm_ps_batched", 0, stream, a_kslow, b_kslow, M, N, K, alpha, A, lda_bytes, a_amax, B, ldb_bytes, b_amax, C, ldc,
                      nullptr, nullptr, 0, 0, nullptr, 0, 1.0f, 0.f, 0ull, 0, 1, nullptr, nullptr, 0, nullptr, 0,
                      batch, a_bs_bytes, b_bs_bytes, c_bs_floats);
}

// qt_gemm_ps that ALSO writes C as an unscaled split image (c_img, row pitch ld_img_bytes: the operand of the next Linear layer, made by
// the epilogue that has the values in registers instead of by a split pass over C).  Returns 3 (C written, image NOT written) when the
// launch cannot do it (K split, unaligned rows, the 32x32x16 form): the caller then splits C as usual.
extern "C" int qt_gemm_ps_img(void* stream, int a_kslow, int b_kslow, int M, int N, int K, float alpha,
                              const void* A, long lda_bytes, const float* a_amax, const void* B, long ldb_bytes, const float* b_amax,
                              float* C, long ldc, const float* bias, const float* add_src, long ld_add, int relu,
                              const float* mask_src, long ld_mask, float mask_scale,
                              float drop_p, unsigned long long drop_seed, int accumulate, int split_k, float* c_amax,
                              void* c_img, long ld_img_bytes) {
    return plane_gemm("qt_gemm_ps_img", 0, stream, a_kslow, b_kslow, M, N, K, alpha, A, lda_bytes, a_amax, B, ldb_bytes, b_amax, C, ldc,
                      bias, add_src, ld_add, relu, mask_src, ld_mask, mask_scale, drop_p, drop_seed, accumulate, split_k, c_amax, nullptr, 0,
                      c_img, ld_img_bytes);
}

// BASELINE configs[4]: C[M,N] = epilogue(alpha/(s_a s_b) * A B^T) with A [M rows][K] and B [N rows][K] (nn.Linear weight) given as
// e4m3 images (qt_quant_fp8: one byte per element, row pitch in bytes a multiple of 128), on v_mfma_scale_f32_32x32x64_f8f6f4
// with fp32 accumulation and the fp32 epilogue of qt_gemm_f32.  K must be a multiple of 128.
extern "C" int qt_gemm_f8(void* stream, int M, int N, int K, float alpha,
                          const void* A, long lda_bytes, const float* a_amax, const void* B, long ldb_bytes, const float* b_amax,
                          float* C, long ldc, const float* bias, const float* add_src, long ld_add, int relu,
                          const float* mask_src, long ld_mask, float mask_scale,
                          float drop_p, unsigned long long drop_seed, int accumulate, int split_k) {
    return plane_gemm("qt_gemm_f8", 1, stream, 0, 0, M, N, K, alpha, A, lda_bytes, a_amax, B, ldb_bytes, b_amax, C, ldc,
                      bias, add_src, ld_add, relu, mask_src, ld_mask, mask_scale, drop_p, drop_seed, accumulate, split_k, nullptr);
}

// fp32 [R, C] -> e4m3 image [R][C bytes] (row pitch ld_img bytes), x * 2^e with 2^e * amax in [128, 256); C a multiple of 16.
__global__ __launch_bounds__(256) void qt_quant_fp8_kernel(const float* __restrict__ x, long ldx, long R, int C, const float* __restrict__ amax,
                                                           char* __restrict__ img, long ld_img) {
    const float s = amax ? qt_fp8_scale(qt_amax_read(amax)) : 1.0f;
    const int g16 = C >> 4;
    const long total = R * g16;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long row = idx / g16; const int c16 = (int)(idx - row * g16);
        const float* src = x + row * ldx + c16 * 16;
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src + 4 * u);
        uint4 o;
        unsigned* ow = reinterpret_cast<unsigned*>(&o);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[u].x * s, v[u].y * s, 0, false);
            pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[u].z * s, v[u].w * s, pk, true);
            ow[u] = (unsigned)pk;
        }
        *reinterpret_cast<uint4*>(img + row * ld_img + c16 * 16) = o;
    }
}

extern "C" int qt_quant_fp8(void* stream, const float* x, long ldx, long R, int C, const float* amax, void* img, long ld_img_bytes) {
    QT_CHECK_ARG(x && img && R > 0 && C > 0 && ldx >= C, "qt_quant_fp8: bad arguments");
    QT_CHECK_ARG(C % 128 == 0, "qt_quant_fp8: the image is made of 128-byte lines (C=%d)", C);
    QT_CHECK_ARG(ldx % 4 == 0 && qt_aligned16(x) && qt_aligned16(img) && ld_img_bytes % 128 == 0 && ld_img_bytes >= C,
                 "qt_quant_fp8: alignment (ldx=%ld, ld_img=%ld)", ldx, ld_img_bytes);
    long blocks = (R * (C >> 4) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(qt_quant_fp8_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, R, C, amax, (char*)img, ld_img_bytes);
    QT_CHECK_LAUNCH("qt_quant_fp8");
    return QT_OK;
}
#endif  // QT_PS_PART == 0
