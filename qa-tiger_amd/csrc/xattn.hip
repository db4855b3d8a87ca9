// Few-query attention with the K/V projections folded away (fp32), forward and backward.
//
// For an attention whose queries are few (Lq = 1: TempMoE.qst_attn, reference
// src/models/modules.py:562; Lq = 2: PatchSelecter.crs_attn, modules.py:692-698) the
// packed K/V in-projection of every key row is wasted work.  With x_p the raw key rows
// (dimension D) and q'_h the projected query of head h:
//     score_h[p] = (W_k,h^T q'_h) · x_p + q'_h·b_k,h        (the bias term is constant in p: softmax drops it)
//     out_h      = W_v,h ( Σ_p a_h[p] x_p ) + b_v,h Σ_p a_h[p]
// (SURVEY.md §7 "Legitimate algebraic savings", Appendix A step 1).  The host computes
// u_h = W_k,h^T q'_h with a head-batched GEMM; THIS kernel does, per item and per (query, head):
//     s_p = scale · u · x_p,   a = softmax_p(s) (+ dropout),   ctx = Σ_p a_p x_p,   asum = Σ_p a_p
// and the host applies W_v,h with a second head-batched GEMM.  At the primary config this
// removes the 215040 x 1536 x 768 GEMM (17 % of all GEMM FLOPs) and its two backward GEMMs.
//
// Layout: U and CTX are [rows, NQH·? ] = element (item n, query i, head h, col c) at
//   ((n*u_is + i*u_rs) * H + h) * D + c;  X rows of item n are rows n*Lk .. n*Lk+Lk-1 of [*, D].
// One workgroup per item.  A wave owns QB (query, head) vectors at a time, each lane D/64 columns;
// rows of X are staged chunk-wise in LDS (coalesced 16-byte loads), scores use an online softmax,
// dot products reduce with wavefront shuffles.  HBM/L2-bound.
#include "qt_common.h"

#define XA_MAXV 4      // float4 per lane: D <= 1024

struct XAttnArgs {
    const float* u; const float* x; float* ctx; float* lse; float* asum;
    long u_is, u_rs; int n_items, H, Lq, Lk, D;
    int bch, gq;             // backward: rows per staged chunk, (query, head) slots held in LDS
    int swap_ug;             // backward: u and dctx rows share ONE LDS buffer (dctx first, then u) so that two workgroups fit a CU
    float scale, drop_p, inv_keep; unsigned long long seed; const unsigned long long* salt;
    // backward
    const float* dctx; const float* dasum; const float* ctx_in;
    float* du; float* dx;
    float* dx_amax;          // nullable: max|dx| record (64 floats), measured while dx is written (the out-projection's dY scale)
    float* du_amax;          // nullable: max|du| record (the head-batched W_k GEMMs' dY scale)
};

__device__ inline float dot4(const float4& a, const float4& b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

// Rows of X are staged into LDS in chunks by the whole workgroup (all loads of a chunk in flight
// at once: a first version that streamed one row at a time per wave was bound by 14 serialised
// load latencies per item), then every wave reads the chunk for its own (query, head) vectors.
// Workgroup = 8 waves x 2 vectors (16 (query, head) vectors per pass = PatchSelecter's 2 x 8).
#define XA_FCH 16      // rows per chunk, forward
#define XA_BCH 8       // rows per chunk, backward (a second LDS tile of the same size accumulates dX)
#define XA_NT 512

// Sum 4 per-lane partials over the wavefront with 7 cross-lane exchanges instead of 24: after the
// xor-32 and xor-16 steps each 16-lane group owns ONE of the four sums (reduce-scatter), four more
// butterfly steps finish it, and the totals are read back from lanes 0/16/32/48.
__device__ inline void xa_reduce4(float (&v)[4], int lane) {
    const bool hi32 = lane & 32, hi16 = lane & 16;
    float k0 = hi32 ? v[2] : v[0], k1 = hi32 ? v[3] : v[1];
    const float s0 = hi32 ? v[0] : v[2], s1 = hi32 ? v[1] : v[3];
    k0 += __shfl_xor(s0, 32, 64); k1 += __shfl_xor(s1, 32, 64);
    float b = hi16 ? k1 : k0;
    const float t = hi16 ? k0 : k1;
    b += __shfl_xor(t, 16, 64);
    b += __shfl_xor(b, 8, 64); b += __shfl_xor(b, 4, 64); b += __shfl_xor(b, 2, 64); b += __shfl_xor(b, 1, 64);
    v[0] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 0));
    v[1] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 16));
    v[2] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 32));
    v[3] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, b), 48));
}

__device__ inline void xa_stage_rows(const float* __restrict__ xbase, float* __restrict__ Xs, int r0, int nrows, int Lk, int D, int nv) {
    constexpr int U = 4;                                           // trips of loads in flight before the first LDS store
    for (int base = threadIdx.x; base < nrows * nv; base += U * XA_NT) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * XA_NT;
            const int r = idx / nv, c4 = idx - r * nv;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < nrows * nv && r0 + r < Lk) v[u] = *reinterpret_cast<const float4*>(xbase + (long)(r0 + r) * D + c4 * 4);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + u * XA_NT;
            const int r = idx / nv, c4 = idx - r * nv;
            if (idx < nrows * nv) *reinterpret_cast<float4*>(Xs + r * D + c4 * 4) = v[u];
        }
    }
}

template <int NV>
__global__ __launch_bounds__(XA_NT) void qt_xattn_fwd_kernel(XAttnArgs p) {
    if (p.salt && p.drop_p > 0.f) p.seed += *p.salt;
    constexpr int QB = 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x;
    const int nv = p.D >> 2, NQ = p.Lq * p.H, D = p.D, Lk = p.Lk;
    const float* xbase = p.x + (long)n * Lk * D;
    float* Xs = lds;                                                // [XA_FCH][D]
    const int ngroups = (NQ + 8 * QB - 1) / (8 * QB);               // same trip count for every wave (barriers inside)
    for (int grp = 0; grp < ngroups; ++grp) {
        const int q0 = (grp * 8 + wave) * QB;
        float4 u[QB][NV], acc[QB][NV];
        float m[QB], l[QB], ls[QB];
        long urow[QB];
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            const int qq = min(q0 + b, NQ - 1);
            const int i = qq / p.H, h = qq - i * p.H;
            urow[b] = (((long)n * p.u_is + (long)i * p.u_rs) * p.H + h);
            m[b] = -INFINITY; l[b] = 0.f; ls[b] = 0.f;
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int c4 = lane + t * 64;
                u[b][t] = c4 < nv ? *reinterpret_cast<const float4*>(p.u + urow[b] * D + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc[b][t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        for (int r0 = 0; r0 < Lk; r0 += XA_FCH) {
            __syncthreads();
            xa_stage_rows(xbase, Xs, r0, XA_FCH, Lk, D, nv);
            __syncthreads();
            const int rn = min(XA_FCH, Lk - r0);
            for (int r = 0; r < rn; r += 2) {                       // two rows per step: 4 partials per reduction
                float4 xv[2][NV];
                float part[4];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int t = 0; t < NV; ++t) {
                        const int c4 = lane + t * 64;
                        xv[rr][t] = (c4 < nv && r + rr < rn) ? *reinterpret_cast<const float4*>(Xs + (r + rr) * D + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int b = 0; b < QB; ++b) {
                        float a = 0.f;
#pragma unroll
                        for (int t = 0; t < NV; ++t) a += dot4(u[b][t], xv[rr][t]);
                        part[rr * 2 + b] = a;
                    }
                xa_reduce4(part, lane);
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    if (r + rr >= rn) continue;
#pragma unroll
                    for (int b = 0; b < QB; ++b) {
                        const float s = part[rr * 2 + b] * p.scale;
                        const float mn = fmaxf(m[b], s);
                        const float alpha = __expf(m[b] - mn);
                        float e = __expf(s - mn);
                        m[b] = mn;
                        l[b] = l[b] * alpha + e;
                        if (p.drop_p > 0.f) e *= qt_drop_mult(p.seed, (unsigned long long)urow[b] * Lk + (r0 + r + rr), p.drop_p, p.inv_keep);
                        ls[b] = ls[b] * alpha + e;
#pragma unroll
                        for (int t = 0; t < NV; ++t) {
                            acc[b][t].x = fmaf(e, xv[rr][t].x, acc[b][t].x * alpha); acc[b][t].y = fmaf(e, xv[rr][t].y, acc[b][t].y * alpha);
                            acc[b][t].z = fmaf(e, xv[rr][t].z, acc[b][t].z * alpha); acc[b][t].w = fmaf(e, xv[rr][t].w, acc[b][t].w * alpha);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            if (q0 + b >= NQ) continue;
            const float inv = 1.0f / l[b];
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int c4 = lane + t * 64;
                if (c4 < nv)
                    *reinterpret_cast<float4*>(p.ctx + urow[b] * D + c4 * 4) =
                        make_float4(acc[b][t].x * inv, acc[b][t].y * inv, acc[b][t].z * inv, acc[b][t].w * inv);
            }
            if (lane == 0) { p.lse[urow[b]] = m[b] + __logf(l[b]); p.asum[urow[b]] = ls[b] * inv; }
        }
    }
}

// MFMA forward for the PatchSelecter shape class (at most 16 (query, head) vectors, at most 16 key rows, D a multiple of 128):
// the two products of an item are small GEMMs — scores S = U X^T (16 x 16 x D) and context C = P X (16 x D x 16) — and the VALU
// kernel above spends ~23 k wave-instructions per item on them (issue-bound at 44 % of the HBM rate).  Here they run on
// v_mfma_f32_16x16x4_f32 (exact fp32 products, same arithmetic class as the VALU dot products):
//   * the D columns are split over the 8 waves in blocks of 16: a wave forms a PARTIAL 16 x 16 score tile over its D/8 columns
//     straight from global memory (16-byte loads: lane (row, kq) takes columns 4 kq .. 4 kq + 3 of a block, used as the k index
//     of four MFMAs — the same assignment on both operands, so the products meet), the 8 partial tiles are summed through LDS;
//   * every wave then holds the full tile in the MFMA result layout (key on the lane, 4 (query, head) rows in registers): softmax
//     over the keys is four 16-lane shuffle reductions per row, dropout uses the same (row, key) index as the VALU kernel
//     (the backward regenerates the mask from it);
//   * the probabilities go through a 1 KiB per-wave LDS tile into operand layout, and the context block of the wave's columns is
//     computed TRANSPOSED (A = X^T from the LDS-staged rows, B = P^T): a lane ends up with four consecutive columns of one
//     (query, head) row -> 16-byte stores.
// No row is staged in LDS (the wave's column blocks of the key rows stay in registers): 25 KiB per workgroup.
typedef float xa_f4 __attribute__((ext_vector_type(4)));
template <int DB>      // 16-column blocks per wave: D = 128 * DB
__global__ __launch_bounds__(XA_NT, 3) void qt_xattn_fwd_mfma_kernel(XAttnArgs p) {
    if (p.salt && p.drop_p > 0.f) p.seed += *p.salt;
    constexpr int D = 128 * DB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Sp = lds;                                   // [8][16][16] partial score tiles
    float* Pw = Sp + 8 * 256;                          // per wave: P [16][17], T [16][17] (transpose scratch)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int n = blockIdx.x, NQ = p.Lq * p.H, Lk = p.Lk;
    const float* xbase = p.x + (long)n * Lk * D;
    auto urow_of = [&](int qh) { const int qq = min(qh, NQ - 1); const int i = qq / p.H, h = qq - i * p.H;
                                 return ((long)n * p.u_is + (long)i * p.u_rs) * p.H + h; };
    // partial scores over this wave's columns; the key rows of those columns stay in registers for the context product
    // (no row is staged in LDS: 10 KiB per workgroup instead of 67, the key rows are read from memory exactly once per item)
    float4 xb[DB];
    xa_f4 sacc = {0.f, 0.f, 0.f, 0.f};
    {
        const float* up = p.u + urow_of(r16) * D + 4 * kq;
        const float* xp = xbase + (long)min(r16, Lk - 1) * D + 4 * kq;
        const bool uok = r16 < NQ, xok = r16 < Lk;
        float4 ua[DB];
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const int cb = (wave * DB + b) * 16;
            ua[b] = uok ? *reinterpret_cast<const float4*>(up + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
            xb[b] = xok ? *reinterpret_cast<const float4*>(xp + cb) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[b].x, xb[b].x, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[b].y, xb[b].y, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[b].z, xb[b].z, sacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[b].w, xb[b].w, sacc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Sp[wave * 256 + (4 * kq + reg) * 16 + r16] = sacc[reg];
    __syncthreads();
    // full tile in every wave: key = r16 on the lane, (query, head) rows 4 kq + reg in registers
    float* Pm = Pw + wave * (2 * 16 * 17);
    float* Tm = Pm + 16 * 17;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int qh = 4 * kq + reg;
        float sv = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) sv += Sp[w * 256 + qh * 16 + r16];
        sv = r16 < Lk ? sv * p.scale : -INFINITY;
        float m = sv;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        float e = r16 < Lk ? __expf(sv - m) : 0.f;
        float l = e;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
        const long ur = urow_of(qh);
        if (p.drop_p > 0.f) e *= qt_drop_mult(p.seed, (unsigned long long)ur * Lk + r16, p.drop_p, p.inv_keep);
        float ls = e;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) ls += __shfl_xor(ls, o, 64);
        const float inv = 1.0f / l;
        Pm[qh * 17 + r16] = e * inv;
        if (wave == 0 && r16 == 0 && qh < NQ) { p.lse[ur] = m + __logf(l); p.asum[ur] = ls * inv; }
    }
    // probabilities in operand layout: B[k = key][col = (query, head)] from lane (qh = r16, key = 4 j + kq)
    float pb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pb[j] = Pm[r16 * 17 + 4 * j + kq];      // (own wave's tile: the LDS writes above are in program order)
    const long ur16 = urow_of(r16);
#pragma unroll
    for (int b = 0; b < DB; ++b) {
        const int cb = (wave * DB + b) * 16;
        // A = X^T of the block: a 16 x 16 register transpose through the wave's LDS tile (lane (key r16, kq) holds x[key][cb + 4 kq ..])
        Tm[r16 * 17 + 4 * kq + 0] = xb[b].x; Tm[r16 * 17 + 4 * kq + 1] = xb[b].y; Tm[r16 * 17 + 4 * kq + 2] = xb[b].z; Tm[r16 * 17 + 4 * kq + 3] = xb[b].w;
        xa_f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(Tm[(4 * j + kq) * 17 + r16], pb[j], acc, 0, 0, 0);
        if (r16 < NQ) *reinterpret_cast<float4*>(p.ctx + ur16 * D + cb + 4 * kq) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }
}

// MFMA backward for the same shape class (at most 16 (query, head) vectors r, at most 16 key rows p, D = 128 DB).  Per item the
// backward is five small products with D as the long dimension:
//     S = U X^T and dA = G X^T (16 x 16 x D, G = dctx),  dU = scale dS X (16 x D x 16),  dX = A_d^T G + scale dS^T U (16 x D x 16, twice)
// with a = exp(scale S - lse), A_d = a mult (dropout), dS = a (mult (dA + dasum) - delta), delta_r = g_r.ctx_r + dasum_r asum_r.
// The VALU kernel below spends its time issuing dot products (33 % of the HBM rate).  Here, as in the forward:
//   * the D columns are split over the 8 waves in 16-column blocks; a lane (r16, kq) loads columns 4 kq .. 4 kq + 3 of row r16 of
//     U, G, ctx and X for each of its blocks ONCE (16-byte loads) and keeps them in registers for the whole item;
//   * partial S and dA tiles (and the partial row dots of delta) over the wave's columns meet in LDS (one __syncthreads per item);
//     every wave then holds a, A_d and scale dS in the MFMA result layout (key on the lane, 4 rows in registers);
//   * dU and dX blocks are computed TRANSPOSED (the A operand is X^T / G^T / U^T of the block: a 16 x 16 register transpose
//     through a 1 KiB per-wave LDS tile; the B operand is the small matrix, read from the wave's LDS copy in operand layout), so a
//     lane ends up with four consecutive columns of one output row -> 16-byte stores.
// No row is staged in LDS: 25 KiB per workgroup.
template <int DB>
__global__ __launch_bounds__(XA_NT, 2) void qt_xattn_bwd_mfma_kernel(XAttnArgs p) {
    if (p.salt && p.drop_p > 0.f) p.seed += *p.salt;
    constexpr int D = 128 * DB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Sp = lds;                                   // [8][16][16] partial S tiles
    float* Dp = Sp + 8 * 256;                          // [8][16][16] partial dA tiles
    float* Dl = Dp + 8 * 256;                          // [8][16] partial g.ctx row dots
    float* Wv = Dl + 8 * 16;                           // per wave: Ad [16][17], Ds [16][17], T [16][17]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kq = lane >> 4;
    const int n = blockIdx.x, NQ = p.Lq * p.H, Lk = p.Lk;
    const float* xbase = p.x + (long)n * Lk * D;
    auto urow_of = [&](int qh) { const int qq = min(qh, NQ - 1); const int i = qq / p.H, h = qq - i * p.H;
                                 return ((long)n * p.u_is + (long)i * p.u_rs) * p.H + h; };
    float* Am = Wv + wave * (3 * 16 * 17);
    float* Dm = Am + 16 * 17;
    float* Tm = Dm + 16 * 17;
    const long ur16 = urow_of(r16);
    const bool uok = r16 < NQ, xok = r16 < Lk;
    float4 ua[DB], ga[DB], xb[DB];
    xa_f4 sacc = {0.f, 0.f, 0.f, 0.f}, dacc = {0.f, 0.f, 0.f, 0.f};
    float dlt = 0.f;
    {
        const float* up = p.u + ur16 * D + 4 * kq;
        const float* gp = p.dctx + ur16 * D + 4 * kq;
        const float* cp = p.ctx_in + ur16 * D + 4 * kq;
        const float* xp = xbase + (long)min(r16, Lk - 1) * D + 4 * kq;
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            const int cb = (wave * DB + b) * 16;
            ua[b] = uok ? *reinterpret_cast<const float4*>(up + cb) : z;
            ga[b] = uok ? *reinterpret_cast<const float4*>(gp + cb) : z;
            xb[b] = xok ? *reinterpret_cast<const float4*>(xp + cb) : z;
            const float4 ca = uok ? *reinterpret_cast<const float4*>(cp + cb) : z;
            dlt += dot4(ga[b], ca);
        }
#pragma unroll
        for (int b = 0; b < DB; ++b) {
            sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[b].x, xb[b].x, sacc, 0, 0, 0);
            dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[b].x, xb[b].x, dacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[b].y, xb[b].y, sacc, 0, 0, 0);
            dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[b].y, xb[b].y, dacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[b].z, xb[b].z, sacc, 0, 0, 0);
            dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[b].z, xb[b].z, dacc, 0, 0, 0);
            sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[b].w, xb[b].w, sacc, 0, 0, 0);
            dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[b].w, xb[b].w, dacc, 0, 0, 0);
        }
    }
    dlt += __shfl_xor(dlt, 16, 64);
    dlt += __shfl_xor(dlt, 32, 64);                    // row r16's dot over this wave's columns, on all four kq lanes
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        Sp[wave * 256 + (4 * kq + reg) * 16 + r16] = sacc[reg];
        Dp[wave * 256 + (4 * kq + reg) * 16 + r16] = dacc[reg];
    }
    if (kq == 0) Dl[wave * 16 + r16] = dlt;
    __syncthreads();
    // full tiles in every wave (key = r16 on the lane, rows 4 kq + reg in registers) -> a, A_d, scale dS into the wave's LDS copy
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int qh = 4 * kq + reg;
        float sv = 0.f, dav = 0.f, dl = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { sv += Sp[w * 256 + qh * 16 + r16]; dav += Dp[w * 256 + qh * 16 + r16]; dl += Dl[w * 16 + qh]; }
        const long ur = urow_of(qh);
        const float das = p.dasum ? p.dasum[ur] : 0.f;
        const float delta = dl + das * p.asum[ur];
        const bool ok = qh < NQ && r16 < Lk;
        const float a = ok ? __expf(sv * p.scale - p.lse[ur]) : 0.f;
        const float mult = (p.drop_p > 0.f && ok) ? qt_drop_mult(p.seed, (unsigned long long)ur * Lk + r16, p.drop_p, p.inv_keep) : 1.0f;
        const float ds = a * ((dav + das) * mult - delta);
        Am[qh * 17 + r16] = a * mult;
        Dm[qh * 17 + r16] = ds * p.scale;
    }
    // operand-layout copies: B[k = 4 j + kq][col = r16]
    float dsb[4], dsk[4], adk[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        dsb[j] = Dm[r16 * 17 + 4 * j + kq];            // dU: k = key 4 j + kq, col = row r16:   scale dS[r16][key]
        dsk[j] = Dm[(4 * j + kq) * 17 + r16];          // dX: k = row 4 j + kq, col = key r16:   scale dS[row][r16]
        adk[j] = Am[(4 * j + kq) * 17 + r16];          //                                         A_d[row][r16]
    }
    float xmax = 0.f, umax = 0.f;
    float* dxrow = p.dx + ((long)n * Lk + min(r16, Lk - 1)) * D + 4 * kq;
    float* durow = p.du + ur16 * D + 4 * kq;
    auto transpose = [&](const float4& v, float (&t)[4]) {          // t[j] = V[row 4 j + kq][column r16 of the block]
        Tm[r16 * 17 + 4 * kq + 0] = v.x; Tm[r16 * 17 + 4 * kq + 1] = v.y; Tm[r16 * 17 + 4 * kq + 2] = v.z; Tm[r16 * 17 + 4 * kq + 3] = v.w;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = Tm[(4 * j + kq) * 17 + r16];
    };
#pragma unroll
    for (int b = 0; b < DB; ++b) {
        const int cb = (wave * DB + b) * 16;
        float xt[4], gt[4], ut[4];
        transpose(xb[b], xt);
        transpose(ga[b], gt);
        transpose(ua[b], ut);
        xa_f4 au = {0.f, 0.f, 0.f, 0.f}, ax = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            au = __builtin_amdgcn_mfma_f32_16x16x4f32(xt[j], dsb[j], au, 0, 0, 0);     // dU[r16][cb + 4 kq + reg]
            ax = __builtin_amdgcn_mfma_f32_16x16x4f32(gt[j], adk[j], ax, 0, 0, 0);     // dX[key r16][cb + 4 kq + reg]
            ax = __builtin_amdgcn_mfma_f32_16x16x4f32(ut[j], dsk[j], ax, 0, 0, 0);
        }
        if (uok) {
            umax = fmaxf(fmaxf(umax, fmaxf(fabsf(au[0]), fabsf(au[1]))), fmaxf(fabsf(au[2]), fabsf(au[3])));
            *reinterpret_cast<float4*>(durow + cb) = make_float4(au[0], au[1], au[2], au[3]);
        }
        if (xok) {
            xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(ax[0]), fabsf(ax[1]))), fmaxf(fabsf(ax[2]), fabsf(ax[3])));
            *reinterpret_cast<float4*>(dxrow + cb) = make_float4(ax[0], ax[1], ax[2], ax[3]);
        }
    }
    if (p.dx_amax) {
        xmax = qt_wave_max(xmax);
        if (lane == 0 && (xmax > 0.f || xmax != xmax))
            atomicMax(reinterpret_cast<unsigned*>(p.dx_amax) + ((blockIdx.x + wave) & (QT_AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, xmax));
    }
    if (p.du_amax) {
        umax = qt_wave_max(umax);
        if (lane == 0 && (umax > 0.f || umax != umax))
            atomicMax(reinterpret_cast<unsigned*>(p.du_amax) + ((blockIdx.x + wave) & (QT_AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, umax));
    }
}

// Backward, any shape.  Phase 1 (wave per pair of (query, head) vectors, rows staged chunk-wise): recompute
// a_p = exp(s_p - lse), da_p = dctx·x_p + dasum, ds_p = a_p (mult_p da_p - delta) with
// delta = dctx·ctx + dasum·asum; dU += ds_p x_p in registers; the scalars a_d[q,p], scale·ds[q,p] and the
// u / dctx rows are parked in LDS.  Phase 2 (wave per key row): dX_p = Σ_q a_d[q,p] dctx_q + scale ds[q,p] u_q
// from LDS.  (LDS float atomics for dX were tried: 8 waves adding to the same 768 addresses ran 7.6 ms.)
// With 16 (query, head) vectors of D = 768 the parked u and dctx rows take 96 KiB: one workgroup per CU, every
// load latency of an item exposed (1.86 ms for the PatchSelecter shape).  When Lk <= 16 (each wave owns at most
// two key rows) phase 2 runs in two halves over ONE buffer — dctx rows first, then the u rows written from
// registers — with the dX rows kept in registers in between: 75 KiB, two workgroups per CU.
template <int NV>
__global__ __launch_bounds__(XA_NT, 4) void qt_xattn_bwd_kernel(XAttnArgs p) {
    if (p.salt && p.drop_p > 0.f) p.seed += *p.salt;
    float xmax = 0.f;                                               // max |dx| written by this thread (dx_amax)
    float umax = 0.f;                                               // max |du| written by this thread (du_amax)
    constexpr int QB = 2, GQ = 16;        // (query, head) vectors per pass
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x;
    const int nv = p.D >> 2, NQ = p.Lq * p.H, D = p.D, Lk = p.Lk;
    const int BCH = p.bch, SL = p.gq;     // host sizes them so that everything fits in 160 KiB
    float* Xs = lds;                      // [BCH][D]
    const bool swap = p.swap_ug != 0;
    float* Us = Xs + BCH * D;             // [SL][D]
    float* Gs = swap ? Us : Us + SL * D;  // [SL][D]   (swap: the same buffer, holding dctx until phase 2a is done)
    float* Ad = Gs + SL * D;              // [SL][Lk]  dropped probabilities
    float* Ds = Ad + SL * Lk;             // [SL][Lk]  scale * dS
    const float* xbase = p.x + (long)n * Lk * D;
    const int ngroups = (NQ + GQ - 1) / GQ;
    for (int grp = 0; grp < ngroups; ++grp) {
        const int ql = wave * QB;                                   // slot inside the pass
        const int q0 = grp * GQ + ql;
        float4 u[QB][NV], g[QB][NV], acc[QB][NV];
        float lse[QB], delta[QB], das[QB];
        long urow[QB];
        bool live[QB];
        __syncthreads();                                            // previous pass finished reading Us/Gs/Ad/Ds
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            live[b] = q0 + b < NQ;
            const int qq = min(q0 + b, NQ - 1);
            const int i = qq / p.H, h = qq - i * p.H;
            urow[b] = (((long)n * p.u_is + (long)i * p.u_rs) * p.H + h);
            float part = 0.f;
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int c4 = lane + t * 64;
                float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
                u[b][t] = cv; g[b][t] = cv;
                if (c4 < nv && live[b]) {
                    u[b][t] = *reinterpret_cast<const float4*>(p.u + urow[b] * D + c4 * 4);
                    g[b][t] = *reinterpret_cast<const float4*>(p.dctx + urow[b] * D + c4 * 4);
                    cv = *reinterpret_cast<const float4*>(p.ctx_in + urow[b] * D + c4 * 4);
                }
                if (c4 < nv && ql + b < SL) {                       // idle waves own no LDS slot
                    if (!swap) *reinterpret_cast<float4*>(Us + (ql + b) * D + c4 * 4) = u[b][t];
                    *reinterpret_cast<float4*>(Gs + (ql + b) * D + c4 * 4) = g[b][t];
                }
                part += dot4(g[b][t], cv);
                acc[b][t] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            das[b] = (p.dasum && live[b]) ? p.dasum[urow[b]] : 0.f;
            delta[b] = qt_wave_sum(part) + das[b] * p.asum[urow[b]];
            lse[b] = p.lse[urow[b]];
        }
        for (int r0 = 0; r0 < Lk; r0 += BCH) {
            __syncthreads();
            xa_stage_rows(xbase, Xs, r0, BCH, Lk, D, nv);
            __syncthreads();
            const int rn = min(BCH, Lk - r0);
#pragma unroll 1
            for (int r = 0; r < rn; ++r) {
                float4 xv[NV];
#pragma unroll
                for (int t = 0; t < NV; ++t) {
                    const int c4 = lane + t * 64;
                    xv[t] = c4 < nv ? *reinterpret_cast<const float4*>(Xs + r * D + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                float part[4];                                      // {s_0, s_1, dctx·x_0, dctx·x_1}
#pragma unroll
                for (int b = 0; b < QB; ++b) {
                    float a = 0.f, c = 0.f;
#pragma unroll
                    for (int t = 0; t < NV; ++t) { a += dot4(u[b][t], xv[t]); c += dot4(g[b][t], xv[t]); }
                    part[b] = a; part[2 + b] = c;
                }
                xa_reduce4(part, lane);
#pragma unroll
                for (int b = 0; b < QB; ++b) {
                    const float a = live[b] ? __expf(part[b] * p.scale - lse[b]) : 0.f;
                    const float mult = p.drop_p > 0.f ? qt_drop_mult(p.seed, (unsigned long long)urow[b] * Lk + (r0 + r), p.drop_p, p.inv_keep) : 1.0f;
                    const float ds = a * ((part[2 + b] + das[b]) * mult - delta[b]);
                    if (lane == 0 && ql + b < SL) { Ad[(ql + b) * Lk + r0 + r] = a * mult; Ds[(ql + b) * Lk + r0 + r] = ds * p.scale; }
#pragma unroll
                    for (int t = 0; t < NV; ++t) {
                        acc[b][t].x = fmaf(ds, xv[t].x, acc[b][t].x); acc[b][t].y = fmaf(ds, xv[t].y, acc[b][t].y);
                        acc[b][t].z = fmaf(ds, xv[t].z, acc[b][t].z); acc[b][t].w = fmaf(ds, xv[t].w, acc[b][t].w);
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            if (!live[b]) continue;
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int c4 = lane + t * 64;
                if (c4 < nv) {
                    const float4 v = make_float4(acc[b][t].x * p.scale, acc[b][t].y * p.scale, acc[b][t].z * p.scale, acc[b][t].w * p.scale);
                    umax = fmaxf(fmaxf(umax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                    *reinterpret_cast<float4*>(p.du + urow[b] * D + c4 * 4) = v;
                }
            }
        }
        __syncthreads();                                            // Ad / Ds / Us / Gs of this pass complete
        const int nq = min(GQ, NQ - grp * GQ);
        auto store_row = [&](int r, const float4 (&dx)[NV]) {      // (the only place dx is written: max|dx| is taken here)
#pragma unroll
            for (int t = 0; t < NV; ++t) {
                const int c4 = lane + t * 64;
                if (c4 < nv) {
                    float4* dst = reinterpret_cast<float4*>(p.dx + ((long)n * Lk + r) * D + c4 * 4);
                    float4 v = dx[t];
                    if (grp > 0) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
                    xmax = fmaxf(fmaxf(xmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                    *dst = v;
                }
            }
        };
        if (!swap) {
            for (int r = wave; r < Lk; r += XA_NT / 64) {
                float4 dx[NV];
#pragma unroll
                for (int t = 0; t < NV; ++t) dx[t] = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int qq = 0; qq < nq; ++qq) {
                    const float a = Ad[qq * Lk + r], d = Ds[qq * Lk + r];
#pragma unroll
                    for (int t = 0; t < NV; ++t) {
                        const int c4 = lane + t * 64;
                        if (c4 < nv) {
                            const float4 gv = *reinterpret_cast<const float4*>(Gs + qq * D + c4 * 4);
                            const float4 uv = *reinterpret_cast<const float4*>(Us + qq * D + c4 * 4);
                            dx[t].x += a * gv.x + d * uv.x; dx[t].y += a * gv.y + d * uv.y;
                            dx[t].z += a * gv.z + d * uv.z; dx[t].w += a * gv.w + d * uv.w;
                        }
                    }
                }
                store_row(r, dx);
            }
        } else {
            // Lk <= 16: this wave owns rows `wave` and `wave + 8`.  Half a: dX += a_d[q,p] dctx_q; then the
            // buffer is refilled with the u rows (still in registers) and half b adds scale*ds[q,p] u_q.
            float4 dx[2][NV];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int t = 0; t < NV; ++t) dx[k][t] = make_float4(0.f, 0.f, 0.f, 0.f);
            auto half = [&](const float* coef) {
#pragma unroll 1
                for (int qq = 0; qq < nq; ++qq) {
                    float w[2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) w[k] = coef[qq * Lk + min(wave + 8 * k, Lk - 1)];
#pragma unroll
                    for (int t = 0; t < NV; ++t) {
                        const int c4 = lane + t * 64;
                        if (c4 < nv) {
                            const float4 v = *reinterpret_cast<const float4*>(Us + qq * D + c4 * 4);
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                dx[k][t].x = fmaf(w[k], v.x, dx[k][t].x); dx[k][t].y = fmaf(w[k], v.y, dx[k][t].y);
                                dx[k][t].z = fmaf(w[k], v.z, dx[k][t].z); dx[k][t].w = fmaf(w[k], v.w, dx[k][t].w);
                            }
                        }
                    }
                }
            };
            half(Ad);
            __syncthreads();                                        // every wave is done with the dctx rows
#pragma unroll
            for (int b = 0; b < QB; ++b)
#pragma unroll
                for (int t = 0; t < NV; ++t) {
                    const int c4 = lane + t * 64;
                    if (c4 < nv && ql + b < SL) *reinterpret_cast<float4*>(Us + (ql + b) * D + c4 * 4) = u[b][t];
                }
            __syncthreads();
            half(Ds);
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (wave + 8 * k < Lk) store_row(wave + 8 * k, dx[k]);
        }
    }
    if (p.dx_amax) {
        xmax = qt_wave_max(xmax);
        if ((threadIdx.x & 63) == 0 && (xmax > 0.f || xmax != xmax))
            atomicMax(reinterpret_cast<unsigned*>(p.dx_amax) + ((blockIdx.x + (threadIdx.x >> 6)) & (QT_AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, xmax));
    }
    if (p.du_amax) {
        umax = qt_wave_max(umax);
        if ((threadIdx.x & 63) == 0 && (umax > 0.f || umax != umax))
            atomicMax(reinterpret_cast<unsigned*>(p.du_amax) + ((blockIdx.x + (threadIdx.x >> 6)) & (QT_AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, umax));
    }
}

static int xattn_check(const char* who, int n_items, int H, int Lq, int Lk, int D) {
    QT_CHECK_ARG(n_items > 0 && H > 0 && Lq > 0 && Lk > 0 && D > 0, "%s: bad sizes", who);
    QT_CHECK_ARG(D % 4 == 0 && D <= 256 * XA_MAXV, "%s: D=%d must be a multiple of 4 and <= %d", who, D, 256 * XA_MAXV);
    return QT_OK;
}

extern "C" int qt_xattn_fwd(void* stream, const float* u, long u_is, long u_rs, const float* x,
                            int n_items, int H, int Lq, int Lk, int D, float scale, float drop_p, unsigned long long seed,
                            float* ctx, float* lse, float* asum) {
    if (int rc = xattn_check("qt_xattn_fwd", n_items, H, Lq, Lk, D)) return rc;
    QT_CHECK_ARG(u && x && ctx && lse && asum, "qt_xattn_fwd: null pointer");
    QT_CHECK_ARG(qt_aligned16(u) && qt_aligned16(x) && qt_aligned16(ctx), "qt_xattn_fwd: 16-byte alignment");
    QT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "qt_xattn_fwd: drop_p=%f", drop_p);
    XAttnArgs a; memset(&a, 0, sizeof(a));
    a.u = u; a.x = x; a.ctx = ctx; a.lse = lse; a.asum = asum; a.u_is = u_is; a.u_rs = u_rs;
    a.n_items = n_items; a.H = H; a.Lq = Lq; a.Lk = Lk; a.D = D; a.scale = scale;
    a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.seed = seed; a.salt = qt_salt_ptr;
    hipStream_t st = (hipStream_t)stream;
    if (Lq * H <= 16 && Lk <= 16 && (D == 768 || D == 1024)) {      // the PatchSelecter shape class: both products on the f32 MFMA
        const size_t mlds = sizeof(float) * ((size_t)8 * 256 + 8 * 2 * 16 * 17);
        static bool attr6 = false, attr8 = false;
        if (D == 768) {
            if (!attr6) { (void)hipFuncSetAttribute((const void*)qt_xattn_fwd_mfma_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds); attr6 = true; }
            hipLaunchKernelGGL(qt_xattn_fwd_mfma_kernel<6>, dim3(n_items), dim3(XA_NT), mlds, st, a);
        } else {
            if (!attr8) { (void)hipFuncSetAttribute((const void*)qt_xattn_fwd_mfma_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlds); attr8 = true; }
            hipLaunchKernelGGL(qt_xattn_fwd_mfma_kernel<8>, dim3(n_items), dim3(XA_NT), mlds, st, a);
        }
        QT_CHECK_LAUNCH("qt_xattn_fwd(mfma)");
        return QT_OK;
    }
    const size_t lds = sizeof(float) * (size_t)XA_FCH * D;
    const int nvl = (D / 4 + 63) / 64;                              // float4 per lane
#define XA_FWD(NV_) do { if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)qt_xattn_fwd_kernel<NV_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                         hipLaunchKernelGGL(qt_xattn_fwd_kernel<NV_>, dim3(n_items), dim3(XA_NT), lds, st, a); } while (0)
    if (nvl == 1) XA_FWD(1); else if (nvl == 2) XA_FWD(2); else if (nvl == 3) XA_FWD(3); else XA_FWD(4);
#undef XA_FWD
    QT_CHECK_LAUNCH("qt_xattn_fwd");
    return QT_OK;
}

extern "C" int qt_xattn_bwd(void* stream, const float* u, long u_is, long u_rs, const float* x,
                            const float* ctx, const float* lse, const float* asum, const float* dctx, const float* dasum,
                            int n_items, int H, int Lq, int Lk, int D, float scale, float drop_p, unsigned long long seed,
                            float* du, float* dx, float* dx_amax, float* du_amax) {
    if (int rc = xattn_check("qt_xattn_bwd", n_items, H, Lq, Lk, D)) return rc;
    QT_CHECK_ARG(u && x && ctx && lse && asum && dctx && du && dx, "qt_xattn_bwd: null pointer");
    QT_CHECK_ARG(qt_aligned16(u) && qt_aligned16(x) && qt_aligned16(ctx) && qt_aligned16(dctx) && qt_aligned16(du) && qt_aligned16(dx),
                 "qt_xattn_bwd: 16-byte alignment");
    XAttnArgs a; memset(&a, 0, sizeof(a));
    a.u = u; a.x = x; a.ctx_in = ctx; a.lse = (float*)lse; a.asum = (float*)asum; a.dctx = dctx; a.dasum = dasum;
    a.u_is = u_is; a.u_rs = u_rs; a.n_items = n_items; a.H = H; a.Lq = Lq; a.Lk = Lk; a.D = D; a.scale = scale;
    a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.seed = seed; a.salt = qt_salt_ptr; a.du = du; a.dx = dx; a.dx_amax = dx_amax; a.du_amax = du_amax;
    const int NQ = Lq * H;
    hipStream_t st = (hipStream_t)stream;
    static int force_valu = -1;
    if (force_valu < 0) { const char* e = getenv("QT_XATTN_BWD_VALU"); force_valu = (e && atoi(e)) ? 1 : 0; }
    if (!force_valu && NQ <= 16 && Lk <= 16 && (D == 768 || D == 1024)) {       // the PatchSelecter shape class: all five products on the f32 MFMA
        const size_t mlds = sizeof(float) * ((size_t)2 * 8 * 256 + 8 * 16 + 8 * 3 * 16 * 17);
        if (D == 768) hipLaunchKernelGGL(qt_xattn_bwd_mfma_kernel<6>, dim3(n_items), dim3(XA_NT), mlds, st, a);
        else hipLaunchKernelGGL(qt_xattn_bwd_mfma_kernel<8>, dim3(n_items), dim3(XA_NT), mlds, st, a);
        QT_CHECK_LAUNCH("qt_xattn_bwd(mfma)");
        return QT_OK;
    }
    a.gq = NQ < 16 ? ((NQ + 1) / 2) * 2 : 16;
    size_t lds = 0;
    // u and dctx rows both parked in LDS when that leaves room for two workgroups per CU; otherwise (and if every
    // wave owns at most two key rows) they take turns in one buffer
    a.swap_ug = (Lk <= 16 && sizeof(float) * ((size_t)XA_BCH * D + (size_t)2 * a.gq * D + (size_t)2 * a.gq * Lk) > 80 * 1024) ? 1 : 0;
    for (a.bch = XA_BCH; a.bch >= 1; a.bch >>= 1) {
        lds = sizeof(float) * ((size_t)a.bch * D + (size_t)(a.swap_ug ? 1 : 2) * a.gq * D + (size_t)2 * a.gq * Lk);
        if (lds <= 160 * 1024) break;
    }
    if (a.bch < 1) QT_FAIL(QT_ERR_UNSUPPORTED, "qt_xattn_bwd: D=%d, Lk=%d, Lq*H=%d needs %zu B of LDS (>160 KiB)", D, Lk, NQ, lds);
    const int nvl = (D / 4 + 63) / 64;
#define XA_BWD(NV_) do { if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)qt_xattn_bwd_kernel<NV_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                         hipLaunchKernelGGL(qt_xattn_bwd_kernel<NV_>, dim3(n_items), dim3(XA_NT), lds, st, a); } while (0)
    if (nvl == 1) XA_BWD(1); else if (nvl == 2) XA_BWD(2); else if (nvl == 3) XA_BWD(3); else XA_BWD(4);
#undef XA_BWD
    QT_CHECK_LAUNCH("qt_xattn_bwd");
    return QT_OK;
}
