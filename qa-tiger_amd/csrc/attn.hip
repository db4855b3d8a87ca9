// Small-sequence multi-head attention core, forward and backward (fp32).
//
// Replaces the `softmax(QK^T/sqrt(d_h)) -> dropout -> ·V` core of every
// nn.MultiheadAttention call on the path (reference src/models/modules.py:
// AVQCrossAttn 217-221, QstGrounding 338, TempMoE 562, PatchSelecter 688,698;
// arithmetic restated in SURVEY.md Appendix C).  The in-/out-projections are
// GEMMs (gemm.hip); this file is the part between them.
//
// All sequences on this path are short (Lk = 77 words, 60 frames, 14 patches,
// 2 grounding tokens), so one workgroup owns one (item, head) problem: K and V
// rows of that head are staged once into LDS with coalesced row reads, scores
// live in LDS, the softmax row reduction runs on wavefront shuffles, and no
// score matrix ever goes to HBM.  Q/K/V are read straight out of the packed
// projection outputs through (leading dim, item stride, row stride) triples, so
// no head-split / permute copies exist.
//
// Item mapping: query item n reads keys/values of item (n + kv_shift) % n_kv.
//   self-attention: shift 0, n_kv = n_items
//   audio<->video cross-attention on the stacked [audio; video] batch: shift B
//   question attention (both directions share the words K/V): n_kv = B
#include "qt_common.h"

struct AttnArgs {
    const float* q; long ldq, q_is, q_rs;      // element (item n, row i, col c) = q[(n*q_is + i*q_rs)*ldq + c]
    const float* k; long ldk;
    const float* v; long ldv; long kv_is;      // kv rows of an item are consecutive
    float* o; long ldo, o_is, o_rs;
    char* o_img; long ld_img;   // nullable: the MFMA forward writes o as a SPLIT IMAGE (gemm_plane.hip: per row and 32 columns one 128-byte line [32 x fp16 hi | 32 x fp16 lo], unscaled) instead of fp32 rows
    int n_items, n_kv, kv_shift, H, Lq, Lk, dh;
    float scale, drop_p, inv_keep;
    unsigned long long seed; const unsigned long long* salt;     // salt: qt_common.h "per-step dropout salt"
    // backward only
    const float* d_o; long lddo, do_is, do_rs;
    float* dq; long lddq, dq_is, dq_rs;
    float* dk; long lddk;
    float* dv; long lddv;
    // streaming (long-sequence) path
    float* g_amax;           // backward, nullable: max|dq,dk,dv| record (64 floats): the in-projection's dY scale, measured while it is written
    float* lse;              // [n_items, H, Lq]  log-sum-exp of the scaled scores (written by fwd, read by bwd)
    const float* o_fwd;      // forward output, same addressing as d_o (bwd: delta = rowsum(dO * O))
    float* delta;            // [n_items, H, Lq] workspace
    int causal;              // forward only: key j visible to query i iff j <= i (CLIP text tower)
    int vec_out;             // MFMA forward: the rows of o are 16-byte aligned -> one float4 store per four columns (set by the launcher)
};

__device__ inline int lds_pad(int dh) { return dh + 1; }   // odd row stride: conflict-free column walks

// Forward: grid = n_items * H, block = NT threads.
template <int NT>
__global__ __launch_bounds__(NT) void qt_attn_fwd_kernel(AttnArgs p) {
    if (p.salt && p.drop_p > 0.f) p.seed += *p.salt;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = blockIdx.x / p.H, hd = blockIdx.x - n * p.H;
    const int kvn = (n + p.kv_shift) % p.n_kv;
    const int Lq = p.Lq, Lk = p.Lk, dh = p.dh, dp = lds_pad(dh), sp = Lk + 1;
    float* Qs = lds;                 // [Lq][dp]
    float* Ks = Qs + Lq * dp;        // [Lk][dp]
    float* Vs = Ks + Lk * dp;        // [Lk][dh]
    float* S = Vs + Lk * dh;         // [Lq][sp]
    const int tid = threadIdx.x;

    for (int idx = tid; idx < Lq * dh; idx += NT) {
        const int i = idx / dh, c = idx - i * dh;
        Qs[i * dp + c] = p.q[((long)n * p.q_is + (long)i * p.q_rs) * p.ldq + hd * dh + c] * p.scale;
    }
    for (int idx = tid; idx < Lk * dh; idx += NT) {
        const int j = idx / dh, c = idx - j * dh;
        const long row = (long)kvn * p.kv_is + j;
        Ks[j * dp + c] = p.k[row * p.ldk + hd * dh + c];
        Vs[j * dh + c] = p.v[row * p.ldv + hd * dh + c];
    }
    __syncthreads();

    for (int idx = tid; idx < Lq * Lk; idx += NT) {           // scores
        const int i = idx / Lk, j = idx - i * Lk;
        const float* qr = Qs + i * dp;
        const float* kr = Ks + j * dp;
        float s = 0.f;
        for (int c = 0; c < dh; ++c) s = fmaf(qr[c], kr[c], s);
        S[i * sp + j] = (p.causal && j > i) ? -INFINITY : s;
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6, nw = NT / 64;
    for (int i = wave; i < Lq; i += nw) {                     // softmax rows: one wave per row
        float* row = S + i * sp;
        float mx = -INFINITY;
        for (int j = lane; j < Lk; j += 64) mx = fmaxf(mx, row[j]);
        mx = qt_wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < Lk; j += 64) { const float e = __expf(row[j] - mx); row[j] = e; sum += e; }
        sum = qt_wave_sum(sum);
        const float inv = 1.0f / sum;
        const unsigned long long base = (((unsigned long long)n * p.H + hd) * Lq + i) * (unsigned long long)Lk;
        for (int j = lane; j < Lk; j += 64) {
            float pr = row[j] * inv;
            if (p.drop_p > 0.f) pr *= qt_drop_mult(p.seed, base + j, p.drop_p, p.inv_keep);
            row[j] = pr;
        }
    }
    __syncthreads();

    for (int idx = tid; idx < Lq * dh; idx += NT) {           // O = P V
        const int i = idx / dh, c = idx - i * dh;
        const float* pr = S + i * sp;
        float acc = 0.f;
        for (int j = 0; j < Lk; ++j) acc = fmaf(pr[j], Vs[j * dh + c], acc);
        p.o[((long)n * p.o_is + (long)i * p.o_rs) * p.ldo + hd * dh + c] = acc;
    }
}

// Backward: grid = n_kv * H.  One workgroup owns the dK/dV rows of its (kv item,
// head) and walks every query item that attends to it, so dK/dV need no atomics.
template <int NT>
__global__ __launch_bounds__(NT) void qt_attn_bwd_kernel(AttnArgs p) {
    if (p.salt && p.drop_p > 0.f) p.seed += *p.salt;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int kvn = blockIdx.x / p.H, hd = blockIdx.x - kvn * p.H;
    const int Lq = p.Lq, Lk = p.Lk, dh = p.dh, dp = lds_pad(dh), sp = Lk + 1;
    float* Qs = lds;                  // [Lq][dp]   scaled q
    float* dOs = Qs + Lq * dp;        // [Lq][dp]
    float* Ks = dOs + Lq * dp;        // [Lk][dp]
    float* Vs = Ks + Lk * dp;         // [Lk][dp]
    float* S = Vs + Lk * dp;          // [Lq][sp]  P (pre-dropout), then dS
    float* dP = S + Lq * sp;          // [Lq][sp]  dropped P for dV
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = NT / 64;

    for (int idx = tid; idx < Lk * dh; idx += NT) {
        const int j = idx / dh, c = idx - j * dh;
        const long row = (long)kvn * p.kv_is + j;
        Ks[j * dp + c] = p.k[row * p.ldk + hd * dh + c];
        Vs[j * dp + c] = p.v[row * p.ldv + hd * dh + c];
    }
    int first = kvn - p.kv_shift;
    first %= p.n_kv;
    if (first < 0) first += p.n_kv;
    bool accumulate_kv = false;
    for (int n = first; n < p.n_items; n += p.n_kv) {
        __syncthreads();
        for (int idx = tid; idx < Lq * dh; idx += NT) {
            const int i = idx / dh, c = idx - i * dh;
            Qs[i * dp + c] = p.q[((long)n * p.q_is + (long)i * p.q_rs) * p.ldq + hd * dh + c] * p.scale;
            dOs[i * dp + c] = p.d_o[((long)n * p.do_is + (long)i * p.do_rs) * p.lddo + hd * dh + c];
        }
        __syncthreads();
        for (int idx = tid; idx < Lq * Lk; idx += NT) {       // S = qK^T ; dPd = dO V^T
            const int i = idx / Lk, j = idx - i * Lk;
            const float* qr = Qs + i * dp; const float* kr = Ks + j * dp;
            const float* gr = dOs + i * dp; const float* vr = Vs + j * dp;
            float s = 0.f, g = 0.f;
            for (int c = 0; c < dh; ++c) { s = fmaf(qr[c], kr[c], s); g = fmaf(gr[c], vr[c], g); }
            S[i * sp + j] = s;
            dP[i * sp + j] = g;
        }
        __syncthreads();
        for (int i = wave; i < Lq; i += nw) {
            float* srow = S + i * sp; float* grow = dP + i * sp;
            float mx = -INFINITY;
            for (int j = lane; j < Lk; j += 64) mx = fmaxf(mx, srow[j]);
            mx = qt_wave_max(mx);
            float sum = 0.f;
            for (int j = lane; j < Lk; j += 64) { const float e = __expf(srow[j] - mx); srow[j] = e; sum += e; }
            sum = qt_wave_sum(sum);
            const float inv = 1.0f / sum;
            const unsigned long long base = (((unsigned long long)n * p.H + hd) * Lq + i) * (unsigned long long)Lk;
            float delta = 0.f;
            for (int j = lane; j < Lk; j += 64) {
                const float pr = srow[j] * inv;
                const float mult = (p.drop_p > 0.f) ? qt_drop_mult(p.seed, base + j, p.drop_p, p.inv_keep) : 1.0f;
                const float dpr = grow[j] * mult;             // grad wrt pre-dropout P
                delta += dpr * pr;
                srow[j] = pr;
                grow[j] = dpr;
            }
            delta = qt_wave_sum(delta);
            for (int j = lane; j < Lk; j += 64) {
                const float pr = srow[j];
                const float mult = (p.drop_p > 0.f) ? qt_drop_mult(p.seed, base + j, p.drop_p, p.inv_keep) : 1.0f;
                const float ds = pr * (grow[j] - delta);
                srow[j] = ds;                                 // dS
                grow[j] = pr * mult;                          // dropped P (forward value)
            }
        }
        __syncthreads();
        for (int idx = tid; idx < Lq * dh; idx += NT) {       // dQ = dS K * scale
            const int i = idx / dh, c = idx - i * dh;
            const float* dsr = S + i * sp;
            float acc = 0.f;
            for (int j = 0; j < Lk; ++j) acc = fmaf(dsr[j], Ks[j * dp + c], acc);
            p.dq[((long)n * p.dq_is + (long)i * p.dq_rs) * p.lddq + hd * dh + c] = acc * p.scale;
        }
        for (int idx = tid; idx < Lk * dh; idx += NT) {       // dK = dS^T (q*scale) ; dV = Pd^T dO
            const int j = idx / dh, c = idx - j * dh;
            float ak = 0.f, av = 0.f;
            for (int i = 0; i < Lq; ++i) {
                ak = fmaf(S[i * sp + j], Qs[i * dp + c], ak);
                av = fmaf(dP[i * sp + j], dOs[i * dp + c], av);
            }
            const long row = (long)kvn * p.kv_is + j;
            float* dkp = p.dk + row * p.lddk + hd * dh + c;
            float* dvp = p.dv + row * p.lddv + hd * dh + c;
            if (accumulate_kv) { ak += *dkp; av += *dvp; }
            *dkp = ak; *dvp = av;
        }
        accumulate_kv = true;
    }
}


// ============================================================================
// Matrix-core path (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32 accumulate)
// ============================================================================
// Used when d_h is a multiple of 16 (<=128), Lk <= 128 and rows are 16-byte aligned;
// the VALU kernels above remain the generic path for odd shapes.
//
// Lane roles of the 16x16x4 MFMA: c = lane&15, g = lane>>4.
//   A operand: A[row=c][k=g]    B operand: B[k=g][col=c]    C/D reg j: [row=4g+j][col=c]
//
// Forward, per wave and 16-query tile:
//   S^T tile (keys x queries) = K_tile · Q_tile^T  — "swapped" so that a lane owns ONE query
//   column: lane (c,g) holds scores of query c for keys {16kt + 4g + j}.  The softmax row
//   reduction is a register reduce + two shuffles (xor 16, 32).  K rows come from LDS as
//   16-byte vectors; the 4 elements of a vector are used in 4 consecutive MFMA k-steps, the
//   same permutation on the Q side, which a dot product does not care about.
//   O tile = P · V: the P registers ARE the A operand (row = query c, k-step j covers keys
//   16kt + 4g + j), V rows are dword LDS reads (row stride d_h+4 keeps the two 16-lane
//   halves on disjoint banks).  No score ever leaves the registers.
// Backward: phase A (wave per query tile) recomputes P, forms dP^T = V·dO^T the same way,
//   dS = P∘(dP − Σ dP∘P), dQ = dS·K from registers, and parks P_dropped^T and dS^T in LDS
//   [key][query]; phase B (wave per key tile) computes dV = P^T·dO and dK = dS^T·Q with the
//   query index as the MFMA k dimension, so dK/dV need no atomics.
typedef float floatx4v __attribute__((ext_vector_type(4)));

template <int DC, int NKT>
__global__ __launch_bounds__(256) void qt_attn_fwd_mfma_kernel(AttnArgs p, int wpp) {
    if (p.salt && p.drop_p > 0.f) p.seed += *p.salt;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int DH = DC * 16, LDK = DH + 4, KR = NKT * 16, V4 = DH / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int ppb = 4 / wpp, pl = wave / wpp, wip = wave - pl * wpp;
    const long prob = (long)blockIdx.x * ppb + pl;
    const bool active = prob < (long)p.n_items * p.H;
    const int n = active ? (int)(prob / p.H) : 0, hd = active ? (int)(prob - (long)n * p.H) : 0;
    const int kvn = (n + p.kv_shift) % p.n_kv;
    const int Lq = p.Lq, Lk = p.Lk;
    float* Ks = lds + pl * (2 * KR * LDK);
    float* Vs = Ks + KR * LDK;
    {   // K|V fill with all trips of loads in flight before the first LDS store (as in the backward): 14x14 launch 0.545 -> 0.466 ms
        constexpr int FU = 6;
        for (int base = wip * 64 + lane; base < KR * V4; base += FU * wpp * 64) {
            float4 kk[FU], vv[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int idx = base + u * wpp * 64;
                const int j = idx / V4, c4 = idx - j * V4;
                kk[u] = make_float4(0.f, 0.f, 0.f, 0.f); vv[u] = kk[u];
                if (active && idx < KR * V4 && j < Lk) {
                    const long row = (long)kvn * p.kv_is + j;
                    kk[u] = *reinterpret_cast<const float4*>(p.k + row * p.ldk + hd * DH + c4 * 4);
                    vv[u] = *reinterpret_cast<const float4*>(p.v + row * p.ldv + hd * DH + c4 * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int idx = base + u * wpp * 64;
                if (idx < KR * V4) {
                    const int j = idx / V4, c4 = idx - j * V4;
                    *reinterpret_cast<float4*>(Ks + j * LDK + c4 * 4) = kk[u];
                    *reinterpret_cast<float4*>(Vs + j * LDK + c4 * 4) = vv[u];
                }
            }
        }
    }
    if (wpp > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();   // one wave per problem: its LDS region is private, LDS ops of a wave execute in order
    if (!active) return;
    for (int q0 = wip * 16; q0 < Lq; q0 += wpp * 16) {
        const int q = q0 + c;
        const bool qok = q < Lq;
        float4 qf[DC];
#pragma unroll
        for (int ch = 0; ch < DC; ++ch) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (qok) t = *reinterpret_cast<const float4*>(p.q + ((long)n * p.q_is + (long)q * p.q_rs) * p.ldq + hd * DH + ch * 16 + g * 4);
            qf[ch] = make_float4(t.x * p.scale, t.y * p.scale, t.z * p.scale, t.w * p.scale);
        }
        floatx4v st[NKT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            floatx4v acc = {0.f, 0.f, 0.f, 0.f};
            if (kt * 16 < Lk) {
#pragma unroll
                for (int ch = 0; ch < DC; ++ch) {
                    const float4 kf = *reinterpret_cast<const float4*>(Ks + (kt * 16 + c) * LDK + ch * 16 + g * 4);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[ch].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[ch].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[ch].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[ch].w, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (kt * 16 + g * 4 + j >= Lk) acc[j] = -INFINITY;
                mx = fmaxf(mx, acc[j]);
            }
            st[kt] = acc;
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float e = __expf(st[kt][j] - mx); st[kt][j] = e; sum += e; }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        const unsigned long long base = (((unsigned long long)n * p.H + hd) * Lq + (qok ? q : 0)) * (unsigned long long)Lk;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float pr = st[kt][j] * inv;
                const int key = kt * 16 + g * 4 + j;
                if (p.drop_p > 0.f && key < Lk) pr *= qt_drop_mult(p.seed, base + key, p.drop_p, p.inv_keep);
                st[kt][j] = pr;
            }
        floatx4v oacc[DC];
#pragma unroll
        for (int dt = 0; dt < DC; ++dt) oacc[dt] = floatx4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            if (kt * 16 >= Lk) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* vrow = Vs + (kt * 16 + g * 4 + j) * LDK + c;
#pragma unroll
                for (int dt = 0; dt < DC; ++dt)     // O^T tile = V^T · P^T (the operands of P · V exchanged): lane (c, g) then owns row q0 + c, columns 16 dt + 4 g .. + 3
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[dt * 16], st[kt][j], oacc[dt], 0, 0, 0);
            }
        }
        // Stores are ISSUE-bound on this chip (~64 cycles of the CU's store path per wave instruction whatever its width,
        // MI355X_MICROARCH.md "store tail"): with the plain orientation a lane held one column of four rows = four dword stores per
        // 16 columns, and the 14x14 patch launch spent 0.38 of its 0.50 ms issuing them.  Four consecutive columns per lane = one
        // 16-byte store (fp32 rows), or after one v_permlane16_swap per dword of (hi, lo) one 16-byte store of eight columns' hi or lo
        // halves (split image): a quarter of the store instructions for the same bytes.
        if (p.o_img) {
            typedef _Float16 h2v __attribute__((ext_vector_type(2)));
            typedef float f2v __attribute__((ext_vector_type(2)));
            const int qo = q0 + c;
            char* irow = p.o_img + ((long)n * p.o_is + (long)min(qo, Lq - 1) * p.o_rs) * p.ld_img;
#pragma unroll
            for (int dt = 0; dt < DC; ++dt) {
                const f2v x01 = {oacc[dt][0], oacc[dt][1]}, x23 = {oacc[dt][2], oacc[dt][3]};
                const h2v h01 = __builtin_convertvector(x01, h2v), h23 = __builtin_convertvector(x23, h2v);
                const h2v l01 = __builtin_convertvector(x01 - __builtin_convertvector(h01, f2v), h2v);
                const h2v l23 = __builtin_convertvector(x23 - __builtin_convertvector(h23, f2v), h2v);
                // rows of 16 lanes = g: odd rows of the first operand <-> even rows of the second.  Even g then holds (own hi, g+1's hi),
                // odd g holds (g-1's lo, own lo): eight consecutive columns' hi halves resp. lo halves, 16 bytes
                const auto r0 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, l01), false, false);
                const auto r1 = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, h23), __builtin_bit_cast(unsigned, l23), false, false);
                const int col = hd * DH + dt * 16 + 4 * (g & ~1);             // first of the eight columns
                if (qo < Lq)
                    *reinterpret_cast<uint4*>(irow + (long)(col >> 5) * 128 + ((g & 1) ? 64 : 0) + (col & 31) * 2) = make_uint4(r0[0], r1[0], r0[1], r1[1]);
            }
        } else if (q0 + c < Lq) {
            float* orow = p.o + ((long)n * p.o_is + (long)(q0 + c) * p.o_rs) * p.ldo + hd * DH + 4 * g;
            if (p.vec_out) {
#pragma unroll
                for (int dt = 0; dt < DC; ++dt) *reinterpret_cast<float4*>(orow + dt * 16) = make_float4(oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]);
            } else {
#pragma unroll
                for (int dt = 0; dt < DC; ++dt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) orow[dt * 16 + j] = oacc[dt][j];
            }
        }
    }
}

// (second launch-bound: waves per SIMD the register allocation must leave room for — three one-wave problems per SIMD for the 14x14
// launch, two workgroups per CU for the multi-wave shapes; unconstrained, the compiler spent 177 resp. > 256 registers per lane)
template <int DC, int NKT>
__global__ __launch_bounds__(256, NKT == 1 ? 3 : (NKT <= 4 ? 2 : 1)) void qt_attn_bwd_mfma_kernel(AttnArgs p, int wpp, int nqt) {
    if (p.salt && p.drop_p > 0.f) p.seed += *p.salt;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int DH = DC * 16, LDK = DH + 4, KR = NKT * 16, V4 = DH / 4;
    const int LDQ = nqt * 16 + 4, QR = nqt * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int ppb = 4 / wpp, pl = wave / wpp, wip = wave - pl * wpp;
    const long prob = (long)blockIdx.x * ppb + pl;
    const bool active = prob < (long)p.n_kv * p.H;
    const int kvn = active ? (int)(prob / p.H) : 0, hd = active ? (int)(prob - (long)kvn * p.H) : 0;
    const int Lq = p.Lq, Lk = p.Lk;
    // LDS holds exactly Lk key rows / Lq query rows, not the 16-row tile multiples: the MFMA tiles' padding rows are read from
    // the last real row instead (their products are multiplied by probabilities / dS that are exactly 0, or land in output rows
    // that are never stored).  60x60 attention with d_h = 96: 78.8 KB per workgroup instead of 86 KB -> TWO workgroups per CU;
    // the 14x14 patch attention: 52.5 KB instead of 60 KB -> three.
    const int mxr = Lk > Lq ? Lk : Lq;
    const int regA = 2 * mxr * LDK;                                     // K|V during phase A, Q|dO during phase B
    float* base_l = lds + pl * (regA + 2 * Lk * LDQ);
    float* Ks = base_l;
    float* Vs = Ks + Lk * LDK;
    float* Qs = base_l;                                                 // overlay (after the phase-A barrier)
    float* Gs = Qs + Lq * LDK;
    float* PT = base_l + regA;                                          // [key][query] dropped P
    float* ST = PT + Lk * LDQ;                                          // [key][query] dS
    int first = (kvn - p.kv_shift) % p.n_kv;
    if (first < 0) first += p.n_kv;
    const int reps = p.n_items / p.n_kv;                                // uniform trip count (host guarantees divisibility)
    float gmax = 0.f;                                                   // max |gradient| this thread writes (g_amax)
    for (int rep = 0; rep < reps; ++rep) {
        const int n = first + rep * p.n_kv;
        if (wpp > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();   // one wave per problem: its LDS region is private, LDS ops of a wave execute in order                                                // previous phase B done with Q|dO and PT|ST
        // (several trips of loads in flight before the first LDS store: one trip at a time exposed a global-load latency per trip,
        // ~6 trips for K|V and 6 for Q|dO per problem.  Also tried: requesting the wave's q / dO fragments of phase A before this
        // fill — the 48 extra live registers cost the 14x14 launch its third workgroup per CU: 1.23 -> 1.65 ms)
        constexpr int FU = 3;
        for (int base = wip * 64 + lane; base < Lk * V4; base += FU * wpp * 64) {
            float4 kk[FU], vv[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int idx = base + u * wpp * 64;
                kk[u] = make_float4(0.f, 0.f, 0.f, 0.f); vv[u] = kk[u];
                if (active && idx < Lk * V4) {
                    const int j = idx / V4, c4 = idx - j * V4;
                    const long row = (long)kvn * p.kv_is + j;
                    kk[u] = *reinterpret_cast<const float4*>(p.k + row * p.ldk + hd * DH + c4 * 4);
                    vv[u] = *reinterpret_cast<const float4*>(p.v + row * p.ldv + hd * DH + c4 * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int idx = base + u * wpp * 64;
                if (idx < Lk * V4) {
                    const int j = idx / V4, c4 = idx - j * V4;
                    *reinterpret_cast<float4*>(Ks + j * LDK + c4 * 4) = kk[u];
                    *reinterpret_cast<float4*>(Vs + j * LDK + c4 * 4) = vv[u];
                }
            }
        }
        if (wpp > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();   // one wave per problem: its LDS region is private, LDS ops of a wave execute in order
        // Multi-wave problems (NKT > 1: the AVQ shapes) have registers to spare (one or two workgroups per CU by LDS): the Q|dO rows
        // that phase B will need in LDS are requested NOW and parked in registers across phase A, so their latency hides behind
        // phase A's MFMAs instead of sitting between the two phases.  (Not for the 14x14 launch: there registers decide occupancy.)
        constexpr bool PF = false && NKT > 1;   // (measured: with the registers this costs, the multi-wave shapes spill)
        constexpr int QTR = PF ? (64 * V4 + 255) / 256 : 1;
        float4 pq[QTR], pg[QTR];
        if constexpr (PF) {
#pragma unroll
            for (int u = 0; u < QTR; ++u) {
                const int idx = wip * 64 + lane + u * wpp * 64;
                pq[u] = make_float4(0.f, 0.f, 0.f, 0.f); pg[u] = pq[u];
                if (active && idx < Lq * V4) {
                    const int qi = idx / V4, c4 = idx - qi * V4;
                    pq[u] = *reinterpret_cast<const float4*>(p.q + ((long)n * p.q_is + (long)qi * p.q_rs) * p.ldq + hd * DH + c4 * 4);
                    pg[u] = *reinterpret_cast<const float4*>(p.d_o + ((long)n * p.do_is + (long)qi * p.do_rs) * p.lddo + hd * DH + c4 * 4);
                }
            }
        }
        // ---------------- phase A: one wave per 16-query tile
        for (int q0 = wip * 16; q0 < QR; q0 += wpp * 16) {
            const int q = q0 + c;
            const bool qok = active && q < Lq;
            float4 qf[DC], gf[DC];
#pragma unroll
            for (int ch = 0; ch < DC; ++ch) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f), u = t;
                if (qok) {
                    t = *reinterpret_cast<const float4*>(p.q + ((long)n * p.q_is + (long)q * p.q_rs) * p.ldq + hd * DH + ch * 16 + g * 4);
                    u = *reinterpret_cast<const float4*>(p.d_o + ((long)n * p.do_is + (long)q * p.do_rs) * p.lddo + hd * DH + ch * 16 + g * 4);
                }
                qf[ch] = make_float4(t.x * p.scale, t.y * p.scale, t.z * p.scale, t.w * p.scale);
                gf[ch] = u;
            }
            floatx4v st[NKT], dp[NKT];
            float mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                floatx4v a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
                if (kt * 16 < Lk) {
                    const int rk = min(kt * 16 + c, Lk - 1);                // (a padding key's score is set to -inf below)
#pragma unroll
                    for (int ch = 0; ch < DC; ++ch) {
                        const float4 kf = *reinterpret_cast<const float4*>(Ks + rk * LDK + ch * 16 + g * 4);
                        const float4 vf = *reinterpret_cast<const float4*>(Vs + rk * LDK + ch * 16 + g * 4);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[ch].x, a1, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, gf[ch].x, a2, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[ch].y, a1, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, gf[ch].y, a2, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[ch].z, a1, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, gf[ch].z, a2, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[ch].w, a1, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, gf[ch].w, a2, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (kt * 16 + g * 4 + j >= Lk) a1[j] = -INFINITY;
                    mx = fmaxf(mx, a1[j]);
                }
                st[kt] = a1; dp[kt] = a2;
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float e = __expf(st[kt][j] - mx); st[kt][j] = e; sum += e; }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            const float inv = qok ? 1.0f / sum : 0.f;                   // padded query columns contribute nothing
            const unsigned long long base = (((unsigned long long)n * p.H + hd) * Lq + (qok ? q : 0)) * (unsigned long long)Lk;
            float delta = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int key = kt * 16 + g * 4 + j;
                    const float pr = st[kt][j] * inv;
                    const float mult = (p.drop_p > 0.f && key < Lk) ? qt_drop_mult(p.seed, base + key, p.drop_p, p.inv_keep) : 1.0f;
                    const float dpr = dp[kt][j] * mult;                 // grad wrt the pre-dropout probability
                    delta += dpr * pr;
                    st[kt][j] = pr;
                    dp[kt][j] = dpr;
                    if (key < Lk) PT[key * LDQ + q] = pr * mult;        // dropped P (forward value), for dV
                }
            delta += __shfl_xor(delta, 16, 64);
            delta += __shfl_xor(delta, 32, 64);
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float ds = st[kt][j] * (dp[kt][j] - delta);
                    st[kt][j] = ds;
                    if (kt * 16 + g * 4 + j < Lk) ST[(kt * 16 + g * 4 + j) * LDQ + q] = ds;
                }
            floatx4v oacc[DC];
#pragma unroll
            for (int dt = 0; dt < DC; ++dt) oacc[dt] = floatx4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                if (kt * 16 >= Lk) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* krow = Ks + min(kt * 16 + g * 4 + j, Lk - 1) * LDK + c;      // dS of a padding key is exactly 0
#pragma unroll
                    for (int dt = 0; dt < DC; ++dt)     // dQ^T tile = K^T · dS^T (operands exchanged): lane (c, g) owns row q0 + c, columns 16 dt + 4 g .. + 3
                        oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(krow[dt * 16], st[kt][j], oacc[dt], 0, 0, 0);
                }
            }
            if (active && q < Lq) {             // one 16-byte store per 16 columns instead of four dword stores (stores are issue-bound: forward kernel)
                float* drow = p.dq + ((long)n * p.dq_is + (long)q * p.dq_rs) * p.lddq + hd * DH + 4 * g;
#pragma unroll
                for (int dt = 0; dt < DC; ++dt) {
                    const float4 dqv = make_float4(oacc[dt][0] * p.scale, oacc[dt][1] * p.scale, oacc[dt][2] * p.scale, oacc[dt][3] * p.scale);
                    gmax = fmaxf(fmaxf(gmax, fmaxf(fabsf(dqv.x), fabsf(dqv.y))), fmaxf(fabsf(dqv.z), fabsf(dqv.w)));
                    *reinterpret_cast<float4*>(drow + dt * 16) = dqv;        // (16-byte aligned rows: attn_bwd_extras_ok)
                }
            }
        }
        if (wpp > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();   // one wave per problem: its LDS region is private, LDS ops of a wave execute in order                                                // K|V dead; PT|ST complete
        if constexpr (PF) {
#pragma unroll
            for (int u = 0; u < QTR; ++u) {
                const int idx = wip * 64 + lane + u * wpp * 64;
                if (idx < Lq * V4) {
                    const int qi = idx / V4, c4 = idx - qi * V4;
                    *reinterpret_cast<float4*>(Qs + qi * LDK + c4 * 4) = make_float4(pq[u].x * p.scale, pq[u].y * p.scale, pq[u].z * p.scale, pq[u].w * p.scale);
                    *reinterpret_cast<float4*>(Gs + qi * LDK + c4 * 4) = pg[u];
                }
            }
        } else
        for (int base = wip * 64 + lane; base < Lq * V4; base += FU * wpp * 64) {
            float4 tq[FU], tg[FU];
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int idx = base + u * wpp * 64;
                tq[u] = make_float4(0.f, 0.f, 0.f, 0.f); tg[u] = tq[u];
                if (active && idx < Lq * V4) {
                    const int qi = idx / V4, c4 = idx - qi * V4;
                    tq[u] = *reinterpret_cast<const float4*>(p.q + ((long)n * p.q_is + (long)qi * p.q_rs) * p.ldq + hd * DH + c4 * 4);
                    tg[u] = *reinterpret_cast<const float4*>(p.d_o + ((long)n * p.do_is + (long)qi * p.do_rs) * p.lddo + hd * DH + c4 * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int idx = base + u * wpp * 64;
                if (idx < Lq * V4) {
                    const int qi = idx / V4, c4 = idx - qi * V4;
                    *reinterpret_cast<float4*>(Qs + qi * LDK + c4 * 4) = make_float4(tq[u].x * p.scale, tq[u].y * p.scale, tq[u].z * p.scale, tq[u].w * p.scale);
                    *reinterpret_cast<float4*>(Gs + qi * LDK + c4 * 4) = tg[u];
                }
            }
        }
        if (wpp > 1) __syncthreads(); else __builtin_amdgcn_wave_barrier();   // one wave per problem: its LDS region is private, LDS ops of a wave execute in order
        // ---------------- phase B: one wave per 16-key tile; MFMA k dimension = query index
        for (int kt = wip; kt * 16 < Lk; kt += wpp) {
            floatx4v av[DC], ak[DC];
#pragma unroll
            for (int dt = 0; dt < DC; ++dt) { av[dt] = floatx4v{0.f, 0.f, 0.f, 0.f}; ak[dt] = av[dt]; }
            for (int qc = 0; qc < nqt; ++qc) {
                const int rk = min(kt * 16 + c, Lk - 1);                    // a padding key's dK / dV rows are never stored
                const float4 pf = *reinterpret_cast<const float4*>(PT + rk * LDQ + qc * 16 + g * 4);
                const float4 sf = *reinterpret_cast<const float4*>(ST + rk * LDQ + qc * 16 + g * 4);
                const float pa[4] = {pf.x, pf.y, pf.z, pf.w}, sa[4] = {sf.x, sf.y, sf.z, sf.w};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int rq = min(qc * 16 + g * 4 + s, Lq - 1);        // P and dS columns of padding queries are exactly 0
                    const float* grow = Gs + rq * LDK + c;
                    const float* qrow = Qs + rq * LDK + c;
#pragma unroll
                    for (int dt = 0; dt < DC; ++dt) {
                        av[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(grow[dt * 16], pa[s], av[dt], 0, 0, 0);     // dV^T, dK^T tiles (operands exchanged):
                        ak[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(qrow[dt * 16], sa[s], ak[dt], 0, 0, 0);     // lane (c, g) owns key 16 kt + c, columns 16 dt + 4 g .. + 3
                    }
                }
            }
            if (active && kt * 16 + c < Lk) {
                const long row = (long)kvn * p.kv_is + kt * 16 + c;
                float* dkp = p.dk + row * p.lddk + hd * DH + 4 * g;
                float* dvp = p.dv + row * p.lddv + hd * DH + 4 * g;
#pragma unroll
                for (int dt = 0; dt < DC; ++dt) {
                    float4 kk = make_float4(ak[dt][0], ak[dt][1], ak[dt][2], ak[dt][3]), vv = make_float4(av[dt][0], av[dt][1], av[dt][2], av[dt][3]);
                    if (rep > 0) {          // a later query item of the same keys: add to what the earlier ones stored
                        const float4 ok = *reinterpret_cast<const float4*>(dkp + dt * 16), ov = *reinterpret_cast<const float4*>(dvp + dt * 16);
                        kk.x += ok.x; kk.y += ok.y; kk.z += ok.z; kk.w += ok.w; vv.x += ov.x; vv.y += ov.y; vv.z += ov.z; vv.w += ov.w;
                    }
                    gmax = fmaxf(gmax, fmaxf(fmaxf(fmaxf(fabsf(kk.x), fabsf(kk.y)), fmaxf(fabsf(kk.z), fabsf(kk.w))), fmaxf(fmaxf(fabsf(vv.x), fabsf(vv.y)), fmaxf(fabsf(vv.z), fabsf(vv.w)))));
                    *reinterpret_cast<float4*>(dkp + dt * 16) = kk;
                    *reinterpret_cast<float4*>(dvp + dt * 16) = vv;
                }
            }
        }
    }
    if (p.g_amax) {
        __shared__ float red[4];
        qt_amax_fold(gmax, red, p.g_amax);
    }
}


// ============================================================================
// Streaming path for long sequences (BASELINE config 4: T = 300, d_h = 128)
// ============================================================================
// Keys/values no longer fit in LDS, so they are streamed in chunks of 64 with an online softmax
// (running max m, running sum l, O rescaled per chunk).  Same MFMA fragments as the resident kernels.
// Backward is the two-kernel form: dQ per query tile (streams K/V), dK/dV per key tile (streams
// Q/dO), both recomputing P = exp(s - lse) from the forward's log-sum-exp, so nothing is atomic.
#define SKC 64      // keys (or queries) per streamed chunk

template <int DC>
__global__ __launch_bounds__(256) void qt_attn_fwd_stream_kernel(AttnArgs p) {
    if (p.salt && p.drop_p > 0.f) p.seed += *p.salt;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int DH = DC * 16, LDK = DH + 4, V4 = DH / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int n = blockIdx.x / p.H, hd = blockIdx.x - n * p.H;
    const int kvn = (n + p.kv_shift) % p.n_kv;
    const int Lq = p.Lq, Lk = p.Lk;
    const int q0 = (blockIdx.y * 4 + wave) * 16;
    const bool wave_on = q0 < Lq;
    const int q = q0 + c;
    const bool qok = wave_on && q < Lq;
    float* Ks = lds;
    float* Vs = Ks + SKC * LDK;
    float4 qf[DC];
#pragma unroll
    for (int ch = 0; ch < DC; ++ch) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
        if (qok) t = *reinterpret_cast<const float4*>(p.q + ((long)n * p.q_is + (long)q * p.q_rs) * p.ldq + hd * DH + ch * 16 + g * 4);
        qf[ch] = make_float4(t.x * p.scale, t.y * p.scale, t.z * p.scale, t.w * p.scale);
    }
    floatx4v oacc[DC];
#pragma unroll
    for (int dt = 0; dt < DC; ++dt) oacc[dt] = floatx4v{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;          // per query column (lane c); l is this lane's partial over its keys
    const unsigned long long base = (((unsigned long long)n * p.H + hd) * Lq + (qok ? q : 0)) * (unsigned long long)Lk;
    for (int kc0 = 0; kc0 < Lk; kc0 += SKC) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < SKC * V4; idx += 256) {
            const int j = idx / V4, c4 = idx - j * V4;
            float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
            if (kc0 + j < Lk) {
                const long row = (long)kvn * p.kv_is + kc0 + j;
                kk = *reinterpret_cast<const float4*>(p.k + row * p.ldk + hd * DH + c4 * 4);
                vv = *reinterpret_cast<const float4*>(p.v + row * p.ldv + hd * DH + c4 * 4);
            }
            *reinterpret_cast<float4*>(Ks + j * LDK + c4 * 4) = kk;
            *reinterpret_cast<float4*>(Vs + j * LDK + c4 * 4) = vv;
        }
        __syncthreads();
        if (!wave_on) continue;
        floatx4v st[4];
        float cm = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            floatx4v acc = {0.f, 0.f, 0.f, 0.f};
            if (kc0 + kt * 16 < Lk) {
#pragma unroll
                for (int ch = 0; ch < DC; ++ch) {
                    const float4 kf = *reinterpret_cast<const float4*>(Ks + (kt * 16 + c) * LDK + ch * 16 + g * 4);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[ch].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[ch].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[ch].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[ch].w, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = kc0 + kt * 16 + g * 4 + j;
                if (key >= Lk || (p.causal && key > q)) acc[j] = -INFINITY;   // causal: the first chunk always holds key 0 <= q
                cm = fmaxf(cm, acc[j]);
            }
            st[kt] = acc;
        }
        cm = fmaxf(cm, __shfl_xor(cm, 16, 64));
        cm = fmaxf(cm, __shfl_xor(cm, 32, 64));
        const float m_new = fmaxf(m_run, cm);                         // finite: every chunk holds a valid key
        const float alpha = __expf(m_run - m_new);                    // 0 on the first chunk
        m_run = m_new;
        float lsum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float e = __expf(st[kt][j] - m_new);
                lsum += e;
                const int key = kc0 + kt * 16 + g * 4 + j;
                if (p.drop_p > 0.f && key < Lk) e *= qt_drop_mult(p.seed, base + key, p.drop_p, p.inv_keep);
                st[kt][j] = e;
            }
        l_run = l_run * alpha + lsum;
        // O rows are queries 4g+j of this tile; their rescale factor lives in lane (c = 4g+j)
        float al[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) al[j] = __shfl(alpha, g * 4 + j, 64);
#pragma unroll
        for (int dt = 0; dt < DC; ++dt)
#pragma unroll
            for (int j = 0; j < 4; ++j) oacc[dt][j] *= al[j];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kc0 + kt * 16 >= Lk) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* vrow = Vs + (kt * 16 + g * 4 + j) * LDK + c;
#pragma unroll
                for (int dt = 0; dt < DC; ++dt)
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[kt][j], vrow[dt * 16], oacc[dt], 0, 0, 0);
            }
        }
    }
    if (!wave_on) return;
    float l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    const float inv = 1.0f / l_tot;
    if (g == 0 && qok && p.lse) p.lse[((long)n * p.H + hd) * Lq + q] = m_run + __logf(l_tot);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float iv = __shfl(inv, g * 4 + j, 64);
        const int qo = q0 + g * 4 + j;
        if (qo < Lq) {
            float* orow = p.o + ((long)n * p.o_is + (long)qo * p.o_rs) * p.ldo + hd * DH + c;
#pragma unroll
            for (int dt = 0; dt < DC; ++dt) orow[dt * 16] = oacc[dt][j] * iv;
        }
    }
}

// dQ (and delta = rowsum(dO*O)) per 16-query tile; streams K/V chunks.
template <int DC>
__global__ __launch_bounds__(256) void qt_attn_bwd_dq_kernel(AttnArgs p) {
    if (p.salt && p.drop_p > 0.f) p.seed += *p.salt;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int DH = DC * 16, LDK = DH + 4, V4 = DH / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int n = blockIdx.x / p.H, hd = blockIdx.x - n * p.H;
    const int kvn = (n + p.kv_shift) % p.n_kv;
    const int Lq = p.Lq, Lk = p.Lk;
    const int q0 = (blockIdx.y * 4 + wave) * 16;
    const bool wave_on = q0 < Lq;
    const int q = q0 + c;
    const bool qok = wave_on && q < Lq;
    float* Ks = lds;
    float* Vs = Ks + SKC * LDK;
    float4 qf[DC], gf[DC];
    float dpart = 0.f;
#pragma unroll
    for (int ch = 0; ch < DC; ++ch) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f), u = t, w = t;
        if (qok) {
            t = *reinterpret_cast<const float4*>(p.q + ((long)n * p.q_is + (long)q * p.q_rs) * p.ldq + hd * DH + ch * 16 + g * 4);
            const long orow = ((long)n * p.do_is + (long)q * p.do_rs) * p.lddo + hd * DH + ch * 16 + g * 4;
            u = *reinterpret_cast<const float4*>(p.d_o + orow);
            w = *reinterpret_cast<const float4*>(p.o_fwd + orow);
        }
        qf[ch] = make_float4(t.x * p.scale, t.y * p.scale, t.z * p.scale, t.w * p.scale);
        gf[ch] = u;
        dpart += u.x * w.x + u.y * w.y + u.z * w.z + u.w * w.w;
    }
    dpart += __shfl_xor(dpart, 16, 64);
    dpart += __shfl_xor(dpart, 32, 64);
    const float delta = dpart;
    const float lse = qok ? p.lse[((long)n * p.H + hd) * Lq + q] : 0.f;
    if (g == 0 && qok) p.delta[((long)n * p.H + hd) * Lq + q] = delta;
    floatx4v oacc[DC];
#pragma unroll
    for (int dt = 0; dt < DC; ++dt) oacc[dt] = floatx4v{0.f, 0.f, 0.f, 0.f};
    const unsigned long long base = (((unsigned long long)n * p.H + hd) * Lq + (qok ? q : 0)) * (unsigned long long)Lk;
    for (int kc0 = 0; kc0 < Lk; kc0 += SKC) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < SKC * V4; idx += 256) {
            const int j = idx / V4, c4 = idx - j * V4;
            float4 kk = make_float4(0.f, 0.f, 0.f, 0.f), vv = kk;
            if (kc0 + j < Lk) {
                const long row = (long)kvn * p.kv_is + kc0 + j;
                kk = *reinterpret_cast<const float4*>(p.k + row * p.ldk + hd * DH + c4 * 4);
                vv = *reinterpret_cast<const float4*>(p.v + row * p.ldv + hd * DH + c4 * 4);
            }
            *reinterpret_cast<float4*>(Ks + j * LDK + c4 * 4) = kk;
            *reinterpret_cast<float4*>(Vs + j * LDK + c4 * 4) = vv;
        }
        __syncthreads();
        if (!wave_on) continue;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            if (kc0 + kt * 16 >= Lk) continue;
            floatx4v a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < DC; ++ch) {
                const float4 kf = *reinterpret_cast<const float4*>(Ks + (kt * 16 + c) * LDK + ch * 16 + g * 4);
                const float4 vf = *reinterpret_cast<const float4*>(Vs + (kt * 16 + c) * LDK + ch * 16 + g * 4);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[ch].x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.x, gf[ch].x, a2, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[ch].y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.y, gf[ch].y, a2, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[ch].z, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.z, gf[ch].z, a2, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[ch].w, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf.w, gf[ch].w, a2, 0, 0, 0);
            }
            floatx4v ds;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int key = kc0 + kt * 16 + g * 4 + j;
                const float pr = (qok && key < Lk) ? __expf(a1[j] - lse) : 0.f;
                const float mult = (p.drop_p > 0.f && key < Lk) ? qt_drop_mult(p.seed, base + key, p.drop_p, p.inv_keep) : 1.0f;
                ds[j] = pr * (a2[j] * mult - delta);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* krow = Ks + (kt * 16 + g * 4 + j) * LDK + c;
#pragma unroll
                for (int dt = 0; dt < DC; ++dt)
                    oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[j], krow[dt * 16], oacc[dt], 0, 0, 0);
            }
        }
    }
    if (!wave_on) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int qo = q0 + g * 4 + j;
        if (qo < Lq) {
            float* drow = p.dq + ((long)n * p.dq_is + (long)qo * p.dq_rs) * p.lddq + hd * DH + c;
#pragma unroll
            for (int dt = 0; dt < DC; ++dt) drow[dt * 16] = oacc[dt][j] * p.scale;
        }
    }
}

// dK/dV per 16-key tile; streams chunks of (scaled) Q and dO of every query item that attends to
// this kv item.  Orientation S[q][key] = Q·K^T so that P (rows = queries on (g,j), column = key on
// the lane) is directly the A operand of dV = P^T·dO and dK = dS^T·Q.
template <int DC>
__global__ __launch_bounds__(256) void qt_attn_bwd_dkv_kernel(AttnArgs p) {
    if (p.salt && p.drop_p > 0.f) p.seed += *p.salt;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int DH = DC * 16, LDK = DH + 4, V4 = DH / 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int kvn = blockIdx.x / p.H, hd = blockIdx.x - kvn * p.H;
    const int Lq = p.Lq, Lk = p.Lk;
    const int k0 = (blockIdx.y * 4 + wave) * 16;
    const bool wave_on = k0 < Lk;
    const int key = k0 + c;
    const bool kok = wave_on && key < Lk;
    float* Qs = lds;
    float* Gs = Qs + SKC * LDK;
    float4 kf[DC], vf[DC];
#pragma unroll
    for (int ch = 0; ch < DC; ++ch) {
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f), u = t;
        if (kok) {
            const long row = (long)kvn * p.kv_is + key;
            t = *reinterpret_cast<const float4*>(p.k + row * p.ldk + hd * DH + ch * 16 + g * 4);
            u = *reinterpret_cast<const float4*>(p.v + row * p.ldv + hd * DH + ch * 16 + g * 4);
        }
        kf[ch] = t; vf[ch] = u;
    }
    floatx4v av[DC], ak[DC];
#pragma unroll
    for (int dt = 0; dt < DC; ++dt) { av[dt] = floatx4v{0.f, 0.f, 0.f, 0.f}; ak[dt] = av[dt]; }
    int first = (kvn - p.kv_shift) % p.n_kv;
    if (first < 0) first += p.n_kv;
    const int reps = p.n_items / p.n_kv;
    for (int rep = 0; rep < reps; ++rep) {
        const int n = first + rep * p.n_kv;
        const long stat0 = ((long)n * p.H + hd) * Lq;
        for (int qc0 = 0; qc0 < Lq; qc0 += SKC) {
            __syncthreads();
            for (int idx = threadIdx.x; idx < SKC * V4; idx += 256) {
                const int qi = idx / V4, c4 = idx - qi * V4;
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f), u = t;
                if (qc0 + qi < Lq) {
                    t = *reinterpret_cast<const float4*>(p.q + ((long)n * p.q_is + (long)(qc0 + qi) * p.q_rs) * p.ldq + hd * DH + c4 * 4);
                    u = *reinterpret_cast<const float4*>(p.d_o + ((long)n * p.do_is + (long)(qc0 + qi) * p.do_rs) * p.lddo + hd * DH + c4 * 4);
                }
                *reinterpret_cast<float4*>(Qs + qi * LDK + c4 * 4) = make_float4(t.x * p.scale, t.y * p.scale, t.z * p.scale, t.w * p.scale);
                *reinterpret_cast<float4*>(Gs + qi * LDK + c4 * 4) = u;
            }
            __syncthreads();
            if (!wave_on) continue;
#pragma unroll
            for (int qt = 0; qt < 4; ++qt) {
                if (qc0 + qt * 16 >= Lq) continue;
                floatx4v a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ch = 0; ch < DC; ++ch) {
                    const float4 qv = *reinterpret_cast<const float4*>(Qs + (qt * 16 + c) * LDK + ch * 16 + g * 4);
                    const float4 gv = *reinterpret_cast<const float4*>(Gs + (qt * 16 + c) * LDK + ch * 16 + g * 4);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.x, kf[ch].x, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.x, vf[ch].x, a2, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.y, kf[ch].y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.y, vf[ch].y, a2, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.z, kf[ch].z, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.z, vf[ch].z, a2, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qv.w, kf[ch].w, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gv.w, vf[ch].w, a2, 0, 0, 0);
                }
                floatx4v pd, ds;                                    // element j <-> query qc0 + qt*16 + 4g + j, column = key (lane c)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qq = qc0 + qt * 16 + g * 4 + j;
                    const bool ok = kok && qq < Lq;
                    const float lse = ok ? p.lse[stat0 + qq] : 0.f;
                    const float dl = ok ? p.delta[stat0 + qq] : 0.f;
                    const float pr = ok ? __expf(a1[j] - lse) : 0.f;
                    const float mult = (p.drop_p > 0.f && ok)
                        ? qt_drop_mult(p.seed, (unsigned long long)(stat0 + qq) * (unsigned long long)Lk + key, p.drop_p, p.inv_keep) : 1.0f;
                    pd[j] = pr * mult;
                    ds[j] = pr * (a2[j] * mult - dl);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float* grow = Gs + (qt * 16 + g * 4 + j) * LDK + c;
                    const float* qrow = Qs + (qt * 16 + g * 4 + j) * LDK + c;
#pragma unroll
                    for (int dt = 0; dt < DC; ++dt) {
                        av[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(pd[j], grow[dt * 16], av[dt], 0, 0, 0);
                        ak[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[j], qrow[dt * 16], ak[dt], 0, 0, 0);
                    }
                }
            }
        }
    }
    if (!wave_on) return;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ko = k0 + g * 4 + j;
        if (ko < Lk) {
            const long row = (long)kvn * p.kv_is + ko;
            float* dkp = p.dk + row * p.lddk + hd * DH + c;
            float* dvp = p.dv + row * p.lddv + hd * DH + c;
#pragma unroll
            for (int dt = 0; dt < DC; ++dt) { dkp[dt * 16] = ak[dt][j]; dvp[dt * 16] = av[dt][j]; }
        }
    }
}

template <int DC>
static int launch_fwd_stream(const AttnArgs& a, hipStream_t st) {
    const size_t lds = sizeof(float) * 2 * SKC * (DC * 16 + 4);
    auto kern = qt_attn_fwd_stream_kernel<DC>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)a.n_items * a.H), (unsigned)(((a.Lq + 15) / 16 + 3) / 4)), dim3(256), lds, st, a);
    return 0;
}
template <int DC>
static int launch_bwd_stream(const AttnArgs& a, hipStream_t st) {
    const size_t lds = sizeof(float) * 2 * SKC * (DC * 16 + 4);
    auto k1 = qt_attn_bwd_dq_kernel<DC>;
    auto k2 = qt_attn_bwd_dkv_kernel<DC>;
    if (lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL(k1, dim3((unsigned)((long)a.n_items * a.H), (unsigned)(((a.Lq + 15) / 16 + 3) / 4)), dim3(256), lds, st, a);
    hipLaunchKernelGGL(k2, dim3((unsigned)((long)a.n_kv * a.H), (unsigned)(((a.Lk + 15) / 16 + 3) / 4)), dim3(256), lds, st, a);
    return 0;
}
#define QT_ATTN_STREAM_DISPATCH(fn, a, st, rc)                    \
    do {                                                          \
        const int dc_ = (a).dh / 16;                              \
        rc = -1;                                                  \
        if (dc_ == 2) rc = fn<2>(a, st); else if (dc_ == 4) rc = fn<4>(a, st); \
        else if (dc_ == 6) rc = fn<6>(a, st); else if (dc_ == 8) rc = fn<8>(a, st); \
    } while (0)

static int nkt_bucket(int Lk) { const int t = (Lk + 15) / 16; return t <= 1 ? 1 : (t <= 4 ? 4 : (t <= 5 ? 5 : 8)); }
// LDS bytes of qt_attn_bwd_mfma_kernel for this geometry (the same arithmetic as launch_bwd_mfma)
static size_t attn_bwd_mfma_lds(int Lq, int Lk, int dh) {
    const int DC = dh / 16, NKT = nkt_bucket(Lk);
    const int nqt = (Lq + 15) / 16;
    const int wpp = (nqt > 1 || NKT > 1) ? 4 : 1;
    const int ppb = 4 / wpp;
    const int LDK = DC * 16 + 4, QR = nqt * 16, LDQ = QR + 4;
    const size_t per = (size_t)(2 * (Lk > Lq ? Lk : Lq)) * LDK + (size_t)2 * Lk * LDQ;
    return sizeof(float) * per * ppb;
}
// ONE predicate decides the kernel family from what forward and backward both know (geometry and the q/k/v
// operands), so the two directions can never disagree: the MFMA forward writes no log-sum-exp, the streaming
// backward needs one.  A geometry whose MFMA BACKWARD does not fit the 160 KiB of LDS (d_h >= 96 with
// 81 <= Lk <= 128 and Lq > 16) therefore takes the streaming kernels in the forward too.
enum { QT_ATTN_MFMA = 0, QT_ATTN_STREAM = 1, QT_ATTN_GENERIC = 2 };
static int attn_path(const AttnArgs& a, bool have_lse) {
    const bool dh_ok = !(a.dh % 16 != 0 || a.dh < 32 || a.dh > 128 || a.dh == 80 || a.dh == 48 || a.dh == 112);
    const bool al = qt_aligned16(a.q) && qt_aligned16(a.k) && qt_aligned16(a.v) && a.ldq % 4 == 0 && a.ldk % 4 == 0 && a.ldv % 4 == 0;
    if (!dh_ok || !al) return QT_ATTN_GENERIC;
    if (a.Lk <= 128 && a.Lq <= 64 && attn_bwd_mfma_lds(a.Lq, a.Lk, a.dh) <= 160 * 1024) return QT_ATTN_MFMA;
    return have_lse ? QT_ATTN_STREAM : QT_ATTN_GENERIC;
}
// what the backward kernels need beyond the forward's conditions
static bool attn_bwd_extras_ok(const AttnArgs& a, bool stream) {
    if (a.n_items % a.n_kv != 0) return false;
    if (!(qt_aligned16(a.d_o) && a.lddo % 4 == 0)) return false;
    if (stream) return a.o_fwd && qt_aligned16(a.o_fwd) && a.delta && a.lse;
    // (the MFMA backward stores four consecutive columns per lane as one float4)
    return qt_aligned16(a.dq) && qt_aligned16(a.dk) && qt_aligned16(a.dv) && a.lddq % 4 == 0 && a.lddk % 4 == 0 && a.lddv % 4 == 0;
}

template <int DC, int NKT>
static int launch_fwd_mfma(const AttnArgs& a, hipStream_t st) {
    const int nqt = (a.Lq + 15) / 16;
    const int wpp = nqt >= 3 || NKT > 1 ? 4 : (nqt == 2 ? 2 : 1);
    const int ppb = 4 / wpp;
    const size_t lds = sizeof(float) * (size_t)ppb * 2 * (NKT * 16) * (DC * 16 + 4);
    const long probs = (long)a.n_items * a.H;
    const long blocks = (probs + ppb - 1) / ppb;
    auto kern = qt_attn_fwd_mfma_kernel<DC, NKT>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    AttnArgs b = a;
    b.vec_out = a.o && qt_aligned16(a.o) && a.ldo % 4 == 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, b, wpp);
    return 0;
}
template <int DC, int NKT>
static int launch_bwd_mfma(const AttnArgs& a, hipStream_t st) {
    const int nqt = (a.Lq + 15) / 16;
    const int wpp = (nqt > 1 || NKT > 1) ? 4 : 1;
    const int ppb = 4 / wpp;
    const int LDK = DC * 16 + 4, QR = nqt * 16, LDQ = QR + 4;
    const size_t per = (size_t)(2 * (a.Lk > a.Lq ? a.Lk : a.Lq)) * LDK + (size_t)2 * a.Lk * LDQ;
    const size_t lds = sizeof(float) * per * ppb;
    if (lds > 160 * 1024) return 1;
    const long probs = (long)a.n_kv * a.H;
    const long blocks = (probs + ppb - 1) / ppb;
    auto kern = qt_attn_bwd_mfma_kernel<DC, NKT>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, st, a, wpp, nqt);
    return 0;
}
#define QT_ATTN_DISPATCH(fn, a, st, rc)                                            \
    do {                                                                           \
        const int dc_ = (a).dh / 16, nk_ = nkt_bucket((a).Lk);                     \
        rc = -1;                                                                   \
        if (dc_ == 2) { if (nk_ == 1) rc = fn<2, 1>(a, st); else if (nk_ == 4) rc = fn<2, 4>(a, st); else if (nk_ == 5) rc = fn<2, 5>(a, st); else rc = fn<2, 8>(a, st); } \
        else if (dc_ == 4) { if (nk_ == 1) rc = fn<4, 1>(a, st); else if (nk_ == 4) rc = fn<4, 4>(a, st); else if (nk_ == 5) rc = fn<4, 5>(a, st); else rc = fn<4, 8>(a, st); } \
        else if (dc_ == 6) { if (nk_ == 1) rc = fn<6, 1>(a, st); else if (nk_ == 4) rc = fn<6, 4>(a, st); else if (nk_ == 5) rc = fn<6, 5>(a, st); else rc = fn<6, 8>(a, st); } \
        else if (dc_ == 8) { if (nk_ == 1) rc = fn<8, 1>(a, st); else if (nk_ == 4) rc = fn<8, 4>(a, st); else if (nk_ == 5) rc = fn<8, 5>(a, st); else rc = fn<8, 8>(a, st); } \
    } while (0)

static int attn_block(int Lq, int Lk, int dh) {
    const long work = (long)Lq * Lk * dh;
    return work <= 14L * 14 * 128 ? 64 : 256;
}

static int attn_fwd_impl(void* stream, const float* q, long ldq, long q_is, long q_rs,
                         const float* k, long ldk, const float* v, long ldv, long kv_is,
                         float* o, long ldo, long o_is, long o_rs,
                         int n_items, int n_kv, int kv_shift, int H, int Lq, int Lk, int dh,
                         float scale, float drop_p, unsigned long long seed, float* lse, int causal,
                         void* o_img = nullptr, long ld_img = 0) {
    QT_CHECK_ARG(q && k && v && o, "qt_attn_fwd: null pointer");
    QT_CHECK_ARG(n_items > 0 && n_kv > 0 && H > 0 && Lq > 0 && Lk > 0 && dh > 0, "qt_attn_fwd: bad sizes");
    QT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "qt_attn_fwd: drop_p=%f", drop_p);
    AttnArgs a; memset(&a, 0, sizeof(a));
    a.q = q; a.ldq = ldq; a.q_is = q_is; a.q_rs = q_rs; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.kv_is = kv_is;
    a.o = o; a.ldo = ldo; a.o_is = o_is; a.o_rs = o_rs;
    a.n_items = n_items; a.n_kv = n_kv; a.kv_shift = kv_shift; a.H = H; a.Lq = Lq; a.Lk = Lk; a.dh = dh;
    a.scale = scale; a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.seed = seed; a.salt = qt_salt_ptr; a.lse = lse; a.causal = causal;
    // causal (text tower, forward only) always streams K/V with an online softmax and needs no log-sum-exp
    int path = attn_path(a, causal || lse != nullptr);
    if (causal && path == QT_ATTN_MFMA) path = QT_ATTN_STREAM;
    if (path == QT_ATTN_MFMA) {
        int rc;
        const bool img = o_img != nullptr && (H * dh) % 32 == 0 && dh % 2 == 0 && ld_img % 128 == 0 && (((uintptr_t)o_img) & 15u) == 0;
        if (img) { a.o_img = (char*)o_img; a.ld_img = ld_img; }
        QT_ATTN_DISPATCH(launch_fwd_mfma, a, (hipStream_t)stream, rc);
        if (rc == 0) { QT_CHECK_LAUNCH("qt_attn_fwd(mfma)"); return img ? 1 : QT_OK; }
        QT_FAIL(QT_ERR_RUNTIME, "qt_attn_fwd: no MFMA kernel instance for d_h=%d", dh);
    }
    if (path == QT_ATTN_STREAM) {
        int rc;
        QT_ATTN_STREAM_DISPATCH(launch_fwd_stream, a, (hipStream_t)stream, rc);
        if (rc == 0) { QT_CHECK_LAUNCH("qt_attn_fwd(stream)"); return QT_OK; }
        QT_FAIL(QT_ERR_RUNTIME, "qt_attn_fwd: no streaming kernel instance for d_h=%d", dh);
    }
    const int dp = dh + 1;
    const size_t lds = sizeof(float) * ((size_t)Lq * dp + (size_t)Lk * dp + (size_t)Lk * dh + (size_t)Lq * (Lk + 1));
    if (lds > 160 * 1024)
        QT_FAIL(QT_ERR_UNSUPPORTED, "qt_attn_fwd: Lq=%d Lk=%d dh=%d needs %zu B of LDS (>160 KiB); "
                "the streaming kernels need d_h in {32,64,96,128}, 16-byte aligned rows and an lse buffer", Lq, Lk, dh, lds);
    hipStream_t st = (hipStream_t)stream;
    const long blocks = (long)n_items * H;
    QT_CHECK_ARG(blocks < (1L << 31), "qt_attn_fwd: grid too large");
    if (attn_block(Lq, Lk, dh) == 64) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)qt_attn_fwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(qt_attn_fwd_kernel<64>, dim3((unsigned)blocks), dim3(64), lds, st, a);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)qt_attn_fwd_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(qt_attn_fwd_kernel<256>, dim3((unsigned)blocks), dim3(256), lds, st, a);
    }
    QT_CHECK_LAUNCH("qt_attn_fwd");
    return QT_OK;
}

extern "C" int qt_attn_fwd(void* stream, const float* q, long ldq, long q_is, long q_rs,
                           const float* k, long ldk, const float* v, long ldv, long kv_is,
                           float* o, long ldo, long o_is, long o_rs,
                           int n_items, int n_kv, int kv_shift, int H, int Lq, int Lk, int dh,
                           float scale, float drop_p, unsigned long long seed, float* lse) {
    return attn_fwd_impl(stream, q, ldq, q_is, q_rs, k, ldk, v, ldv, kv_is, o, ldo, o_is, o_rs,
                         n_items, n_kv, kv_shift, H, Lq, Lk, dh, scale, drop_p, seed, lse, 0);
}

// The same, with the option of writing o as a split image (include/qatiger.h qt_split_f16 format, unscaled) for the out-projection GEMM
// that consumes it: returns 1 when the image was written INSTEAD of the fp32 rows (the MFMA kernels: all shapes of the path), 0 when the
// fp32 rows were written as by qt_attn_fwd (other kernels; the caller then splits them as before).
extern "C" int qt_attn_fwd_img(void* stream, const float* q, long ldq, long q_is, long q_rs,
                               const float* k, long ldk, const float* v, long ldv, long kv_is,
                               float* o, long ldo, long o_is, long o_rs,
                               int n_items, int n_kv, int kv_shift, int H, int Lq, int Lk, int dh,
                               float scale, float drop_p, unsigned long long seed, float* lse, void* o_img, long ld_img_bytes) {
    return attn_fwd_impl(stream, q, ldq, q_is, q_rs, k, ldk, v, ldv, kv_is, o, ldo, o_is, o_rs,
                         n_items, n_kv, kv_shift, H, Lq, Lk, dh, scale, drop_p, seed, lse, 0, o_img, ld_img_bytes);
}

// Causal self-attention, forward only: the frozen CLIP text tower (reference
// src/models/base/clip_base.py:211-214 with the additive -inf mask of :390-396).
extern "C" int qt_attn_causal_fwd(void* stream, const float* q, long ldq, long q_is, long q_rs,
                                  const float* k, long ldk, const float* v, long ldv, long kv_is,
                                  float* o, long ldo, long o_is, long o_rs,
                                  int n_items, int H, int L, int dh, float scale) {
    return attn_fwd_impl(stream, q, ldq, q_is, q_rs, k, ldk, v, ldv, kv_is, o, ldo, o_is, o_rs,
                         n_items, n_items, 0, H, L, L, dh, scale, 0.f, 0ull, nullptr, 1);
}

extern "C" int qt_attn_bwd(void* stream, const float* q, long ldq, long q_is, long q_rs,
                           const float* k, long ldk, const float* v, long ldv, long kv_is,
                           const float* d_o, long lddo, long do_is, long do_rs,
                           float* dq, long lddq, long dq_is, long dq_rs,
                           float* dk, long lddk, float* dv, long lddv,
                           int n_items, int n_kv, int kv_shift, int H, int Lq, int Lk, int dh,
                           float scale, float drop_p, unsigned long long seed,
                           const float* o_fwd, const float* lse, float* delta_ws, float* g_amax) {
    QT_CHECK_ARG(q && k && v && d_o && dq && dk && dv, "qt_attn_bwd: null pointer");
    QT_CHECK_ARG(n_items > 0 && n_kv > 0 && H > 0 && Lq > 0 && Lk > 0 && dh > 0, "qt_attn_bwd: bad sizes");
    QT_CHECK_ARG(n_kv <= n_items, "qt_attn_bwd: n_kv=%d > n_items=%d leaves kv rows without a gradient", n_kv, n_items);
    AttnArgs a; memset(&a, 0, sizeof(a));
    a.q = q; a.ldq = ldq; a.q_is = q_is; a.q_rs = q_rs; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.kv_is = kv_is;
    a.d_o = d_o; a.lddo = lddo; a.do_is = do_is; a.do_rs = do_rs;
    a.dq = dq; a.lddq = lddq; a.dq_is = dq_is; a.dq_rs = dq_rs; a.dk = dk; a.lddk = lddk; a.dv = dv; a.lddv = lddv;
    a.n_items = n_items; a.n_kv = n_kv; a.kv_shift = kv_shift; a.H = H; a.Lq = Lq; a.Lk = Lk; a.dh = dh;
    a.scale = scale; a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.seed = seed; a.salt = qt_salt_ptr;
    a.o_fwd = o_fwd; a.lse = (float*)lse; a.delta = delta_ws;
    a.g_amax = nullptr;
    // the family the FORWARD took for this geometry (attn_path); a backward whose extra conditions fail falls to the
    // generic kernel, which recomputes the softmax itself and needs no log-sum-exp, never to another family
    const int path = attn_path(a, lse != nullptr);
    if (path == QT_ATTN_MFMA && attn_bwd_extras_ok(a, false)) {
        int rc;
        if (g_amax) {                                   // only this family measures max|dq,dk,dv| while writing them: return 1 = recorded
            a.g_amax = g_amax;                          // (a zeroed record from the caller)
        }
        QT_ATTN_DISPATCH(launch_bwd_mfma, a, (hipStream_t)stream, rc);
        if (rc == 0) { QT_CHECK_LAUNCH("qt_attn_bwd(mfma)"); return g_amax ? 1 : QT_OK; }
        QT_FAIL(QT_ERR_RUNTIME, "qt_attn_bwd: MFMA kernel refused Lq=%d Lk=%d d_h=%d (rc %d)", Lq, Lk, dh, rc);
    }
    if (path == QT_ATTN_STREAM && attn_bwd_extras_ok(a, true)) {
        int rc;
        QT_ATTN_STREAM_DISPATCH(launch_bwd_stream, a, (hipStream_t)stream, rc);
        if (rc == 0) { QT_CHECK_LAUNCH("qt_attn_bwd(stream)"); return QT_OK; }
        QT_FAIL(QT_ERR_RUNTIME, "qt_attn_bwd: no streaming kernel instance for d_h=%d", dh);
    }
    const int dp = dh + 1;
    const size_t lds = sizeof(float) * (2 * (size_t)Lq * dp + 2 * (size_t)Lk * dp + 2 * (size_t)Lq * (Lk + 1));
    if (lds > 160 * 1024)
        QT_FAIL(QT_ERR_UNSUPPORTED, "qt_attn_bwd: Lq=%d Lk=%d dh=%d needs %zu B of LDS (>160 KiB); "
                "the streaming kernels need d_h in {32,64,96,128}, 16-byte aligned rows and an lse buffer", Lq, Lk, dh, lds);
    hipStream_t st = (hipStream_t)stream;
    const long blocks = (long)n_kv * H;
    QT_CHECK_ARG(blocks < (1L << 31), "qt_attn_bwd: grid too large");
    if (attn_block(Lq, Lk, dh) == 64) {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)qt_attn_bwd_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(qt_attn_bwd_kernel<64>, dim3((unsigned)blocks), dim3(64), lds, st, a);
    } else {
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)qt_attn_bwd_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(qt_attn_bwd_kernel<256>, dim3((unsigned)blocks), dim3(256), lds, st, a);
    }
    QT_CHECK_LAUNCH("qt_attn_bwd");
    return QT_OK;
}
