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
    int n_items, n_kv, kv_shift, H, Lq, Lk, dh;
    float scale, drop_p, inv_keep;
    unsigned long long seed;
    // backward only
    const float* d_o; long lddo, do_is, do_rs;
    float* dq; long lddq, dq_is, dq_rs;
    float* dk; long lddk;
    float* dv; long lddv;
};

__device__ inline int lds_pad(int dh) { return dh + 1; }   // odd row stride: conflict-free column walks

// Forward: grid = n_items * H, block = NT threads.
template <int NT>
__global__ __launch_bounds__(NT) void qt_attn_fwd_kernel(AttnArgs p) {
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
        S[i * sp + j] = s;
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

static int attn_block(int Lq, int Lk, int dh) {
    const long work = (long)Lq * Lk * dh;
    return work <= 14L * 14 * 128 ? 64 : 256;
}

extern "C" int qt_attn_fwd(void* stream, const float* q, long ldq, long q_is, long q_rs,
                           const float* k, long ldk, const float* v, long ldv, long kv_is,
                           float* o, long ldo, long o_is, long o_rs,
                           int n_items, int n_kv, int kv_shift, int H, int Lq, int Lk, int dh,
                           float scale, float drop_p, unsigned long long seed) {
    QT_CHECK_ARG(q && k && v && o, "qt_attn_fwd: null pointer");
    QT_CHECK_ARG(n_items > 0 && n_kv > 0 && H > 0 && Lq > 0 && Lk > 0 && dh > 0, "qt_attn_fwd: bad sizes");
    QT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "qt_attn_fwd: drop_p=%f", drop_p);
    AttnArgs a; memset(&a, 0, sizeof(a));
    a.q = q; a.ldq = ldq; a.q_is = q_is; a.q_rs = q_rs; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.kv_is = kv_is;
    a.o = o; a.ldo = ldo; a.o_is = o_is; a.o_rs = o_rs;
    a.n_items = n_items; a.n_kv = n_kv; a.kv_shift = kv_shift; a.H = H; a.Lq = Lq; a.Lk = Lk; a.dh = dh;
    a.scale = scale; a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.seed = seed;
    const int dp = dh + 1;
    const size_t lds = sizeof(float) * ((size_t)Lq * dp + (size_t)Lk * dp + (size_t)Lk * dh + (size_t)Lq * (Lk + 1));
    if (lds > 160 * 1024)
        QT_FAIL(QT_ERR_UNSUPPORTED, "qt_attn_fwd: Lq=%d Lk=%d dh=%d needs %zu B of LDS (>160 KiB); "
                "long-sequence tiling is not built yet", Lq, Lk, dh, lds);
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

extern "C" int qt_attn_bwd(void* stream, const float* q, long ldq, long q_is, long q_rs,
                           const float* k, long ldk, const float* v, long ldv, long kv_is,
                           const float* d_o, long lddo, long do_is, long do_rs,
                           float* dq, long lddq, long dq_is, long dq_rs,
                           float* dk, long lddk, float* dv, long lddv,
                           int n_items, int n_kv, int kv_shift, int H, int Lq, int Lk, int dh,
                           float scale, float drop_p, unsigned long long seed) {
    QT_CHECK_ARG(q && k && v && d_o && dq && dk && dv, "qt_attn_bwd: null pointer");
    QT_CHECK_ARG(n_items > 0 && n_kv > 0 && H > 0 && Lq > 0 && Lk > 0 && dh > 0, "qt_attn_bwd: bad sizes");
    QT_CHECK_ARG(n_kv <= n_items, "qt_attn_bwd: n_kv=%d > n_items=%d leaves kv rows without a gradient", n_kv, n_items);
    AttnArgs a; memset(&a, 0, sizeof(a));
    a.q = q; a.ldq = ldq; a.q_is = q_is; a.q_rs = q_rs; a.k = k; a.ldk = ldk; a.v = v; a.ldv = ldv; a.kv_is = kv_is;
    a.d_o = d_o; a.lddo = lddo; a.do_is = do_is; a.do_rs = do_rs;
    a.dq = dq; a.lddq = lddq; a.dq_is = dq_is; a.dq_rs = dq_rs; a.dk = dk; a.lddk = lddk; a.dv = dv; a.lddv = lddv;
    a.n_items = n_items; a.n_kv = n_kv; a.kv_shift = kv_shift; a.H = H; a.Lq = Lq; a.Lk = Lk; a.dh = dh;
    a.scale = scale; a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.seed = seed;
    const int dp = dh + 1;
    const size_t lds = sizeof(float) * (2 * (size_t)Lq * dp + 2 * (size_t)Lk * dp + 2 * (size_t)Lq * (Lk + 1));
    if (lds > 160 * 1024)
        QT_FAIL(QT_ERR_UNSUPPORTED, "qt_attn_bwd: Lq=%d Lk=%d dh=%d needs %zu B of LDS (>160 KiB); "
                "long-sequence tiling is not built yet", Lq, Lk, dh, lds);
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
