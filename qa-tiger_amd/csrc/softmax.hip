// Row softmax with dropout, forward and backward, and head-averaged attention weights (fp32, HBM/latency-bound).
// Used by the attention path built from batched GEMMs for head dimensions the fused attention kernels do not
// cover (nn.MultiheadAttention(512, nhead=1) in the reference's TSPM model, src/models/tspm.py:20-21, 194), and for
// the averaged weights nn.MultiheadAttention returns by default, which TSPM turns into top-K time indices
// (src/models/tspm.py:99-100, 118-125).
#include "qt_common.h"
#include <math.h>

// one wave per row; element (r, j) uses dropout index r*L + j
__global__ __launch_bounds__(256) void qt_softmax_fwd_kernel(const float* __restrict__ s, long lds_, long rows, int L, float scale,
                                                             float drop_p, float inv_keep, unsigned long long seed,
                                                             float* __restrict__ p, long ldp, float* __restrict__ y, long ldy,
                                                             const unsigned long long* __restrict__ salt) {
    if (salt) seed += *salt;
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* sr = s + r * lds_;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) mx = fmaxf(mx, sr[j] * scale);
    mx = qt_wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) sum += __expf(sr[j] * scale - mx);
    sum = qt_wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < L; j += 64) {
        const float pr = __expf(sr[j] * scale - mx) * inv;
        if (p) p[r * ldp + j] = pr;
        const float m = drop_p > 0.f ? qt_drop_mult(seed, (unsigned long long)r * L + j, drop_p, inv_keep) : 1.0f;
        y[r * ldy + j] = pr * m;
    }
}

// ds = scale * p * (dp - sum_j dp_j p_j),  dp = dy * dropout multiplier
__global__ __launch_bounds__(256) void qt_softmax_bwd_kernel(const float* __restrict__ dy, long lddy, const float* __restrict__ p, long ldp,
                                                             long rows, int L, float scale, float drop_p, float inv_keep,
                                                             unsigned long long seed, float* __restrict__ ds, long ldds,
                                                             const unsigned long long* __restrict__ salt) {
    if (salt) seed += *salt;
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float dot = 0.f;
    for (int j = lane; j < L; j += 64) {
        const float m = drop_p > 0.f ? qt_drop_mult(seed, (unsigned long long)r * L + j, drop_p, inv_keep) : 1.0f;
        dot += dy[r * lddy + j] * m * p[r * ldp + j];
    }
    dot = qt_wave_sum(dot);
    for (int j = lane; j < L; j += 64) {
        const float m = drop_p > 0.f ? qt_drop_mult(seed, (unsigned long long)r * L + j, drop_p, inv_keep) : 1.0f;
        const float pr = p[r * ldp + j];
        ds[r * ldds + j] = scale * pr * (dy[r * lddy + j] * m - dot);
    }
}

extern "C" int qt_softmax_fwd(void* stream, const float* s, long lds_, long rows, int L, float scale, float drop_p,
                              unsigned long long seed, float* p, long ldp, float* y, long ldy) {
    QT_CHECK_ARG(s && y && rows > 0 && L > 0 && lds_ >= L && ldy >= L && (!p || ldp >= L), "qt_softmax_fwd: bad arguments");
    QT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "qt_softmax_fwd: drop_p=%f", drop_p);
    hipLaunchKernelGGL(qt_softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       s, lds_, rows, L, scale, drop_p, 1.0f / (1.0f - drop_p), seed, p, ldp, y, ldy, qt_salt_ptr);
    QT_CHECK_LAUNCH("qt_softmax_fwd");
    return QT_OK;
}

extern "C" int qt_softmax_bwd(void* stream, const float* dy, long lddy, const float* p, long ldp, long rows, int L, float scale,
                              float drop_p, unsigned long long seed, float* ds, long ldds) {
    QT_CHECK_ARG(dy && p && ds && rows > 0 && L > 0 && lddy >= L && ldp >= L && ldds >= L, "qt_softmax_bwd: bad arguments");
    QT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "qt_softmax_bwd: drop_p=%f", drop_p);
    hipLaunchKernelGGL(qt_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       dy, lddy, p, ldp, rows, L, scale, drop_p, 1.0f / (1.0f - drop_p), seed, ds, ldds, qt_salt_ptr);
    QT_CHECK_LAUNCH("qt_softmax_bwd");
    return QT_OK;
}

// out[n, i, j] = mean_h dropout(softmax_j(scale * q_h[n,i] . k_h[kvn,j])): one wave per (item, query row), heads in turn.
// Addressing and dropout indices as qt_attn_fwd (so the weights are those the attention kernel applied).
__global__ __launch_bounds__(256) void qt_attn_avg_weights_kernel(const float* __restrict__ q, long ldq, long q_is, long q_rs,
                                                                  const float* __restrict__ k, long ldk, long kv_is,
                                                                  int n_items, int n_kv, int kv_shift, int H, int Lq, int Lk, int dh,
                                                                  float scale, float drop_p, float inv_keep, unsigned long long seed,
                                                                  float* __restrict__ out, const unsigned long long* __restrict__ salt) {
    if (salt) seed += *salt;
    const int lane = threadIdx.x & 63;
    const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (long)n_items * Lq) return;
    const int n = (int)(wid / Lq), i = (int)(wid - (long)n * Lq);
    const int kvn = (n + kv_shift) % n_kv;
    const float* qrow = q + ((long)n * q_is + (long)i * q_rs) * ldq;
    float* orow = out + ((long)n * Lq + i) * Lk;
    for (int j = lane; j < Lk; j += 64) orow[j] = 0.f;
    for (int hd = 0; hd < H; ++hd) {
        float mx = -INFINITY;
        for (int j = lane; j < Lk; j += 64) {                      // scores: lanes over keys
            const float* kr = k + ((long)kvn * kv_is + j) * ldk + hd * dh;
            float s = 0.f;
            for (int c = 0; c < dh; ++c) s = fmaf(qrow[hd * dh + c] * scale, kr[c], s);
            mx = fmaxf(mx, s);
        }
        mx = qt_wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < Lk; j += 64) {
            const float* kr = k + ((long)kvn * kv_is + j) * ldk + hd * dh;
            float s = 0.f;
            for (int c = 0; c < dh; ++c) s = fmaf(qrow[hd * dh + c] * scale, kr[c], s);
            sum += __expf(s - mx);
        }
        sum = qt_wave_sum(sum);
        const unsigned long long base = (((unsigned long long)n * H + hd) * Lq + i) * (unsigned long long)Lk;
        for (int j = lane; j < Lk; j += 64) {
            const float* kr = k + ((long)kvn * kv_is + j) * ldk + hd * dh;
            float s = 0.f;
            for (int c = 0; c < dh; ++c) s = fmaf(qrow[hd * dh + c] * scale, kr[c], s);
            float pr = __expf(s - mx) / sum;
            if (drop_p > 0.f) pr *= qt_drop_mult(seed, base + j, drop_p, inv_keep);
            orow[j] += pr / (float)H;
        }
    }
}

extern "C" int qt_attn_avg_weights(void* stream, const float* q, long ldq, long q_is, long q_rs,
                                   const float* k, long ldk, long kv_is,
                                   int n_items, int n_kv, int kv_shift, int H, int Lq, int Lk, int dh,
                                   float scale, float drop_p, unsigned long long seed, float* out) {
    QT_CHECK_ARG(q && k && out && n_items > 0 && n_kv > 0 && H > 0 && Lq > 0 && Lk > 0 && dh > 0, "qt_attn_avg_weights: bad arguments");
    QT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "qt_attn_avg_weights: drop_p=%f", drop_p);
    const long waves = (long)n_items * Lq;
    hipLaunchKernelGGL(qt_attn_avg_weights_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       q, ldq, q_is, q_rs, k, ldk, kv_is, n_items, n_kv, kv_shift, H, Lq, Lk, dh, scale, drop_p,
                       1.0f / (1.0f - drop_p), seed, out, qt_salt_ptr);
    QT_CHECK_LAUNCH("qt_attn_avg_weights");
    return QT_OK;
}
