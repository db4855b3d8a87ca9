// Small HBM-bound helpers of the train step (fp32):
//   qt_colsum          bias gradients  db[n] = Σ_m dY[m,n]              (backward of every nn.Linear bias)
//   qt_add_rows        out[r,:] = a[r % ra,:] + b[r,:]                   (TempMoE `data + sub_data`, reference modules.py:587,596)
//   qt_xent_fwd_bwd    nn.CrossEntropyLoss(mean) + its gradient          (reference train.py:41, trainutils.py:278)
//   qt_adam_step       torch.optim.Adam on one flat buffer               (reference trainutils.py:141-144)
//   qt_version / qt_last_error
#include "qt_common.h"
#include <math.h>

thread_local char qt_err_buf[512] = "";
const unsigned long long* qt_salt_ptr = nullptr;

extern "C" int qt_version(void) { return QT_ABI_VERSION; }
extern "C" const char* qt_last_error(void) { return qt_err_buf; }

// ---- dropout salt (see qt_common.h) and the device-side step counter of the captured train step ---------------------
extern "C" int qt_set_dropout_salt(const unsigned long long* salt_dev) { qt_salt_ptr = salt_dev; return QT_OK; }
static __global__ void qt_bump_salt_kernel(unsigned long long* salt, unsigned long long inc) { if (threadIdx.x == 0) *salt += inc; }
extern "C" int qt_bump_dropout_salt(void* stream, unsigned long long* salt_dev, unsigned long long inc) {
    QT_CHECK_ARG(salt_dev, "qt_bump_dropout_salt: null pointer");
    hipLaunchKernelGGL(qt_bump_salt_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, salt_dev, inc);
    QT_CHECK_LAUNCH("qt_bump_dropout_salt");
    return QT_OK;
}

// zero n floats (max|x| record pools, ...): an ordinary kernel, capturable like everything else here
extern "C" int qt_zero_f32(void* stream, float* p, long n) {
    QT_CHECK_ARG(p && n > 0, "qt_zero_f32: bad arguments");
    hipError_t e = qt_zero_1d(p, n, (hipStream_t)stream);
    if (e != hipSuccess) QT_FAIL(QT_ERR_RUNTIME, "qt_zero_f32: %s", hipGetErrorString(e));
    return QT_OK;
}

// ---- column sums ---------------------------------------------------------------
// grid = (ceil(N/256), row_chunks); each thread owns one column of its chunk:
// consecutive lanes read consecutive columns (coalesced), partials meet in atomics.
__global__ __launch_bounds__(256) void qt_colsum_kernel(const float* __restrict__ x, long ldx, int M, int N,
                                                        int rows_per_chunk, float* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int m_beg = blockIdx.y * rows_per_chunk;
    const int m_end = min(M, m_beg + rows_per_chunk);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int m = m_beg;
    for (; m + 3 < m_end; m += 4) {
        s0 += x[(long)(m + 0) * ldx + n]; s1 += x[(long)(m + 1) * ldx + n];
        s2 += x[(long)(m + 2) * ldx + n]; s3 += x[(long)(m + 3) * ldx + n];
    }
    for (; m < m_end; ++m) s0 += x[(long)m * ldx + n];
    atomicAdd(out + n, (s0 + s1) + (s2 + s3));
}

extern "C" int qt_colsum(void* stream, const float* x, long ldx, int M, int N, float* out, int accumulate) {
    QT_CHECK_ARG(x && out && M > 0 && N > 0 && ldx >= N, "qt_colsum: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = qt_zero_1d(out, N, st);
        if (e != hipSuccess) QT_FAIL(QT_ERR_RUNTIME, "qt_colsum: memset: %s", hipGetErrorString(e));
    }
    const int col_blocks = (N + 255) / 256;
    int chunks = (2048 + col_blocks - 1) / col_blocks;        // ~2048 workgroups in flight
    if (chunks > (M + 63) / 64) chunks = (M + 63) / 64;
    if (chunks < 1) chunks = 1;
    const int rpc = (M + chunks - 1) / chunks;
    chunks = (M + rpc - 1) / rpc;
    hipLaunchKernelGGL(qt_colsum_kernel, dim3(col_blocks, chunks), dim3(256), 0, st, x, ldx, M, N, rpc, out);
    QT_CHECK_LAUNCH("qt_colsum");
    return QT_OK;
}

// ---- out[r,:] = alpha*a[r % ra,:] + beta*b[r % rb,:] ----------------------------------
__global__ __launch_bounds__(256) void qt_add_rows_kernel(const float* __restrict__ a, long lda, int ra, float alpha,
                                                          const float* __restrict__ b, long ldb, int rb, float beta,
                                                          float* __restrict__ out, long ldo, long rows, int nv) {
    const long total = rows * nv;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / nv; const int c4 = (int)(idx - r * nv);
        const float4 x = *reinterpret_cast<const float4*>(a + (r % ra) * lda + c4 * 4);
        const float4 y = *reinterpret_cast<const float4*>(b + (r % rb) * ldb + c4 * 4);
        float4 o = make_float4(alpha * x.x + beta * y.x, alpha * x.y + beta * y.y, alpha * x.z + beta * y.z, alpha * x.w + beta * y.w);
        *reinterpret_cast<float4*>(out + r * ldo + c4 * 4) = o;
    }
}

extern "C" int qt_add_rows(void* stream, const float* a, long lda, int ra, float alpha,
                           const float* b, long ldb, int rb, float beta,
                           float* out, long ldo, long rows, int D) {
    QT_CHECK_ARG(a && b && out && rows > 0 && D > 0 && ra > 0 && rb > 0, "qt_add_rows: bad arguments");
    QT_CHECK_ARG(D % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldo % 4 == 0 && qt_aligned16(a) && qt_aligned16(b) && qt_aligned16(out),
                 "qt_add_rows: rows must be 16-byte aligned (D=%d)", D);
    const int nv = D / 4;
    long blocks = (rows * nv + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(qt_add_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       a, lda, ra, alpha, b, ldb, rb, beta, out, ldo, rows, nv);
    QT_CHECK_LAUNCH("qt_add_rows");
    return QT_OK;
}

// ---- gradient masking for y = dropout(relu(z)) / dropout(z) --------------------------------
// y given : out = g * (y > 0 ? inv_keep : 0)      (y>0 <=> relu passed AND element kept)
// y null  : out = g * dropout_multiplier(seed, r*N + c)   (same index space as the forward epilogue)
__global__ __launch_bounds__(256) void qt_mask_grad_kernel(const float* __restrict__ g, long ldg, const float* __restrict__ y, long ldy,
                                                           float drop_p, float inv_keep, unsigned long long seed,
                                                           float* __restrict__ out, long ldo, long rows, int N, float* __restrict__ amax,
                                                           const unsigned long long* __restrict__ salt) {
    if (salt) seed += *salt;
    const long total = rows * N;
    float gm = 0.f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / N; const int c = (int)(idx - r * N);
        float m;
        if (y) m = y[r * ldy + c] > 0.f ? inv_keep : 0.f;
        else   m = qt_drop_mult(seed, (unsigned long long)idx, drop_p, inv_keep);
        const float v = g[r * ldg + c] * m;
        gm = fmaxf(gm, fabsf(v));
        out[r * ldo + c] = v;
    }
    if (amax) {                                                     // max|out| record for the layer's backward GEMMs (zeroed by the caller)
        __shared__ float red[4];
        qt_amax_fold(gm, red, amax);
    }
}

extern "C" int qt_mask_grad(void* stream, const float* g, long ldg, const float* y, long ldy, float drop_p,
                            unsigned long long seed, float* out, long ldo, long rows, int N, float* amax) {
    QT_CHECK_ARG(g && out && rows > 0 && N > 0, "qt_mask_grad: bad arguments");
    QT_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "qt_mask_grad: drop_p=%f", drop_p);
    QT_CHECK_ARG(y || drop_p > 0.f, "qt_mask_grad: nothing to mask");
    long blocks = (rows * N + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(qt_mask_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       g, ldg, y, ldy, drop_p, 1.0f / (1.0f - drop_p), seed, out, ldo, rows, N, amax, qt_salt_ptr);
    QT_CHECK_LAUNCH("qt_mask_grad");
    return QT_OK;
}

// ---- softmax cross-entropy, mean reduction: loss and dlogits in one pass ---------------
// one wavefront per sample (A = 42 classes fits one wave; larger A strides)
__global__ __launch_bounds__(256) void qt_xent_kernel(const float* __restrict__ logits, long ld, const long long* __restrict__ label,
                                                      int B, int A, float grad_scale, float* __restrict__ loss_rows,
                                                      float* __restrict__ dlogits, long ldd) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* row = logits + (long)b * ld;
    float mx = -INFINITY;
    for (int a = lane; a < A; a += 64) mx = fmaxf(mx, row[a]);
    mx = qt_wave_max(mx);
    float sum = 0.f;
    for (int a = lane; a < A; a += 64) sum += expf(row[a] - mx);
    sum = qt_wave_sum(sum);
    const int y = (int)label[b];
    if (lane == 0) loss_rows[b] = (y >= 0 && y < A) ? (logf(sum) + mx - row[y]) : NAN;   // out-of-range label: loud NaN
    if (dlogits) {
        const float inv = 1.0f / sum;
        for (int a = lane; a < A; a += 64)
            dlogits[(long)b * ldd + a] = grad_scale * (expf(row[a] - mx) * inv - (a == y ? 1.f : 0.f));
    }
}
// fixed-order mean of the per-sample losses (single wave; B is a few hundred)
__global__ __launch_bounds__(64) void qt_mean_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) s += x[i];
    s = qt_wave_sum(s);
    if (threadIdx.x == 0) out[0] = s / (float)n;
}

extern "C" int qt_xent_fwd_bwd(void* stream, const float* logits, long ld, const long long* label, int B, int A,
                               float* loss_rows, float* loss_mean, float* dlogits, long ldd) {
    QT_CHECK_ARG(logits && label && loss_rows && loss_mean && B > 0 && A > 0, "qt_xent_fwd_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(qt_xent_kernel, dim3((B + 3) / 4), dim3(256), 0, st, logits, ld, label, B, A, 1.0f / (float)B,
                       loss_rows, dlogits, ldd);
    QT_CHECK_LAUNCH("qt_xent");
    hipLaunchKernelGGL(qt_mean_kernel, dim3(1), dim3(64), 0, st, loss_rows, B, loss_mean);
    QT_CHECK_LAUNCH("qt_mean");
    return QT_OK;
}

// ---- Adam on a flat buffer (L2 weight decay added to the gradient, bias-corrected exactly as torch.optim.Adam) --
__global__ __launch_bounds__(256) void qt_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                      float* __restrict__ v, long n, float lr, float b1, float b2, float eps,
                                                      float bc1, float bc2_sqrt, float grad_scale, float wd) {
    const long nv = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
#define QT_ADAM1(c)                                                              \
        { const float gr = gg.c * grad_scale + wd * pp.c;                                   \
          mm.c = b1 * mm.c + (1.f - b1) * gr;                                      \
          vv.c = b2 * vv.c + (1.f - b2) * gr * gr;                                 \
          pp.c -= (lr / bc1) * mm.c / (sqrtf(vv.c) / bc2_sqrt + eps); }
        QT_ADAM1(x) QT_ADAM1(y) QT_ADAM1(z) QT_ADAM1(w)
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {          // tail
        const long i = (nv << 2) + threadIdx.x;
        const float gr = g[i] * grad_scale + wd * p[i];
        m[i] = b1 * m[i] + (1.f - b1) * gr;
        v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
        p[i] -= (lr / bc1) * m[i] / (sqrtf(v[i]) / bc2_sqrt + eps);
    }
}

// Device-side optimizer state for a CAPTURED train step.  Kernel arguments are frozen inside a HIP graph, so everything that changes from
// step to step lives in memory: step_dev[0] (the step count) and hp_dev[0..7] = { 1 - b1^t, sqrt(1 - b2^t), lr, b1, b2, eps, weight_decay,
// grad_scale }.  qt_adam_set_dev (launched OUTSIDE the graph, before a replay) writes the count and hp[2..7] from the host's values, so a
// learning-rate scheduler or a loaded optimizer state reaches every replay; the captured qt_adam_step_dev increments the count, derives
// the two bias corrections and updates the parameters reading all of it from memory.
#define QT_HP_BC1 0
#define QT_HP_BC2 1
#define QT_HP_LR 2
#define QT_HP_B1 3
#define QT_HP_B2 4
#define QT_HP_EPS 5
#define QT_HP_WD 6
#define QT_HP_GS 7
static __global__ void qt_adam_set_kernel(long long* step_dev, float* hp, long long step, float lr, float b1, float b2, float eps, float wd, float gs) {
    if (threadIdx.x == 0) {
        *step_dev = step;
        hp[QT_HP_LR] = lr; hp[QT_HP_B1] = b1; hp[QT_HP_B2] = b2; hp[QT_HP_EPS] = eps; hp[QT_HP_WD] = wd; hp[QT_HP_GS] = gs;
    }
}
extern "C" int qt_adam_set_dev(void* stream, long long* step_dev, float* hp_dev, long long step, float lr, float beta1, float beta2, float eps,
                               float weight_decay, float grad_scale) {
    QT_CHECK_ARG(step_dev && hp_dev && step >= 0, "qt_adam_set_dev: bad arguments");
    hipLaunchKernelGGL(qt_adam_set_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_dev, hp_dev, step, lr, beta1, beta2, eps, weight_decay, grad_scale);
    QT_CHECK_LAUNCH("qt_adam_set_dev");
    return QT_OK;
}
static __global__ void qt_adam_prep_kernel(long long* step_dev, float* hp) {
    if (threadIdx.x == 0) {
        const long long t = ++(*step_dev);
        hp[QT_HP_BC1] = (float)(1.0 - pow((double)hp[QT_HP_B1], (double)t));
        hp[QT_HP_BC2] = (float)sqrt(1.0 - pow((double)hp[QT_HP_B2], (double)t));
    }
}
__global__ __launch_bounds__(256) void qt_adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, long n, const float* __restrict__ hp) {
    const float bc1 = hp[QT_HP_BC1], bc2_sqrt = hp[QT_HP_BC2], lr = hp[QT_HP_LR], b1 = hp[QT_HP_B1], b2 = hp[QT_HP_B2], eps = hp[QT_HP_EPS],
                wd = hp[QT_HP_WD], grad_scale = hp[QT_HP_GS];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float gr = g[i] * grad_scale + wd * p[i];
        const float mm = b1 * m[i] + (1.f - b1) * gr, vv = b2 * v[i] + (1.f - b2) * gr * gr;
        m[i] = mm; v[i] = vv;
        p[i] -= (lr / bc1) * mm / (sqrtf(vv) / bc2_sqrt + eps);
    }
}
extern "C" int qt_adam_step_dev(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                                long long* step_dev, float* hp_dev) {
    QT_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step_dev && hp_dev, "qt_adam_step_dev: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(qt_adam_prep_kernel, dim3(1), dim3(64), 0, st, step_dev, hp_dev);
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(qt_adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, st, param, grad, exp_avg, exp_avg_sq, n, hp_dev);
    QT_CHECK_LAUNCH("qt_adam_step_dev");
    return QT_OK;
}

extern "C" int qt_adam_step(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                            int step, float lr, float beta1, float beta2, float eps, float grad_scale, float weight_decay) {
    QT_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1, "qt_adam_step: bad arguments");
    QT_CHECK_ARG(qt_aligned16(param) && qt_aligned16(grad) && qt_aligned16(exp_avg) && qt_aligned16(exp_avg_sq),
                 "qt_adam_step: flat buffers must be 16-byte aligned");
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    long blocks = ((n >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(qt_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale, weight_decay);
    QT_CHECK_LAUNCH("qt_adam_step");
    return QT_OK;
}

// ---- value-bias term of a folded attention (ops.folded_attention; reference nn.MultiheadAttention in-projection bias b_v) ------------
// With W_v applied AFTER the attention-weighted sum, the bias contributes  out[m, e] = o[m, e] + asum[m, head(e)] * b_v[e]  where
// asum[m, h] is the sum of head h's (dropped) attention probabilities of query row m (1 without dropout).  Forward: one elementwise pass;
// backward: d asum[m, h] = sum_{e in head h} dOut[m, e] b_v[e]  and  d b_v[e] += sum_m dOut[m, e] asum[m, head(e)]  (d o = dOut itself).
// (Round 2 did this as a K = H GEMM on a masked [E, H] weight built with ATen ops: three tiny-K GEMM launches and ~10 glue launches.)
__global__ __launch_bounds__(256) void qt_headbias_fwd_kernel(const float* __restrict__ o, long ldo, const float* __restrict__ asum, const float* __restrict__ bv,
                                                              long rows, int E, int dh, int H, float* __restrict__ out, long ldout) {
    const long n4 = rows * (E >> 2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long m = i / (E >> 2); const int e = (int)(i - m * (E >> 2)) * 4;
        const float4 x = *reinterpret_cast<const float4*>(o + m * ldo + e);
        const float4 b = *reinterpret_cast<const float4*>(bv + e);
        const float a = asum[m * H + e / dh];                                    // dh % 4 == 0: the four columns share a head
        *reinterpret_cast<float4*>(out + m * ldout + e) = make_float4(fmaf(a, b.x, x.x), fmaf(a, b.y, x.y), fmaf(a, b.z, x.z), fmaf(a, b.w, x.w));
    }
}
extern "C" int qt_headbias_fwd(void* stream, const float* o, long ldo, const float* asum, const float* bv, long rows, int E, int H,
                               float* out, long ldout) {
    QT_CHECK_ARG(o && asum && bv && out && rows > 0 && E > 0 && H > 0 && E % H == 0 && (E / H) % 4 == 0 && ldo % 4 == 0 && ldout % 4 == 0,
                 "qt_headbias_fwd: bad arguments");
    QT_CHECK_ARG(qt_aligned16(o) && qt_aligned16(bv) && qt_aligned16(out), "qt_headbias_fwd: 16-byte alignment");
    long blocks = (rows * (E >> 2) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(qt_headbias_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, o, ldo, asum, bv, rows, E, E / H, H, out, ldout);
    QT_CHECK_LAUNCH("qt_headbias_fwd");
    return QT_OK;
}
// One WAVE per row at a time (4 waves per workgroup, QT_HB_ROWS rows per workgroup): a lane owns the float4 columns 4 lane + 256 c of the
// row (coalesced 1 KiB loads), keeps its share of d b_v in registers over its rows, and the row's per-head sums are formed from the
// lanes' partials through a per-wave LDS row (a head is 24 consecutive float4 slots when d_h = 96: no power-of-two segment) — no LDS
// atomics (a first version let 96 threads hit one LDS address per head and row: 87 us per launch at 30720 x 768).  The four waves'
// column sums meet in LDS, then one global atomic per column per workgroup into d b_v.
#define QT_HB_ROWS 64
#define QT_HB_MAXC 8            // 64-float4 chunks per row: E <= 2048
__global__ __launch_bounds__(256) void qt_headbias_bwd_kernel(const float* __restrict__ dout, long ld, const float* __restrict__ asum, const float* __restrict__ bv,
                                                              long rows, int E, int dh, int H, float* __restrict__ dasum, float* __restrict__ dbv) {
    __shared__ float part[4][QT_HB_MAXC * 64];                                  // per wave: one partial per float4 slot of the row
    __shared__ float4 colsum[4][QT_HB_MAXC * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.x * QT_HB_ROWS;
    const int nr = (int)min((long)QT_HB_ROWS, rows - r0);
    const int nslot = E >> 2, spf = dh >> 2;                                    // float4 slots per row / per head
    float4 acc[QT_HB_MAXC], b4[QT_HB_MAXC];
    int hs[QT_HB_MAXC];
#pragma unroll
    for (int c = 0; c < QT_HB_MAXC; ++c) {
        const int slot = lane + c * 64;
        acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        b4[c] = slot < nslot ? *reinterpret_cast<const float4*>(bv + slot * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        hs[c] = slot < nslot ? slot / spf : 0;
    }
    for (int r = wave; r < nr; r += 4) {
        const float* drow = dout + (r0 + r) * ld;
        const float* arow = asum + (r0 + r) * H;
#pragma unroll
        for (int c = 0; c < QT_HB_MAXC; ++c) {
            const int slot = lane + c * 64;
            float pv = 0.f;
            if (slot < nslot) {
                const float4 d = *reinterpret_cast<const float4*>(drow + slot * 4);
                const float a = arow[hs[c]];
                acc[c].x = fmaf(d.x, a, acc[c].x); acc[c].y = fmaf(d.y, a, acc[c].y); acc[c].z = fmaf(d.z, a, acc[c].z); acc[c].w = fmaf(d.w, a, acc[c].w);
                pv = d.x * b4[c].x + d.y * b4[c].y + d.z * b4[c].z + d.w * b4[c].w;
            }
            part[wave][slot] = pv;
        }
        __builtin_amdgcn_wave_barrier();                                        // the wave's own LDS row: its LDS operations execute in order
        if (lane < H) {
            float s = 0.f;
            for (int i = 0; i < spf; ++i) s += part[wave][lane * spf + i];
            dasum[(r0 + r) * H + lane] = s;
        }
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int c = 0; c < QT_HB_MAXC; ++c) colsum[wave][lane + c * 64] = acc[c];
    __syncthreads();
    for (int slot = threadIdx.x; slot < nslot; slot += 256) {
        const float4 a = colsum[0][slot], b = colsum[1][slot], c2 = colsum[2][slot], d = colsum[3][slot];
        atomicAdd(dbv + slot * 4 + 0, (a.x + b.x) + (c2.x + d.x)); atomicAdd(dbv + slot * 4 + 1, (a.y + b.y) + (c2.y + d.y));
        atomicAdd(dbv + slot * 4 + 2, (a.z + b.z) + (c2.z + d.z)); atomicAdd(dbv + slot * 4 + 3, (a.w + b.w) + (c2.w + d.w));
    }
}
extern "C" int qt_headbias_bwd(void* stream, const float* dout, long ld, const float* asum, const float* bv, long rows, int E, int H,
                               float* dasum, float* dbv, int dbv_accumulate) {
    QT_CHECK_ARG(dout && asum && bv && dasum && dbv && rows > 0 && E > 0 && H > 0 && E % H == 0 && (E / H) % 4 == 0 && H <= 64 && E <= 256 * QT_HB_MAXC &&
                 ld % 4 == 0 && qt_aligned16(dout) && qt_aligned16(bv), "qt_headbias_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (!dbv_accumulate) {
        hipError_t e = qt_zero_1d(dbv, E, st);
        if (e != hipSuccess) QT_FAIL(QT_ERR_RUNTIME, "qt_headbias_bwd: zeroing: %s", hipGetErrorString(e));
    }
    const long blocks = (rows + QT_HB_ROWS - 1) / QT_HB_ROWS;
    QT_CHECK_ARG(blocks < (1L << 31), "qt_headbias_bwd: too many rows");
    hipLaunchKernelGGL(qt_headbias_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dout, ld, asum, bv, rows, E, E / H, H, dasum, dbv);
    QT_CHECK_LAUNCH("qt_headbias_bwd");
    return QT_OK;
}

// ---- feature shards stored as fp16: widen to the fp32 the path computes in -----------
// (the reference casts on the host, src/trainutils.py:223-250: `.float()` after `.to(device)`)
__global__ __launch_bounds__(256) void qt_cast_f16_f32_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, long n8, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n8) {                                                   // 8 halves (16 B) in, 2 x 16 B out
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        const h8 v = *reinterpret_cast<const h8*>(src + i * 8);
        float4 a = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
        float4 b = make_float4((float)v[4], (float)v[5], (float)v[6], (float)v[7]);
        *reinterpret_cast<float4*>(dst + i * 8) = a;
        *reinterpret_cast<float4*>(dst + i * 8 + 4) = b;
    }
    if (i == 0) for (long j = n8 * 8; j < n; ++j) dst[j] = (float)src[j];   // tail (< 8 elements)
}

extern "C" int qt_cast_f16_f32(void* stream, const void* src_f16, float* dst, long n) {
    QT_CHECK_ARG(src_f16 && dst && n > 0, "qt_cast_f16_f32: bad arguments");
    QT_CHECK_ARG(qt_aligned16(src_f16) && qt_aligned16(dst), "qt_cast_f16_f32: 16-byte alignment");
    const long n8 = n / 8;
    const long blocks = (n8 + 255) / 256 > 0 ? (n8 + 255) / 256 : 1;
    QT_CHECK_ARG(blocks < (1L << 31), "qt_cast_f16_f32: too large");
    hipLaunchKernelGGL(qt_cast_f16_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16*>(src_f16), dst, n8, n);
    QT_CHECK_LAUNCH("qt_cast_f16_f32");
    return QT_OK;
}
