// Fused  y = LayerNorm( sum_i  s_i * dropout_i(x_i) )  forward and backward (fp32).
//
// Replaces the residual + dropout + nn.LayerNorm groups of the path (reference
// src/models/modules.py:223-229, 233-235 AVQCrossAttn; 344-346 QstGrounding incl.
// its `data.mean(dim=0)` as two addends scaled 0.5; 603, 612 TempMoE; 713-716
// PatchSelecter), eps = 1e-5 (torch default).
//
// HBM-bound: one wavefront per row, 16-byte coalesced loads of up to four addend
// rows, the row kept in registers, mean / variance by wavefront shuffles, one
// write of y plus the normalised row (x̂) and 1/σ that the backward needs.
#include "qt_common.h"
#include <stdlib.h>

#define LN_MAXV 4          // float4 per lane  => D <= 1024
#define LN_MAX_ADD 4

struct LnArgs {
    int rows, D;
    const float* x[LN_MAX_ADD]; long ldx[LN_MAX_ADD]; float sx[LN_MAX_ADD];
    int drop[LN_MAX_ADD]; unsigned long long seed[LN_MAX_ADD]; const unsigned long long* salt;
    float drop_p, inv_keep, eps;
    const float* gamma; const float* beta;
    float* y; long ldy; int relu_out;
    float* xhat; float* rstd;            // [rows, D], [rows]  (forward: written; backward: read)
    // backward
    const float* dy; long lddy;
    float* dx[LN_MAX_ADD]; long lddx[LN_MAX_ADD]; int dx_acc[LN_MAX_ADD];
    float* dx_amax[LN_MAX_ADD];          // nullable per addend: max|dx_a| record (the producing Linear layer's dY scale), zeroed by the caller
    float* dgamma; float* dbeta;
};

__global__ __launch_bounds__(256) void qt_add_ln_fwd_kernel(LnArgs p) {
    if (p.salt && p.drop_p > 0.f) {
        const unsigned long long sl = *p.salt;
#pragma unroll
        for (int a = 0; a < LN_MAX_ADD; ++a) p.seed[a] += sl;
    }
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int nv = p.D >> 2;                              // float4 per row
    for (int row = wave_global; row < p.rows; row += nwaves) {
        float4 v[LN_MAXV];
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < LN_MAXV; ++t) {
            const int c4 = lane + t * 64;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c4 < nv) {
#pragma unroll
                for (int a = 0; a < LN_MAX_ADD; ++a) {
                    if (!p.x[a]) continue;
                    const float4 xv = *reinterpret_cast<const float4*>(p.x[a] + (long)row * p.ldx[a] + c4 * 4);
                    float m0 = p.sx[a], m1 = m0, m2 = m0, m3 = m0;
                    if (p.drop[a]) {
                        const unsigned long long base = (unsigned long long)row * p.D + c4 * 4;
                        m0 *= qt_drop_mult(p.seed[a], base + 0, p.drop_p, p.inv_keep);
                        m1 *= qt_drop_mult(p.seed[a], base + 1, p.drop_p, p.inv_keep);
                        m2 *= qt_drop_mult(p.seed[a], base + 2, p.drop_p, p.inv_keep);
                        m3 *= qt_drop_mult(p.seed[a], base + 3, p.drop_p, p.inv_keep);
                    }
                    acc.x = fmaf(xv.x, m0, acc.x); acc.y = fmaf(xv.y, m1, acc.y);
                    acc.z = fmaf(xv.z, m2, acc.z); acc.w = fmaf(xv.w, m3, acc.w);
                }
                sum += acc.x + acc.y + acc.z + acc.w;
            }
            v[t] = acc;
        }
        const float mean = qt_wave_sum(sum) / (float)p.D;
        float sq = 0.f;
#pragma unroll
        for (int t = 0; t < LN_MAXV; ++t) {
            if (lane + t * 64 < nv) {
                v[t].x -= mean; v[t].y -= mean; v[t].z -= mean; v[t].w -= mean;
                sq += v[t].x * v[t].x + v[t].y * v[t].y + v[t].z * v[t].z + v[t].w * v[t].w;
            }
        }
        const float rstd = rsqrtf(qt_wave_sum(sq) / (float)p.D + p.eps);
        if (lane == 0 && p.rstd) p.rstd[row] = rstd;
#pragma unroll
        for (int t = 0; t < LN_MAXV; ++t) {
            const int c4 = lane + t * 64;
            if (c4 < nv) {
                const float4 g = *reinterpret_cast<const float4*>(p.gamma + c4 * 4);
                const float4 b = *reinterpret_cast<const float4*>(p.beta + c4 * 4);
                float4 xh = make_float4(v[t].x * rstd, v[t].y * rstd, v[t].z * rstd, v[t].w * rstd);
                float4 o = make_float4(fmaf(xh.x, g.x, b.x), fmaf(xh.y, g.y, b.y), fmaf(xh.z, g.z, b.z), fmaf(xh.w, g.w, b.w));
                if (p.relu_out) { o.x = qt_relu(o.x); o.y = qt_relu(o.y); o.z = qt_relu(o.z); o.w = qt_relu(o.w); }
                *reinterpret_cast<float4*>(p.y + (long)row * p.ldy + c4 * 4) = o;
                if (p.xhat) *reinterpret_cast<float4*>(p.xhat + (long)row * p.D + c4 * 4) = xh;
            }
        }
    }
}

// Backward.  dy is the gradient wrt the (pre-ReLU) LayerNorm output.
//   ds = rstd * (g·dy - mean(g·dy) - x̂ · mean(g·dy·x̂));   dx_i = s_i * mask_i * ds
// dgamma/dbeta: each wave accumulates its rows in registers, then one atomicAdd
// per column per wave (dgamma/dbeta are zeroed by the launcher).
__global__ __launch_bounds__(256) void qt_add_ln_bwd_kernel(LnArgs p) {
    if (p.salt && p.drop_p > 0.f) {
        const unsigned long long sl = *p.salt;
#pragma unroll
        for (int a = 0; a < LN_MAX_ADD; ++a) p.seed[a] += sl;
    }
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;
    const int nv = p.D >> 2;
    float4 ag[LN_MAXV], ab[LN_MAXV];
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) { ag[t] = make_float4(0.f, 0.f, 0.f, 0.f); ab[t] = ag[t]; }
    float gmx[LN_MAX_ADD];
#pragma unroll
    for (int a = 0; a < LN_MAX_ADD; ++a) gmx[a] = 0.f;
    // gamma stays in registers; the NEXT row's dy / x̂ are requested before the current row is reduced (a wave owns ~4 rows of a
    // 30720-row launch: without this every row exposed a full load latency behind two wave reductions)
    float4 gm[LN_MAXV], nd[LN_MAXV], nx[LN_MAXV];
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        const int c4 = lane + t * 64;
        gm[t] = c4 < nv ? *reinterpret_cast<const float4*>(p.gamma + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        nd[t] = nx[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < nv && wave_global < p.rows) {
            nd[t] = *reinterpret_cast<const float4*>(p.dy + (long)wave_global * p.lddy + c4 * 4);
            nx[t] = *reinterpret_cast<const float4*>(p.xhat + (long)wave_global * p.D + c4 * 4);
        }
    }
    for (int row = wave_global; row < p.rows; row += nwaves) {
        float4 dxh[LN_MAXV], xh[LN_MAXV], dcur[LN_MAXV];
#pragma unroll
        for (int t = 0; t < LN_MAXV; ++t) { dcur[t] = nd[t]; xh[t] = nx[t]; }
        const int nrow = row + nwaves;
        if (nrow < p.rows) {
#pragma unroll
            for (int t = 0; t < LN_MAXV; ++t) {
                const int c4 = lane + t * 64;
                if (c4 < nv) {
                    nd[t] = *reinterpret_cast<const float4*>(p.dy + (long)nrow * p.lddy + c4 * 4);
                    nx[t] = *reinterpret_cast<const float4*>(p.xhat + (long)nrow * p.D + c4 * 4);
                }
            }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int t = 0; t < LN_MAXV; ++t) {
            const int c4 = lane + t * 64;
            if (c4 < nv) {
                const float4 d = dcur[t];
                const float4 g = gm[t];
                dxh[t] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
                s1 += dxh[t].x + dxh[t].y + dxh[t].z + dxh[t].w;
                s2 += dxh[t].x * xh[t].x + dxh[t].y * xh[t].y + dxh[t].z * xh[t].z + dxh[t].w * xh[t].w;
                ag[t].x = fmaf(d.x, xh[t].x, ag[t].x); ag[t].y = fmaf(d.y, xh[t].y, ag[t].y);
                ag[t].z = fmaf(d.z, xh[t].z, ag[t].z); ag[t].w = fmaf(d.w, xh[t].w, ag[t].w);
                ab[t].x += d.x; ab[t].y += d.y; ab[t].z += d.z; ab[t].w += d.w;
            }
        }
        const float m1 = qt_wave_sum(s1) / (float)p.D, m2 = qt_wave_sum(s2) / (float)p.D;
        const float rstd = p.rstd[row];
#pragma unroll
        for (int t = 0; t < LN_MAXV; ++t) {
            const int c4 = lane + t * 64;
            if (c4 < nv) {
                const float4 ds = make_float4(rstd * (dxh[t].x - m1 - xh[t].x * m2), rstd * (dxh[t].y - m1 - xh[t].y * m2),
                                              rstd * (dxh[t].z - m1 - xh[t].z * m2), rstd * (dxh[t].w - m1 - xh[t].w * m2));
#pragma unroll
                for (int a = 0; a < LN_MAX_ADD; ++a) {
                    if (!p.dx[a]) continue;
                    float m0 = p.sx[a], mm1 = m0, mm2 = m0, mm3 = m0;
                    if (p.drop[a]) {
                        const unsigned long long base = (unsigned long long)row * p.D + c4 * 4;
                        m0 *= qt_drop_mult(p.seed[a], base + 0, p.drop_p, p.inv_keep);
                        mm1 *= qt_drop_mult(p.seed[a], base + 1, p.drop_p, p.inv_keep);
                        mm2 *= qt_drop_mult(p.seed[a], base + 2, p.drop_p, p.inv_keep);
                        mm3 *= qt_drop_mult(p.seed[a], base + 3, p.drop_p, p.inv_keep);
                    }
                    float4* dst = reinterpret_cast<float4*>(p.dx[a] + (long)row * p.lddx[a] + c4 * 4);
                    float4 o = make_float4(ds.x * m0, ds.y * mm1, ds.z * mm2, ds.w * mm3);
                    if (p.dx_acc[a]) { const float4 old = *dst; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
                    gmx[a] = fmaxf(fmaxf(gmx[a], fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                    *dst = o;
                }
            }
        }
    }
    // the block's four waves meet in LDS first, so the chip sends one global atomic per column per BLOCK
    __shared__ float red[2 * 256 * LN_MAXV];
    for (int c = threadIdx.x; c < 2 * p.D; c += 256) red[c] = 0.f;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < LN_MAXV; ++t) {
        const int c4 = lane + t * 64;
        if (c4 < nv) {
            float* dg = red + c4 * 4; float* db = red + p.D + c4 * 4;
            atomicAdd(dg + 0, ag[t].x); atomicAdd(dg + 1, ag[t].y); atomicAdd(dg + 2, ag[t].z); atomicAdd(dg + 3, ag[t].w);
            atomicAdd(db + 0, ab[t].x); atomicAdd(db + 1, ab[t].y); atomicAdd(db + 2, ab[t].z); atomicAdd(db + 3, ab[t].w);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < p.D; c += 256) {
        atomicAdd(p.dgamma + c, red[c]);
        atomicAdd(p.dbeta + c, red[p.D + c]);
    }
#pragma unroll
    for (int a = 0; a < LN_MAX_ADD; ++a) {
        if (p.dx[a] && p.dx_amax[a]) {                              // block-uniform
            __syncthreads();
            qt_amax_fold(gmx[a], red, p.dx_amax[a]);
        }
    }
}

static int ln_check(const char* who, int rows, int D) {
    QT_CHECK_ARG(rows > 0 && D > 0, "%s: empty", who);
    QT_CHECK_ARG(D % 4 == 0 && D <= 256 * LN_MAXV, "%s: D=%d must be a multiple of 4 and <= %d", who, D, 256 * LN_MAXV);
    return QT_OK;
}

extern "C" int qt_add_ln_fwd(void* stream, int rows, int D, int n_add,
                             const float* const* x, const long* ldx, const float* sx, const int* drop,
                             const unsigned long long* seed, float drop_p,
                             const float* gamma, const float* beta, float eps,
                             float* y, long ldy, int relu_out, float* xhat, float* rstd) {
    if (int rc = ln_check("qt_add_ln_fwd", rows, D)) return rc;
    QT_CHECK_ARG(n_add >= 1 && n_add <= LN_MAX_ADD, "qt_add_ln_fwd: n_add=%d", n_add);
    QT_CHECK_ARG(gamma && beta && y, "qt_add_ln_fwd: null pointer");
    LnArgs a; memset(&a, 0, sizeof(a));
    a.rows = rows; a.D = D; a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.eps = eps; a.salt = qt_salt_ptr;
    for (int i = 0; i < n_add; ++i) {
        QT_CHECK_ARG(x[i] && qt_aligned16(x[i]) && ldx[i] % 4 == 0, "qt_add_ln_fwd: addend %d must be 16-byte aligned with ld%%4==0", i);
        a.x[i] = x[i]; a.ldx[i] = ldx[i]; a.sx[i] = sx[i]; a.drop[i] = (drop[i] && drop_p > 0.f); a.seed[i] = seed[i];
    }
    QT_CHECK_ARG(qt_aligned16(y) && ldy % 4 == 0 && qt_aligned16(gamma) && qt_aligned16(beta), "qt_add_ln_fwd: alignment");
    a.gamma = gamma; a.beta = beta; a.y = y; a.ldy = ldy; a.relu_out = relu_out; a.xhat = xhat; a.rstd = rstd;
    const int blocks = (int)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
    hipLaunchKernelGGL(qt_add_ln_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    QT_CHECK_LAUNCH("qt_add_ln_fwd");
    return QT_OK;
}

extern "C" int qt_add_ln_bwd(void* stream, int rows, int D, int n_add,
                             const float* dy, long lddy, const float* xhat, const float* rstd, const float* gamma,
                             const float* sx, const int* drop, const unsigned long long* seed, float drop_p,
                             float* const* dx, const long* lddx, const int* dx_acc,
                             float* dgamma, float* dbeta, float* const* dx_amax, int accumulate_params) {
    if (int rc = ln_check("qt_add_ln_bwd", rows, D)) return rc;
    QT_CHECK_ARG(n_add >= 1 && n_add <= LN_MAX_ADD, "qt_add_ln_bwd: n_add=%d", n_add);
    QT_CHECK_ARG(dy && xhat && rstd && gamma && dgamma && dbeta, "qt_add_ln_bwd: null pointer");
    QT_CHECK_ARG(qt_aligned16(dy) && lddy % 4 == 0, "qt_add_ln_bwd: dy alignment");
    LnArgs a; memset(&a, 0, sizeof(a));
    a.rows = rows; a.D = D; a.drop_p = drop_p; a.inv_keep = 1.0f / (1.0f - drop_p); a.salt = qt_salt_ptr;
    a.dy = dy; a.lddy = lddy; a.xhat = (float*)xhat; a.rstd = (float*)rstd; a.gamma = gamma; a.dgamma = dgamma; a.dbeta = dbeta;
    for (int i = 0; i < n_add; ++i) {
        a.sx[i] = sx[i]; a.drop[i] = (drop[i] && drop_p > 0.f); a.seed[i] = seed[i];
        a.dx[i] = dx[i]; a.lddx[i] = lddx[i]; a.dx_acc[i] = dx_acc[i]; a.dx_amax[i] = dx_amax ? dx_amax[i] : nullptr;
        if (dx[i]) QT_CHECK_ARG(qt_aligned16(dx[i]) && lddx[i] % 4 == 0, "qt_add_ln_bwd: dx %d alignment", i);
    }
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (accumulate_params) {}                                        // dgamma / dbeta hold earlier contributions (a pre-zeroed gradient buffer): add to them
    else if (dbeta == dgamma + D) e = qt_zero_1d(dgamma, 2L * D, st);     // one launch when the two live side by side (ops.py allocates them so)
    else { e = qt_zero_1d(dgamma, D, st); if (e == hipSuccess) e = qt_zero_1d(dbeta, D, st); }
    if (e != hipSuccess) QT_FAIL(QT_ERR_RUNTIME, "qt_add_ln_bwd: memset: %s", hipGetErrorString(e));
    // 512 blocks x 4 waves (8 waves per CU, each prefetching its next row): every block ends with 2 D global atomics into the SAME 2 D
    // addresses, and same-address atomics serialise in the L2 — with 2048 blocks that tail was ~80 us of every launch, whatever its size
    // (15360 x 768, one addend: 107 us; 512 blocks: 44 us; `tools/ln_blocks_sweep.sh`).  QT_LN_BLOCKS overrides (A/B).
    int cap = 512;
    { const char* e = getenv("QT_LN_BLOCKS"); if (e && atoi(e) > 0) cap = atoi(e); }
    const int blocks = (int)((rows + 3) / 4 < cap ? (rows + 3) / 4 : cap);
    hipLaunchKernelGGL(qt_add_ln_bwd_kernel, dim3(blocks), dim3(256), 0, st, a);
    QT_CHECK_LAUNCH("qt_add_ln_bwd");
    return QT_OK;
}
