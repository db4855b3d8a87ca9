// Question-aware Gaussian-expert kernels (TempMoE), forward and backward (fp32).
//
// Replaces reference src/models/modules.py:565-580 (router softmax, sorted top-k,
// renormalisation, tanh/sigmoid Gaussian parameters), 420-471 `generate_gaussian`
// and 473-534 `get_output`.  Closed form checked against the reference:
// SURVEY.md Appendix A.
//
// The reference materialises all K expert outputs for every token as a
// (T,B,K,D) stack and gathers from it.  Here the expert second layer is pulled
// through the temporal sum (it is linear):
//     out[b] = Σ_e W2_e ( Σ_t c[b,t,e] · h_e[b,t] ) + b2_e Σ_t c[b,t,e]
// with the coefficient table
//     c[b,t,e] = Σ_i [ topk_inds[(b*T+t) % B, i] == e ] · topk_probs[b,i] · gauss[b,i,t]
// which reproduces the reference's row/index mismatch (`repeat(T,1)` against
// rows b*T+t, modules.py:495-502) exactly.
//
// Kernels:
//   qt_moe_router_fwd   one wavefront per sample: K-way softmax, rank-by-count
//                       sorted top-k (descending, first index wins ties),
//                       renormalise, Gaussian over T with discrete-peak
//                       normalisation (quirk Q6)
//   qt_moe_coef_fwd     c[B,T,K] and csum[B,K] (one thread per (b,t) row)
//   qt_moe_treduce_fwd  R[s,b,e,:] = Σ_t c[b,t,e] · H[(s*B+b)*T+t, e, :]   (HBM-bound stream over H)
//   qt_moe_treduce_bwd  dH = c · dR · [H>0],  dc = <dR, H>
//   qt_moe_router_bwd   dc (+ dcsum) -> d router_logits, d gauss_raw
#include "qt_common.h"

// sigma (reference TempMoE.__init__ `sigma`, modules.py:361, default 9) is a kernel argument
#define MOE_WFLOOR 0.09f    // reference modules.py:457
#define MOE_MAXK 64

__device__ inline float center_of(int e, int K) {           // linspace(margin, 1-margin, K), modules.py:405-407
    const float margin = 1.0f / (2.0f * K);
    return K > 1 ? margin + (float)e * ((1.0f - 2.0f * margin) / (float)(K - 1)) : margin;
}

// grid = ceil(B/4), block = 256 (4 waves, one sample each)
__global__ __launch_bounds__(256) void qt_moe_router_fwd_kernel(
    const float* __restrict__ logits, const float* __restrict__ raw, int B, int T, int K, int k, float sigma,
    float* __restrict__ probs, long long* __restrict__ topk_inds, float* __restrict__ topk_probs,
    float* __restrict__ gauss) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;                                      // wave-uniform exit
    const float lg = lane < K ? logits[(long)b * K + lane] : -INFINITY;
    const float mx = qt_wave_max(lg);
    const float ex = lane < K ? expf(lg - mx) : 0.f;
    const float pr = ex / qt_wave_sum(ex);
    if (lane < K) probs[(long)b * K + lane] = pr;
    // rank = number of experts that sort before this one (descending; lower index first on ties).  Non-finite router
    // outputs (an overflowed or NaN input upstream) must still give a PERMUTATION of the experts: the indices written below
    // address memory in qt_moe_coef_fwd / the backward, so NaNs are ranked as equal lowest values instead of comparing false
    // against everything (which left slots unwritten and sent garbage indices downstream: a GPU memory fault).
    const float key = (pr == pr) ? pr : -1.0f;
    int rank = 0;
    for (int e = 0; e < K; ++e) {
        const float pe = __shfl(key, e, 64);
        rank += (pe > key || (pe == key && e < lane)) ? 1 : 0;
    }
    const bool sel = lane < K && rank < k;
    const float denom = qt_wave_sum(sel ? pr : 0.f);
    if (sel) {
        topk_inds[(long)b * k + rank] = lane;
        topk_probs[(long)b * k + rank] = pr / denom;
    }
    const float margin = 1.0f / (2.0f * K);
    for (int i = 0; i < k; ++i) {
        // expert occupying slot i: the lane whose rank == i
        const unsigned long long m = __ballot(lane < K && rank == i);
        const int e = __ffsll((long long)m) - 1;
        const float u = raw[(long)b * 2 * K + 2 * e], w_raw = raw[(long)b * 2 * K + 2 * e + 1];
        const float c = fminf(fmaxf(center_of(e, K) + tanhf(u) * margin, 0.f), 1.f);
        const float sg = 1.0f / (1.0f + expf(-w_raw));
        const float w = fmaxf(sg, MOE_WFLOOR) / sigma;
        const float inv2w2 = 1.0f / (2.0f * w * w);
        float zmax = -INFINITY;
        for (int t = lane; t < T; t += 64) {
            const float tau = T > 1 ? (float)t / (float)(T - 1) : 0.f;
            const float d = tau - c;
            zmax = fmaxf(zmax, -d * d * inv2w2);
        }
        zmax = qt_wave_max(zmax);
        for (int t = lane; t < T; t += 64) {
            const float tau = T > 1 ? (float)t / (float)(T - 1) : 0.f;
            const float d = tau - c;
            gauss[((long)b * k + i) * T + t] = expf(-d * d * inv2w2 - zmax);
        }
    }
}

// one thread per (b,t): c row of K entries
__global__ __launch_bounds__(256) void qt_moe_coef_fwd_kernel(
    const long long* __restrict__ topk_inds, const float* __restrict__ topk_probs,
    const float* __restrict__ gauss, int B, int T, int K, int k,
    float* __restrict__ coef) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= (long)B * T) return;
    const int b = (int)(row / T), t = (int)(row - (long)b * T);
    const int src = (int)(row % B);                          // quirk Q1: index table row = (b*T+t) % B
    float* cr = coef + row * K;
    for (int e = 0; e < K; ++e) cr[e] = 0.f;
    for (int i = 0; i < k; ++i) {
        int e = (int)topk_inds[(long)src * k + i];
        e = e < 0 ? 0 : (e >= K ? K - 1 : e);               // indices come from device memory: never address outside the row
        const float val = topk_probs[(long)b * k + i] * gauss[((long)b * k + i) * T + t];
        cr[e] += val;                                        // experts of one sample are distinct: no aliasing
    }
}

// csum[b,e] = Σ_t c[b,t,e]  (fixed summation order: bitwise reproducible)
__global__ __launch_bounds__(256) void qt_moe_csum_kernel(const float* __restrict__ coef, int B, int T, int K,
                                                          float* __restrict__ csum) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * K) return;
    const int b = (int)(idx / K), e = (int)(idx - (long)b * K);
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += coef[((long)b * T + t) * K + e];
    csum[idx] = s;
}

// grid = (S*B, column chunks of 256 float4) blocks (one block per sample put a single workgroup on each CU: 1.8 TB/s); H [(S*B*T), K*Dh]; R [S*B, K*Dh]
__global__ __launch_bounds__(256) void qt_moe_treduce_fwd_kernel(
    const float* __restrict__ H, const float* __restrict__ coef, int B, int T, int K, int Dh,
    float* __restrict__ R) {
    const int sb = blockIdx.x, b = sb % B;
    const int nv = (K * Dh) >> 2, dv = Dh >> 2;
    const int c4 = blockIdx.y * 256 + threadIdx.x;
    if (c4 < nv) {
        const int e = c4 / dv;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
        for (int t = 0; t < T; ++t) {                        // (same summation order over t as before: bitwise the same result)
            const float w = coef[((long)b * T + t) * K + e];
            const float4 h = *reinterpret_cast<const float4*>(H + ((long)sb * T + t) * (long)(K * Dh) + c4 * 4);
            acc.x = fmaf(w, h.x, acc.x); acc.y = fmaf(w, h.y, acc.y); acc.z = fmaf(w, h.z, acc.z); acc.w = fmaf(w, h.w, acc.w);
        }
        *reinterpret_cast<float4*>(R + (long)sb * (K * Dh) + c4 * 4) = acc;
    }
}

// grid = (B, ceil(T / MOE_TRB_TPB)) blocks (both streams of a sample in one block so dc has one owner; a first
// version with one block per sample put 4 waves on each CU and ran at 2 TB/s).
#define MOE_TRB_TPB 8
// dH[(s*B+b)*T+t, e, :] = coef[b,t,e] * dR[s*B+b, e, :] * [H>0]
// dcoef[b,t,e] = Σ_s <dR[s*B+b,e,:], H[(s*B+b)*T+t, e, :]>
__global__ __launch_bounds__(256) void qt_moe_treduce_bwd_kernel(
    const float* __restrict__ H, const float* __restrict__ coef, const float* __restrict__ dR,
    int S, int B, int T, int K, int Dh, float* __restrict__ dH, float* __restrict__ dcoef, float* __restrict__ dh_amax) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int dv = Dh >> 2;
    float hmax = 0.f;                                        // max |dH| written by this thread (dh_amax: the expert layer's dY scale)
    // work item = (t, e): one wave each, lanes over Dh
    const int t0 = blockIdx.y * MOE_TRB_TPB, tn = min(T - t0, MOE_TRB_TPB);
    for (int item = wave; item < tn * K; item += 4) {
        const int t = t0 + item / K, e = item % K;
        const float w = coef[((long)b * T + t) * K + e];
        float dot = 0.f;
        for (int s = 0; s < S; ++s) {
            const long sb = (long)s * B + b;
            const float* hrow = H + (sb * T + t) * (long)(K * Dh) + (long)e * Dh;
            float* drow = dH + (sb * T + t) * (long)(K * Dh) + (long)e * Dh;
            const float* gr = dR + sb * (long)(K * Dh) + (long)e * Dh;
            for (int c4 = lane; c4 < dv; c4 += 64) {
                const float4 h = *reinterpret_cast<const float4*>(hrow + c4 * 4);
                const float4 g = *reinterpret_cast<const float4*>(gr + c4 * 4);
                dot += g.x * h.x + g.y * h.y + g.z * h.z + g.w * h.w;
                float4 o;
                o.x = h.x > 0.f ? w * g.x : 0.f; o.y = h.y > 0.f ? w * g.y : 0.f;
                o.z = h.z > 0.f ? w * g.z : 0.f; o.w = h.w > 0.f ? w * g.w : 0.f;
                hmax = fmaxf(fmaxf(hmax, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
                *reinterpret_cast<float4*>(drow + c4 * 4) = o;
            }
        }
        dot = qt_wave_sum(dot);
        if (lane == 0) dcoef[((long)b * T + t) * K + e] = dot;
    }
    if (dh_amax) {
        hmax = qt_wave_max(hmax);
        if (lane == 0 && (hmax > 0.f || hmax != hmax))
            atomicMax(reinterpret_cast<unsigned*>(dh_amax) + ((blockIdx.x + blockIdx.y + wave) & (QT_AMAX_SLOTS - 1)), __builtin_bit_cast(unsigned, hmax));
    }
}

// one wavefront per sample
__global__ __launch_bounds__(256) void qt_moe_router_bwd_kernel(
    const float* __restrict__ dcoef, const long long* __restrict__ topk_inds, const float* __restrict__ topk_probs,
    const float* __restrict__ gauss, const float* __restrict__ probs, const float* __restrict__ raw,
    const float* __restrict__ dcsum, int B, int T, int K, int k, float sigma, float* __restrict__ dlogits, float* __restrict__ draw) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const float margin = 1.0f / (2.0f * K);
    float du_e = 0.f, dv_e = 0.f;                            // lane e owns expert e; unselected experts stay exactly 0
    float dp_e = 0.f;                                        // d probs[e] held by lane e
    float sum_dtp_tp = 0.f;
    // first pass over slots: d topk_probs and the Gaussian parameter gradients
    float dtp_slot = 0.f;                                    // lane i (< k) keeps dtp[i]
    for (int i = 0; i < k; ++i) {
        const int e = (int)topk_inds[(long)b * k + i];
        const float tp = topk_probs[(long)b * k + i];
        const float u = raw[(long)b * 2 * K + 2 * e], w_raw = raw[(long)b * 2 * K + 2 * e + 1];
        const float th = tanhf(u);
        const float cc = center_of(e, K) + th * margin;
        const float c = fminf(fmaxf(cc, 0.f), 1.f);
        const float sg = 1.0f / (1.0f + expf(-w_raw));
        const float w = fmaxf(sg, MOE_WFLOOR) / sigma;
        float dtp = 0.f, sgg = 0.f, a1 = 0.f, a2 = 0.f;
        float zbest = -INFINITY; int tbest = 0x7fffffff;
        for (int t = lane; t < T; t += 64) {
            const long row = (long)b * T + t;
            const int es = (int)topk_inds[(row % B) * k + i];             // expert this coefficient landed on
            const float dcv = dcoef[row * K + es] + (dcsum ? dcsum[(long)b * K + es] : 0.f);   // csum = Σ_t coef
            const float g = gauss[((long)b * k + i) * T + t];
            dtp += dcv * g;
            const float gbar = dcv * tp;                                   // d gauss[b,i,t]
            const float tau = T > 1 ? (float)t / (float)(T - 1) : 0.f;
            const float d = tau - c;
            const float gg = gbar * g;
            sgg += gg;
            a1 += gg * d; a2 += gg * d * d;
            const float z = -d * d;                                        // argmax of z over t (same ordering)
            if (z > zbest) { zbest = z; tbest = t; }
        }
        dtp = qt_wave_sum(dtp); sgg = qt_wave_sum(sgg); a1 = qt_wave_sum(a1); a2 = qt_wave_sum(a2);
        // arg-max frame t* (first maximum, as torch.max): the peak-normalisation routes -Σ ḡ g to it
        const float zb = qt_wave_max(zbest);
        int cand = (zbest == zb) ? tbest : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
        const float tau_s = T > 1 ? (float)cand / (float)(T - 1) : 0.f;
        const float ds = tau_s - c;
        // dz_t = ḡ_t g_t - δ(t,t*) Σ ḡ g ;  ∂z/∂c = (τ-c)/w² ; ∂z/∂w = (τ-c)²/w³
        const float dc_center = (a1 - sgg * ds) / (w * w);
        const float dw = (a2 - sgg * ds * ds) / (w * w * w);
        const float d_off = (cc >= 0.f && cc <= 1.f) ? dc_center : 0.f;   // clamp gate (inclusive, as torch.clamp)
        const float d_sg = (sg >= MOE_WFLOOR) ? dw / sigma : 0.f;
        if (lane == e) {
            du_e = d_off * margin * (1.0f - th * th);
            dv_e = d_sg * sg * (1.0f - sg);
        }
        sum_dtp_tp += dtp * tp;
        if (lane == i) dtp_slot = dtp;
    }
    // renormalisation tp_i = p_{e_i} / S  ->  dp_{e_i} = (dtp_i - Σ_j dtp_j tp_j) / S
    float S = 0.f;
    for (int i = 0; i < k; ++i) S += probs[(long)b * K + (int)topk_inds[(long)b * k + i]];
    for (int i = 0; i < k; ++i) {
        const int e = (int)topk_inds[(long)b * k + i];
        const float dtp = __shfl(dtp_slot, i, 64);
        if (lane == e) dp_e = (dtp - sum_dtp_tp) / S;
    }
    // softmax backward over K
    const float pe = lane < K ? probs[(long)b * K + lane] : 0.f;
    const float dot = qt_wave_sum(dp_e * pe);
    if (lane < K) {
        dlogits[(long)b * K + lane] = pe * (dp_e - dot);
        draw[(long)b * 2 * K + 2 * lane] = du_e;
        draw[(long)b * 2 * K + 2 * lane + 1] = dv_e;
    }
}

extern "C" int qt_moe_router_fwd(void* stream, const float* logits, const float* gauss_raw, int B, int T, int K, int k, float sigma,
                                 float* probs, long long* topk_inds, float* topk_probs, float* gauss) {
    QT_CHECK_ARG(sigma > 0.f, "qt_moe_router_fwd: sigma=%f", sigma);
    QT_CHECK_ARG(logits && gauss_raw && probs && topk_inds && topk_probs && gauss, "qt_moe_router_fwd: null pointer");
    QT_CHECK_ARG(B > 0 && T > 0 && K > 0 && k > 0 && k <= K && K <= MOE_MAXK, "qt_moe_router_fwd: need 0<k<=K<=%d (K=%d k=%d)", MOE_MAXK, K, k);
    hipLaunchKernelGGL(qt_moe_router_fwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       logits, gauss_raw, B, T, K, k, sigma, probs, topk_inds, topk_probs, gauss);
    QT_CHECK_LAUNCH("qt_moe_router_fwd");
    return QT_OK;
}

extern "C" int qt_moe_coef_fwd(void* stream, const long long* topk_inds, const float* topk_probs, const float* gauss,
                               int B, int T, int K, int k, float* coef, float* csum) {
    QT_CHECK_ARG(topk_inds && topk_probs && gauss && coef && csum, "qt_moe_coef_fwd: null pointer");
    QT_CHECK_ARG(B > 0 && T > 0 && K > 0 && k > 0 && k <= K, "qt_moe_coef_fwd: bad sizes");
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)B * T;
    hipLaunchKernelGGL(qt_moe_coef_fwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st,
                       topk_inds, topk_probs, gauss, B, T, K, k, coef);
    QT_CHECK_LAUNCH("qt_moe_coef_fwd");
    hipLaunchKernelGGL(qt_moe_csum_kernel, dim3((unsigned)(((long)B * K + 255) / 256)), dim3(256), 0, st, coef, B, T, K, csum);
    QT_CHECK_LAUNCH("qt_moe_csum");
    return QT_OK;
}

extern "C" int qt_moe_treduce_fwd(void* stream, const float* H, const float* coef, int S, int B, int T, int K, int Dh,
                                  float* R) {
    QT_CHECK_ARG(H && coef && R, "qt_moe_treduce_fwd: null pointer");
    QT_CHECK_ARG(S > 0 && B > 0 && T > 0 && K > 0 && Dh > 0 && Dh % 4 == 0, "qt_moe_treduce_fwd: Dh=%d must be a multiple of 4", Dh);
    QT_CHECK_ARG(qt_aligned16(H) && qt_aligned16(R), "qt_moe_treduce_fwd: alignment");
    hipLaunchKernelGGL(qt_moe_treduce_fwd_kernel, dim3(S * B, ((K * Dh >> 2) + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, coef, B, T, K, Dh, R);
    QT_CHECK_LAUNCH("qt_moe_treduce_fwd");
    return QT_OK;
}

extern "C" int qt_moe_treduce_bwd(void* stream, const float* H, const float* coef, const float* dR,
                                  int S, int B, int T, int K, int Dh, float* dH, float* dcoef, float* dh_amax) {
    QT_CHECK_ARG(H && coef && dR && dH && dcoef, "qt_moe_treduce_bwd: null pointer");
    QT_CHECK_ARG(S > 0 && B > 0 && T > 0 && K > 0 && Dh > 0 && Dh % 4 == 0, "qt_moe_treduce_bwd: Dh=%d must be a multiple of 4", Dh);
    QT_CHECK_ARG(qt_aligned16(H) && qt_aligned16(dR) && qt_aligned16(dH), "qt_moe_treduce_bwd: alignment");
    hipLaunchKernelGGL(qt_moe_treduce_bwd_kernel, dim3(B, (T + MOE_TRB_TPB - 1) / MOE_TRB_TPB), dim3(256), 0, (hipStream_t)stream,
                       H, coef, dR, S, B, T, K, Dh, dH, dcoef, dh_amax);
    QT_CHECK_LAUNCH("qt_moe_treduce_bwd");
    return QT_OK;
}

extern "C" int qt_moe_router_bwd(void* stream, const float* dcoef, const long long* topk_inds, const float* topk_probs,
                                 const float* gauss, const float* probs, const float* gauss_raw, const float* dcsum,
                                 int B, int T, int K, int k, float sigma, float* dlogits, float* dgauss_raw) {
    QT_CHECK_ARG(sigma > 0.f, "qt_moe_router_bwd: sigma=%f", sigma);
    QT_CHECK_ARG(dcoef && topk_inds && topk_probs && gauss && probs && gauss_raw && dlogits && dgauss_raw, "qt_moe_router_bwd: null pointer");
    QT_CHECK_ARG(B > 0 && T > 0 && K > 0 && k > 0 && k <= K && K <= MOE_MAXK, "qt_moe_router_bwd: need 0<k<=K<=%d", MOE_MAXK);
    hipLaunchKernelGGL(qt_moe_router_bwd_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       dcoef, topk_inds, topk_probs, gauss, probs, gauss_raw, dcsum, B, T, K, k, sigma, dlogits, dgauss_raw);
    QT_CHECK_LAUNCH("qt_moe_router_bwd");
    return QT_OK;
}
