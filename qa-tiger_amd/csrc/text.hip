// Text-tower glue and on-device evaluation counters (fp32 / integer, HBM-bound):
//   qt_text_embed_fwd   x[b,l,:] = token_embedding[text[b,l],:] + positional_embedding[l,:]
//                       (reference src/models/encoders.py:31-32)
//   qt_text_eot_rows    out[b,:] = x[b, argmax_l text[b,l], :]      -- the EOT feature that is
//                       multiplied by text_projection (reference src/models/encoders.py:37)
//   qt_eval_counts      argmax over the 42 answer logits + per-question-type hit/total counters
//                       (reference src/trainutils.py:338-346, 417-423: one .item() sync per sample there)
#include "qt_common.h"
#include <limits.h>

// one wave per row, float4 per lane
__global__ __launch_bounds__(256) void qt_text_embed_kernel(const int* __restrict__ text, long rows, int L, int W,
                                                            int vocab, const float* __restrict__ emb,
                                                            const float* __restrict__ pos, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    int tok = text[row];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);          // nn.Embedding would raise; ids are clamped here and checked on the host
    const int l = (int)(row % L);
    const float4* e = reinterpret_cast<const float4*>(emb + (long)tok * W);
    const float4* q = reinterpret_cast<const float4*>(pos + (long)l * W);
    float4* o = reinterpret_cast<float4*>(out + row * W);
    for (int c = lane; c < (W >> 2); c += 64) {
        const float4 a = e[c], b = q[c];
        o[c] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

extern "C" int qt_text_embed_fwd(void* stream, const int* text, int B, int L, int W, int vocab,
                                 const float* token_embedding, const float* positional_embedding, float* out) {
    QT_CHECK_ARG(text && token_embedding && positional_embedding && out, "qt_text_embed_fwd: null pointer");
    QT_CHECK_ARG(B > 0 && L > 0 && W > 0 && W % 4 == 0 && vocab > 0, "qt_text_embed_fwd: bad sizes (W must be a multiple of 4)");
    QT_CHECK_ARG(qt_aligned16(token_embedding) && qt_aligned16(positional_embedding) && qt_aligned16(out),
                 "qt_text_embed_fwd: tables and output must be 16-byte aligned");
    const long rows = (long)B * L;
    hipLaunchKernelGGL(qt_text_embed_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       text, rows, L, W, vocab, token_embedding, positional_embedding, out);
    QT_CHECK_LAUNCH("qt_text_embed_fwd");
    return QT_OK;
}

// one wave per sample: wave-wide argmax of the token ids (first maximum, as torch.argmax on CPU), then the row copy
__global__ __launch_bounds__(256) void qt_text_eot_kernel(const int* __restrict__ text, int B, int L, int W,
                                                          const float* __restrict__ x, float* __restrict__ out,
                                                          int* __restrict__ eot_index) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    int best = INT_MIN, arg = 0;
    for (int l = lane; l < L; l += 64) {
        const int t = text[(long)b * L + l];
        if (t > best) { best = t; arg = l; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int ob = __shfl_xor(best, o, 64), oa = __shfl_xor(arg, o, 64);
        if (ob > best || (ob == best && oa < arg)) { best = ob; arg = oa; }
    }
    if (lane == 0 && eot_index) eot_index[b] = arg;
    const float* src = x + ((long)b * L + arg) * W;
    for (int c = lane; c < W; c += 64) out[(long)b * W + c] = src[c];
}

extern "C" int qt_text_eot_rows(void* stream, const int* text, int B, int L, int W, const float* x, float* out, int* eot_index) {
    QT_CHECK_ARG(text && x && out && B > 0 && L > 0 && W > 0, "qt_text_eot_rows: bad arguments");
    hipLaunchKernelGGL(qt_text_eot_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, text, B, L, W, x, out, eot_index);
    QT_CHECK_LAUNCH("qt_text_eot_rows");
    return QT_OK;
}

// one thread per sample (C = 42 logits); counters are int64 like the reference's tot_tensor / correct_tensor
__global__ __launch_bounds__(256) void qt_eval_counts_kernel(const float* __restrict__ logits, long ld, int B, int C,
                                                             const long long* __restrict__ label, const int* __restrict__ qtype,
                                                             int n_types, long long* __restrict__ predicted,
                                                             unsigned long long* __restrict__ totals,
                                                             unsigned long long* __restrict__ correct_by_type,
                                                             unsigned long long* __restrict__ total_by_type) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const float* row = logits + (long)b * ld;
    float best = row[0]; int arg = 0;
    for (int c = 1; c < C; ++c) { const float v = row[c]; if (v > best) { best = v; arg = c; } }   // first maximum (torch.max)
    if (predicted) predicted[b] = arg;
    const bool hit = (long long)arg == label[b];
    atomicAdd(totals + 0, 1ull);
    if (hit) atomicAdd(totals + 1, 1ull);
    if (qtype) {
        const int t = qtype[b];
        if (t >= 0 && t < n_types) {
            atomicAdd(total_by_type + t, 1ull);
            if (hit) atomicAdd(correct_by_type + t, 1ull);
        }
    }
}

extern "C" int qt_eval_counts(void* stream, const float* logits, long ld, int B, int C, const long long* label,
                              const int* qtype, int n_types, long long* predicted,
                              long long* totals, long long* correct_by_type, long long* total_by_type) {
    QT_CHECK_ARG(logits && label && totals && B > 0 && C > 0 && ld >= C, "qt_eval_counts: bad arguments");
    QT_CHECK_ARG(!qtype || (n_types > 0 && correct_by_type && total_by_type), "qt_eval_counts: qtype needs n_types and both per-type counters");
    hipLaunchKernelGGL(qt_eval_counts_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       logits, ld, B, C, label, qtype, n_types, predicted,
                       reinterpret_cast<unsigned long long*>(totals), reinterpret_cast<unsigned long long*>(correct_by_type),
                       reinterpret_cast<unsigned long long*>(total_by_type));
    QT_CHECK_LAUNCH("qt_eval_counts");
    return QT_OK;
}
