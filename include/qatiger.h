/* libqatiger_hip.so — C ABI of the MI355X (gfx950) kernels behind the QA-TIGER
 * fusion path.
 *
 * The reference (FightSH/QA-TIGER) has no FFI or operator-plugin API: its hot
 * path is `QA_TIGER.forward` (src/models/net.py:106-144) composed of torch.nn
 * modules (src/models/modules.py).  The drop-in boundary is therefore the
 * nn.Module surface (mirrored in qa-tiger_amd/src/models/), and this header is
 * the native layer underneath it: each entry point replaces the ATen op group
 * named in its comment.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (PyTorch): the
 *    library never allocates, frees or synchronises, and keeps no global state;
 *  - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); all
 *    work is enqueued on it, so the calls are hipGraph-capturable;
 *  - tensors are row-major fp32 with explicit leading dimensions (in elements);
 *  - return 0 on success, <0 on error (-2 bad argument, -3 launch failure,
 *    -4 unsupported shape); `qt_last_error()` returns a thread-local message;
 *  - dropout masks come from a stateless counter RNG keyed by (seed, element
 *    index): passing the same seed to the backward call regenerates the mask.
 */
#ifndef QATIGER_H
#define QATIGER_H
#ifdef __cplusplus
extern "C" {
#endif

int qt_version(void);
const char* qt_last_error(void);
/* Per-step dropout salt: dropout seeds are call arguments (frozen inside a captured HIP graph), so every dropout-carrying kernel adds the
 * device word registered here to its seed; bump it once per step (a one-thread kernel, part of the captured step) and every replay draws
 * fresh masks while the forward and backward of one step still agree.  NULL (default): seeds are used as passed.  This pointer is the one
 * piece of library-global state. */
int qt_set_dropout_salt(const unsigned long long* salt_dev);
int qt_bump_dropout_salt(void* stream, unsigned long long* salt_dev, unsigned long long inc);
/* p[0..n) = 0 (a kernel, not a runtime memset: every call of this library is an ordinary capturable launch). */
int qt_zero_f32(void* stream, float* p, long n);

/* C[M,N] = epilogue(alpha * op(A) op(B)).
 * Replaces nn.Linear / packed MHA in- and out-projections: modules.py:33,43
 * (Projection), 174-182 (AVQCrossAttn MHAs, linear1/2), 284-290, 382-398,
 * 637-645; net.py:60 (head); and their autograd backward.
 *   transA=0: A is [M,K] (lda>=K)   transA=1: A is [K,M] (lda>=M)
 *   transB=1: B is [N,K] (Linear weight layout)   transB=0: B is [K,N]
 * epilogue order: +bias[n], +add_src[m,n], relu, mask (mask_src[m,n]>0 ? v*mask_scale : 0),
 * dropout(drop_p, drop_seed, element index m*N+n), accumulate (C += v).
 * split_k: 0 = auto, 1 = off, >1 = K split over workgroups with fp32 atomics
 * (linear epilogues only).
 * batch > 1: `batch` independent problems of the same shape in one launch (one per attention head);
 * problem z uses A + z*a_bs, B + z*b_bs, C + z*c_bs (element strides); no bias/residual/mask then. */
int qt_gemm_f32(void* stream, int transA, int transB, int M, int N, int K, float alpha,
                const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                const float* bias, const float* add_src, long ld_add, int relu /* 0 none, 1 ReLU, 2 QuickGELU (clip_base.py:166-168) */,
                const float* mask_src, long ld_mask, float mask_scale,
                float drop_p, unsigned long long drop_seed, int accumulate, int split_k,
                int batch, long a_bs, long b_bs, long c_bs);
/* The same with a caller-owned split-K workspace (as qt_gemm_x3's `ws`): K splits of one problem store their partial tiles there and a
 * second launch sums them in a fixed order — no atomics, no zeroing launch, reproducible bits. */
int qt_gemm_f32_ws(void* stream, int transA, int transB, int M, int N, int K, float alpha,
                const float* A, long lda, const float* B, long ldb, float* C, long ldc,
                const float* bias, const float* add_src, long ld_add, int relu /* 0 none, 1 ReLU, 2 QuickGELU (clip_base.py:166-168) */,
                const float* mask_src, long ld_mask, float mask_scale,
                float drop_p, unsigned long long drop_seed, int accumulate, int split_k,
                int batch, long a_bs, long b_bs, long c_bs, float* ws, long ws_floats);

/* Same contract as qt_gemm_f32, computed with three 16-bit MFMAs per product on operands split
 * on the fly into hi+lo parts (half_type 0: fp16 parts, ~fp32-accurate, operands must be inside
 * fp16 range — forward GEMMs; 1: bf16 parts, ~1.5e-5 relative, full exponent range — backward GEMMs;
 * 3: ONE bf16 product per term, ~4e-3 relative — the reduced-precision mode of BASELINE config 5, never the parity path).
 * a_colsum (nullable, transA only; with accumulate != 0 it is ADDED to, like C): a_colsum[m] = sum_k A[k,m] — the bias gradient of a Linear layer
 * computed while the dW GEMM streams dY (A = dY given as [rows, N_out]).
 * Returns -4 without touching C or a_colsum when the shape or alignment does not fit its tiles;
 * callers then use qt_gemm_f32 (+ qt_colsum). */
int qt_gemm_x3(void* stream, int half_type, int transA, int transB, int M, int N, int K, float alpha,
               const float* A, long lda, const float* B, long ldb, float* C, long ldc,
               const float* bias, const float* add_src, long ld_add, int relu,
               const float* mask_src, long ld_mask, float mask_scale,
               float drop_p, unsigned long long drop_seed, int accumulate, int split_k, float* a_colsum,
               int batch, long a_bs, long b_bs, long c_bs,
               const float* a_amax /* nullable: device max|A| (qt_amax_f32); A is scaled by the power of two that puts it in fp16's normal
                                      range before its split and the result is scaled back: backward GEMMs pass max|dY| here */,
               const float* b_amax /* the same for B */,
               float* c_amax /* nullable: max|C| record of the values written — the next layer's max|dY| for free.  A launch that had to split K
                                 does not fill it and returns 2 instead of 0: measure C with qt_amax_f32 then */,
               float* ws /* nullable: caller-owned workspace; when the launch splits K (one problem) and split * M * N floats fit, the splits
                            store their partial tiles there and a second launch sums them into C in a fixed order (no atomics, no zeroing) */,
               long ws_floats);

/* softmax(q k^T * scale) -> dropout -> · v for every (item, head).
 * Replaces the core of nn.MultiheadAttention (torch F.multi_head_attention_forward
 * math path) at modules.py:217-221, 338, 562, 688, 698.
 * Element (item n, row i, col c) of q is q[(n*q_is + i*q_rs)*ldq + c]; key/value rows
 * of kv item m are rows m*kv_is .. m*kv_is+Lk-1.  Query item n attends to kv item
 * (n + kv_shift) % n_kv.  Head h uses columns h*dh .. h*dh+dh-1.
 * Short sequences (Lk <= 128) keep K/V resident in LDS; longer ones stream K/V in chunks with an
 * online softmax and need `lse` [n_items,H,Lq] (written by fwd), and in bwd the forward output
 * `o_fwd` (same addressing as d_o) plus a workspace `delta_ws` [n_items,H,Lq]. */
int qt_attn_fwd(void* stream, const float* q, long ldq, long q_is, long q_rs,
                const float* k, long ldk, const float* v, long ldv, long kv_is,
                float* o, long ldo, long o_is, long o_rs,
                int n_items, int n_kv, int kv_shift, int H, int Lq, int Lk, int dh,
                float scale, float drop_p, unsigned long long seed, float* lse);
/* The same with the option of writing o as a SPLIT IMAGE (qt_split_f16's format, unscaled; row pitch ld_img_bytes, a multiple of 128)
 * for the out-projection GEMM that is its only consumer (qt_gemm_ps): returns 1 when the image was written INSTEAD of the fp32 rows
 * (MFMA kernels: every attention shape of the path), 0 when the fp32 rows were written as by qt_attn_fwd. */
int qt_attn_fwd_img(void* stream, const float* q, long ldq, long q_is, long q_rs,
                    const float* k, long ldk, const float* v, long ldv, long kv_is,
                    float* o, long ldo, long o_is, long o_rs,
                    int n_items, int n_kv, int kv_shift, int H, int Lq, int Lk, int dh,
                    float scale, float drop_p, unsigned long long seed, float* lse, void* o_img, long ld_img_bytes);
int qt_attn_bwd(void* stream, const float* q, long ldq, long q_is, long q_rs,
                const float* k, long ldk, const float* v, long ldv, long kv_is,
                const float* d_o, long lddo, long do_is, long do_rs,
                float* dq, long lddq, long dq_is, long dq_rs,
                float* dk, long lddk, float* dv, long lddv,
                int n_items, int n_kv, int kv_shift, int H, int Lq, int Lk, int dh,
                float scale, float drop_p, unsigned long long seed,
                const float* o_fwd, const float* lse, float* delta_ws,
                float* g_amax /* nullable: max|dq,dk,dv| record, written by the MFMA kernels only: the call then returns 1 instead of 0 */);

/* Few-query attention with the K/V projections folded away (modules.py:562 TempMoE.qst_attn, Lq=1;
 * modules.py:692-698 PatchSelecter.crs_attn, Lq=2).  u = W_k,h^T q'_h per (query, head) is the "query",
 * the raw rows x (dimension D) are keys AND values:  s_p = scale*u.x_p, a = softmax_p(s) (+dropout),
 * ctx = sum_p a_p x_p, asum = sum_p a_p, lse = log-sum-exp.  Element (item n, query i, head h) of
 * u / ctx is row ((n*u_is + i*u_rs)*H + h) of a [*, D] matrix; lse / asum are indexed by that row.
 * The caller applies W_k / W_v with head-batched qt_gemm_* launches. */
int qt_xattn_fwd(void* stream, const float* u, long u_is, long u_rs, const float* x,
                 int n_items, int H, int Lq, int Lk, int D, float scale, float drop_p, unsigned long long seed,
                 float* ctx, float* lse, float* asum);
int qt_xattn_bwd(void* stream, const float* u, long u_is, long u_rs, const float* x,
                 const float* ctx, const float* lse, const float* asum, const float* dctx, const float* dasum,
                 int n_items, int H, int Lq, int Lk, int D, float scale, float drop_p, unsigned long long seed,
                 float* du, float* dx, float* dx_amax /* nullable: max|dx| record */, float* du_amax /* nullable: max|du| record */);

/* y = LayerNorm(sum_i sx[i] * dropout_i(x[i])) (+ optional ReLU on y); n_add <= 4.
 * Replaces modules.py:223-229, 233-235, 344-346, 603, 612, 713-716.
 * x, ldx, sx, drop, seed, dx, lddx, dx_acc are HOST arrays of length n_add.
 * xhat [rows,D] and rstd [rows] are saved for the backward. */
int qt_add_ln_fwd(void* stream, int rows, int D, int n_add,
                  const float* const* x, const long* ldx, const float* sx, const int* drop,
                  const unsigned long long* seed, float drop_p,
                  const float* gamma, const float* beta, float eps,
                  float* y, long ldy, int relu_out, float* xhat, float* rstd);
int qt_add_ln_bwd(void* stream, int rows, int D, int n_add,
                  const float* dy, long lddy, const float* xhat, const float* rstd, const float* gamma,
                  const float* sx, const int* drop, const unsigned long long* seed, float drop_p,
                  float* const* dx, const long* lddx, const int* dx_acc,
                  float* dgamma, float* dbeta,
                  float* const* dx_amax /* nullable; per addend a nullable max|dx_a| record, zeroed by the caller */,
                  int accumulate_params /* != 0: dgamma / dbeta are ADDED to (slices of a pre-zeroed gradient buffer), not overwritten */);

/* TempMoE router / Gaussian weights (modules.py:565-580, 420-471):
 * logits [B,K], gauss_raw [B,2K] (pairs (u_e, v_e)) -> probs [B,K], topk_inds [B,k] int64
 * (sorted by descending probability), topk_probs [B,k] (renormalised), gauss [B,k,T]. */
int qt_moe_router_fwd(void* stream, const float* logits, const float* gauss_raw, int B, int T, int K, int k,
                      float sigma /* TempMoE(sigma=...), modules.py:361; the Gaussian width is max(sigmoid(v), 0.09) / sigma */,
                      float* probs, long long* topk_inds, float* topk_probs, float* gauss);
/* coef[b,t,e] = sum_i [topk_inds[(b*T+t)%B, i]==e] topk_probs[b,i] gauss[b,i,t]; csum[b,e] = sum_t coef
 * (modules.py:495-533 including the repeat(T,1) index rule). */
int qt_moe_coef_fwd(void* stream, const long long* topk_inds, const float* topk_probs, const float* gauss,
                    int B, int T, int K, int k, float* coef, float* csum);
/* R[s*B+b, e, :] = sum_t coef[b,t,e] * H[(s*B+b)*T+t, e, :]   (H = expert first-layer activations
 * [S*B*T, K*Dh]; modules.py:590-599 + 517-533 with the second layer pulled out of the sum). */
int qt_moe_treduce_fwd(void* stream, const float* H, const float* coef, int S, int B, int T, int K, int Dh, float* R);
int qt_moe_treduce_bwd(void* stream, const float* H, const float* coef, const float* dR,
                       int S, int B, int T, int K, int Dh, float* dH, float* dcoef,
                       float* dh_amax /* nullable: zeroed max|dH| record (the stacked expert layer's dY scale) */);
/* dcsum (nullable) is the gradient of csum; it is folded into dcoef on the fly. */
int qt_moe_router_bwd(void* stream, const float* dcoef, const long long* topk_inds, const float* topk_probs,
                      const float* gauss, const float* probs, const float* gauss_raw, const float* dcsum,
                      int B, int T, int K, int k, float sigma, float* dlogits, float* dgauss_raw);

/* out[n] (+)= sum_m x[m,n]  — bias gradients of every nn.Linear. */
int qt_colsum(void* stream, const float* x, long ldx, int M, int N, float* out, int accumulate);
/* out[r,:] = alpha*a[r % ra,:] + beta*b[r % rb,:]  — `data + sub_data` (modules.py:587,596) and gradient folds. */
int qt_add_rows(void* stream, const float* a, long lda, int ra, float alpha,
                const float* b, long ldb, int rb, float beta, float* out, long ldo, long rows, int D);
/* out = g * mask: y given -> (y>0 ? 1/(1-p) : 0)  [backward of dropout(relu(z)) from its output];
 * y null -> the forward dropout multiplier regenerated from (seed, r*N+c). */
int qt_mask_grad(void* stream, const float* g, long ldg, const float* y, long ldy, float drop_p,
                 unsigned long long seed, float* out, long ldo, long rows, int N, float* amax /* nullable: max|out| record, zeroed by the caller */);
/* nn.CrossEntropyLoss(mean) and d loss / d logits (train.py:41, trainutils.py:278,286). */
int qt_xent_fwd_bwd(void* stream, const float* logits, long ld, const long long* label, int B, int A,
                    float* loss_rows, float* loss_mean, float* dlogits, long ldd);
/* torch.optim.Adam on one flat buffer (trainutils.py:141-144; the reference passes weight_decay from its config);
 * grad_scale multiplies the gradient first (1/world_size after a SUM all-reduce), then weight_decay * param is added
 * (torch's L2 form, not AdamW). */
int qt_adam_step(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                 int step, float lr, float beta1, float beta2, float eps, float grad_scale, float weight_decay);
/* The same with everything that changes from step to step in DEVICE memory (a captured train step's kernel arguments are frozen):
 * step_dev[0] = the step count, hp_dev[0..7] = { 1 - beta1^t, sqrt(1 - beta2^t), lr, beta1, beta2, eps, weight_decay, grad_scale }.
 * qt_adam_set_dev writes the count and hp_dev[2..7] from host values; call it OUTSIDE the graph before every replay, so a learning-rate
 * scheduler (StepLR / ReduceLROnPlateau, trainutils.py:147-182) or a loaded optimizer state reaches the replayed update.
 * qt_adam_step_dev (inside the graph): ++step_dev[0], bias corrections into hp_dev[0..1] (one-thread kernel), then the update. */
int qt_adam_set_dev(void* stream, long long* step_dev, float* hp_dev, long long step, float lr, float beta1, float beta2, float eps,
                    float weight_decay, float grad_scale);
int qt_adam_step_dev(void* stream, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                     long long* step_dev, float* hp_dev);

/* Value-bias term of an attention whose value projection is applied AFTER the attention-weighted sum (qt_xattn_fwd):
 *   out[m, e] = o[m, e] + asum[m, e / (E/H)] * b_v[e]      (asum = qt_xattn_fwd's per-head sum of the dropped probabilities)
 * replaces the `+ b_v` of nn.MultiheadAttention's value in-projection (modules.py:562, 698).  Backward: d o = d out;
 * dasum[m, h] = sum_{e in head h} dout[m, e] b_v[e];  dbv[e] (+)= sum_m dout[m, e] asum[m, head(e)]  (dbv_accumulate: add to what dbv holds). */
int qt_headbias_fwd(void* stream, const float* o, long ldo, const float* asum, const float* bv, long rows, int E, int H, float* out, long ldout);
int qt_headbias_bwd(void* stream, const float* dout, long ld, const float* asum, const float* bv, long rows, int E, int H,
                    float* dasum, float* dbv, int dbv_accumulate);

/* ---- pre-split operands (csrc/gemm_plane.hip): the same Linear / projection GEMMs and their backward, on operands whose
 * fp32 -> (hi, lo) fp16 split was done ONCE by a producer pass instead of inside every tile of the GEMM. ---- */

/* max |x| over an fp32 [R, C] tensor (row pitch ldx elements) into a device RECORD of 64 floats (256 bytes): each workgroup folds its
 * maximum into slot (block index & 63) with one atomic max on the bit pattern, consumers take the maximum of the 64 slots.  Every
 * `amax` argument of this header is such a 64-float record.  zero_first != 0 clears the record first, 0 folds into an earlier one.
 * Kernels that fill a record as a by-product (qt_gemm_x3 / qt_gemm_ps c_amax, qt_attn_bwd g_amax, qt_xattn_bwd dx_amax / du_amax) expect it ZEROED by the
 * caller (the host side hands out records from a pool it zeroes half at a time: no per-record launch). */
int qt_amax_f32(void* stream, const float* x, long ldx, long R, int C, float* amax, int zero_first);
/* the same for n_segments tensors of one flat buffer in one launch: amax[s] = max |flat[offsets[s] .. offsets[s] + counts[s])|
 * (offsets / counts: device int64 arrays; amax: n_segments records of 64 floats).  Every parameter's scale after an optimizer step. */
int qt_amax_segments(void* stream, const float* flat, const long* offsets, const long* counts, int n_segments, float* amax);
/* Split images of n_segments tensors of one flat buffer in one launch (all weight matrices after an optimizer step): the image of
 * flat[offsets[s] .. + counts[s]) is written to the same element range of `img` (128-byte aligned, as large as `flat`), scaled by
 * record amax[s] (qt_amax_segments).  Segments whose start or length is not a multiple of 32 elements are skipped. */
int qt_split_f16_segments(void* stream, const float* flat, const long* offsets, const long* counts, int n_segments,
                          const float* amax, void* img);
/* fp32 [R, C] -> split image: per row and per block of 32 columns one 128-byte line [32 x fp16 hi | 32 x fp16 lo] with
 * hi = fp16(s x), lo = fp16(s x - hi); s = 2^e chosen from *amax (nullable: s = 1) so that s * amax lies in [2^13, 2^14).
 * img: R rows of ld_img_bytes (multiple of 128, >= 4 C).  colsum (nullable): colsum[c] = sum_r x[r, c], the bias gradient of a
 * Linear layer whose dY is being split for its dX / dW GEMMs (overwritten; colsum_accumulate != 0: added to what is there, e.g. the
 * parameter's slice of a gradient buffer the caller has zeroed once for the whole step). */
int qt_split_f16(void* stream, const float* x, long ldx, long R, int C, const float* amax, void* img, long ld_img_bytes,
                 float* colsum, int colsum_accumulate);
/* C[M,N] = epilogue(alpha / (s_a s_b) * op(A) op(B)) with the epilogue of qt_gemm_f32, on split images streamed to LDS by LDS-DMA.
 *   a_kslow = 0: A image holds [M rows][K];  a_kslow = 1: [K rows][M]  (dW = dY^T X reads dY and X this way)
 *   b_kslow = 0: B image holds [N rows][K] (nn.Linear weight layout);  b_kslow = 1: [K rows][N]
 * a_amax / b_amax: the device floats the images were scaled by (nullable = unscaled).
 * K must be a multiple of 32 (and M / N of a K-slow operand); returns -4 without touching C when the shape does not fit.
 * c_amax (nullable): a zeroed max|x| record (64 floats) that receives max|C| of the values as stored (after the whole epilogue,
 * `accumulate` included), so the next GEMM that takes C as its gradient operand needs no pass of its own; a split-K launch
 * (atomics) cannot know it and returns 2 with the record untouched.
 * Tile height: 256 rows, or 192 / 128 for row-form A operands when the launcher's cost model finds that the shorter tile fills the
 * rounds of the persistent walk better (QT_PS_WMT=2|3|4 in the environment pins it).
 * ws / ws_floats (nullable / 0): a caller-owned workspace.  When the launch splits K (small outputs over a long K: the dW GEMMs) and
 * split * M * N floats fit, every K split stores its partial tile into its own slab and a second launch sums the slabs into C in a
 * fixed order — no fp32 atomics, no zeroing launch, bit-reproducible; without a workspace the splits meet in C through atomics. */
int qt_gemm_ps(void* stream, int a_kslow, int b_kslow, int M, int N, int K, float alpha,
               const void* A, long lda_bytes, const float* a_amax, const void* B, long ldb_bytes, const float* b_amax,
               float* C, long ldc, const float* bias, const float* add_src, long ld_add, int relu,
               const float* mask_src, long ld_mask, float mask_scale,
               float drop_p, unsigned long long drop_seed, int accumulate, int split_k, float* c_amax,
               float* ws, long ws_floats);
/* `batch` independent products of one shape in ONE launch: problem z multiplies the images at A + z a_bs_bytes and B + z b_bs_bytes
 * (strides multiples of 128) and writes C + z c_bs_floats; plain epilogue (alpha), no K split.  The H head slabs of the fold GEMMs
 * (u[:, h, :] = q[:, h-th d_h columns] W[h-th d_h rows, :]) are one such launch: the column block of q's image and the row block of
 * W's image of head h lie h * 4 d_h resp. h * d_h * ldb_bytes bytes further on, the slab of u h * D floats. */
int qt_gemm_ps_batched(void* stream, int a_kslow, int b_kslow, int M, int N, int K, float alpha,
                       const void* A, long lda_bytes, const float* a_amax, const void* B, long ldb_bytes, const float* b_amax,
                       float* C, long ldc, int batch, long a_bs_bytes, long b_bs_bytes, long c_bs_floats);
/* qt_gemm_ps that ALSO writes C as an unscaled split image (row pitch ld_img_bytes, a multiple of 128): the operand of the Linear layer
 * that consumes C, produced by the epilogue that holds the values in registers instead of by a qt_split_f16 pass over C.  Returns 3
 * (C written, image NOT written) when the launch cannot do it (a K split, unaligned rows): the caller then splits C as usual. */
int qt_gemm_ps_img(void* stream, int a_kslow, int b_kslow, int M, int N, int K, float alpha,
                   const void* A, long lda_bytes, const float* a_amax, const void* B, long ldb_bytes, const float* b_amax,
                   float* C, long ldc, const float* bias, const float* add_src, long ld_add, int relu,
                   const float* mask_src, long ld_mask, float mask_scale,
                   float drop_p, unsigned long long drop_seed, int accumulate, int split_k, float* c_amax,
                   void* c_img, long ld_img_bytes);

/* ---- BASELINE configs[4] ("fp8 MFMA on projection GEMMs"): e4m3 operand images and the GEMM on
 * v_mfma_scale_f32_32x32x64_f8f6f4 (fp32 accumulate, fp32 epilogue).  Outside the 1e-3 parity gate by construction: judged by
 * arg-max / top-k agreement with the fp32 path. ---- */

/* fp32 [R, C] -> e4m3 (OCP) image, one byte per element, R rows of ld_img_bytes (multiple of 128, >= C; C multiple of 128):
 * x * 2^e with 2^e * (*amax) in [128, 256) (amax nullable: unscaled). */
int qt_quant_fp8(void* stream, const float* x, long ldx, long R, int C, const float* amax, void* img, long ld_img_bytes);
/* C[M,N] = epilogue(alpha / (s_a s_b) * A B^T), A image [M rows][K], B image [N rows][K] (nn.Linear weight layout), K % 128 == 0;
 * epilogue of qt_gemm_f32; returns -4 without touching C when the shape does not fit the 256x256 tile. */
int qt_gemm_f8(void* stream, int M, int N, int K, float alpha,
               const void* A, long lda_bytes, const float* a_amax, const void* B, long ldb_bytes, const float* b_amax,
               float* C, long ldc, const float* bias, const float* add_src, long ld_add, int relu,
               const float* mask_src, long ld_mask, float mask_scale,
               float drop_p, unsigned long long drop_seed, int accumulate, int split_k);

/* ---- frozen CLIP text tower (forward only) and evaluation counters ---- */

/* Causal self-attention core of the text transformer: softmax(q k^T * scale + mask) v with the additive
 * -inf upper-triangular mask of reference src/models/base/clip_base.py:390-396, called from
 * ResidualAttentionBlock.attention (:211-214).  Addressing as qt_attn_fwd; Lq = Lk = L; no dropout. */
int qt_attn_causal_fwd(void* stream, const float* q, long ldq, long q_is, long q_rs,
                       const float* k, long ldk, const float* v, long ldv, long kv_is,
                       float* o, long ldo, long o_is, long o_rs,
                       int n_items, int H, int L, int dh, float scale);
/* out[b,l,:] = token_embedding[text[b,l],:] + positional_embedding[l,:]  (src/models/encoders.py:31-32).
 * text is int32 [B,L] (the dtype the reference feeds, src/trainutils.py:226); ids outside [0,vocab) are clamped. */
int qt_text_embed_fwd(void* stream, const int* text, int B, int L, int W, int vocab,
                      const float* token_embedding, const float* positional_embedding, float* out);
/* out[b,:] = x[b, argmax_l text[b,l], :] — the EOT-token feature (src/models/encoders.py:37); first maximum wins.
 * eot_index (nullable) receives the chosen positions. */
int qt_text_eot_rows(void* stream, const int* text, int B, int L, int W, const float* x, float* out, int* eot_index);
/* Evaluation counters (src/trainutils.py:338-346, 417-423): predicted[b] = argmax_c logits[b,c] (nullable),
 * totals[0] += B, totals[1] += #correct; with qtype (int32 [B], values in [0,n_types)) also
 * total_by_type[t] += 1 and correct_by_type[t] += hit.  Counters are int64 and are ADDED to (zero them first). */
int qt_eval_counts(void* stream, const float* logits, long ld, int B, int C, const long long* label,
                   const int* qtype, int n_types, long long* predicted,
                   long long* totals, long long* correct_by_type, long long* total_by_type);

/* ---- feature shards ---- */
/* dst[i] = (float)src[i] for fp16 feature shards (the reference casts per batch in get_items, src/trainutils.py:223-250). */
int qt_cast_f16_f32(void* stream, const void* src_f16, float* dst, long n);

/* ---- TSPM (src/models/tspm.py) — attention pieces its nn.MultiheadAttention(512, nhead=1) and its
 * use of the returned attention weights need beyond qt_attn_fwd/bwd ---- */
/* p = softmax(scale * s) per row (nullable output, saved for backward), y = dropout(p); one row = one (item, query). */
int qt_softmax_fwd(void* stream, const float* s, long lds, long rows, int L, float scale, float drop_p,
                   unsigned long long seed, float* p, long ldp, float* y, long ldy);
/* ds = scale * p * (dp - sum_j dp_j p_j) with dp = dy * (dropout multiplier regenerated from seed). */
int qt_softmax_bwd(void* stream, const float* dy, long lddy, const float* p, long ldp, long rows, int L, float scale,
                   float drop_p, unsigned long long seed, float* ds, long ldds);
/* out[n,i,:] = head-averaged attention weights of query row i of item n (what nn.MultiheadAttention returns with
 * need_weights=True, average_attn_weights=True: tspm.py:99-100 -> top-K time indices at :118-125).  Addressing, kv item
 * mapping and dropout indices as qt_attn_fwd; forward only. */
int qt_attn_avg_weights(void* stream, const float* q, long ldq, long q_is, long q_rs,
                        const float* k, long ldk, long kv_is,
                        int n_items, int n_kv, int kv_shift, int H, int Lq, int Lk, int dh,
                        float scale, float drop_p, unsigned long long seed, float* out);

#ifdef __cplusplus
}
#endif
#endif /* QATIGER_H */
