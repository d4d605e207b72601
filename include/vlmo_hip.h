/* C-ABI of libvlmo_hip.so: the MI355X (gfx950) engine under the VLMo
 * pretraining forward/backward path of fanzhongyi/ExploreMultiModal.
 *
 * The reference has no native/FFI boundary for this path (it is a Python
 * nn.Module contract: models/build.py:4-12, models/vlmo/vlmo_module.py:395-436,
 * models/vlmo/vlmo.py:357-414); these entry points are what a binding of that
 * path to this engine calls.  Each one cites the reference code it replaces.
 *
 * Conventions
 *  - plain pointers and sizes only; every buffer (including workspaces) is
 *    owned by the caller and lives in device memory unless stated otherwise;
 *  - every function only ENQUEUES work on `stream` (asynchronous, graph-capturable,
 *    no allocation, no synchronisation) and returns 0 on success, <0 for an
 *    argument error, >0 for a hipError_t; vlmo_last_error() gives the message;
 *  - token-major activations: row m of an [M, d] matrix is one token; rows are
 *    "packed" (all text tokens of the batch, then all image tokens) and attention
 *    finds its keys through per-sequence segment descriptors, so the reference's
 *    torch.cat([txt, img], dim=1) (vlmo.py:406) never materialises;
 *  - dropout masks come from a counter-based generator keyed by (seed, element),
 *    so backward regenerates them; prob = thresh/65536, thresh 0 = disabled.
 */
#ifndef VLMO_HIP_H
#define VLMO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

enum { VLMO_BF16 = 0, VLMO_F16 = 1, VLMO_F32 = 2 };

/* GEMM epilogues (fused into the accumulator write-back) */
enum {
    VLMO_EPI_BIAS = 0,      /* out[T]   = acc + bias            (relu optional)             */
    VLMO_EPI_BIAS_GELU = 1, /* out[T]   = u = acc + bias ; out2[T] = dropout(gelu_erf(u))    */
    VLMO_EPI_RESID = 2,     /* zd = dropout(acc + bias); out2[T] = zd;                        */
                            /* out[f32] = resid + gamma * zd * row_scale[m]                   */
    VLMO_EPI_DGELU = 3,     /* out[T]   = dropout_mask(acc) * gelu_erf'(aux[m,n])            */
    VLMO_EPI_F32 = 4,       /* out[f32] = acc + bias + beta * out                             */
    VLMO_EPI_DUAL = 5,      /* v = resid[T] + beta*(acc + bias); out[T] = v; out2[T] = relu(v)  */
    VLMO_EPI_ARGMAX = 6,    /* out = partial (max, argmax) of acc + bias per row and 64-column   */
                            /* chunk: float/int32 pairs [M, ldo, 2]; finish with vlmo_argmax_reduce */
    VLMO_EPI_CE = 7,        /* fused cross-entropy forward of a vocabulary head (heads.py:86-112,    */
                            /* objectives.py:57-68,571-582): out = partial {max, sum exp(x - max),    */
                            /* argmax, logit of label row_index[m]} per row and 64-column chunk,      */
                            /* floats [M, ldo, 4]; finish with vlmo_ce_reduce.  No logits in HBM.     */
    VLMO_EPI_CE_BWD = 8     /* out[T] = (exp(acc + bias - lse[m]) - [n == label[m]]) * row_scale[m]:  */
                            /* d loss / d logits recomputed; resid = lse [M] (1-D!), row_index = labels */
};

typedef struct VlmoEpilogue {
    void* out;              /* [M, ldo]                                     */
    void* out2;             /* [M, ld2] (see epilogue table), may be NULL   */
    const float* bias;      /* [N] or NULL                                  */
    const float* gamma;     /* [N] layer-scale or NULL (=1)                 */
    const float* resid;     /* [M, ldo] fp32 residual stream                */
    const float* row_scale; /* drop-path scale: per token [M], or per group */
                            /* when row_index is set (scale[row_index[m]])  */
    const int32_t* row_index; /* [M] token -> scale group, or NULL          */
    const void* aux;        /* [M, ld2] pre-activation for VLMO_EPI_DGELU   */
    int32_t ldo, ld2;
    int32_t relu;           /* bit 0: ReLU on the output (VLMO_EPI_BIAS); bit 1: ReLU on the INPUT activations (vlmo_conv2d_nhwc, f16);
                             * bit 2: VLMO_EPI_BIAS_GELU writes d h / d u = GELU'(u) * dropout mask / (1 - p) to `out` (not u),
                             * and VLMO_EPI_DGELU takes exactly that as `aux`: out = acc * aux, no erf / hash in the backward */
    uint32_t drop_thresh;   /* round(p * 65536); 0 disables dropout         */
    float inv_keep;         /* 1 / (1 - p)                                  */
    float beta;
    uint64_t seed;
    float* colpart;         /* VLMO_EPI_DGELU: [ceil(M/16), N] fp32 or NULL: column-sum partials of   */
                            /* the values written to out; EVERY row is written (an epilogue pass     */
                            /* puts the sums of its 16 or 32 output rows into the row of its first   */
                            /* 16-row block and zeros into the other) -- the fc1 bias gradient is    */
                            /* the fold of these rows (vlmo_colwork_multi kind 0); plain stores      */
} VlmoEpilogue;

const char* vlmo_last_error(void);
int vlmo_abi_version(void);

/* C[M,N] = A[M,K] . B[N,K]^T with a fused epilogue.  tile: -1 = pick by shape, 0 = 128x128x64
 * (two workgroups per CU), 3 = 256x256x64 with the two-wave-group ping-pong schedule (one per CU),
 * 4 = 256x128x32 (two per CU; bf16 with the bias / bias+GELU epilogues, else it falls back to 0),
 * 8 = 192x256x64 ping-pong (bf16 with the bias / bias+GELU / residual / GELU-derivative epilogues, else 3): picked when its tile count
 * needs fewer dispatch rounds than 256x256 (VLMo-Large at 32 pairs per GPU),
 * 309..320 = (16 * (tile - 300)) x 256 x 64 ping-pong on v_mfma_f32_16x16x32 (144 .. 320 rows in 16-row steps; bf16, the
 * same four epilogues; 106..110 = the even heights 32 * (tile - 100)): tile height chosen per (M, N) so that the tiles
 * fill whole dispatch rounds of 256 CUs.
 * Replaces nn.functional.linear at vlmo.py:76-78 (qkv), vlmo.py:96 (proj), timm
 * Mlp fc1/fc2 (vlmo.py:141-157, 195-196), the PatchEmbed conv (vlmo.py:304) and,
 * with pre-transposed weights, their input gradients. K % 64 == 0, N % 4 == 0. */
int vlmo_gemm_nt(int epi, int dtype, int tile, const void* A, int lda, const void* B, int ldb,
                 int M, int N, int K, const VlmoEpilogue* e, hipStream_t stream);
/* C = epilogue((seg_scale * A[:, :k1] . B[:, :k1]^T) + A2[:, :K-k1] . B[:, k1:]^T): the reduction runs over two activation
 * matrices with the same rows (own leading dimensions) against ONE weight matrix [N, K].  dall_e EncoderBlock tail
 * (dall_e/encoder.py:45-46): id_path(x) + post_gain * res_path(x) with a convolutional id_path = one GEMM over
 * [conv_3 output | x] against [conv_4.w | id_path.w]; output convolution with the fp32 weight split [w_hi | w_lo]:
 * A2 = A (the activations are read twice instead of being concatenated in HBM).  k1 % 64 == 0; dtype VLMO_F16 only. */
int vlmo_gemm_nt_2src(int epi, int dtype, int tile, const void* A, int lda, int k1, float seg_scale, const void* A2,
                      int lda2, const void* B, int ldb, int M, int N, int K, const VlmoEpilogue* e,
                      hipStream_t stream);
/* 1..4 problems C_g[M_g,N] = A_g[M_g,K] . B_g[N,K]^T with the same N, K, leading dimensions and epilogue kind
 * in ONE launch: the per-modality expert FFNs of a Block below the fusion layer (mlp['l'] on the text rows,
 * mlp['v'] on the image rows: vlmo.py:141-157, 195-196).  e[g] is group g's epilogue (its own outputs, bias,
 * dropout seed).  A launch takes at least one tile time however few tiles it has, so two half-empty launches
 * cost twice one. */
int vlmo_gemm_nt_grouped(int epi, int dtype, int tile, int ngroups, const void* const* A, int lda,
                         const void* const* B, int ldb, const int32_t* M, int N, int K,
                         const VlmoEpilogue* e, hipStream_t stream);

/* C[N1,N2] += alpha * A[M,N1]^T . B[M,N2]  (weight gradients of the linears above, i.e. autograd of
 * vlmo.py:76-78,96,195-196).  The token dimension is split over workgroups; partial products go through
 * the caller's workspace `ws` (>= vlmo_gemm_tn_ws_bytes(M,N1,N2)) and one reduction pass, or, with
 * ws = NULL, straight into C with fp32 atomics. */
int64_t vlmo_gemm_tn_ws_bytes(int M, int N1, int N2);
int vlmo_gemm_tn(int dtype, const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                 int M, int N1, int N2, float alpha, int splits, float* ws, int64_t ws_bytes,
                 hipStream_t stream);

/* Several weight gradients C_q[N1,N2] (+)= alpha * A_q[M,N1]^T . B_q[M,N2] in ONE launch of 256x256 tiles:
 * the qkv / proj / fc1 / fc2 gradients of one or two transformer blocks (autograd of vlmo.py:76-78,96,195-196)
 * together fill the chip without splitting the token dimension, so there are no partial slabs and no
 * reduction pass.  accumulate != 0: C += ..., else C = ...  No workspace. */
typedef struct VlmoTnProblem {
    const void* A;
    const void* B;
    float* C;
    int32_t lda, ldb, ldc;
    int32_t M, N1, N2;
    float alpha;
    int32_t accumulate;
} VlmoTnProblem;
int vlmo_gemm_tn_multi(int dtype, const VlmoTnProblem* probs, int n, hipStream_t stream);

/* LayerNorm over the last dim (eps = 1e-12 in VLMo: vlmo_module.py:21-23; vlmo.py:188,192,413).
 * x fp32 [M,d] -> y (bf16, or fp32 when out_f32) at row rowmap[m] (or m), + mean/rstd [M]. */
int vlmo_ln_fwd(const float* x, const float* w, const float* b, void* y, int out_f32,
                float* mean, float* rstd, const int32_t* rowmap, int M, int d, float eps,
                hipStream_t stream);
/* Workspace (bytes) the column-reducing kernels below need for `ncols` reduced columns
 * (ln_bwd and resid_bwd reduce 2*d columns, colsum N).  Caller-owned, reusable across calls
 * on one stream. */
int64_t vlmo_reduce_ws_bytes(int ncols);

/* dx = dres + LN'(dy);  dw += sum dy*xhat;  db += sum dy  (dres may be NULL). */
int vlmo_ln_bwd(const void* dy, int dy_f32, const int32_t* rowmap, const float* x, const float* w,
                const float* mean, const float* rstd, const float* dres, float* dx, float* dw,
                float* db, int M, int d, float* ws, int64_t ws_bytes, hipStream_t stream);

/* vlmo_ln_bwd (dy bf16, no row map) fused with the vlmo_resid_bwd of the residual branch that the LayerNorm's
 * input gradient feeds next (Block backward: norm2, then the attention branch x1 = x + gamma_1 * rs * zd):
 * dz = dropout_mask * dx * gamma * rs, dgamma += sum dx * rs * zd, dbias += sum dz, where dx is the value this
 * call writes.  Saves re-reading dx (fp32 [M, d]) and one launch.  ws: vlmo_reduce_ws_bytes(4 * d). */
int vlmo_ln_resid_bwd(const void* dy, const float* x, const float* w, const float* mean, const float* rstd,
                      const float* dres, float* dx, float* dw, float* db, const void* zd, const float* gamma,
                      const float* row_scale, const int32_t* row_index, void* dz, float* dgamma, float* dbias,
                      uint32_t drop_thresh, float inv_keep, uint64_t seed, int M, int d, float* ws,
                      int64_t ws_bytes, hipStream_t stream);

/* Fused softmax attention over packed rows (vlmo.py:79-95).
 * qkv [M, 3*d] (q | k | v, head-major inside each third), ctx [M, d].
 * seg[s] = {rowA, lenA, rowB, lenB}: sequence s = rows [rowA,rowA+lenA) ++ [rowB,rowB+lenB).
 * keymask [M] int32 (0 = padded key, vlmo.py:89-91) or NULL.  lse [S, heads, NPAD] fp32.
 * Attention dropout (vlmo.py:93) is a counter hash of (seed, mask_seq0 + s, head, query, key): a backward launch over
 * the sequences [s0, s0 + n) of a forward launch passes mask_seq0 = s0 and the forward's seed to regenerate its mask. */
int vlmo_attn_fwd(const void* qkv, const int32_t* seg, int num_seq, const int32_t* keymask,
                  void* ctx, float* lse, int lse_stride, int heads, int d, int max_len,
                  float scale, uint32_t drop_thresh, float inv_keep, uint64_t seed, int mask_seq0,
                  hipStream_t stream);
/* qv_colsum (optional, [num_seq][2 d] fp32, written): per sequence the column sums of its tokens' dq | dv rows -- the
 * q_bias / v_bias gradient (vlmo.py:71-75) is their sum over the sequences, so nobody re-reads dqkv for it. */
int vlmo_attn_bwd(const void* qkv, const void* ctx, const void* dctx, const float* lse,
                  int lse_stride, const int32_t* seg, int num_seq, const int32_t* keymask,
                  void* dqkv, float* qv_colsum, int heads, int d, int max_len, float scale,
                  uint32_t drop_thresh, float inv_keep, uint64_t seed, int mask_seq0, hipStream_t stream);

/* Residual-branch backward (vlmo.py:194-196): dz = dx * gamma * row_scale * dropmask/(1-p);
 * dgamma += sum_m dx * row_scale * zd;  dbias += sum_m dz. */
int vlmo_resid_bwd(const float* dx, const void* zd, const float* gamma, const float* row_scale,
                   const int32_t* row_index, void* dz, float* dgamma, float* dbias, int M, int d,
                   uint32_t drop_thresh,
                   float inv_keep, uint64_t seed, float* ws, int64_t ws_bytes, hipStream_t stream);

/* out[c] += sum_m x[m, c]  (bias gradients), x is bf16/f16 [M, ld]. */
int vlmo_colsum(int dtype, const void* x, int ld, float* out, int M, int N, float* ws,
                int64_t ws_bytes, hipStream_t stream);

/* Column work of several producers in ONE launch (the bias / layer-scale / LayerNorm-weight gradients of one or
 * two transformer blocks):  kind 0 folds fp32 partial rows ws[rows, ncols] (written by vlmo_ln_bwd /
 * vlmo_ln_resid_bwd / vlmo_resid_bwd with defer), kind 1 sums the columns of a bf16/f16 matrix x[rows, ld].
 * Column c goes to out[c / n0][c % n0] += sum (out[k] may be NULL). */
typedef struct VlmoColJob {
    int32_t kind, ld;
    const void* src;
    int32_t rows, ncols;
    float* out[4];
    int32_t n0, pad_;
} VlmoColJob;
int vlmo_colwork_multi(int dtype, const VlmoColJob* jobs, int n, hipStream_t stream);

/* fp32 -> bf16/f16 weight shadow copies: dst = cast(src), dstT = cast(src)^T (either may be NULL). */
int vlmo_cast_weight(int dtype, const float* src, int rows, int cols, void* dst, void* dstT,
                     hipStream_t stream);
/* n weights in one launch (every weight of the model is stale after an optimizer step): job q casts src[q] [rows[q], cols[q]]
 * fp32 into dst[q] (same layout, may be NULL) and dstT[q] (transposed, may be NULL). */
int vlmo_cast_weight_multi(int dtype, int n, const float* const* src, const int32_t* rows, const int32_t* cols,
                           void* const* dst, void* const* dstT, hipStream_t stream);

/* Image embedding (vlmo.py:298-319, timm PatchEmbed):
 *  patchify: image f32 [B,C,H,W] -> bf16 [B*gh*gw, C*p*p] rows ordered (b, py, px), cols (c, ky, kx)
 *  finish:   x[b,0] = cls + pos[0] + type;  x[b,1+i] = (masked? mask_token : proj[b,i]) + pos[1+i] + type */
int vlmo_patchify(const float* img, void* out, int B, int C, int H, int W, int patch,
                  hipStream_t stream);
int vlmo_embed_img_finish(const void* proj, const float* cls_tok, const float* mask_tok,
                          const float* pos, const float* type_row, const uint8_t* masked_pos,
                          float* x, int B, int npatch, int d, uint32_t drop_thresh,
                          float inv_keep, uint64_t seed, hipStream_t stream);
/* backward of finish: dproj[b,i] = masked? 0 : g; dcls += ; dmask += ; dpos += ; dtype_row += */
int vlmo_embed_img_bwd(const float* dx, const uint8_t* masked_pos, void* dproj, float* dcls,
                       float* dmask, float* dpos, float* dtype_row, int B, int npatch, int d,
                       uint32_t drop_thresh, float inv_keep, uint64_t seed, hipStream_t stream);

/* Text embedding (vlmo.py:321-324 + transformers BertEmbeddings):
 *  e = word[ids] + btype0 + pos[t];  x = LN_eps(e) (+dropout) + type0 */
int vlmo_embed_txt_fwd(const int64_t* ids, const float* word, const float* pos, const float* btype0,
                       const float* ln_w, const float* ln_b, const float* type0, float* x,
                       float* xhat, float* rstd, int B, int T, int d, float eps,
                       uint32_t drop_thresh, float inv_keep, uint64_t seed, hipStream_t stream);
int vlmo_embed_txt_bwd(const float* dx, const int64_t* ids, const float* xhat, const float* rstd,
                       const float* ln_w, float* dword, float* dpos, float* dbtype0, float* dln_w,
                       float* dln_b, float* dtype0, int B, int T, int d, uint32_t drop_thresh,
                       float inv_keep, uint64_t seed, hipStream_t stream);

/* Optional timing of the GEMM launches with HIP event pairs on their launch stream (bench.py's roofline).
 * stop() sums per tag: tag = epilogue id (+16 for the 256x256 tile, +48 for the 256x128 tile) for vlmo_gemm_nt, 32 + epilogue for
 * vlmo_conv2d_nhwc, 64 / 72 for vlmo_gemm_tn (128x128 / 256x256 tiles), 73 for vlmo_gemm_tn_multi; returns the number of recorded launches. Synchronise first. */
int vlmo_profile_start(int max_records);
int vlmo_profile_stop(int ntags, double* ms, double* flops, int64_t* launches);

/* ---- optimizer step ---------------------------------------------------------------
 * Multi-tensor Adam / AdamW over a list of fp32 tensors, replacing apex FusedAdam(adam_w_mode=True)
 * (utils/optim_factory.py:185-186) and the unscale + clip_grad_norm_ of NativeScalerWithGradNormCount
 * (utils/utils.py:343-364).  All tables are DEVICE arrays owned by the caller: per tensor the device
 * addresses of parameter / gradient / exp_avg / exp_avg_sq, its element count, learning rate and weight
 * decay; the work list is cut into chunks of `chunk` elements (chunk_tensor[c], chunk_start[c]). */
typedef struct {
    const int64_t* p;
    const int64_t* g;
    const int64_t* m;
    const int64_t* v;
    const int64_t* numel;
    const float* lr;
    const float* wd;
    const int32_t* chunk_tensor;
    const int64_t* chunk_start;
    int32_t n_chunks, chunk;
} VlmoTensorList;

typedef struct {
    float beta1, beta2, eps;
    float inv_bc1, inv_bc2;     /* 1 / (1 - beta^step), or 1 without bias correction */
    int32_t adam_w_mode;        /* 1: decoupled weight decay (AdamW); 0: L2 (decay added to the gradient) */
} VlmoAdamArgs;

/* out[0] = || inv_scale * g ||_2 over every tensor of the list, out[1] = the factor to apply to the raw
 * gradients = inv_scale * min(1, max_norm / (norm + 1e-6)) (torch.nn.utils.clip_grad_norm_; max_norm <= 0:
 * no clipping), out[2] = 1 if the norm is not finite.  partial: n_chunks floats of scratch.  Deterministic. */
int vlmo_mt_grad_norm(const VlmoTensorList* tl, float inv_scale, float max_norm, float* partial, float* out,
                      hipStream_t stream);
/* One Adam step on every tensor.  ctl = the 3 floats of vlmo_mt_grad_norm (gradients are multiplied by
 * ctl[1]; the whole step is skipped when ctl[2] != 0, like GradScaler.step) or NULL. */
int vlmo_mt_adam(const VlmoTensorList* tl, const VlmoAdamArgs* a, const float* ctl, hipStream_t stream);

/* The stream vlmo_block_bwd's weight-gradient work runs on (VlmoBlockDesc.side_stream).  It has the
 * whole step of slack while the activation-gradient chain on the caller's stream is the critical
 * path, so it is created with the LOWEST dispatch priority of the device (low_priority != 0), and/or
 * confined to the compute units of cu_mask (cu_mask_words 32-bit words; NULL/0 = all CUs). */
int vlmo_side_stream_create(int low_priority, const uint32_t* cu_mask, int cu_mask_words, hipStream_t* out);

/* ---- one transformer Block in ONE call (vlmo.py:187-197 and its autograd) ---------------------
 * Enqueues norm1 -> qkv -> attention -> proj(+gamma_1, residual) -> norm2 -> expert FFN(s)
 * (+gamma_2, residual), resp. the whole backward of that, from native code: the host pays one
 * FFI call per block instead of ~12 / ~30.  Expert e works on rows [exp_row0[e], +exp_rows[e]);
 * attention launch a on seg[a] (sequences of at most maxlen[a] tokens).  All buffers caller-owned.
 * Backward runs the weight-gradient GEMMs and bias column sums on `side_stream` (if not NULL)
 * beside the input-gradient chain and joins before returning control of the buffers. */
typedef struct VlmoBlockDesc {
    int32_t M, d, hidden, heads;
    int32_t n_experts, exp_row0[2], exp_rows[2];   /* 1..2 experts */
    int32_t n_attn, nseq[2], maxlen[2], lse_stride[2];
    int32_t attn_seed_idx[2], attn_seq0[2];         /* dropout mask of launch a: seed + 11 + attn_seed_idx[a], first sequence attn_seq0[a]
                                                     * (a backward launch split off a shared forward launch keeps the forward's mask) */
    const int32_t* seg[2];
    const int32_t* keymask;
    float eps;
    uint32_t drop_thresh, attn_drop_thresh;
    float inv_keep, attn_inv_keep;
    uint64_t seed;
    const float* rs1;           /* drop-path scales of the two residual branches: per row, or per */
    const float* rs2;           /* group when row_index is set                                     */
    const int32_t* row_index;
    int32_t tile, need_bwd;
    /* parameters: fp32 vectors, bf16 shadow matrices W [out,in] and W^T [in,out] */
    const float *g1, *g2, *n1w, *n1b, *n2w, *n2b, *qkv_bias, *proj_b;
    const void *qkv_w, *qkv_wT, *proj_w, *proj_wT;
    const float *b1[2], *b2[2];
    const void *w1[2], *w1T[2], *w2[2], *w2T[2];
    /* forward activations (saved for backward) */
    const float* x;
    float *x1, *x2;
    void *y1, *qkv, *ctx, *zd1, *y2, *u, *h, *zd2;
    float *mean1, *rstd1, *mean2, *rstd2, *lse[2];
    /* backward */
    const float* dx2;
    float *dx1, *dx0;
    void *dz2, *du, *dy2, *dz1, *dctx, *dqkv, *dy1;
    float *dg1, *dg2, *dn1w, *dn1b, *dn2w, *dn2b, *dqkv_w, *dqkv_b, *dproj_w, *dproj_b;
    float *dw1[2], *db1[2], *dw2[2], *db2[2];
    float *ws_main, *ws_side;   /* column-reduction scratch per stream (vlmo_reduce_ws_bytes)            */
    int64_t ws_bytes;
    float* ws_tn;               /* weight-gradient slab scratch (vlmo_gemm_tn_ws_bytes), used on the     */
    int64_t ws_tn_bytes;        /* side stream; NULL = atomics                                           */
    hipStream_t side_stream;
} VlmoBlockDesc;
int vlmo_block_fwd(const VlmoBlockDesc* b, hipStream_t stream);
int vlmo_block_bwd(const VlmoBlockDesc* b, hipStream_t stream);

/* ---- all blocks of one backbone pass in ONE call per direction (the loops at vlmo.py:402-411) -------------
 * blocks[] are in forward order; block i's x2 is block i+1's x.  vlmo_stack_fwd = vlmo_block_fwd per block.
 * vlmo_stack_bwd runs the activation-gradient chains of all blocks back to back on `stream` and defers the
 * parameter-gradient work (weight-gradient GEMMs, bias / layer-scale / LayerNorm column sums) to
 * `side_stream` in batches of `wgrad_batch` blocks, one vlmo_gemm_tn_multi + one vlmo_colwork_multi launch
 * per batch (parameter gradients are ACCUMULATED: zero or pre-load them; with wgrad_store only the vector ones).  Because that work runs later than
 * the block's own chain, the caller gives the k-th block in backward order its own backward temporaries
 * (dz2, du, dz1, dqkv) and column workspace ws_main (>= (3 + n_experts) * vlmo_reduce_ws_bytes(2*d) bytes),
 * rotating over n_tmp_sets > wgrad_batch sets (set k % n_tmp_sets); the call waits for a set's deferred readers
 * before reusing it and joins the side stream before it returns control of the buffers.  ws_side, ws_tn and
 * the blocks' own side_stream field are ignored.  grad_ready (optional, [n_blocks] hipEvent_t handles from
 * vlmo_event_create): event i is recorded when block i's parameter gradients are complete, so a gradient
 * all-reduce on another stream can start per block (vlmo_stream_wait_event). */
typedef struct VlmoStackDesc {
    int32_t n_blocks, wgrad_batch, n_tmp_sets;
    int32_t wgrad_store;        /* != 0: weight-gradient MATRICES are written (C = ...), not accumulated: no zero-fill, no
                                 * read of C.  The vector gradients (biases, LayerNorm, layer-scale) always accumulate. */
    const VlmoBlockDesc* blocks;
    hipStream_t side_stream;
    void* const* grad_ready;
} VlmoStackDesc;
int vlmo_stack_fwd(const VlmoStackDesc* s, hipStream_t stream);
int vlmo_stack_bwd(const VlmoStackDesc* s, hipStream_t stream);
int vlmo_event_create(void** out);
int vlmo_event_destroy(void* ev);
int vlmo_stream_wait_event(hipStream_t stream, void* ev);

/* ---- gradient exchange (replaces torch DDP / DeepSpeed ZeRO-2 over NCCL, train/pretrain/multimodal.py:61-95,
 * conf/ds_stage/l2.yaml) ----
 * RCCL over xGMI, one communicator per process, collectives enqueued on the caller's stream, sum reduction.  RCCL is
 * resolved at run time (the copy already in the process, else librccl.so); without it every entry returns -1 (an argument
 * error, message in vlmo_last_error); RCCL's own failures come back as 1000 + ncclResult_t.
 * Bootstrap: rank 0 calls vlmo_comm_unique_id and hands the 128 bytes to the other ranks by any means (the Python host
 * uses the torch.distributed store); every rank then calls vlmo_comm_init (blocks until all ranks arrived). */
#define VLMO_COMM_ID_BYTES 128
int vlmo_comm_available(void);
int vlmo_comm_unique_id(void* id128);
int vlmo_comm_init(void** comm, const void* id128, int rank, int world);
/* waits for the device to drain (every collective enqueued through the communicator has finished), then frees it */
int vlmo_comm_destroy(void* comm);
/* the size of the communicator as RCCL reports it (ncclCommCount) and this process' rank in it (rank may be NULL) */
int vlmo_comm_count(void* comm, int* ranks, int* rank);
int vlmo_comm_all_reduce(void* comm, const void* send, void* recv, int64_t count, int dtype, hipStream_t stream);
/* send [world * recv_count] -> recv [recv_count] = this rank's slice of the sum (ZeRO-2 gradient partition) */
int vlmo_comm_reduce_scatter(void* comm, const void* send, void* recv, int64_t recv_count, int dtype,
                             hipStream_t stream);
/* send [send_count] -> recv [world * send_count]; send may be recv + rank * send_count (in place) */
int vlmo_comm_all_gather(void* comm, const void* send, void* recv, int64_t send_count, int dtype,
                         hipStream_t stream);
/* the reducer's two passes over a gradient arena: dst bf16 = src * scale (1 / world), and back.  16-B aligned. */
int vlmo_grad_pack(const float* src, void* dst_bf16, int64_t n, float scale, hipStream_t stream);
int vlmo_grad_unpack(const void* src_bf16, float* dst, int64_t n, hipStream_t stream);

/* ---- dall_e dVAE encoder (dall_e/encoder.py:49-133), fp16 NHWC activations [B*H*W, C] ---- */

/* Conv2d, stride 1, same padding (kw-1)/2 (dall_e/utils.py:37-48) as implicit GEMM:
 * x [B*H*W, Cin], w [Cout, kw*kw*Cin] (tap-major, channel-minor), zero_page = >= 128 zero bytes.
 * Epilogues: VLMO_EPI_BIAS (relu flag), VLMO_EPI_DUAL (EncoderBlock tail, encoder.py:45-46), VLMO_EPI_F32. */
int vlmo_conv2d_nhwc(int epi, int dtype, const void* x, int B, int H, int W, int Cin, int kw,
                     const void* w, int Cout, const void* zero_page, const VlmoEpilogue* e,
                     hipStream_t stream);
/* stem input: image f32 NCHW -> f16 [B*H*W, Kpad] patches, column = c*kw*kw + ky*kw + kx. */
int vlmo_dvae_im2col(const float* x, void* out, int B, int C, int H, int W, int kw, int Kpad,
                     hipStream_t stream);
/* MaxPool2d(2) (encoder.py:85,95,105): raw pooled map + relu of it (relu may be NULL). */
int vlmo_maxpool2_nhwc(const void* x, void* raw, void* relu, int B, int H, int W, int C,
                       hipStream_t stream);
/* finish of VLMO_EPI_CE: partial [M, nchunk, 4] -> lse [M], loss [M] (lse - label logit; 0 where
 * labels[m] == ignore_index; may be NULL), pred [M] (arg-max, first maximum wins; may be NULL). */
int vlmo_ce_reduce(const float* partial, int nchunk, const int32_t* labels, int ignore_index, float* lse,
                   float* loss, int32_t* pred, int M, hipStream_t stream);
/* final step of Dalle_VAE.get_codebook_indices (modeling_discrete_vae.py:246-248):
 * partial [M, nchunk, 2] from VLMO_EPI_ARGMAX -> ids int64 [M] (first maximum wins). */
int vlmo_argmax_reduce(const float* partial, int nchunk, int64_t* ids, int M, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VLMO_HIP_H */
