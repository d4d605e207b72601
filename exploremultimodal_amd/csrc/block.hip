// One VLMo Block forward / backward per C-ABI call (vlmo.py:187-197): native orchestration
// of the kernels in gemm.hip / attention.hip / layernorm.hip / elementwise.hip, so the Python
// host issues one FFI call per block and stays far ahead of the GPU.
#include "common.h"
#include "vlmo_hip.h"
#include <algorithm>
#include <map>
#include <vector>

#include <stdlib.h>
namespace {
// VlmoEpilogue.relu bit 2: the fc1 epilogue saves GELU'(u) * dropout mask / (1 - p) in place of the pre-activation u, and the
// GELU-derivative epilogue of dgrad_fc2 multiplies by it (measurement aid: VLMO_SAVE_GELU_DERIV=0 = the pre-activation path)
int gelu_deriv_flag() {
    static const int v = [] {
        const char* e = getenv("VLMO_SAVE_GELU_DERIV");
        return (e && e[0] == '0') ? 0 : 4;
    }();
    return v;
}


// fork / join events of the calling thread, one set per device (a process may drive several GPUs)
struct Events {
    hipEvent_t fork = nullptr, join = nullptr;
    std::vector<hipEvent_t> pool;      // vlmo_stack_bwd: fork / done event per deferred batch
    hipEvent_t get(size_t i) {
        while (pool.size() <= i) {
            hipEvent_t e = nullptr;
            (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
            pool.push_back(e);
        }
        return pool[i];
    }
};
Events& events() {
    static thread_local std::map<int, Events> per_device;
    int dev = 0;
    (void)hipGetDevice(&dev);
    Events& e = per_device[dev];
    if (!e.fork) {
        (void)hipEventCreateWithFlags(&e.fork, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&e.join, hipEventDisableTiming);
    }
    return e;
}

inline const char* bp(const void* p, size_t row, size_t ld, size_t esz) { return (const char*)p + row * ld * esz; }
inline char* bp(void* p, size_t row, size_t ld, size_t esz) { return (char*)p + row * ld * esz; }

#define TRY(x)              \
    do {                    \
        int rc__ = (x);     \
        if (rc__) {         \
            vlmo_defer_reduce = nullptr; \
            return rc__;    \
        }                   \
    } while (0)

VlmoEpilogue epi() {
    VlmoEpilogue e{};
    e.inv_keep = 1.f;
    return e;
}

}  // namespace

extern "C" int vlmo_block_fwd(const VlmoBlockDesc* b, hipStream_t st) {
    VLMO_CHECK_ARG(b && b->x && b->x1 && b->x2, "vlmo_block_fwd: null descriptor/buffers");
    VLMO_CHECK_ARG(b->n_experts >= 1 && b->n_experts <= 2 && b->n_attn >= 1 && b->n_attn <= 2, "vlmo_block_fwd: bad counts");
    const int M = b->M, d = b->d, hid = b->hidden;
    TRY(vlmo_ln_fwd(b->x, b->n1w, b->n1b, b->y1, 0, b->mean1, b->rstd1, nullptr, M, d, b->eps, st));
    {
        VlmoEpilogue e = epi();
        e.out = b->qkv;
        e.ldo = 3 * d;
        e.bias = b->qkv_bias;
        TRY(vlmo_gemm_nt(VLMO_EPI_BIAS, VLMO_BF16, b->tile, b->y1, d, b->qkv_w, d, M, 3 * d, d, &e, st));
    }
    const float scale = 1.0f / sqrtf((float)(d / b->heads));
    for (int a = 0; a < b->n_attn; ++a)
        TRY(vlmo_attn_fwd(b->qkv, b->seg[a], b->nseq[a], b->keymask, b->ctx, b->lse[a], b->lse_stride[a], b->heads, d,
                          b->maxlen[a], scale, b->attn_drop_thresh, b->attn_inv_keep, b->seed + 11 + b->attn_seed_idx[a], b->attn_seq0[a], st));
    {
        VlmoEpilogue e = epi();
        e.out = b->x1;
        e.ldo = d;
        e.out2 = b->need_bwd ? b->zd1 : nullptr;
        e.ld2 = d;
        e.bias = b->proj_b;
        e.gamma = b->g1;
        e.resid = b->x;
        e.row_scale = b->rs1;
        e.row_index = b->row_index;
        e.drop_thresh = b->drop_thresh;
        e.inv_keep = b->inv_keep;
        e.seed = b->seed + 1;
        TRY(vlmo_gemm_nt(VLMO_EPI_RESID, VLMO_BF16, b->tile, b->ctx, d, b->proj_w, d, M, d, d, &e, st));
    }
    TRY(vlmo_ln_fwd(b->x1, b->n2w, b->n2b, b->y2, 0, b->mean2, b->rstd2, nullptr, M, d, b->eps, st));
    // expert FFNs: all experts of the block in ONE grouped launch per linear (row ranges, weights and biases
    // differ per expert; a launch costs at least one tile time, so per-expert launches of the 4 096 text rows
    // and the 12 608 image rows would cost two)
    const void *a1[4], *w1[4], *a2[4], *w2[4];
    int32_t rows[4];
    VlmoEpilogue e1[4], e2[4];
    for (int x = 0; x < b->n_experts; ++x) {
        const size_t r0 = b->exp_row0[x];
        rows[x] = b->exp_rows[x];
        a1[x] = bp(b->y2, r0, d, 2);
        w1[x] = b->w1[x];
        VlmoEpilogue& e = e1[x] = epi();
        e.out = bp(b->u, r0, hid, 2);
        e.out2 = bp(b->h, r0, hid, 2);
        e.ldo = e.ld2 = hid;
        e.bias = b->b1[x];
        e.drop_thresh = b->drop_thresh;
        e.inv_keep = b->inv_keep;
        e.seed = b->seed + 20 + 2 * x;
        e.relu = gelu_deriv_flag();         // `u` holds GELU'(u) * mask / (1 - p): see the backward's EPI_DGELU
        a2[x] = bp(b->h, r0, hid, 2);
        w2[x] = b->w2[x];
        VlmoEpilogue& f = e2[x] = epi();
        f.out = bp(b->x2, r0, d, 4);
        f.ldo = d;
        f.out2 = b->need_bwd ? bp(b->zd2, r0, d, 2) : nullptr;
        f.ld2 = d;
        f.bias = b->b2[x];
        f.gamma = b->g2;
        f.resid = (const float*)bp((const void*)b->x1, r0, d, 4);
        f.row_scale = b->row_index ? b->rs2 : (b->rs2 ? b->rs2 + r0 : nullptr);
        f.row_index = b->row_index ? b->row_index + r0 : nullptr;
        f.drop_thresh = b->drop_thresh;
        f.inv_keep = b->inv_keep;
        f.seed = b->seed + 21 + 2 * x;
    }
    TRY(vlmo_gemm_nt_grouped(VLMO_EPI_BIAS_GELU, VLMO_BF16, b->tile, b->n_experts, a1, d, w1, d, rows, hid, d, e1, st));
    TRY(vlmo_gemm_nt_grouped(VLMO_EPI_RESID, VLMO_BF16, b->tile, b->n_experts, a2, hid, w2, hid, rows, d, hid, e2, st));
    return 0;
}

extern "C" int vlmo_side_stream_create(int low_priority, const uint32_t* cu_mask, int cu_mask_words, hipStream_t* out) {
    VLMO_CHECK_ARG(out, "vlmo_side_stream_create: null out");
    hipError_t rc;
    if (cu_mask && cu_mask_words > 0) {
        rc = hipExtStreamCreateWithCUMask(out, (uint32_t)cu_mask_words, cu_mask);
    } else {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        rc = hipStreamCreateWithPriority(out, hipStreamNonBlocking, low_priority ? least : 0);
    }
    if (rc != hipSuccess) {
        vlmo_set_error("vlmo_side_stream_create: %s", hipGetErrorString(rc));
        return (int)rc;
    }
    return 0;
}

extern "C" int vlmo_block_bwd(const VlmoBlockDesc* b, hipStream_t st) {
    VLMO_CHECK_ARG(b && b->dx2 && b->dx1 && b->dx0, "vlmo_block_bwd: null descriptor/buffers");
    const int M = b->M, d = b->d, hid = b->hidden;
    hipStream_t side = b->side_stream ? b->side_stream : st;
    float* ws_side = b->side_stream ? b->ws_side : b->ws_main;
    Events& ev = events();
    auto fork = [&]() {     // side stream may read everything enqueued on the main stream so far
        if (side != st) {
            (void)hipEventRecord(ev.fork, st);
            (void)hipStreamWaitEvent(side, ev.fork, 0);
        }
    };
    // The column folds of the LayerNorm / LayerScale / bias gradients are deferred to the side stream (after
    // the next fork) when the workspace has a slot per producer; else they run in place on the main stream.
    const int64_t slot = reduce_ws_need(2 * d);
    const bool defer = side != st && b->ws_bytes >= (int64_t)(3 + b->n_experts) * slot;
    auto ws_slot = [&](int k) { return defer ? (float*)((char*)b->ws_main + k * slot) : b->ws_main; };
    PartialReduce pend[6];
    auto arm = [&](int k) { vlmo_defer_reduce = defer ? &pend[k] : nullptr; };
    // ---- FFN half: per-expert residual-branch backward, then the two dgrad GEMMs of ALL experts as grouped
    // launches on the main stream and every expert's weight gradients on the side stream
    VLMO_CHECK_ARG(b->n_experts >= 1 && b->n_experts <= 2, "vlmo_block_bwd: 1..2 experts per block");
    const void *ag[4], *wg[4], *af[4], *wf[4];
    int32_t rows[4];
    VlmoEpilogue eg[4], ef[4];
    for (int x = 0; x < b->n_experts; ++x) {
        const size_t r0 = b->exp_row0[x];
        const int n = rows[x] = b->exp_rows[x];
        arm(2 + x);
        TRY(vlmo_resid_bwd(b->dx2 + r0 * d, bp(b->zd2, r0, d, 2), b->g2,
                           b->row_index ? b->rs2 : (b->rs2 ? b->rs2 + r0 : nullptr),
                           b->row_index ? b->row_index + r0 : nullptr, bp(b->dz2, r0, d, 2), b->dg2, b->db2[x], n, d, b->drop_thresh, b->inv_keep,
                           b->seed + 21 + 2 * x, ws_slot(2 + x), slot, st));
        vlmo_defer_reduce = nullptr;
        ag[x] = bp(b->dz2, r0, d, 2);
        wg[x] = b->w2T[x];
        VlmoEpilogue& e = eg[x] = epi();
        e.out = bp(b->du, r0, hid, 2);
        e.ldo = hid;
        e.aux = bp(b->u, r0, hid, 2);
        e.ld2 = hid;
        e.drop_thresh = b->drop_thresh;
        e.inv_keep = b->inv_keep;
        e.seed = b->seed + 20 + 2 * x;
        e.relu = gelu_deriv_flag();
        af[x] = bp(b->du, r0, hid, 2);
        wf[x] = b->w1T[x];
        VlmoEpilogue& f = ef[x] = epi();
        f.out = bp(b->dy2, r0, d, 2);
        f.ldo = d;
    }
    TRY(vlmo_gemm_nt_grouped(VLMO_EPI_DGELU, VLMO_BF16, b->tile, b->n_experts, ag, d, wg, d, rows, hid, d, eg, st));
    fork();
    for (int x = 0; x < b->n_experts; ++x) {
        const size_t r0 = b->exp_row0[x];
        const int n = b->exp_rows[x];
        TRY(reduce_partials(pend[2 + x], side));
        TRY(vlmo_gemm_tn(VLMO_BF16, bp(b->dz2, r0, d, 2), d, bp(b->h, r0, hid, 2), hid, b->dw2[x], hid, n, d, hid, 1.f, 0, b->ws_tn, b->ws_tn_bytes, side));
        TRY(vlmo_colsum(VLMO_BF16, bp(b->du, r0, hid, 2), hid, b->db1[x], n, hid, ws_side, b->ws_bytes, side));
        TRY(vlmo_gemm_tn(VLMO_BF16, bp(b->du, r0, hid, 2), hid, bp(b->y2, r0, d, 2), d, b->dw1[x], d, n, hid, d, 1.f, 0, b->ws_tn, b->ws_tn_bytes, side));
    }
    TRY(vlmo_gemm_nt_grouped(VLMO_EPI_BIAS, VLMO_BF16, b->tile, b->n_experts, af, hid, wf, hid, rows, d, hid, ef, st));
    // norm2 backward fused with the attention branch's residual backward (it consumes the dx1 this writes):
    // four column partials (dn2w, dn2b, dgamma_1, dproj_b) in slots 0-1
    arm(0);
    TRY(vlmo_ln_resid_bwd(b->dy2, b->x1, b->n2w, b->mean2, b->rstd2, b->dx2, b->dx1, b->dn2w, b->dn2b, b->zd1, b->g1,
                          b->rs1, b->row_index, b->dz1, b->dg1, b->dproj_b, b->drop_thresh, b->inv_keep, b->seed + 1, M, d,
                          ws_slot(0), 2 * slot, st));
    vlmo_defer_reduce = nullptr;
    {
        VlmoEpilogue e = epi();
        e.out = b->dctx;
        e.ldo = d;
        TRY(vlmo_gemm_nt(VLMO_EPI_BIAS, VLMO_BF16, b->tile, b->dz1, d, b->proj_wT, d, M, d, d, &e, st));
    }
    const float scale = 1.0f / sqrtf((float)(d / b->heads));
    for (int a = 0; a < b->n_attn; ++a)
        TRY(vlmo_attn_bwd(b->qkv, b->ctx, b->dctx, b->lse[a], b->lse_stride[a], b->seg[a], b->nseq[a], b->keymask,
                          b->dqkv, nullptr, b->heads, d, b->maxlen[a], scale, b->attn_drop_thresh, b->attn_inv_keep,
                          b->seed + 11 + b->attn_seed_idx[a], b->attn_seq0[a], st));
    fork();     // one fork for the whole attention half: the proj gradient waits for it too (the side stream has slack)
    TRY(reduce_partials(pend[0], side));
    TRY(vlmo_gemm_tn(VLMO_BF16, b->dz1, d, b->ctx, d, b->dproj_w, d, M, d, d, 1.f, 0, b->ws_tn, b->ws_tn_bytes, side));
    TRY(vlmo_colsum(VLMO_BF16, b->dqkv, 3 * d, b->dqkv_b, M, 3 * d, ws_side, b->ws_bytes, side));
    TRY(vlmo_gemm_tn(VLMO_BF16, b->dqkv, 3 * d, b->y1, d, b->dqkv_w, d, M, 3 * d, d, 1.f, 0, b->ws_tn, b->ws_tn_bytes, side));
    {
        VlmoEpilogue e = epi();
        e.out = b->dy1;
        e.ldo = d;
        TRY(vlmo_gemm_nt(VLMO_EPI_BIAS, VLMO_BF16, b->tile, b->dqkv, 3 * d, b->qkv_wT, 3 * d, M, d, 3 * d, &e, st));
    }
    // the last fold stays on the main stream (a fork for it would cost what it saves); its own slot: the
    // side stream may still be reading slot 0
    TRY(vlmo_ln_bwd(b->dy1, 0, nullptr, b->x, b->n1w, b->mean1, b->rstd1, b->dx1, b->dx0, b->dn1w, b->dn1b, M, d,
                    ws_slot(2 + b->n_experts), slot, st));
    if (side != st) {       // join: gradients complete, every buffer the side stream read is reusable
        (void)hipEventRecord(ev.join, side);
        (void)hipStreamWaitEvent(st, ev.join, 0);
    }
    return 0;
}


// ================================================================ whole block stacks in one call
// vlmo_stack_fwd / vlmo_stack_bwd run the blocks of one backbone pass (vlmo.py:402-411) from native code:
// one FFI call per pass and direction.  The backward enqueues the activation-gradient chains of all blocks
// back to back on the caller's stream and DEFERS every block's parameter-gradient work -- the four to six
// weight-gradient GEMMs and all bias / layer-scale / LayerNorm-weight column sums -- to the side stream in
// batches of `wgrad_batch` blocks: ONE vlmo_gemm_tn_multi launch (216 output tiles for two VLMo-Base blocks:
// the chip is full without splitting the token dimension) and ONE vlmo_colwork_multi launch per batch.
namespace {

struct Deferred {
    std::vector<VlmoTnProblem> tn;
    std::vector<VlmoColJob> col;
    std::vector<int> blocks;
    bool store = false;         // weight-gradient matrices are written, not accumulated (VlmoStackDesc.wgrad_store)
};

void push_fold(Deferred& D, const PartialReduce& r) {
    if (!r.ws) return;
    VlmoColJob j{};
    j.kind = 0;
    j.ld = r.ncols;
    j.src = r.ws;
    j.rows = r.nblk;
    j.ncols = r.ncols;
    j.out[0] = r.out0, j.out[1] = r.out1, j.out[2] = r.out2, j.out[3] = r.out3;
    j.n0 = r.n0;
    D.col.push_back(j);
}
void push_colsum(Deferred& D, const void* x, int ld, int rows, int ncols, float* out) {
    VlmoColJob j{};
    j.kind = 1;
    j.ld = ld;
    j.src = x;
    j.rows = rows;
    j.ncols = ncols;
    j.out[0] = out;
    j.n0 = ncols;
    D.col.push_back(j);
}
void push_tn(Deferred& D, const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N1, int N2) {
    D.tn.push_back(VlmoTnProblem{A, B, C, lda, ldb, ldc, M, N1, N2, 1.f, D.store ? 0 : 1});
}

// activation-gradient chain of one block on `st`; parameter-gradient work is appended to D.
// `below` = the block processed next (the one under this one in the stack), or NULL: when its FFN residual branch can
// take its incoming gradient straight from this block's last kernel, norm1's backward and that branch's backward run as
// ONE kernel (the 51 MB fp32 gradient is not re-read) and *fused_below is set; the chain of `below` is then started
// with skip_resid.
int block_dgrad_chain(const VlmoBlockDesc* b, hipStream_t st, Deferred& D, const VlmoBlockDesc* below, bool skip_resid,
                      bool* fused_below) {
    const int M = b->M, d = b->d, hid = b->hidden;
    const int64_t slot = reduce_ws_need(2 * d);
    VLMO_CHECK_ARG(b->dx2 && b->dx1 && b->dx0, "vlmo_stack_bwd: null gradient buffers");
    VLMO_CHECK_ARG(b->n_experts >= 1 && b->n_experts <= 2, "vlmo_stack_bwd: 1..2 experts per block");
    VLMO_CHECK_ARG(b->ws_main && b->ws_bytes >= (int64_t)(3 + b->n_experts) * slot,
                   "vlmo_stack_bwd: column workspace too small (need %lld bytes per block)",
                   (long long)((3 + b->n_experts) * slot));
    auto ws_slot = [&](int k) { return (float*)((char*)b->ws_main + k * slot); };
    PartialReduce pend;
    auto arm = [&]() {
        pend = PartialReduce{};
        vlmo_defer_reduce = &pend;
    };
    auto collect = [&]() {
        vlmo_defer_reduce = nullptr;
        push_fold(D, pend);
    };
    const void *ag[4], *wg[4], *af[4], *wf[4];
    int32_t rows[4];
    VlmoEpilogue eg[4], ef[4];
    static const bool fold_db1 = !getenv("VLMO_FOLD_DB1") || atoi(getenv("VLMO_FOLD_DB1")) != 0;   // measurement aid
    int64_t colpart_off = (int64_t)(4 + b->n_experts) * slot / 4;       // floats (behind the fold slots)
    for (int x = 0; x < b->n_experts; ++x) {
        const size_t r0 = b->exp_row0[x];
        const int n = rows[x] = b->exp_rows[x];
        if (!skip_resid) {
            arm();
            TRY(vlmo_resid_bwd(b->dx2 + r0 * d, bp(b->zd2, r0, d, 2), b->g2,
                               b->row_index ? b->rs2 : (b->rs2 ? b->rs2 + r0 : nullptr),
                               b->row_index ? b->row_index + r0 : nullptr, bp(b->dz2, r0, d, 2), b->dg2, b->db2[x], n, d,
                               b->drop_thresh, b->inv_keep, b->seed + 21 + 2 * x, ws_slot(2 + x), slot, st));
            collect();
        }
        ag[x] = bp(b->dz2, r0, d, 2);
        wg[x] = b->w2T[x];
        VlmoEpilogue& e = eg[x] = epi();
        e.out = bp(b->du, r0, hid, 2);
        e.ldo = hid;
        e.aux = bp(b->u, r0, hid, 2);
        e.ld2 = hid;
        e.drop_thresh = b->drop_thresh;
        e.inv_keep = b->inv_keep;
        e.seed = b->seed + 20 + 2 * x;
        e.relu = gelu_deriv_flag();
        // fc1 bias gradient: the DGELU epilogue leaves column-sum partials (one row per 16 output rows, see
        // VlmoEpilogue.colpart) behind the (4 + experts) fold slots when the workspace has room; they join the block's
        // other column folds.  Else: a pass over du.
        const int nblk64 = (n + 15) / 16;       // 16-row blocks
        if (fold_db1 && b->db1[x] && (colpart_off + (int64_t)nblk64 * hid) * 4 <= b->ws_bytes) {
            e.colpart = (float*)b->ws_main + colpart_off;
            colpart_off += (int64_t)nblk64 * hid;
            VlmoColJob j{};
            j.kind = 0;
            j.ld = hid;
            j.src = e.colpart;
            j.rows = nblk64;
            j.ncols = hid;
            j.out[0] = b->db1[x];
            j.n0 = hid;
            D.col.push_back(j);
        } else {
            push_colsum(D, bp(b->du, r0, hid, 2), hid, n, hid, b->db1[x]);
        }
        af[x] = bp(b->du, r0, hid, 2);
        wf[x] = b->w1T[x];
        VlmoEpilogue& f = ef[x] = epi();
        f.out = bp(b->dy2, r0, d, 2);
        f.ldo = d;
        push_tn(D, bp(b->dz2, r0, d, 2), d, bp(b->h, r0, hid, 2), hid, b->dw2[x], hid, n, d, hid);
        push_tn(D, bp(b->du, r0, hid, 2), hid, bp(b->y2, r0, d, 2), d, b->dw1[x], d, n, hid, d);
    }
    TRY(vlmo_gemm_nt_grouped(VLMO_EPI_DGELU, VLMO_BF16, b->tile, b->n_experts, ag, d, wg, d, rows, hid, d, eg, st));
    TRY(vlmo_gemm_nt_grouped(VLMO_EPI_BIAS, VLMO_BF16, b->tile, b->n_experts, af, hid, wf, hid, rows, d, hid, ef, st));
    arm();
    TRY(vlmo_ln_resid_bwd(b->dy2, b->x1, b->n2w, b->mean2, b->rstd2, b->dx2, b->dx1, b->dn2w, b->dn2b, b->zd1, b->g1,
                          b->rs1, b->row_index, b->dz1, b->dg1, b->dproj_b, b->drop_thresh, b->inv_keep, b->seed + 1, M, d,
                          ws_slot(0), 2 * slot, st));
    collect();
    {
        VlmoEpilogue e = epi();
        e.out = b->dctx;
        e.ldo = d;
        TRY(vlmo_gemm_nt(VLMO_EPI_BIAS, VLMO_BF16, b->tile, b->dz1, d, b->proj_wT, d, M, d, d, &e, st));
    }
    const float scale = 1.0f / sqrtf((float)(d / b->heads));
    // q_bias / v_bias gradient = column sums of the q and v thirds of dqkv (the k third of the bias is a constant zero,
    // vlmo.py:71-75, and its gradient vanishes anyway: the soft-max is invariant to a shift of the scores along the
    // keys).  The attention backward leaves them per sequence ([sequences][2d], behind the other column partials of
    // this block's workspace) and the batched column kernel folds 64-128 rows instead of reading 51 MB of dqkv again.
    int nseq_all = 0;
    for (int a = 0; a < b->n_attn; ++a) nseq_all += b->nseq[a];
    float* qvsum = nullptr;
    if ((colpart_off + (int64_t)nseq_all * 2 * d) * 4 <= b->ws_bytes) qvsum = (float*)b->ws_main + colpart_off;
    for (int a = 0, s0 = 0; a < b->n_attn; s0 += b->nseq[a], ++a)
        TRY(vlmo_attn_bwd(b->qkv, b->ctx, b->dctx, b->lse[a], b->lse_stride[a], b->seg[a], b->nseq[a], b->keymask,
                          b->dqkv, qvsum ? qvsum + (size_t)s0 * 2 * d : nullptr, b->heads, d, b->maxlen[a], scale,
                          b->attn_drop_thresh, b->attn_inv_keep, b->seed + 11 + b->attn_seed_idx[a], b->attn_seq0[a], st));
    push_tn(D, b->dz1, d, b->ctx, d, b->dproj_w, d, M, d, d);
    push_tn(D, b->dqkv, 3 * d, b->y1, d, b->dqkv_w, d, M, 3 * d, d);
    if (qvsum) {
        VlmoColJob j{};
        j.kind = 0;
        j.ld = 2 * d;
        j.src = qvsum;
        j.rows = nseq_all;
        j.ncols = 2 * d;
        j.out[0] = b->dqkv_b, j.out[1] = b->dqkv_b + 2 * d;
        j.n0 = d;
        D.col.push_back(j);
    } else {
        push_colsum(D, b->dqkv, 3 * d, M, d, b->dqkv_b);
        push_colsum(D, bp(b->dqkv, 0, 0, 2) + 4 * (size_t)d, 3 * d, M, d, b->dqkv_b + 2 * d);
    }
    {
        VlmoEpilogue e = epi();
        e.out = b->dy1;
        e.ldo = d;
        TRY(vlmo_gemm_nt(VLMO_EPI_BIAS, VLMO_BF16, b->tile, b->dqkv, 3 * d, b->qkv_wT, 3 * d, M, d, 3 * d, &e, st));
    }
    static const bool fuse_below = !getenv("VLMO_FUSE_BELOW") || atoi(getenv("VLMO_FUSE_BELOW")) != 0;   // measurement aid
    const VlmoBlockDesc* n = fuse_below ? below : nullptr;
    *fused_below = false;
    if (n && n->M == M && n->d == d && n->dx2 == b->dx0 && n->zd2 && n->dz2 && n->n_experts >= 1 && n->n_experts <= 2 &&
        n->exp_row0[0] == 0 && (n->n_experts == 1 ? n->exp_rows[0] == M
                                                   : (n->exp_row0[1] == n->exp_rows[0] && n->exp_rows[0] + n->exp_rows[1] == M)) &&
        b->ws_bytes >= (int64_t)(4 + b->n_experts) * slot) {
        const int seg = n->n_experts == 1 ? M : n->exp_row0[1];
        float* wsf = ws_slot(2 + b->n_experts);        // [workgroups][4d]: two slots
        int nb0 = 0, nblk = 0;
        arm();
        TRY(ln_resid_seg_bwd(b->dy1, b->x, b->n1w, b->mean1, b->rstd1, b->dx1, b->dx0, b->dn1w, b->dn1b, n->zd2, n->g2,
                             n->rs2, n->row_index, n->dz2, n->dg2, n->drop_thresh, n->inv_keep, n->seed + 21, n->seed + 23,
                             seg, M, d, wsf, 2 * slot, &nb0, &nblk, st));
        collect();
        // the FFN bias gradients of the block below: columns [3d, 4d) of the partial rows of each segment
        for (int x = 0; x < n->n_experts; ++x) {
            if (!n->db2[x]) continue;
            VlmoColJob j{};
            j.kind = 0;
            j.ld = 4 * d;
            j.src = wsf + 3 * d + (x ? (size_t)nb0 * 4 * d : 0);
            j.rows = x ? nblk - nb0 : nb0;
            j.ncols = d;
            j.out[0] = n->db2[x];
            j.n0 = d;
            if (j.rows > 0) D.col.push_back(j);
        }
        *fused_below = true;
        return 0;
    }
    arm();
    TRY(vlmo_ln_bwd(b->dy1, 0, nullptr, b->x, b->n1w, b->mean1, b->rstd1, b->dx1, b->dx0, b->dn1w, b->dn1b, M, d,
                    ws_slot(2 + b->n_experts), slot, st));
    collect();
    return 0;
}

}  // namespace

extern "C" int vlmo_stack_fwd(const VlmoStackDesc* s, hipStream_t st) {
    VLMO_CHECK_ARG(s && s->blocks && s->n_blocks >= 1, "vlmo_stack_fwd: empty stack");
    for (int i = 0; i < s->n_blocks; ++i)
        if (int rc = vlmo_block_fwd(&s->blocks[i], st)) return rc;
    return 0;
}

extern "C" int vlmo_stack_bwd(const VlmoStackDesc* s, hipStream_t st) {
    VLMO_CHECK_ARG(s && s->blocks && s->n_blocks >= 1, "vlmo_stack_bwd: empty stack");
    const int nb = s->n_blocks;
    const int batch = s->wgrad_batch >= 1 ? s->wgrad_batch : 1;
    const int nsets = s->n_tmp_sets >= 1 ? s->n_tmp_sets : 1;
    hipStream_t side = s->side_stream ? s->side_stream : st;
    const bool two = side != st;
    VLMO_CHECK_ARG(!two || nsets > batch || nb <= nsets,
                   "vlmo_stack_bwd: %d temporary sets cannot cover deferred batches of %d blocks", nsets, batch);
    Events& ev = events();
    Deferred D;
    D.store = s->wgrad_store != 0;
    std::vector<int> batch_of(nb, -1);
    int n_batches = 0, waited_upto = -1;
    auto flush = [&]() -> int {
        if (D.blocks.empty()) return 0;
        if (two) {
            hipEvent_t f = ev.get(2 * n_batches);
            (void)hipEventRecord(f, st);
            (void)hipStreamWaitEvent(side, f, 0);
        }
        // longest reductions first: with more tiles than CUs the short ones back-fill
        std::stable_sort(D.tn.begin(), D.tn.end(), [](const VlmoTnProblem& a, const VlmoTnProblem& b) { return a.M > b.M; });
        TRY(vlmo_gemm_tn_multi(VLMO_BF16, D.tn.data(), (int)D.tn.size(), side));
        TRY(vlmo_colwork_multi(VLMO_BF16, D.col.data(), (int)D.col.size(), side));
        if (two) (void)hipEventRecord(ev.get(2 * n_batches + 1), side);
        if (s->grad_ready)
            for (int i : D.blocks)
                if (s->grad_ready[i]) (void)hipEventRecord((hipEvent_t)s->grad_ready[i], side);
        ++n_batches;
        D.tn.clear();
        D.col.clear();
        D.blocks.clear();
        return 0;
    };
    auto set_free = [&](int kk) -> int {
        // block kk (processing order) reuses the temporaries of the block processed nsets steps earlier: that block's
        // deferred work must be done
        if (!two || kk < nsets || kk >= nb) return 0;
        int need = batch_of[kk - nsets];
        if (need < 0 || need >= n_batches) {
            TRY(flush());
            need = n_batches - 1;
        }
        if (need >= 0 && need > waited_upto) {
            (void)hipStreamWaitEvent(st, ev.get(2 * need + 1), 0);
            waited_upto = need;
        }
        return 0;
    };
    bool skip_resid = false;
    for (int k = 0; k < nb; ++k) {
        const int i = nb - 1 - k;
        TRY(set_free(k));
        TRY(set_free(k + 1));       // the last kernel of this chain may already write dz2 of the block below
        bool fused = false;
        TRY(block_dgrad_chain(&s->blocks[i], st, D, i > 0 ? &s->blocks[i - 1] : nullptr, skip_resid, &fused));
        skip_resid = fused;
        D.blocks.push_back(i);
        batch_of[k] = n_batches;
        if ((int)D.blocks.size() >= batch || k == nb - 1) TRY(flush());
    }
    if (two && n_batches > 0) (void)hipStreamWaitEvent(st, ev.get(2 * (n_batches - 1) + 1), 0);
    return 0;
}

extern "C" int vlmo_event_create(void** out) {
    VLMO_CHECK_ARG(out, "vlmo_event_create: null out");
    hipEvent_t e = nullptr;
    hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) {
        vlmo_set_error("vlmo_event_create: %s", hipGetErrorString(rc));
        return (int)rc;
    }
    *out = e;
    return 0;
}
extern "C" int vlmo_event_destroy(void* ev) { return ev ? (int)hipEventDestroy((hipEvent_t)ev) : 0; }
extern "C" int vlmo_stream_wait_event(hipStream_t stream, void* ev) {
    VLMO_CHECK_ARG(ev, "vlmo_stream_wait_event: null event");
    return (int)hipStreamWaitEvent(stream, (hipEvent_t)ev, 0);
}
