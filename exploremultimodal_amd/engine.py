"""Autograd glue between the nn.Module mirror (vlmo.py) and the HIP C-ABI.

Layout ("packed rows"): the token matrix of one backbone pass is [M, d] with
all text tokens of the batch first (B*T rows), then all image tokens (B*P rows).
LayerNorm / linear layers do not care about row order, so below the fusion
layer the shared-weight ops (norm1, qkv, proj, norm2) run ONCE over both
modalities and only the expert FFNs and attention look at row ranges; above it
attention finds a fused sequence through a two-segment descriptor, so the
reference's torch.cat([txt, img], dim=1) (vlmo.py:406) never materialises.

The residual stream is fp32 (as under the reference's autocast), GEMM operands
bf16 with fp32 accumulation, parameters fp32 masters with cached bf16 shadows.
"""
import ctypes
import sys

import torch

from . import hip

LN_EPS = 1e-12          # vlmo_module.py:21-23
GRAD_SINK = None         # set by dp.GradReducer: block gradients are accumulated straight into its flat buckets
import os as _os
# Weight-gradient GEMMs + bias column sums on a side stream: '1' always, '0' never, unset = by the rule below.
# The GEMM kernels take whole CUs (128 KB of LDS, every vector register), so two streams time-slice the chip instead of
# sharing it: the overlap pays only where the main stream's kernels leave CUs idle (short kernels, partial dispatch
# rounds).  In-session A/Bs, side stream against one stream, no reducer: VLMo-Base at 64 pairs (16 704 rows per pass)
# 14.70 / 14.66 / 14.66 against 14.63 / 14.64 / 14.64 ms and 14.25 / 14.25 against 14.18 / 14.20 on another box; the
# four-loss objective at 32 pairs (passes of 33 408 and 12 608 rows) 48.1 / 47.9 against 47.3 / 47.3; VLMo-Large at 32
# pairs (8 352 rows) 25.02 / 25.03 against 25.58 / 25.55.  So: one stream for passes of ONE_STREAM_ROWS rows and more, the
# side stream below -- and always under a gradient reducer, where the side stream is what lets a block's gradients
# finish (and their collective start) while the activation gradients of the blocks below are still being computed
# (dp.GradReducer waits on the per-block grad_ready events).
_ov = _os.environ.get('VLMO_OVERLAP_WGRAD')
OVERLAP_WGRAD = None if _ov is None else _ov != '0'
ONE_STREAM_ROWS = 12288


def _use_side_stream(sink, rows):
    if OVERLAP_WGRAD is not None:
        return OVERLAP_WGRAD
    return sink is not None or rows < ONE_STREAM_ROWS
SIDE_MODE = _os.environ.get('VLMO_SIDE_STREAM', 'low')
PROBE_STREAMS = _os.environ.get('VLMO_PROBE_STREAMS', '1') != '0'
MERGE_SEPARATE_ATTENTION = _os.environ.get('VLMO_MERGE_ATTN', '1') != '0'
DEFAULT_TILE = -1        # GEMM tile: -1 = chosen per shape by the library (see vlmo_gemm_nt)


class ShadowCache:
    """bf16 copies (W and W^T) of fp32 master weights, refreshed when the
    parameter's version counter changes (optimizer step / load_state_dict)."""

    def __init__(self):
        self._c = {}

    def get(self, p, need_t=True):
        key = id(p)
        ent = self._c.get(key)
        ver = (p._version, p.data_ptr())
        if ent is None or ent[0] != ver or (need_t and ent[2] is None):
            w2 = p.detach().reshape(p.shape[0], -1)
            w = torch.empty(w2.shape, dtype=torch.bfloat16, device=p.device)
            wt = torch.empty((w2.shape[1], w2.shape[0]), dtype=torch.bfloat16, device=p.device) if need_t else None
            hip.cast_weight(w2, w, wt)
            ent = (ver, w, wt)
            self._c[key] = ent
        return ent[1], ent[2]

    def refresh(self, params):
        """Bring the shadows of `params` (2-D weights, W and W^T) up to date in ONE launch: after an optimizer step every
        weight is stale, and one cast launch per weight was 0.65 ms of an 18 ms training step.  Buffers of an unchanged
        shape are overwritten in place (the cast is ordered behind the step's kernels on the caller's stream)."""
        jobs = []
        for p in params:
            key = id(p)
            ent = self._c.get(key)
            ver = (p._version, p.data_ptr())
            if ent is not None and ent[0] == ver and ent[2] is not None:
                continue
            w2 = p.detach().reshape(p.shape[0], -1)
            # in place only when nobody else holds the pair: a graph whose forward ran BEFORE the parameter update keeps its
            # shadows for its input-gradient GEMMs (ctx.keep), and overwriting them would make that backward use the NEW
            # weights (forward A, optimizer step, forward B, backward A).  The cache entry is the only other owner.
            if (ent is not None and ent[1].shape == w2.shape and ent[2] is not None
                    and sys.getrefcount(ent[1]) <= 2 and sys.getrefcount(ent[2]) <= 2):
                w, wt = ent[1], ent[2]
            else:
                w = torch.empty(w2.shape, dtype=torch.bfloat16, device=p.device)
                wt = torch.empty((w2.shape[1], w2.shape[0]), dtype=torch.bfloat16, device=p.device)
            jobs.append((w2, w, wt))
            self._c[key] = (ver, w, wt)
        if jobs:
            hip.cast_weight_multi(jobs)

    def qkv_bias(self, q_bias, v_bias):
        """cat(q_bias, 0, v_bias) (vlmo.py:72-75), cached until either parameter changes."""
        key = ('qkvb', id(q_bias))
        ver = (q_bias._version, v_bias._version, q_bias.data_ptr())
        ent = self._c.get(key)
        if ent is None or ent[0] != ver:
            ent = (ver, torch.cat([q_bias.detach(), torch.zeros_like(q_bias), v_bias.detach()]))
            self._c[key] = ent
        return ent[1]

    def clear(self):
        self._c.clear()


class _PlanStatic:
    """Shape-only part of a plan (segment descriptors, row maps): built once per (B, T, P, device).
    Building it copies a few small host tensors to the device, which would stall the host every pass."""
    _cache = {}

    @classmethod
    def get(cls, B, T, P, device):
        key = (B, T, P, str(device))
        st = cls._cache.get(key)
        if st is None:
            st = cls(B, T, P, device)
            cls._cache[key] = st
        return st

    def __init__(self, B, T, P, device):
        nt = B * T
        ar = torch.arange(B, dtype=torch.int32)
        z = torch.zeros(B, dtype=torch.int32)

        def seg(a0, la, b0, lb):
            return torch.stack([a0, la, b0, lb], 1).contiguous().to(device)

        tl, pl = torch.full((B,), T, dtype=torch.int32), torch.full((B,), P, dtype=torch.int32)
        self.seg_txt = seg(ar * T, tl, z, z) if T else None
        self.seg_img = seg(nt + ar * P, pl, z, z) if P else None
        self.seg_vl = seg(ar * T, tl, nt + ar * P, pl) if (T and P) else None
        # below the fusion layer text and image sequences attend separately but in ONE launch (longest first:
        # the workgroups of the short text sequences fill the tail of the image ones)
        self.seg_sep = torch.cat([self.seg_img, self.seg_txt]).contiguous() if (T and P) else None
        # packed row -> row of the [B, T+P, d] output (text first, vlmo.py:406)
        N = T + P
        rm_t = (torch.arange(B).view(B, 1) * N + torch.arange(T).view(1, T)).reshape(-1)
        rm_i = (torch.arange(B).view(B, 1) * N + T + torch.arange(P).view(1, P)).reshape(-1)
        self.rowmap = torch.cat([rm_t, rm_i]).to(torch.int32).to(device)
        # drop-path group of every packed row: text rows of sample b -> b, image rows -> B + b (a per-sample
        # scale vector [2B] is expanded inside the kernels through this map)
        self.row_group = torch.cat([torch.arange(B).repeat_interleave(T),
                                    B + torch.arange(B).repeat_interleave(P)]).to(torch.int32).to(device)


class Plan:
    """Row layout + attention launches of one backbone pass."""

    def __init__(self, B, T, P, device, txt_mask=None, img_mask=None):
        self.B, self.T, self.P = B, T, P
        self.nt, self.ni = B * T, B * P
        self.M = self.nt + self.ni
        self.device = device
        st = _PlanStatic.get(B, T, P, device)
        self.seg_txt, self.seg_img, self.seg_vl, self.seg_sep = st.seg_txt, st.seg_img, st.seg_vl, st.seg_sep
        self.rowmap, self.row_group = st.rowmap, st.row_group
        # key-padding mask over packed rows (vlmo.py:89-91); None = all valid
        parts = []
        if T:
            parts.append(txt_mask.reshape(-1).to(torch.int32) if txt_mask is not None
                         else torch.ones(self.nt, dtype=torch.int32, device=device))
        if P:
            parts.append(img_mask.reshape(-1).to(torch.int32) if img_mask is not None
                         else torch.ones(self.ni, dtype=torch.int32, device=device))
        self.keymask = torch.cat(parts).contiguous() if (txt_mask is not None or img_mask is not None) else None

    def attn_launches(self, fused):
        if fused and self.seg_vl is not None:
            return [(self.seg_vl, self.B, self.T + self.P)]
        if self.seg_sep is not None and MERGE_SEPARATE_ATTENTION:
            return [(self.seg_sep, 2 * self.B, max(self.T, self.P))]
        out = []
        if self.seg_txt is not None:
            out.append((self.seg_txt, self.B, self.T))
        if self.seg_img is not None:
            out.append((self.seg_img, self.B, self.P))
        return out


class BlockMeta:
    """Static (non-tensor) description of one Block call."""

    def __init__(self, plan, heads, d, hidden, fused, expert_ranges, training, drop, attn_drop,
                 row_scale1, row_scale2, seed, eps=LN_EPS):
        self.plan, self.heads, self.d, self.hidden = plan, heads, d, hidden
        self.eps = eps
        self.fused = fused
        self.expert_ranges = expert_ranges      # [(row0, nrows)], one per expert in param order
        self.training = training
        self.drop = hip.drop_params(drop, training)
        self.attn_drop = hip.drop_params(attn_drop, training)
        self.rs1, self.rs2 = row_scale1, row_scale2
        self.seed = seed
        self.shadows = None
        self.tile = DEFAULT_TILE


def _empty(shape, dtype, dev):
    return torch.empty(shape, dtype=dtype, device=dev)


_SIDE = {}


def _side_stream(dev):
    """The weight-gradient stream of a device: lowest dispatch priority, so the activation-gradient chain
    on the caller's stream (the critical path) wins every freed compute-unit slot.  VLMO_SIDE_STREAM=
    'low' (default) | 'normal' | 'cumask:<hex words, comma separated>' (measurement aid)."""
    s = _SIDE.get(dev)
    if s is None:
        def make():
            if SIDE_MODE.startswith('cumask:'):
                raw = hip.side_stream_create(False, [int(w, 16) for w in SIDE_MODE[7:].split(',')])
            else:
                raw = hip.side_stream_create(SIDE_MODE != 'normal')
            return torch.cuda.ExternalStream(raw, device=dev)
        with torch.cuda.device(dev):
            s = pick_stream(dev, make, [torch.cuda.current_stream(dev)]) if PROBE_STREAMS else make()
        _SIDE[dev] = s
    return s


_PROBE_SCRATCH = {}


def _queued_behind(dev, busy, cand):
    """True when work on stream `cand` waits for kernels on stream `busy` (the two share a hardware queue, or their
    queues share a command-processor pipe): a long memory-bound kernel sequence goes to `busy`, a one-element kernel
    to `cand` right behind it; sharing shows as the small kernel finishing only when the long ones have."""
    # one 64 MB scratch per device, kept: a probe per rank and candidate at start-up must not allocate 256 MB each time
    # (8 ranks x up to 16 probes); 24 passes of ~25 us keep `busy` occupied for the same ~0.6 ms
    scratch = _PROBE_SCRATCH.get(dev)
    if scratch is None:
        scratch = _PROBE_SCRATCH[dev] = torch.empty(1 << 24, device=dev)
    tiny = torch.empty(64, device=dev)
    votes = 0
    for _ in range(2):
        torch.cuda.synchronize(dev)
        t0, t1, t2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        with torch.cuda.stream(busy):
            scratch.zero_()                                     # the timer starts once `busy` is running
            t0.record()
            for _ in range(24):
                scratch.mul_(1.0)
            t1.record()
        with torch.cuda.stream(cand):
            tiny.zero_()
            t2.record()
        torch.cuda.synchronize(dev)
        votes += t0.elapsed_time(t2) > 0.5 * t0.elapsed_time(t1)
    return votes == 2


def pick_stream(dev, make, against, tries=8):
    """A stream from make() whose work does not queue behind any stream of `against`.  HIP hands streams their
    hardware queues round-robin and the queues are spread over the command processor's pipes by creation order, so
    whether two streams can run side by side depends on what else (RCCL, the framework) created streams before:
    measured here, a step with the reducer took 22.9 instead of 16.5 ms when the communication stream's queue shared the
    main stream's pipe (rocprofv3: every kernel on it started ~50 us late), and one stream in four of a fresh batch
    finishes a one-element kernel 1.4 ms late behind a busy main stream (tools/pipe_probe.py).  Probing is the only
    portable way to know.  Falls back to the first candidate when every one collides."""
    first = None
    for _ in range(tries):
        s = make()
        first = first or s
        with torch.cuda.stream(s):
            torch.zeros(1, device=dev)                          # binds the hardware queue
        if not any(_queued_behind(dev, a, s) for a in against):
            return s
    import warnings
    warnings.warn('no stream found that runs beside the main stream: side-stream work will serialise')
    return first


class _Fork:
    """Run weight-gradient work on a side stream: fork() after the producers were enqueued on the main
    stream, join() before the buffers it reads may be reused.  Off = everything on the main stream."""

    def __init__(self, dev, enabled):
        self.main = torch.cuda.current_stream(dev)
        self.side = _side_stream(dev) if enabled else None

    def __enter__(self):
        if self.side is not None:
            self.side.wait_stream(self.main)
            self._ctx = torch.cuda.stream(self.side)
            self._ctx.__enter__()
        return self

    def __exit__(self, *a):
        if self.side is not None:
            self._ctx.__exit__(*a)

    def join(self):
        if self.side is not None:
            self.main.wait_stream(self.side)


def _fill_forward(D, meta, params, launches, lse_sizes, M, pb, pf, need_bwd, keep):
    """Forward part of a VlmoBlockDesc: geometry, parameters (fp32 vectors, bf16 weight shadows) and the saved-
    activation slabs at pb (bf16) / pf (fp32).  params in BlockFn's order; x / x2 are set by the caller."""
    (g1, g2, n1w, n1b, qkv_w, q_bias, v_bias, proj_w, proj_b, n2w, n2b) = params[:11]
    nexp = len(meta.expert_ranges)
    pl, d, H, hid = meta.plan, meta.d, meta.heads, meta.hidden
    sh = meta.shadows
    qkv_bias = sh.qkv_bias(q_bias, v_bias)
    D.M, D.d, D.hidden, D.heads = M, d, hid, H
    D.n_experts = nexp
    for i, (r0, n) in enumerate(meta.expert_ranges):
        D.exp_row0[i], D.exp_rows[i] = r0, n
    D.n_attn = len(launches)
    md2 = M * d * 2
    D.y1, D.qkv, D.ctx, D.zd1, D.y2 = pb, pb + md2, pb + 4 * md2, pb + 5 * md2, pb + 6 * md2
    D.u, D.h, D.zd2 = pb + 7 * md2, pb + 11 * md2, pb + 15 * md2
    D.x1 = pf
    st = pf + M * d * 4
    D.mean1, D.rstd1, D.mean2, D.rstd2 = st, st + 4 * M, st + 8 * M, st + 12 * M
    off = st + 16 * M
    for i, ((seg, nseq, ml), sz) in enumerate(zip(launches, lse_sizes)):
        D.seg[i], D.nseq[i], D.maxlen[i] = seg.data_ptr(), nseq, ml
        D.lse_stride[i] = ((ml + 31) // 32) * 32
        D.lse[i] = off
        D.attn_seed_idx[i], D.attn_seq0[i] = i, 0
        off += sz * 4
    D.keymask = hip._p(pl.keymask)
    D.eps = meta.eps
    D.drop_thresh, D.inv_keep = meta.drop
    D.attn_drop_thresh, D.attn_inv_keep = meta.attn_drop
    D.seed = meta.seed & 0xFFFFFFFFFFFFFFFF
    D.rs1, D.rs2 = hip._p(meta.rs1), hip._p(meta.rs2)
    D.row_index = pl.row_group.data_ptr() if meta.rs1 is not None else None
    D.tile, D.need_bwd = meta.tile, int(need_bwd)
    D.g1, D.g2, D.n1w, D.n1b, D.n2w, D.n2b = (t.data_ptr() for t in (g1, g2, n1w, n1b, n2w, n2b))
    D.qkv_bias, D.proj_b = qkv_bias.data_ptr(), proj_b.data_ptr()
    keep.append(qkv_bias)
    w, wt = sh.get(qkv_w)
    D.qkv_w, D.qkv_wT = w.data_ptr(), wt.data_ptr()
    keep += [w, wt]
    w, wt = sh.get(proj_w)
    D.proj_w, D.proj_wT = w.data_ptr(), wt.data_ptr()
    keep += [w, wt]
    for i in range(nexp):
        w1, b1, w2, b2 = params[11 + 4 * i: 15 + 4 * i]
        a, at = sh.get(w1)
        c, ct = sh.get(w2)
        D.w1[i], D.w1T[i], D.w2[i], D.w2T[i] = a.data_ptr(), at.data_ptr(), c.data_ptr(), ct.data_ptr()
        D.b1[i], D.b2[i] = b1.data_ptr(), b2.data_ptr()
        keep += [a, at, c, ct]


SPLIT_BWD_ATTENTION = _os.environ.get('VLMO_SPLIT_BWD_ATTN', '1') != '0'


def _split_backward_attention(D, meta):
    """Below the fusion layer the image and the text sequences share ONE forward attention launch (Plan.seg_sep, image
    sequences first).  The single-pass backward sizes a workgroup (one wave per key tile, LDS images) for the launch's
    LONGEST sequence, so in the shared launch every 64-token text sequence would hold a whole CU with 2 of 7 waves
    working (122 us for the pair at Base B=64); as two launches the text one takes 34 KB of LDS and 128 threads per
    workgroup, four to a CU (79 + 16 us).  The forward's log-sum-exp buffer is addressed per sequence with the shared
    launch's stride, so the two backward launches are views of it."""
    pl = meta.plan
    if (not SPLIT_BWD_ATTENTION or meta.fused or D.n_attn != 1 or pl.seg_sep is None or not (pl.T and pl.P)
            or D.nseq[0] != 2 * pl.B or max(pl.T, pl.P) > 256):
        return
    stride = D.lse_stride[0]
    D.n_attn = 2
    D.nseq[0], D.nseq[1] = pl.B, pl.B
    D.maxlen[0], D.maxlen[1] = pl.P, pl.T
    D.lse_stride[1] = stride
    D.seg[1] = D.seg[0] + pl.B * 16                                       # 4 int32 per sequence
    # the text launch regenerates the attention-dropout mask of ITS sequences of the shared forward launch
    # (sequences [B, 2B) under the forward's seed), not the mask of a second forward launch
    D.attn_seed_idx[1], D.attn_seq0[1] = D.attn_seed_idx[0], pl.B
    D.lse[1] = D.lse[0] + pl.B * meta.heads * stride * 4


def _carve(flat, shapes):
    out, off = [], 0
    for shp in shapes:
        n = 1
        for s_ in shp:
            n *= s_
        out.append(flat[off:off + n].view(*shp))
        off += n
    return out


def shared_layout(params, d):
    """[(parameter, offset)] of a block's shared-parameter group inside its flat gradient bucket (the carve order of
    _fill_grads; params in BlockFn's order).  q_bias / v_bias sit at the two ends of the 3d-wide qkv-bias slot."""
    (g1, g2, n1w, n1b, qkv_w, q_bias, v_bias, proj_w, proj_b, n2w, n2b) = params[:11]
    o = 6 * d
    out = [(g1, 0), (g2, d), (n1w, 2 * d), (n1b, 3 * d), (n2w, 4 * d), (n2b, 5 * d), (qkv_w, o)]
    o += 3 * d * d
    out.append((proj_w, o))
    o += d * d
    out.append((proj_b, o))
    o += d
    out += [(q_bias, o), (v_bias, o + 2 * d)]
    return out


def expert_layout(params, d, hid):
    w1, b1, w2, b2 = params
    return [(w1, 0), (b1, hid * d), (w2, hid * d + hid), (b2, 2 * hid * d + hid)]


def _fill_grads(D, flats, d, hid, nexp):
    """Parameter-gradient pointers of a VlmoBlockDesc from flat fp32 storage (flats[0]: shared parameters,
    flats[1 + e]: expert e) -> gradient tensors in BlockFn's parameter order."""
    (dg1, dg2, dn1w, dn1b, dn2w, dn2b, dqkv_w, dproj_w, dproj_b, dqkv_b) = _carve(
        flats[0], [(d,)] * 6 + [(3 * d, d), (d, d), (d,), (3 * d,)])
    D.dg1, D.dg2, D.dn1w, D.dn1b, D.dn2w, D.dn2b = (t.data_ptr() for t in (dg1, dg2, dn1w, dn1b, dn2w, dn2b))
    D.dqkv_w, D.dproj_w, D.dproj_b, D.dqkv_b = (t.data_ptr() for t in (dqkv_w, dproj_w, dproj_b, dqkv_b))
    dexp = []
    for i in range(nexp):
        dw1, db1, dw2, db2 = _carve(flats[1 + i], [(hid, d), (hid,), (d, hid), (d,)])
        D.dw1[i], D.db1[i], D.dw2[i], D.db2[i] = dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(), db2.data_ptr()
        dexp += [dw1, db1, dw2, db2]
    return [dg1, dg2, dn1w, dn1b, dqkv_w, dqkv_b[:d], dqkv_b[2 * d:], dproj_w, dproj_b, dn2w, dn2b] + dexp


class BlockFn(torch.autograd.Function):
    """One VLMo Block (vlmo.py:187-197) = norm1 -> qkv -> attention -> proj(+gamma_1, residual) -> norm2 ->
    expert FFN(+gamma_2, residual), forward and backward each ONE native call (vlmo_block_fwd / vlmo_block_bwd).
    params order: gamma_1, gamma_2, n1w, n1b, qkv_w, q_bias, v_bias, proj_w, proj_b, n2w, n2b,
    then (fc1_w, fc1_b, fc2_w, fc2_b) per expert range."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        nexp = len(meta.expert_ranges)
        experts = [params[11 + 4 * i: 15 + 4 * i] for i in range(nexp)]
        pl, d, H = meta.plan, meta.d, meta.heads
        M, dev = x.shape[0], x.device
        need_bwd = any(ctx.needs_input_grad)   # grad mode is off inside forward; this reflects the caller's
        x = x.contiguous()
        launches = pl.attn_launches(meta.fused)
        lse_sizes = [nseq * H * (((ml + 31) // 32) * 32) for _, nseq, ml in launches]
        # one bf16 slab: y1 | qkv(3) | ctx | zd1 | y2 | u(4) | h(4) | zd2  (units of M*d) ; one fp32 slab
        sb = torch.empty(16 * M * d, dtype=torch.bfloat16, device=dev)
        sf = torch.empty(M * d + 4 * M + sum(lse_sizes), dtype=torch.float32, device=dev)
        x2 = torch.empty((M, d), dtype=torch.float32, device=dev)
        D = hip.BlockDesc()
        keep = [sb, sf, pl]
        _fill_forward(D, meta, params, launches, lse_sizes, M, sb.data_ptr(), sf.data_ptr(), need_bwd, keep)
        D.x, D.x2 = x.data_ptr(), x2.data_ptr()
        hip.block_fwd(D)
        if need_bwd:
            ctx.meta, ctx.desc, ctx.keep = meta, D, keep
            ctx.save_for_backward(x, *params)
            ctx.sink = GRAD_SINK
            if ctx.sink is not None:
                # one bucket for the block's shared parameters, one per expert used by this call
                ctx.sink_groups = [tuple(params[:11])] + [tuple(e) for e in experts]
                for g_ in ctx.sink_groups:
                    ctx.sink.expect(g_)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        meta, D = ctx.meta, ctx.desc
        x, *params = ctx.saved_tensors
        nexp = len(meta.expert_ranges)
        d, hid = meta.d, meta.hidden
        M, dev = x.shape[0], x.device
        f32 = torch.float32
        dx2 = dx2.contiguous()
        # every parameter gradient of this block lives in ONE zero-filled flat buffer (one memset)
        sink = ctx.sink
        shared_n = 6 * d + 3 * d * d + d * d + d + 3 * d
        exp_n = 2 * hid * d + hid + d
        if sink is not None:    # data-parallel run: accumulate straight into the reducer's persistent flat buckets
            akey, aroom = id(params[0]), shared_n + 3 * exp_n       # one arena per block: shared + up to 3 experts
            flats = [sink.acquire(ctx.sink_groups[0], shared_n, dev, akey, aroom,
                                  layout=shared_layout(ctx.sink_groups[0], d))] + \
                    [sink.acquire(g_, exp_n, dev, akey, aroom, layout=expert_layout(g_, d, hid))
                     for g_ in ctx.sink_groups[1:]]
        else:                   # ONE zero-filled flat buffer (one memset) carved into all gradients of the block
            reg = _task_flats()
            if reg:             # fresh gradients for these groups: later StackFn nodes of this backward must not add into
                for g_ in [params[:11]] + [params[11 + 4 * e: 15 + 4 * e] for e in range(nexp)]:      # an earlier node's buffer
                    reg.pop(_group_key(g_), None)
            whole = torch.zeros(shared_n + nexp * exp_n, dtype=f32, device=dev)
            flats = [whole[:shared_n]] + [whole[shared_n + i * exp_n: shared_n + (i + 1) * exp_n] for i in range(nexp)]

        grads = _fill_grads(D, flats, d, hid, nexp)
        _split_backward_attention(D, meta)
        # temporaries: dz2 | du(4) | dy2(=dctx) | dz1 | dqkv(3) | dy1  bf16 ; dx1, dx0 fp32
        tb = torch.empty(11 * M * d, dtype=torch.bfloat16, device=dev)
        dx1 = torch.empty((M, d), dtype=f32, device=dev)
        dx0 = torch.empty((M, d), dtype=f32, device=dev)
        pb, md2 = tb.data_ptr(), M * d * 2
        D.dz2, D.du, D.dy2, D.dz1, D.dqkv, D.dy1 = pb, pb + md2, pb + 5 * md2, pb + 6 * md2, pb + 7 * md2, pb + 10 * md2
        D.dctx = D.dy2      # dy2 is consumed by ln_bwd on the main stream before dctx is written; never read on the side
        D.dx2, D.dx1, D.dx0 = dx2.data_ptr(), dx1.data_ptr(), dx0.data_ptr()
        # column-partial workspace: one 2d-wide slot per deferred fold of vlmo_block_bwd (3 + experts of them)
        ncols = max(3 * d, hid, 2 * d * (3 + len(meta.expert_ranges)))
        ws_main = hip.workspace(dev, ncols)
        D.ws_main, D.ws_bytes = ws_main.data_ptr(), ws_main.numel() * 4
        side = _side_stream(dev) if _use_side_stream(sink, M) else None
        tn_need = hip.lib().vlmo_gemm_tn_ws_bytes(M, hid, d)
        if side is not None:
            with torch.cuda.stream(side):
                ws_side = hip.workspace(dev, ncols)
                ws_tn = hip.tn_workspace(dev, tn_need)
            D.ws_side, D.side_stream = ws_side.data_ptr(), side.cuda_stream
            D.ws_bytes = min(D.ws_bytes, ws_side.numel() * 4)
        else:
            D.ws_side, D.side_stream = None, None
            ws_tn = hip.tn_workspace(dev, tn_need)
        D.ws_tn, D.ws_tn_bytes = ws_tn.data_ptr(), ws_tn.numel() * 4
        hip.block_bwd(D)
        ctx.desc = ctx.keep = None
        if sink is not None:
            # the bucket IS the gradient storage: p.grad become views of it, autograd gets nothing to add
            for p_, g_ in zip(params, grads):
                if p_.requires_grad:
                    if p_.grad is None:
                        p_.grad = g_
                    elif p_.grad.data_ptr() != g_.data_ptr():
                        raise RuntimeError('a parameter of a data-parallel block already holds a foreign .grad; '
                                           'use zero_grad(set_to_none=True)')
            sink.release_all(ctx.sink_groups)
            return (dx0, None) + (None,) * len(grads)
        return (dx0, None, *grads)


WGRAD_BATCH = int(_os.environ.get('VLMO_WGRAD_BATCH', '2'))     # blocks per deferred weight-gradient launch; 0 = by tile count


def wgrad_batch_for(d, hid, cus=256, max_batch=4):
    """Blocks per batched weight-gradient launch.  Default: 2 (VLMo-Base: 108 output tiles of 256 x 256 per block -> 216 =
    one dispatch round).  VLMO_WGRAD_BATCH=0 picks the count whose tiles fill whole rounds (VLMo-Large: 192 per block -> 4
    blocks = 768 = exactly three rounds, where 2 blocks take two rounds for 1.5): measured, the side stream's rounds are not
    what the step waits for -- Large, 32 pairs, one box: (batch, temporary sets) = (2, 4) 24.80 ms, (4, 5) 25.35, (4, 6)
    24.91, (4, 8) 24.92, (2, 6) 24.85; Base (2, 4) 14.38, (4, 6) 14.48 -- the CUs a partial round leaves idle are taken by the
    main stream's kernels, and a batch of four holds its blocks' temporaries longer."""
    if WGRAD_BATCH > 0:
        return WGRAD_BATCH
    c = lambda n: -(-n // 256)
    tiles = c(d) * c(3 * d) + c(d) * c(d) + 2 * c(d) * c(hid)
    best, best_cost = 1, None
    for b in range(1, max_batch + 1):
        cost = -(-b * tiles // cus) / b
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = b, cost
    return best
TMP_SETS = int(_os.environ.get('VLMO_TMP_SETS', '4'))            # rotation depth of the backward temporaries
USE_STACK = _os.environ.get('VLMO_STACK', '1') != '0'            # one native call per pass (else one per block)
WGRAD_STORE = _os.environ.get('VLMO_WGRAD_STORE', '1') != '0'    # weight-gradient matrices written, not zero-filled + accumulated

_PERSIST = {}


def _persist(dev, tag, numel, dtype):
    """Scratch that lives across passes (backward temporaries, column workspaces): keyed by the caller's stream,
    because reuse is ordered by that stream (vlmo_stack_bwd joins its side stream before it returns)."""
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, tag)
    t = _PERSIST.get(key)
    if t is None or t.numel() < numel or t.dtype != dtype:
        t = _PERSIST[key] = torch.empty(numel, dtype=dtype, device=dev)
    return t


_EVENTS = {}


def _ready_events(dev, n):
    evs = _EVENTS.setdefault(dev, [])
    while len(evs) < n:
        evs.append(hip.event_create())
    return evs[:n]


INPLACE_ACCUM = _os.environ.get('VLMO_INPLACE_ACCUM', '1') != '0'      # kill switch for both mechanisms below

# ---- in-place gradient accumulation across the backward passes of one step -------------------------------------
# A four-objective step runs the block stack three to seven times; without a reducer every pass hands autograd a fresh
# gradient per parameter and the engine's input buffers add them: 337 elementwise launches = 1.4 ms of a 61 ms step.
#
# (1) WITHIN one backward() / autograd.grad() call (always on): the first StackFn node of the graph task that reaches a
#     parameter GROUP (a block's shared parameters, or one expert) returns views of a flat buffer and registers the
#     buffer under the task's id; every later node of the same task accumulates INTO that buffer inside the
#     weight-gradient kernels and returns None for the group.  The engine still holds the first node's tensors (in the
#     AccumulateGrad input buffer, or as the captured result of autograd.grad), so what it finally accumulates or
#     returns is the sum: .grad is never touched by us, which makes this correct under autograd.grad(...) and
#     backward(inputs=[...]) too.  Every engine node of a task must cooperate: a BlockFn node that returns fresh
#     gradients for a group drops the group's entry (the input buffer would add out of place and orphan our buffer).
# (2) ACROSS backward() calls (gradient-accumulation micro-steps, zero_grad(set_to_none=False)): a pass accumulates
#     straight into the views the parameters already hold as .grad.  That is only right in an ordinary accumulating
#     backward(), so it is OPT-IN: `with engine.accumulate_into_grad():` around loss.backward() -- the package's
#     NativeScalerWithGradNormCount (the reference loop's backward, utils.py:343-364) and bench.py do that.
_TASK_FLATS = {'task': None, 'flats': {}}
_INTO_GRAD = [False]


class accumulate_into_grad:
    """Context manager: backward passes inside it may add into the gradient views parameters already hold."""

    def __init__(self, enabled=True):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.prev = _INTO_GRAD[0]
        _INTO_GRAD[0] = self.enabled
        return self

    def __exit__(self, *a):
        _INTO_GRAD[0] = self.prev


def _task_flats():
    """{group key: flat gradient buffer} of the running graph task (emptied when another task starts)."""
    tid = torch._C._current_graph_task_id()
    if _TASK_FLATS['task'] != tid:
        _TASK_FLATS['task'], _TASK_FLATS['flats'] = tid, {}
    return _TASK_FLATS['flats'] if tid >= 0 else None


def _group_key(gparams):
    # a weight matrix of the group (qkv weight / fc1 weight): gamma_1 may be a stand-in shared by all blocks (init_values=None)
    return gparams[4].data_ptr() if len(gparams) == 11 else gparams[0].data_ptr()


def _grad_flat(gparams, n):
    """The flat fp32 storage the group's parameters ALREADY hold as .grad in this engine's layout (views an earlier
    backward returned, possibly zeroed by zero_grad(set_to_none=False)), or None."""
    g = gparams[0].grad
    base = g._base if g is not None else None
    if base is None or g.dtype != torch.float32 or base.dim() != 1 or not base.is_contiguous():
        return None
    off = (g.data_ptr() - base.data_ptr()) // 4
    if off < 0 or off + n > base.numel():
        return None
    return base[off:off + n]


def _same_views(layout, flat):
    """Every parameter of the group holds as .grad exactly the view the layout carves out of `flat`, and nobody hooks it."""
    for p_, off in layout:
        if not p_.requires_grad:
            continue
        pg = p_.grad
        if pg is None or pg.data_ptr() != flat.data_ptr() + 4 * off or not pg.is_contiguous() or pg.dtype != torch.float32:
            return False
        if getattr(p_, '_post_accumulate_grad_hooks', None) or getattr(p_, '_backward_hooks', None):
            return False
    return True


class StackFn(torch.autograd.Function):
    """All Blocks of one backbone pass (the loops at vlmo.py:402-411) as ONE native call per direction
    (vlmo_stack_fwd / vlmo_stack_bwd).  metas: one BlockMeta per block, in forward order; params: the blocks'
    parameter lists concatenated, each in BlockFn's order."""

    @staticmethod
    def forward(ctx, x, metas, *params):
        nb = len(metas)
        m0 = metas[0]
        pl, d, H, hid = m0.plan, m0.d, m0.heads, m0.hidden
        M, dev = x.shape[0], x.device
        need_bwd = any(ctx.needs_input_grad)
        x = x.contiguous()
        md = M * d
        # per block: one bf16 slab  y1 | qkv(3) | ctx | zd1 | y2 | u(4) | h(4) | zd2  (units of M*d) and one fp32
        # slab  x1 | mean1 rstd1 mean2 rstd2 | lse...  ; the blocks' outputs x2 (= the next block's saved input)
        lse_sizes, launches = [], []
        for mt in metas:
            ln = pl.attn_launches(mt.fused)
            launches.append(ln)
            lse_sizes.append([nseq * H * (((ml + 31) // 32) * 32) for _, nseq, ml in ln])
        sf_n = [md + 4 * M + sum(ls) for ls in lse_sizes]
        live = nb if need_bwd else 1        # without a backward every block reuses the first block's slabs
        SB = torch.empty(live * 16 * md, dtype=torch.bfloat16, device=dev)
        SF = torch.empty(sum(sf_n[:live]) if need_bwd else max(sf_n), dtype=torch.float32, device=dev)
        X2 = torch.empty((nb if need_bwd else min(nb, 2), M, d), dtype=torch.float32, device=dev)
        descs = (hip.BlockDesc * nb)()
        keep = [SB, SF, X2, pl]
        if m0.shadows is not None:          # every stale weight shadow of the pass in one launch (ShadowCache.refresh)
            stale, po = [], 0
            for mt in metas:
                ne = len(mt.expert_ranges)
                bp = params[po:po + 11 + 4 * ne]
                po += 11 + 4 * ne
                stale += [bp[4], bp[7]] + [bp[11 + 4 * e] for e in range(ne)] + [bp[13 + 4 * e] for e in range(ne)]
            m0.shadows.refresh(stale)
        pofs, sf_off = 0, 0
        xin = x.data_ptr()
        spans = []
        for i, mt in enumerate(metas):
            nexp = len(mt.expert_ranges)
            npar = 11 + 4 * nexp
            bp = params[pofs:pofs + npar]
            spans.append((pofs, npar))
            pofs += npar
            D = descs[i]
            pb = SB.data_ptr() + (i * 16 * md * 2 if need_bwd else 0)
            pf = SF.data_ptr() + (sf_off * 4 if need_bwd else 0)
            sf_off += sf_n[i]
            x2 = X2[i if need_bwd else i % 2]
            _fill_forward(D, mt, bp, launches[i], lse_sizes[i], M, pb, pf, need_bwd, keep)
            D.x, D.x2 = xin, x2.data_ptr()
            xin = x2.data_ptr()
        S = hip.StackDesc()
        S.n_blocks = nb
        S.blocks = ctypes.cast(descs, ctypes.POINTER(hip.BlockDesc))
        hip.stack_fwd(S)
        out = X2[(nb - 1) if need_bwd else (nb - 1) % 2]
        if need_bwd:
            ctx.metas, ctx.descs, ctx.keep, ctx.spans = metas, descs, keep, spans
            ctx.save_for_backward(x, *params)
            ctx.sink = GRAD_SINK
            if ctx.sink is not None:
                ctx.sink_groups = []
                for (o, n_) in spans:
                    bp = params[o:o + n_]
                    groups = [tuple(bp[:11])] + [tuple(bp[11 + 4 * e: 15 + 4 * e]) for e in range((n_ - 11) // 4)]
                    ctx.sink_groups.append(groups)
                    for g_ in groups:
                        ctx.sink.expect(g_)
        return out

    @staticmethod
    def backward(ctx, dxo):
        metas, descs, spans = ctx.metas, ctx.descs, ctx.spans
        x, *params = ctx.saved_tensors
        nb = len(metas)
        m0 = metas[0]
        d, hid = m0.d, m0.hidden
        M, dev = x.shape[0], x.device
        md = M * d
        f32 = torch.float32
        dxo = dxo.contiguous()
        sink = ctx.sink
        shared_n = 6 * d + 3 * d * d + d * d + d + 3 * d
        exp_n = 2 * hid * d + hid + d
        want = wgrad_batch_for(d, hid)
        nsets = max(1, min(max(TMP_SETS, want + 1), nb))
        batch = max(1, min(want, nsets - 1)) if nb > nsets else max(1, want)
        # persistent scratch: backward temporaries (dz2 | du(4) | dy2=dctx | dz1 | dqkv(3) | dy1 bf16) and column
        # workspaces, one set per block in flight; dx1 and the dx ping-pong
        tb = _persist(dev, 'tb', nsets * 11 * md, torch.bfloat16)
        slot = hip.lib().vlmo_reduce_ws_bytes(2 * d)
        # column-fold slots + the du column partials of the DGELU epilogue + the attention backward's per-sequence dq | dv sums
        ws_n = 6 * slot // 4 + (M // 16 + 4) * hid + 4 * m0.plan.B * d
        ws = _persist(dev, 'ws', nsets * ws_n, f32)
        dxs = _persist(dev, 'dx', 3 * md, f32)
        dx_in = torch.empty((M, d), dtype=f32, device=dev)
        if sink is None:
            tot = sum(shared_n + ((n_ - 11) // 4) * exp_n for (_, n_) in spans)
            # no memset of the weight-gradient matrices: the deferred launches WRITE them (wgrad_store); only the vector
            # gradients (accumulated with atomics by the column folds) are zeroed, in one multi-tensor fill
            whole = None        # allocated when the first group needs fresh gradient storage
            reg = _task_flats()
        grads_all = [None] * len(params)
        goff = 0
        store_ok, acquired = WGRAD_STORE, []      # (flat, fresh, is_expert) of every gradient bucket of the pass
        for k in range(nb):                 # backward order: k-th processed block is i = nb-1-k
            i = nb - 1 - k
            D = descs[i]
            o, n_ = spans[i]
            nexp = (n_ - 11) // 4
            if sink is not None:
                akey, aroom = id(params[o]), shared_n + 3 * exp_n
                groups = ctx.sink_groups[i]
                got = [sink.acquire(groups[0], shared_n, dev, akey, aroom, layout=shared_layout(groups[0], d), lazy_zero=True)] + \
                      [sink.acquire(g_, exp_n, dev, akey, aroom, layout=expert_layout(g_, d, hid), lazy_zero=True)
                       for g_ in groups[1:]]
                flats = [f_ for f_, _ in got]
                store_ok = store_ok and all(fr for _, fr in got)      # a bucket an earlier pass of the step already fed: accumulate
                acquired += [(f_, fr, j > 0) for j, (f_, fr) in enumerate(got)]
            else:
                bp = params[o:o + n_]
                groups = [(bp[:11], shared_n)] + [(bp[11 + 4 * e: 15 + 4 * e], exp_n) for e in range(nexp)]
                ni = ctx.needs_input_grad[2 + o: 2 + o + n_]
                flats, kept = [], []
                for gi, (gp_, gn_) in enumerate(groups):
                    lo = 0 if gi == 0 else 11 + 4 * (gi - 1)
                    wanted = all(w or not p_.requires_grad for p_, w in zip(gp_, ni[lo:lo + len(gp_)]))
                    have, how = None, 0
                    if INPLACE_ACCUM and wanted:
                        if reg is not None and _group_key(gp_) in reg:
                            have, how = reg[_group_key(gp_)], 1                     # an earlier node of this backward
                        elif _INTO_GRAD[0]:
                            # an earlier backward of this step: only when the parameters hold exactly this layout's views
                            have = _grad_flat(gp_, gn_)
                            lay = (shared_layout(gp_, d) if gi == 0 else expert_layout(gp_, d, hid)) if have is not None else None
                            if have is not None and _same_views(lay, have):
                                how = 2
                            else:
                                have = None
                    flats.append(have)
                    kept.append(how)
                for gi, (gp_, gn_) in enumerate(groups):
                    if flats[gi] is None:
                        if whole is None:
                            whole = torch.empty(tot, dtype=f32, device=dev) if WGRAD_STORE else torch.zeros(tot, dtype=f32, device=dev)
                        flats[gi] = whole[goff:goff + gn_]
                        lo = 0 if gi == 0 else 11 + 4 * (gi - 1)
                        if reg is not None and INPLACE_ACCUM and all(w or not p_.requires_grad for p_, w in zip(gp_, ni[lo:lo + len(gp_)])):
                            reg[_group_key(gp_)] = flats[gi]
                    goff += gn_
                    acquired.append((flats[gi], kept[gi] == 0, gi > 0))
                if any(kept):
                    store_ok = False
            grads = _fill_grads(D, flats, d, hid, nexp)
            if sink is not None:
                grads_all[o:o + n_] = grads
            else:
                # groups accumulated in place hand nothing to autograd
                for gi in range(len(groups)):
                    lo = 0 if gi == 0 else 11 + 4 * (gi - 1)
                    hi = 11 if gi == 0 else lo + 4
                    if not kept[gi]:
                        grads_all[o + lo:o + hi] = grads[lo:hi]
            _split_backward_attention(D, metas[i])
            st_ = k % nsets
            pb = tb.data_ptr() + st_ * 11 * md * 2
            md2 = md * 2
            D.dz2, D.du, D.dy2, D.dz1, D.dqkv, D.dy1 = pb, pb + md2, pb + 5 * md2, pb + 6 * md2, pb + 7 * md2, pb + 10 * md2
            D.dctx = D.dy2
            D.ws_main, D.ws_bytes = ws.data_ptr() + st_ * ws_n * 4, ws_n * 4
            D.ws_side, D.side_stream, D.ws_tn, D.ws_tn_bytes = None, None, None, 0
            D.dx1 = dxs.data_ptr() + 2 * md * 4
            D.dx2 = dxo.data_ptr() if k == 0 else dxs.data_ptr() + ((k - 1) % 2) * md * 4
            D.dx0 = dx_in.data_ptr() if k == nb - 1 else dxs.data_ptr() + (k % 2) * md * 4
        if store_ok:
            # vector gradients only: [g1 g2 n1w n1b n2w n2b | qkv_w proj_w | proj_b qkv_b] and [w1 | b1 | w2 | b2]
            vecs = []
            for f_, _, is_exp in acquired:
                vecs += [f_[hid * d:hid * d + hid], f_[2 * hid * d + hid:]] if is_exp else [f_[:6 * d], f_[6 * d + 4 * d * d:]]
            torch._foreach_zero_(vecs)
        else:
            for f_, fr, _ in acquired:
                if fr and (sink is not None or WGRAD_STORE):
                    f_.zero_()
        S = hip.StackDesc()
        S.n_blocks, S.wgrad_batch, S.n_tmp_sets, S.wgrad_store = nb, batch, nsets, int(store_ok)
        S.blocks = ctypes.cast(descs, ctypes.POINTER(hip.BlockDesc))
        side = _side_stream(dev) if _use_side_stream(sink, M) else None
        S.side_stream = side.cuda_stream if side is not None else None
        evs = None
        if sink is not None and side is not None:
            evs = _ready_events(dev, nb)
            arr = (ctypes.c_void_p * nb)(*evs)
            S.grad_ready = ctypes.cast(arr, ctypes.POINTER(ctypes.c_void_p))
        hip.stack_bwd(S)
        ctx.descs = ctx.keep = None
        if sink is not None:
            for k in range(nb):
                i = nb - 1 - k
                o, n_ = spans[i]
                for p_, g_ in zip(params[o:o + n_], grads_all[o:o + n_]):
                    if p_.requires_grad:
                        if p_.grad is None:
                            p_.grad = g_
                        elif p_.grad.data_ptr() != g_.data_ptr():
                            raise RuntimeError('a parameter of a data-parallel block already holds a foreign .grad; '
                                               'use zero_grad(set_to_none=True)')
                sink.release_all(ctx.sink_groups[i], ready_event=evs[i] if evs is not None else None)
            return (dx_in, None) + (None,) * len(params)
        return (dx_in, None, *grads_all)


class FinalNormFn(torch.autograd.Function):
    """self.norm (vlmo.py:413) writing the [B, T+P, d] fp32 output through the row map."""

    @staticmethod
    def forward(ctx, x, w, b, plan, out_shape, eps=LN_EPS):
        M, d = x.shape
        out = torch.empty(out_shape, dtype=torch.float32, device=x.device)
        mean, rstd = torch.empty(M, device=x.device), torch.empty(M, device=x.device)
        hip.ln_fwd(x, w, b, out, mean, rstd, plan.rowmap, M, d, eps)
        ctx.plan = plan
        ctx.save_for_backward(x, w, mean, rstd)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, mean, rstd = ctx.saved_tensors
        M, d = x.shape
        dout = dout.contiguous().float()
        dx = torch.empty_like(x)
        dw, db = torch.zeros_like(w), torch.zeros_like(w)
        hip.ln_bwd(dout, ctx.plan.rowmap, x, w, mean, rstd, None, dx, dw, db, M, d)
        return dx, dw, db, None, None, None


class EmbedFn(torch.autograd.Function):
    """embed_txt + embed_img (vlmo.py:298-324) into one packed fp32 [M, d] matrix.

    tensor args: patch_w [d,C,p,p], patch_b, cls_tok, mask_tok, pos_embed [1,P,d], type_emb [2or3,d],
                 word, tpos, btype, ln_w, ln_b   (any of the two groups may be unused)"""

    @staticmethod
    def forward(ctx, meta, patch_w, patch_b, cls_tok, mask_tok, pos_embed, type_emb, word, tpos, btype,
                ln_w, ln_b):
        pl, d, dev = meta['plan'], meta['d'], meta['device']
        img, ids, masked = meta['img'], meta['ids'], meta['masked']
        B, T, P = pl.B, pl.T, pl.P
        x = torch.empty((pl.M, d), dtype=torch.float32, device=dev)
        drop, seed, sh = meta['drop'], meta['seed'], meta['shadows']
        saved = {}
        if T:
            xhat = torch.empty((pl.nt, d), dtype=torch.float32, device=dev)
            rstd = torch.empty((pl.nt,), dtype=torch.float32, device=dev)
            hip.embed_txt_fwd(ids, word, tpos, btype[0], ln_w, ln_b, type_emb[0], x[:pl.nt], xhat, rstd, B, T, d,
                              meta['txt_eps'], drop=drop, seed=seed + 1)
            saved['txt'] = (xhat, rstd)
        if P:
            npatch = P - 1
            p = meta['patch']
            patches = torch.empty((B * npatch, patch_w[0].numel()), dtype=torch.bfloat16, device=dev)
            hip.patchify(img, patches, p)
            proj = torch.empty((B * npatch, d), dtype=torch.bfloat16, device=dev)
            K = patches.shape[1]
            hip.gemm_nt(hip.EPI_BIAS, patches, sh.get(patch_w, need_t=False)[0], B * npatch, d, K, proj,
                        bias=patch_b)
            hip.embed_img_finish(proj, cls_tok, mask_tok, pos_embed, type_emb[meta['img_type']], masked,
                                 x[pl.nt:], B, npatch, d, drop=drop, seed=seed + 2)
            saved['img'] = patches
        ctx.meta, ctx.saved = meta, saved
        ctx.save_for_backward(patch_w, type_emb, word, ln_w)
        return x

    @staticmethod
    def backward(ctx, dx):
        meta, saved = ctx.meta, ctx.saved
        patch_w, type_emb, word, ln_w = ctx.saved_tensors
        pl, d, dev = meta['plan'], meta['d'], meta['device']
        B, T, P = pl.B, pl.T, pl.P
        drop, seed = meta['drop'], meta['seed']
        dx = dx.contiguous()
        g = [None] * 11   # patch_w, patch_b, cls, mask, pos, type, word, tpos, btype, ln_w, ln_b
        # every parameter gradient of the embeddings is a view of ONE zero-filled buffer (one fill launch instead of eleven)
        npatch = P - 1 if P else 0
        shapes = [tuple(type_emb.shape)]
        if T:
            shapes += [tuple(word.shape), (meta['tpos_rows'], d), (2, d), (d,), (d,)]
        if P:
            shapes += [(d,), (d,), (P, d), (d, saved['img'].shape[1]), (d,)]
        sizes = [int(torch.Size(sh).numel()) for sh in shapes]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        views, o = [], 0
        for sh, n in zip(shapes, sizes):
            views.append(flat[o:o + n].view(sh))
            o += n
        dtype_emb = views[0]
        k = 1
        if T:
            xhat, rstd = saved['txt']
            dword, dtpos_full, dbtype, dlnw, dlnb = views[k:k + 5]
            k += 5
            hip.embed_txt_bwd(dx[:pl.nt], meta['ids'], xhat, rstd, ln_w, dword, dtpos_full[:T], dbtype[0], dlnw, dlnb,
                              dtype_emb[0], B, T, d, drop=drop, seed=seed + 1)
            g[6], g[7], g[8], g[9], g[10] = dword, dtpos_full, dbtype, dlnw, dlnb
        if P:
            patches = saved['img']
            dproj = torch.empty((B * npatch, d), dtype=torch.bfloat16, device=dev)
            dcls, dmask, dpos, dpw, dpb = views[k:k + 5]
            hip.embed_img_bwd(dx[pl.nt:], meta['masked'], dproj, dcls, dmask, dpos, dtype_emb[meta['img_type']],
                              B, npatch, d, drop=drop, seed=seed + 2)
            hip.gemm_tn(dproj, patches, dpw, B * npatch, d, patches.shape[1])
            hip.colsum(dproj, dpb, B * npatch, d)
            g[0], g[1] = dpw.view_as(patch_w), dpb
            g[2], g[4] = dcls, dpos
            g[3] = dmask if meta['masked'] is not None else None
        g[5] = dtype_emb
        return (None, *g)
