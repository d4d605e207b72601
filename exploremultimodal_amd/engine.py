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
import torch

from . import hip

LN_EPS = 1e-12          # vlmo_module.py:21-23
OVERLAP_WGRAD = True     # weight-gradient GEMMs + bias column sums on a side stream beside the dgrad chain
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

    def clear(self):
        self._c.clear()


class Plan:
    """Row layout + attention launches of one backbone pass."""

    def __init__(self, B, T, P, device, txt_mask=None, img_mask=None):
        self.B, self.T, self.P = B, T, P
        self.nt, self.ni = B * T, B * P
        self.M = self.nt + self.ni
        self.device = device
        ar = torch.arange(B, dtype=torch.int32)
        z = torch.zeros(B, dtype=torch.int32)

        def seg(a0, la, b0, lb):
            return torch.stack([a0, la, b0, lb], 1).contiguous().to(device)

        tl, pl = torch.full((B,), T, dtype=torch.int32), torch.full((B,), P, dtype=torch.int32)
        self.seg_txt = seg(ar * T, tl, z, z) if T else None
        self.seg_img = seg(self.nt + ar * P, pl, z, z) if P else None
        self.seg_vl = seg(ar * T, tl, self.nt + ar * P, pl) if (T and P) else None
        # key-padding mask over packed rows (vlmo.py:89-91); None = all valid
        parts = []
        if T:
            parts.append(txt_mask.reshape(-1).to(torch.int32) if txt_mask is not None
                         else torch.ones(self.nt, dtype=torch.int32, device=device))
        if P:
            parts.append(img_mask.reshape(-1).to(torch.int32) if img_mask is not None
                         else torch.ones(self.ni, dtype=torch.int32, device=device))
        self.keymask = torch.cat(parts).contiguous() if (txt_mask is not None or img_mask is not None) else None
        # packed row -> row of the [B, T+P, d] output (text first, vlmo.py:406)
        N = T + P
        rm_t = (torch.arange(B).view(B, 1) * N + torch.arange(T).view(1, T)).reshape(-1)
        rm_i = (torch.arange(B).view(B, 1) * N + T + torch.arange(P).view(1, P)).reshape(-1)
        self.rowmap = torch.cat([rm_t, rm_i]).to(torch.int32).to(device)
        # sample index of every packed row (drop-path scale expansion)
        self.row_sample_txt = torch.arange(B).repeat_interleave(T).to(device) if T else None
        self.row_sample_img = torch.arange(B).repeat_interleave(P).to(device) if P else None

    def attn_launches(self, fused):
        if fused and self.seg_vl is not None:
            return [(self.seg_vl, self.B, self.T + self.P)]
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
    s = _SIDE.get(dev)
    if s is None:
        s = torch.cuda.Stream(device=dev)
        _SIDE[dev] = s
    return s


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


class BlockFn(torch.autograd.Function):
    """One VLMo Block (vlmo.py:187-197) = norm1 -> qkv -> attention -> proj(+gamma_1,
    residual) -> norm2 -> expert FFN(+gamma_2, residual).  params order:
    gamma_1, gamma_2, n1w, n1b, qkv_w, q_bias, v_bias, proj_w, proj_b, n2w, n2b,
    then (fc1_w, fc1_b, fc2_w, fc2_b) per expert range."""

    @staticmethod
    def forward(ctx, x, meta, *params):
        (g1, g2, n1w, n1b, qkv_w, q_bias, v_bias, proj_w, proj_b, n2w, n2b) = params[:11]
        experts = [params[11 + 4 * i: 15 + 4 * i] for i in range(len(meta.expert_ranges))]
        pl, d, H, hid = meta.plan, meta.d, meta.heads, meta.hidden
        M, dev = x.shape[0], x.device
        bf, f32 = torch.bfloat16, torch.float32
        sh = meta.shadows
        need_bwd = any(ctx.needs_input_grad)   # grad mode is off inside forward; this reflects the caller's
        seed = meta.seed

        y1, mean1, rstd1 = _empty((M, d), bf, dev), _empty((M,), f32, dev), _empty((M,), f32, dev)
        hip.ln_fwd(x, n1w, n1b, y1, mean1, rstd1, None, M, d, meta.eps)
        qkv_bias = torch.cat([q_bias.detach(), torch.zeros_like(q_bias), v_bias.detach()])   # vlmo.py:72-75
        qkv = _empty((M, 3 * d), bf, dev)
        hip.gemm_nt(hip.EPI_BIAS, y1, sh.get(qkv_w)[0], M, 3 * d, d, qkv, bias=qkv_bias, tile=meta.tile)
        ctxb = _empty((M, d), bf, dev)
        launches = pl.attn_launches(meta.fused)
        lses = []
        for li, (seg, nseq, maxlen) in enumerate(launches):
            npad = ((maxlen + 31) // 32) * 32
            lse = _empty((nseq * H, npad), f32, dev)
            hip.attn_fwd(qkv, seg, nseq, pl.keymask, ctxb, lse, H, d, maxlen, (d // H) ** -0.5,
                         drop=meta.attn_drop, seed=seed + 11 + li)
            lses.append(lse)
        x1 = _empty((M, d), f32, dev)
        zd1 = _empty((M, d), bf, dev) if need_bwd else None
        hip.gemm_nt(hip.EPI_RESID, ctxb, sh.get(proj_w)[0], M, d, d, x1, out2=zd1, bias=proj_b, gamma=g1,
                    resid=x, row_scale=meta.rs1, drop=meta.drop, seed=seed + 1, tile=meta.tile)
        y2, mean2, rstd2 = _empty((M, d), bf, dev), _empty((M,), f32, dev), _empty((M,), f32, dev)
        hip.ln_fwd(x1, n2w, n2b, y2, mean2, rstd2, None, M, d, meta.eps)
        u, hh = _empty((M, hid), bf, dev), _empty((M, hid), bf, dev)
        x2 = _empty((M, d), f32, dev)
        zd2 = _empty((M, d), bf, dev) if need_bwd else None
        for ei, ((r0, n), (w1, b1, w2, b2)) in enumerate(zip(meta.expert_ranges, experts)):
            hip.gemm_nt(hip.EPI_BIAS_GELU, y2[r0:r0 + n], sh.get(w1)[0], n, hid, d, u[r0:r0 + n],
                        out2=hh[r0:r0 + n], bias=b1, drop=meta.drop, seed=seed + 20 + 2 * ei, tile=meta.tile)
            hip.gemm_nt(hip.EPI_RESID, hh[r0:r0 + n], sh.get(w2)[0], n, d, hid, x2[r0:r0 + n],
                        out2=zd2[r0:r0 + n] if zd2 is not None else None, bias=b2, gamma=g2,
                        resid=x1[r0:r0 + n], row_scale=meta.rs2[r0:r0 + n] if meta.rs2 is not None else None,
                        drop=meta.drop, seed=seed + 21 + 2 * ei, tile=meta.tile)
        if need_bwd:
            ctx.meta = meta
            ctx.save_for_backward(x, *params)
            ctx.saved = (y1, mean1, rstd1, qkv, ctxb, lses, zd1, x1, y2, mean2, rstd2, u, hh, zd2)
        return x2

    @staticmethod
    def backward(ctx, dx2):
        meta = ctx.meta
        x, *params = ctx.saved_tensors
        (g1, g2, n1w, n1b, qkv_w, q_bias, v_bias, proj_w, proj_b, n2w, n2b) = params[:11]
        experts = [params[11 + 4 * i: 15 + 4 * i] for i in range(len(meta.expert_ranges))]
        (y1, mean1, rstd1, qkv, ctxb, lses, zd1, x1, y2, mean2, rstd2, u, hh, zd2) = ctx.saved
        ctx.saved = None
        pl, d, H, hid = meta.plan, meta.d, meta.heads, meta.hidden
        M, dev = x.shape[0], x.device
        bf, f32 = torch.bfloat16, torch.float32
        sh, seed = meta.shadows, meta.seed
        dx2 = dx2.contiguous()
        # every parameter gradient of this block lives in ONE zero-filled flat buffer (one memset)
        nexp = len(meta.expert_ranges)
        total = 6 * d + 3 * d * d + d * d + d + 3 * d + nexp * (2 * hid * d + hid + d)
        flat = torch.zeros(total, dtype=f32, device=dev)
        off = [0]

        def z(*shape):
            n = 1
            for s_ in shape:
                n *= s_
            t = flat[off[0]:off[0] + n].view(*shape)
            off[0] += n
            return t

        dg1, dg2, dn1w, dn1b, dn2w, dn2b = z(d), z(d), z(d), z(d), z(d), z(d)
        dqkv_w, dproj_w, dproj_b = z(3 * d, d), z(d, d), z(d)
        dexp = []
        fork = _Fork(dev, OVERLAP_WGRAD)
        # ---- FFN half
        dz2 = _empty((M, d), bf, dev)
        du = _empty((M, hid), bf, dev)
        dy2 = _empty((M, d), bf, dev)
        for ei, ((r0, n), (w1, b1, w2, b2)) in enumerate(zip(meta.expert_ranges, experts)):
            dw1, db1, dw2, db2 = z(hid, d), z(hid), z(d, hid), z(d)
            sl = slice(r0, r0 + n)
            hip.resid_bwd(dx2[sl], zd2[sl], g2, meta.rs2[sl] if meta.rs2 is not None else None, dz2[sl], dg2, db2,
                          n, d, drop=meta.drop, seed=seed + 21 + 2 * ei)
            hip.gemm_nt(hip.EPI_DGELU, dz2[sl], sh.get(w2)[1], n, hid, d, du[sl], aux=u[sl], drop=meta.drop,
                        seed=seed + 20 + 2 * ei, tile=meta.tile)
            with fork:      # off the critical path: dW2, db1, dW1
                hip.gemm_tn(dz2[sl], hh[sl], dw2, n, d, hid)
                hip.colsum(du[sl], db1, n, hid)
                hip.gemm_tn(du[sl], y2[sl], dw1, n, hid, d)
            hip.gemm_nt(hip.EPI_BIAS, du[sl], sh.get(w1)[1], n, d, hid, dy2[sl], tile=meta.tile)
            dexp += [dw1, db1, dw2, db2]
        dx1 = _empty((M, d), f32, dev)
        hip.ln_bwd(dy2, None, x1, n2w, mean2, rstd2, dx2, dx1, dn2w, dn2b, M, d)
        # ---- attention half
        dz1 = _empty((M, d), bf, dev)      # not aliased with dz2: the side stream may still read dz2
        hip.resid_bwd(dx1, zd1, g1, meta.rs1, dz1, dg1, dproj_b, M, d, drop=meta.drop, seed=seed + 1)
        dctx = dy2         # reuse: dy2 was consumed by ln_bwd above on this stream, never read on the side
        hip.gemm_nt(hip.EPI_BIAS, dz1, sh.get(proj_w)[1], M, d, d, dctx, tile=meta.tile)
        with fork:
            hip.gemm_tn(dz1, ctxb, dproj_w, M, d, d)
        dqkv = _empty((M, 3 * d), bf, dev)
        for li, ((seg, nseq, maxlen), lse) in enumerate(zip(pl.attn_launches(meta.fused), lses)):
            hip.attn_bwd(qkv, ctxb, dctx, lse, seg, nseq, pl.keymask, dqkv, H, d, maxlen, (d // H) ** -0.5,
                         drop=meta.attn_drop, seed=seed + 11 + li)
        dqkv_b = z(3 * d)
        with fork:
            hip.colsum(dqkv, dqkv_b, M, 3 * d)
            hip.gemm_tn(dqkv, y1, dqkv_w, M, 3 * d, d)
        dy1 = _empty((M, d), bf, dev)
        hip.gemm_nt(hip.EPI_BIAS, dqkv, sh.get(qkv_w)[1], M, d, 3 * d, dy1, tile=meta.tile)
        dx0 = _empty((M, d), f32, dev)
        hip.ln_bwd(dy1, None, x, n1w, mean1, rstd1, dx1, dx0, dn1w, dn1b, M, d)
        fork.join()        # gradients are complete (and every buffer the side stream read is free) from here
        grads = [dg1, dg2, dn1w, dn1b, dqkv_w, dqkv_b[:d], dqkv_b[2 * d:], dproj_w, dproj_b, dn2w, dn2b] + dexp
        return (dx0, None, *grads)


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
        z = lambda t: torch.zeros_like(t, dtype=torch.float32)
        g = [None] * 11   # patch_w, patch_b, cls, mask, pos, type, word, tpos, btype, ln_w, ln_b
        dtype_emb = z(type_emb)
        if T:
            xhat, rstd = saved['txt']
            dword = z(word)
            dtpos = torch.zeros((T, d), dtype=torch.float32, device=dev)
            dbtype = torch.zeros((2, d), dtype=torch.float32, device=dev)
            dlnw, dlnb = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
            hip.embed_txt_bwd(dx[:pl.nt], meta['ids'], xhat, rstd, ln_w, dword, dtpos, dbtype[0], dlnw, dlnb,
                              dtype_emb[0], B, T, d, drop=drop, seed=seed + 1)
            g[6], g[8], g[9], g[10] = dword, dbtype, dlnw, dlnb
            g[7] = dtpos if meta['tpos_rows'] == T else torch.cat(
                [dtpos, torch.zeros((meta['tpos_rows'] - T, d), device=dev)])
        if P:
            npatch = P - 1
            patches = saved['img']
            dproj = torch.empty((B * npatch, d), dtype=torch.bfloat16, device=dev)
            dcls, dmask = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
            dpos = torch.zeros((P, d), device=dev)
            hip.embed_img_bwd(dx[pl.nt:], meta['masked'], dproj, dcls, dmask, dpos, dtype_emb[meta['img_type']],
                              B, npatch, d, drop=drop, seed=seed + 2)
            dpw = torch.zeros((d, patches.shape[1]), device=dev)
            dpb = torch.zeros(d, device=dev)
            hip.gemm_tn(dproj, patches, dpw, B * npatch, d, patches.shape[1])
            hip.colsum(dproj, dpb, B * npatch, d)
            g[0], g[1] = dpw.view_as(patch_w), dpb
            g[2], g[4] = dcls, dpos
            g[3] = dmask if meta['masked'] is not None else None
        g[5] = dtype_emb
        return (None, *g)
