"""Host-side mirror of the reference backbone (models/vlmo/vlmo.py).

Same class names, constructor arguments, parameter names (state-dict keys),
method signatures and error behaviour as the reference's ``VLMO`` / ``Block`` /
``Attention``; the compute goes through the HIP engine (engine.py -> hip.py ->
libvlmo_hip.so).  The nn.Linear / nn.LayerNorm / nn.Embedding / nn.Conv2d
children are parameter containers only (never called), so checkpoints of the
reference load unchanged.

Documented deviations:
  * ``Attention`` / ``Block`` return ``None`` for the attention-probability
    tensor (vlmo.py:98): every caller in the reference drops it and the fused
    kernel never materialises [B,h,N,N];
  * inputs must live on a gfx950 device; there is no CPU path.
"""
import math
from functools import partial
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import engine, hip


def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    # timm.models.layers.trunc_normal_: truncation bounds are ABSOLUTE
    return nn.init.trunc_normal_(tensor, mean, std, a, b)


def LayerNorm(normalized_shape, eps=1e-5, elementwise_affine=True, export=False):
    """vlmo.py:26-36.  The apex/nn.LayerNorm choice is irrelevant here: the
    module only holds weight/bias; the HIP kernel does the arithmetic."""
    return nn.LayerNorm(normalized_shape, eps, elementwise_affine)


class Mlp(nn.Module):
    """timm Mlp parameter container (fc1 -> GELU(erf) -> drop -> fc2 -> drop)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop = drop


class PatchEmbed(nn.Module):
    """timm PatchEmbed parameter container (conv k = s = patch)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)


class BertEmbeddings(nn.Module):
    """transformers BertEmbeddings parameter container."""

    def __init__(self, vocab_size, hidden_size, max_position_embeddings, layer_norm_eps=1e-12):
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab_size, hidden_size, padding_idx=0)
        self.position_embeddings = nn.Embedding(max_position_embeddings, hidden_size)
        self.token_type_embeddings = nn.Embedding(2, hidden_size)
        self.LayerNorm = nn.LayerNorm(hidden_size, eps=layer_norm_eps)


class BertPooler(nn.Module):
    """tanh(W x[:, 0] + b) (vlmo_module.py:379); tiny, runs as stock torch ops."""

    def __init__(self, hidden_size):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)

    def forward(self, hidden_states):
        return torch.tanh(nn.functional.linear(hidden_states[:, 0], self.dense.weight, self.dense.bias))


class Attention(nn.Module):
    """vlmo.py:39-98 (parameters: qkv.weight, q_bias, v_bias, proj.*)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        self.num_heads = num_heads
        head_dim = dim // num_heads
        if head_dim != 64:
            raise NotImplementedError(f'the gfx950 attention kernel is built for head_dim 64, got {head_dim}')
        if qk_scale is not None:
            raise NotImplementedError('qk_scale override is not supported')
        self.scale = head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=False)
        if qkv_bias:
            self.q_bias = nn.Parameter(torch.zeros(dim))
            self.v_bias = nn.Parameter(torch.zeros(dim))
        else:       # vlmo.py:60-62: no bias parameters; the engine is handed a constant zero vector instead
            self.q_bias = None
            self.v_bias = None
            self.register_buffer('_zero_bias', torch.zeros(dim), persistent=False)
        self.attn_drop = attn_drop
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = proj_drop


class Block(nn.Module):
    """vlmo.py:101-197."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, init_values=None, act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.num_heads = dim, num_heads
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                              attn_drop=attn_drop, proj_drop=drop)
        self.drop_path_rate = float(drop_path)
        self.drop = drop
        self.norm2 = norm_layer(dim)
        self.mlp_hidden_dim = int(dim * mlp_ratio)
        self.mlp = nn.ModuleDict({k: Mlp(in_features=dim, hidden_features=self.mlp_hidden_dim,
                                         act_layer=act_layer, drop=drop) for k in ('v', 'l', 'vl')})
        if init_values is not None and init_values > 0:
            self.gamma_1 = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)
            self.gamma_2 = nn.Parameter(init_values * torch.ones((dim)), requires_grad=True)
        else:       # vlmo.py:185-186, 190-192: no layer-scale; the engine multiplies by a constant vector of ones
            self.gamma_1, self.gamma_2 = None, None
            self.register_buffer('_unit_scale', torch.ones(dim), persistent=False)

    # -- engine plumbing -----------------------------------------------------
    def _params(self, routes):
        a = self.attn
        g1, g2 = (self.gamma_1, self.gamma_2) if self.gamma_1 is not None else (self._unit_scale, self._unit_scale)
        qb, vb = (a.q_bias, a.v_bias) if a.q_bias is not None else (a._zero_bias, a._zero_bias)
        ps = [g1, g2, self.norm1.weight, self.norm1.bias, a.qkv.weight, qb, vb,
              a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias]
        for r in routes:
            m = self.mlp[r]
            ps += [m.fc1.weight, m.fc1.bias, m.fc2.weight, m.fc2.bias]
        return ps

    def meta(self, x, plan, routes, ranges, fused, shadows, seed, drop_scales=None):
        """engine.BlockMeta + parameter list of one call of this block.  routes/ranges: experts and their row
        ranges; drop_scales: this block's drop-path scales [branch, stream, B] when the caller drew them for all
        blocks at once."""
        rs1 = rs2 = None
        if drop_scales is not None:
            if self.drop_path_rate > 0.0:
                rs1, rs2 = drop_scales[0].reshape(-1), drop_scales[1].reshape(-1)
        elif self.training and self.drop_path_rate > 0.0:
            # timm DropPath: per-sample Bernoulli(keep)/keep, one draw per residual branch.  Below the fusion
            # layer the reference calls the block once per modality (vlmo.py:402-404), so the text and image
            # streams of a sample draw independently; above it the fused sequence shares one draw.
            # Layout [branch, stream, B]; rows find their entry through plan.row_group inside the kernels.
            keep = 1.0 - self.drop_path_rate
            sc = torch.empty((2, 2, plan.B), device=x.device).bernoulli_(keep).div_(keep)
            if fused or not (plan.T and plan.P):
                sc[:, 1 if plan.T else 0] = sc[:, 0 if plan.T else 1]
            rs1, rs2 = sc[0].reshape(-1), sc[1].reshape(-1)
        meta = engine.BlockMeta(plan, self.num_heads, self.dim, self.mlp_hidden_dim, fused, ranges, self.training,
                                self.drop, self.attn.attn_drop, rs1, rs2, seed, eps=self.norm1.eps)
        meta.shadows = shadows
        return meta, self._params(routes)

    def run(self, x, plan, routes, ranges, fused, shadows, seed, drop_scales=None):
        """x: packed fp32 [M, d] -> this block's output (one native call per direction)."""
        meta, params = self.meta(x, plan, routes, ranges, fused, shadows, seed, drop_scales)
        if engine.USE_STACK:
            return engine.StackFn.apply(x, [meta], *params)
        return engine.BlockFn.apply(x, meta, *params)

    def forward(self, x, mask=None, route='vl'):
        """Reference signature (vlmo.py:187): x [B, N, d] -> (x, attn=None)."""
        owner = getattr(self, '_owner', None)
        shadows = owner()._shadows if owner is not None else engine.ShadowCache()
        B, N, d = x.shape
        plan = engine.Plan(B, 0, N, x.device, None, mask)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if self.training else 0
        out = self.run(x.reshape(B * N, d).float().contiguous(), plan, [route], [(0, B * N)], False, shadows, seed)
        return out.view(B, N, d), None


_CONSTS = {}


def _const_to(t, device):
    """Small host constant on the device, uploaded once per (values, device)."""
    key = (tuple(t.reshape(-1).tolist()), tuple(t.shape), str(device))
    v = _CONSTS.get(key)
    if v is None:
        v = _CONSTS[key] = t.to(device)
    return v


class VLMO(nn.Module):
    """vlmo.py:200-477."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, num_classes=1000, embed_dim=768, depth=12,
                 num_heads=12, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, norm_layer=None, init_values=None, vocab_size=30000, max_text_len=27,
                 fusion_layer=3):
        super().__init__()
        self.num_classes = num_classes
        self.num_features = self.embed_dim = embed_dim
        norm_layer = norm_layer or partial(nn.LayerNorm, eps=1e-6)

        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans,
                                      embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.patch_size = patch_size
        self.patch_dim = img_size // patch_size
        self.pos_embed = nn.parameter.Parameter(torch.zeros(1, num_patches + 1, embed_dim))
        self.drop_rate = drop_rate

        self.max_text_len = max_text_len
        # attribute-compatible stand-in for transformers.BertConfig (vlmo.py:248-259)
        self.bert_config = SimpleNamespace(
            vocab_size=vocab_size, hidden_size=embed_dim, num_hidden_layers=depth, num_attention_heads=num_heads,
            intermediate_size=int(embed_dim * mlp_ratio), max_position_embeddings=max_text_len,
            hidden_dropout_prob=drop_rate, attention_probs_dropout_prob=drop_rate, layer_norm_eps=1e-12,
            hidden_act='gelu')
        self.txt_embeddings = BertEmbeddings(vocab_size, embed_dim, max_text_len)
        self.token_type_embeddings = nn.Embedding(2, embed_dim)
        self.img_cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.img_mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.fusion_layer = fusion_layer

        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                  drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer,
                  init_values=init_values) for i in range(depth)])
        self.norm = norm_layer(embed_dim)

        self.pooler = BertPooler(embed_dim)
        self.head = nn.Identity()

        trunc_normal_(self.pos_embed, std=0.02)
        trunc_normal_(self.img_cls_token, std=0.02)
        self.apply(self._init_weights)

        self._shadows = engine.ShadowCache()
        self._plans = {}
        import weakref
        for b in self.blocks:
            object.__setattr__(b, '_owner', weakref.ref(self))

    # ------------------------------------------------------------------ utils
    def _init_weights(self, m):
        if isinstance(m, (nn.Linear, nn.Embedding)):
            trunc_normal_(m.weight, std=0.02)
            if isinstance(m, nn.Linear) and m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Conv2d):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, (nn.LayerNorm, nn.GroupNorm, nn.BatchNorm2d)):
            nn.init.zeros_(m.bias)
            nn.init.ones_(m.weight)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_embed", "img_cls_token"}

    def _seed(self):
        # one draw from torch's CPU generator per pass keeps torch.manual_seed reproducibility
        return int(torch.randint(0, 2 ** 62, (1,)).item()) if self.training else 0

    def _routes(self, i, mode, fusion_layer, plan):
        """experts and packed row ranges of block i."""
        if mode == 'v':
            return ['v'], [(0, plan.M)], False
        if mode == 'l':
            return ['l'], [(0, plan.M)], False
        if mode == 'vl_all':
            return ['vl'], [(0, plan.M)], True
        if i < fusion_layer:
            return ['l', 'v'], [(0, plan.nt), (plan.nt, plan.ni)], False
        return ['vl'], [(0, plan.M)], True

    def _embed(self, plan, img, txt, bool_masked_pos, img_token_type_idx, seed):
        te = self.txt_embeddings
        if img is not None:
            if img.dtype != torch.float32:
                img = img.float()
            img = img.contiguous()
        masked = None
        if bool_masked_pos is not None and img is not None:
            masked = bool_masked_pos.reshape(plan.B, -1).to(torch.uint8).contiguous()
        if img_token_type_idx >= self.token_type_embeddings.weight.shape[0]:
            raise IndexError('img_token_type_idx out of range for token_type_embeddings')
        meta = dict(plan=plan, d=self.embed_dim, device=plan.device, img=img,
                    ids=txt.contiguous() if txt is not None else None, masked=masked,
                    drop=hip.drop_params(self.drop_rate, self.training), seed=seed, shadows=self._shadows,
                    patch=self.patch_size, img_type=img_token_type_idx, txt_eps=te.LayerNorm.eps,
                    tpos_rows=te.position_embeddings.weight.shape[0])
        return engine.EmbedFn.apply(
            meta, self.patch_embed.proj.weight, self.patch_embed.proj.bias, self.img_cls_token.view(-1),
            self.img_mask_token.view(-1), self.pos_embed.view(-1, self.embed_dim),
            self.token_type_embeddings.weight, te.word_embeddings.weight, te.position_embeddings.weight,
            te.token_type_embeddings.weight, te.LayerNorm.weight, te.LayerNorm.bias)

    def _check_inputs(self, img, txt, img_attn_masks, txt_attn_masks):
        if img is None and txt is None:
            raise ValueError('forward_features needs img and/or txt')
        dev = (img if img is not None else txt).device
        if dev.type != 'cuda':
            raise RuntimeError('exploremultimodal_amd runs on MI355X only: inputs must be on a cuda (ROCm) device')
        if txt is not None and txt.shape[1] > self.max_text_len:
            raise IndexError(f'text length {txt.shape[1]} exceeds max_text_len {self.max_text_len}')
        return dev

    def invalidate_weight_shadows(self):
        """Drop the cached bf16 copies of the GEMM weights.  They are re-cast automatically when a parameter's
        autograd version counter changes (optimizer steps, ``load_state_dict``, any in-place torch op); code that
        writes weights through ``param.data`` or raw pointers must call this (or
        ``torch.autograd.graph.increment_version``)."""
        self._shadows.clear()

    def _run_blocks(self, x, plan, mode, fusion_layer, layers, seed):
        layers = list(layers)
        todo = [(i,) + tuple(self._routes(i, mode, fusion_layer, plan)) for i in layers]
        # drop-path draws of ALL blocks in three tiny kernels instead of three per block (each one is a launch
        # and a dependency bubble on the main stream): Bernoulli(keep_i) / keep_i, layout [block, branch, stream, B]
        scales = None
        rates = [self.blocks[i].drop_path_rate for i in layers]
        if self.training and layers and max(rates) > 0.0:
            keep = 1.0 - torch.tensor(rates, dtype=torch.float32).view(-1, 1, 1, 1)
            share = torch.tensor([1.0 if (f or not (plan.T and plan.P)) else 0.0 for _, _, _, f in todo]).view(-1, 1, 1)
            kd, sd = _const_to(keep, x.device), _const_to(share, x.device)
            u = torch.rand((len(layers), 2, 2, plan.B), device=x.device)
            first = 0 if plan.T else 1
            # fused / single-stream blocks: both streams of a sample share the draw of the stream that exists
            u = torch.where(sd.unsqueeze(-1) > 0, u[:, :, first:first + 1].expand_as(u), u)
            scales = (u < kd).to(torch.float32) / kd
        if engine.USE_STACK and todo:
            # the whole stack in ONE native call per direction (engine.StackFn)
            metas, params = [], []
            for n, (i, routes, ranges, fused) in enumerate(todo):
                mt, ps = self.blocks[i].meta(x, plan, routes, ranges, fused, self._shadows, seed + 1000 * (i + 1),
                                             drop_scales=scales[n] if scales is not None else None)
                metas.append(mt)
                params += ps
            return engine.StackFn.apply(x, metas, *params)
        for n, (i, routes, ranges, fused) in enumerate(todo):
            x = self.blocks[i].run(x, plan, routes, ranges, fused, self._shadows, seed + 1000 * (i + 1),
                                   drop_scales=scales[n] if scales is not None else None)
        return x

    # ------------------------------------------------------------ reference API
    def embed_img(self, x, img_masks, bool_masked_pos=None, img_token_type_idx=1):
        """vlmo.py:298-319 -> fp32 [B, P, d]."""
        B = x.shape[0]
        P = self.patch_embed.num_patches + 1
        plan = engine.Plan(B, 0, P, x.device)
        return self._embed(plan, x, None, bool_masked_pos, img_token_type_idx, self._seed()).view(B, P, -1)

    def embed_txt(self, x, txt_attn_masks):
        """vlmo.py:321-324 -> fp32 [B, T, d]."""
        B, T = x.shape
        plan = engine.Plan(B, T, 0, x.device)
        return self._embed(plan, None, x, None, 1, self._seed()).view(B, T, -1)

    def forward_interval(self, x, attn_masks, route=None, need_embed=False, bool_masked_pos=None, in_layer=None,
                         out_layer=None, img_token_type_idx=1, need_norm=False):
        """vlmo.py:326-355."""
        assert route in ['v', 'l', 'vl']
        B = x.size(0)
        seed = self._seed()
        P = self.patch_embed.num_patches + 1
        if need_embed and route in ['v']:
            if attn_masks is None:
                attn_masks = torch.ones([B, P], dtype=torch.int64, device=x.device)
            plan = engine.Plan(B, 0, P, x.device, None, attn_masks)
            h = self._embed(plan, x, None, bool_masked_pos, img_token_type_idx, seed)
            N = P
        elif need_embed and route in ['l']:
            N = x.shape[1]
            plan = engine.Plan(B, N, 0, x.device, attn_masks, None)
            h = self._embed(plan, None, x, None, 1, seed)
        else:
            N = x.shape[1]
            plan = engine.Plan(B, 0, N, x.device, None, attn_masks)
            h = x.reshape(B * N, -1).float().contiguous()
        mode = {'v': 'v', 'l': 'l', 'vl': 'vl_all'}[route]
        layers = list(range(len(self.blocks)))[in_layer:out_layer]
        h = self._run_blocks(h, plan, mode, 0, layers, seed)
        if need_norm:
            # single-stream plans have an identity row map
            return engine.FinalNormFn.apply(h, self.norm.weight, self.norm.bias, plan, (B, N, self.embed_dim),
                                            self.norm.eps)
        return h.view(B, N, self.embed_dim)

    def forward_features(self, img=None, txt=None, img_attn_masks=None, txt_attn_masks=None, bool_masked_pos=None,
                         fusion_layer=None, img_token_type_idx=1):
        """vlmo.py:357-414 -> (x fp32 [B, N, d], mask)."""
        dev = self._check_inputs(img, txt, img_attn_masks, txt_attn_masks)
        seed = self._seed()
        L = len(self.blocks)
        P = self.patch_embed.num_patches + 1
        if txt is None:
            B = img.shape[0]
            plan = engine.Plan(B, 0, P, dev, None, img_attn_masks)
            x = self._embed(plan, img, None, bool_masked_pos, img_token_type_idx, seed)
            x = self._run_blocks(x, plan, 'v', 0, range(L), seed)
            out = engine.FinalNormFn.apply(x, self.norm.weight, self.norm.bias, plan, (B, P, self.embed_dim),
                                              self.norm.eps)
            return out, img_attn_masks
        if img is None:
            B, T = txt.shape
            plan = engine.Plan(B, T, 0, dev, txt_attn_masks, None)
            x = self._embed(plan, None, txt, None, 1, seed)
            x = self._run_blocks(x, plan, 'l', 0, range(L), seed)
            out = engine.FinalNormFn.apply(x, self.norm.weight, self.norm.bias, plan, (B, T, self.embed_dim),
                                              self.norm.eps)
            return out, txt_attn_masks
        fusion_layer = fusion_layer or self.fusion_layer
        assert 0 <= fusion_layer <= self.bert_config.num_hidden_layers
        B, T = txt.shape
        plan = engine.Plan(B, T, P, dev, txt_attn_masks, img_attn_masks)
        x = self._embed(plan, img, txt, bool_masked_pos, img_token_type_idx, seed)
        x = self._run_blocks(x, plan, 'vl', fusion_layer, range(L), seed)
        out = engine.FinalNormFn.apply(x, self.norm.weight, self.norm.bias, plan, (B, T + P, self.embed_dim),
                                          self.norm.eps)
        co_attn_masks = torch.cat([txt_attn_masks, img_attn_masks], dim=1)
        return out, co_attn_masks

    def forward(self, img=None, txt=None, img_attn_masks=None, txt_attn_masks=None, fusion_layer=None,
                img_token_type_idx=1):
        """vlmo.py:416-435."""
        x, _ = self.forward_features(img=img, txt=txt, img_attn_masks=img_attn_masks,
                                     txt_attn_masks=txt_attn_masks, fusion_layer=fusion_layer,
                                     img_token_type_idx=img_token_type_idx)
        return self.head(x[:, 0])
