"""Host-side mirror of VlmoModule (models/vlmo/vlmo_module.py:14-442): same constructor
(attribute-style config), parameter names, `infer` / `forward` / `load_from_ckpt` /
`no_weight_decay` contracts.  Pretraining losses [mlm, mim, itc, itm]; the downstream
(vqa / nlvr2 / irtr / mpp), EMA and negative-queue branches raise NotImplementedError
(out of scope, off in conf/train/pretrain_mum.yaml)."""
import math
from collections import defaultdict
from functools import partial

import numpy as np
import os

import torch
import torch.nn as nn

from . import objectives
from .heads import ITCHead, ITMHead, MIMHead, MLMHead
from .vlmo import VLMO, LayerNorm


class VlmoModule(nn.Module):

    def __init__(self, config):
        super().__init__()
        self.config = config
        model_cfg = config.model
        norm_layer = partial(LayerNorm, eps=1e-12, export='fused' not in model_cfg.norm_layer)
        self.transformer = VLMO(
            img_size=model_cfg.img_size, patch_size=model_cfg.patch_size, in_chans=model_cfg.in_chans,
            num_classes=model_cfg.num_classes, embed_dim=model_cfg.embed_dim, depth=model_cfg.depth,
            num_heads=model_cfg.num_heads, mlp_ratio=model_cfg.mlp_ratio, qkv_bias=model_cfg.qkv_bias,
            qk_scale=None, drop_rate=model_cfg.drop_rate, attn_drop_rate=model_cfg.attn_drop_rate,
            drop_path_rate=model_cfg.drop_path_rate, norm_layer=norm_layer, init_values=model_cfg.init_values,
            vocab_size=model_cfg.vocab_size, max_text_len=model_cfg.max_text_len,
            fusion_layer=model_cfg.fusion_layer)
        self._freeze_params()

        self.loss_names = config.train.loss_names
        hs = model_cfg.embed_dim
        for unsupported in ('mpp', 'vqa', 'nlvr2', 'irtr', 'refcoco'):
            if unsupported in self.loss_names:
                raise NotImplementedError(f'loss {unsupported!r} belongs to a downstream phase outside the '
                                          'pretraining hot path')
        if 'mlm' in self.loss_names:
            self.mlm_head = MLMHead(self.transformer.bert_config,
                                    weight=self.transformer.txt_embeddings.word_embeddings.weight)
            self.mlm_head.apply(self.transformer._init_weights)
        if 'itc' in self.loss_names:
            self.itc_head = ITCHead(hs, model_cfg.itc_dim)
            self.itc_head.apply(self.transformer._init_weights)
            self.itc_temp = nn.Parameter(torch.ones([]) * np.log(1 / model_cfg.itc_temp))
        if 'itm' in self.loss_names:
            self.itm_head = ITMHead(hs)
            self.itm_head.apply(self.transformer._init_weights)
        if 'mim' in self.loss_names:
            self.d_vae = objectives.create_d_vae(weight_path=config.train.discrete_vae_weight_path,
                                                 d_vae_type=config.train.discrete_vae_type, device='cpu',
                                                 image_size=model_cfg.img_size // 2,
                                                 vocab_size=model_cfg.img_vocab_size)
            for p in self.d_vae.parameters():
                p.requires_grad = False
            self.mim_head = MIMHead(hs, model_cfg.img_vocab_size)
            self.mim_head.apply(self.transformer._init_weights)

        self.transformer_m = None
        if getattr(self.config, 'vlmo_ema', False):
            raise NotImplementedError('vlmo_ema (momentum twin) is out of scope; config.yaml:136 default is off')
        if hasattr(config.train, 'neg_queue') and config.train.neg_queue:
            raise NotImplementedError('neg_queue is out of scope; pretrain_mum.yaml:40 default is off')
        self.q_size = 0
        self.img_queue, self.txt_queue = None, None

    def _freeze_params(self):
        """vlmo_module.py:148-167."""
        if self.config.train.phase in ['pretrain_txt']:
            for b in self.transformer.blocks:
                del b.mlp['vl']
                if self.config.train.fixed_attn:
                    b.gamma_1.requires_grad = False
                    b.gamma_2.requires_grad = False
                    for p in b.attn.parameters():
                        p.requires_grad = False
                    for p in b.norm1.parameters():
                        p.requires_grad = False
                    for p in b.norm2.parameters():
                        p.requires_grad = False
            for p in self.transformer.norm.parameters():
                p.requires_grad = False
        elif self.config.train.phase in ['pretrain_mum', 'finetune_vqa']:
            for b in self.transformer.blocks[:self.transformer.fusion_layer]:
                del b.mlp['vl']

    def _adjust_downstream_params(self):
        """vlmo_module.py:169-185: only touches nlvr2 / irtr heads, which this module never builds."""

    # ------------------------------------------------------------- checkpoints
    def interpolate_pos_embedding(self, state_dict):
        """Fit a checkpoint's position tables to this model (behaviour of vlmo_module.py:187-232): the image table is
        resampled bicubically when the checkpoint was trained on another patch grid (class / extra tokens kept as they
        are), the text table is cut to ``max_text_len`` rows."""
        tr = self.transformer
        grid = math.isqrt(tr.patch_embed.num_patches)
        n_extra = tr.pos_embed.shape[-2] - tr.patch_embed.num_patches
        for key in ('pos_embed', 'transformer.pos_embed'):
            table = state_dict.get(key)
            if table is None:
                continue
            src = math.isqrt(table.shape[-2] - n_extra)
            if src == grid:
                continue
            head, body = table[:, :n_extra], table[:, n_extra:]
            body = body.unflatten(1, (src, src)).movedim(-1, 1)                       # [1, C, src, src]
            body = torch.nn.functional.interpolate(body, size=(grid, grid), mode='bicubic', align_corners=False)
            state_dict[key] = torch.cat([head, body.movedim(1, -1).flatten(1, 2)], dim=1)
        txt = 'transformer.txt_embeddings.position_embeddings.weight'
        if txt in state_dict:
            state_dict[txt] = state_dict[txt][:tr.max_text_len]
        state_dict.pop('transformer.txt_embeddings.position_ids', None)   # non-persistent buffer upstream
        return state_dict

    def _load_vlmo(self, state_dict):
        for k in list(state_dict.keys()):
            for old, new in (('.mlp.v_mlp', '.mlp.v'), ('.mlp.l_mlp', '.mlp.l'), ('.mlp.vl_mlp', '.mlp.vl')):
                if old in k:
                    state_dict[k.replace(old, new)] = state_dict.pop(k)
                    break
        matching = self.load_state_dict(state_dict, strict=False)
        self._adjust_downstream_params()
        return matching

    def _load_beit(self, state_dict):
        for k in list(state_dict.keys()):
            nk = k
            if 'mlp' in nk:
                nk = nk.replace('.mlp', '.mlp.v')
            if 'mask_token' in nk:
                nk = nk.replace('mask_token', 'img_mask_token')
            elif 'cls_token' in nk:
                nk = nk.replace('cls_token', 'img_cls_token')
            if 'lm_head' in nk:
                nk = nk.replace('lm_head', 'fc')
            if nk != k:
                state_dict[nk] = state_dict.pop(k)
        if 'mim' in self.loss_names:
            self.mim_head.load_state_dict(state_dict, strict=False)
        matching = self.transformer.load_state_dict(state_dict, strict=False)
        self._adjust_downstream_params()
        return matching

    def load_from_ckpt(self, state_dict):
        """vlmo_module.py:300-319 -> (matching, is_beit)."""
        state_dict = self.interpolate_pos_embedding(state_dict)
        is_beit = not any(('.mlp.v' in k or '.mlp.l' in k or '.mlp.vl' in k) for k in state_dict)
        matching = (self._load_beit if is_beit else self._load_vlmo)(state_dict)
        return matching, is_beit

    # ------------------------------------------------------------------ infer
    def infer(self, batch, infer_mode='img-txt', mask_txt=False, mask_img=False, image_token_type_idx=1,
              momentum_mode=False):
        """vlmo_module.py:321-393."""
        assert infer_mode in ['img_only', 'txt_only', 'img-txt']
        if momentum_mode:
            assert self.transformer_m is not None
        transformer = self.transformer
        img, img_attn_masks, bool_masked_pos = None, None, None
        txt_ids, txt_labels, txt_attn_masks = None, None, None
        if 'img' in infer_mode:
            imgkey = f'image_{image_token_type_idx - 1}' if f'image_{image_token_type_idx - 1}' in batch else 'image'
            img = batch[imgkey]
            B = img.size(0)
            img_attn_masks = torch.ones([B, transformer.patch_embed.num_patches + 1], dtype=torch.int64,
                                        device=img.device)
            bool_masked_pos = batch['image_bool_masked_pos'] if mask_img else None
        if 'txt' in infer_mode:
            do_mlm = '_mlm' if mask_txt else ''
            txt_ids = batch[f'text_ids{do_mlm}']
            txt_labels = batch[f'text_labels{do_mlm}'] if mask_txt else None
            txt_attn_masks = batch['text_mask']
        co_feats, _ = transformer.forward_features(img=img, txt=txt_ids, img_attn_masks=img_attn_masks,
                                                   txt_attn_masks=txt_attn_masks, bool_masked_pos=bool_masked_pos,
                                                   fusion_layer=None)
        if txt_ids is not None:
            txt_feats, img_feats = (co_feats[:, :transformer.max_text_len], co_feats[:, transformer.max_text_len:])
        else:
            txt_feats, img_feats = None, co_feats
        cls_feats = transformer.pooler(co_feats)
        return {'txt_feats': txt_feats, 'img_feats': img_feats, 'co_feats': co_feats, 'cls_feats': cls_feats,
                'img_masks': img_attn_masks, 'img_bool_masked_pos': bool_masked_pos, 'txt_labels': txt_labels,
                'txt_ids': txt_ids, 'txt_masks': txt_attn_masks}

    # ------------------------------------------------- merged backbone passes
    @staticmethod
    def _split_infer(out, sizes):
        """Cut the batch dimension of one infer() result into consecutive pieces of the given sizes."""
        parts, b0 = [], 0
        for n in sizes:
            parts.append({k: (v[b0:b0 + n] if torch.is_tensor(v) and v.dim() > 0 else v) for k, v in out.items()})
            b0 += n
        return parts

    def _forward_merged(self, batch):
        """The same objectives as forward(), with the backbone passes that share a mode batched into ONE pass each
        (SURVEY 8f-1): V = [ITC image | MIM masked image], L = ITC text, VL = [MLM | ITM positive | ITM negatives].
        Every sample goes through exactly the arithmetic of its own pass (rows are independent: LayerNorm per token,
        attention per sequence), so the results equal the pass-by-pass ones except for the dropout streams; 3
        launches of the 12/24-block stack instead of 7 halve the host time per step."""
        ret = dict()
        names = self.loss_names
        B = batch['image'].size(0)
        dev = batch['image'].device
        if 'mim' in names:
            with torch.no_grad():
                batch['image_bool_masked_pos'] = batch['image_bool_masked_pos'].flatten(1).to(torch.bool)
        # ---- V and L passes as ONE pass of unfused (image, text) pairs: forward_features with fusion_layer = depth runs the
        # image tokens through the 'v' experts and the text tokens through the 'l' experts of EVERY block and never joins them
        # (vlmo.py:400-413 with an empty 'vl' range), i.e. exactly the image-only and text-only passes side by side.  The
        # text-only pass of B sequences alone is ~300 launches of 2 048-row kernels; riding with the 2B images it costs its rows.
        # Pairs: (ITC image_i, text_i), (MIM masked image_i, text_i again -- its text half is discarded, no gradient enters it).
        unfused = ('itc' in names and 'mim' in names and self.transformer_m is None
                   and getattr(self.config.train, 'merge_unfused', os.environ.get('VLMO_MERGE_UNFUSED', '1') != '0'))
        if unfused:
            tr = self.transformer
            T, depth = tr.max_text_len, len(tr.blocks)
            bmp = batch['image_bool_masked_pos']
            img2 = torch.cat([batch['image'], batch['image']], 0)
            bmp2 = torch.cat([torch.zeros_like(bmp), bmp], 0)
            ids2 = torch.cat([batch['text_ids'], batch['text_ids']], 0)
            tm2 = torch.cat([batch['text_mask'], batch['text_mask']], 0)
            ones = torch.ones([2 * B, tr.patch_embed.num_patches + 1], dtype=torch.int64, device=dev)
            co, _ = tr.forward_features(img=img2, txt=ids2, img_attn_masks=ones, txt_attn_masks=tm2, bool_masked_pos=bmp2,
                                        fusion_layer=depth)
            img_itc, img_mim, txt_itc = co[:B, T:], co[B:, T:], co[:B, :T]
            batch['_itc_img_infer'] = {'txt_feats': None, 'img_feats': img_itc, 'co_feats': img_itc, 'cls_feats': None,
                                       'img_masks': ones[:B], 'img_bool_masked_pos': None, 'txt_labels': None,
                                       'txt_ids': None, 'txt_masks': None}
            batch['_mim_infer'] = {'txt_feats': None, 'img_feats': img_mim, 'co_feats': img_mim, 'cls_feats': None,
                                   'img_masks': ones[B:], 'img_bool_masked_pos': bmp, 'txt_labels': None,
                                   'txt_ids': None, 'txt_masks': None}
            batch['_itc_txt_infer'] = {'txt_feats': txt_itc, 'img_feats': None, 'co_feats': txt_itc, 'cls_feats': None,
                                       'img_masks': None, 'img_bool_masked_pos': None, 'txt_labels': None,
                                       'txt_ids': batch['text_ids'], 'txt_masks': batch['text_mask']}
        # ---- V pass
        v_parts = []
        if 'itc' in names and not unfused:
            v_parts.append('itc')
        if 'mim' in names and not unfused:
            v_parts.append('mim')
        if v_parts:
            nb = len(v_parts)
            bmp = batch['image_bool_masked_pos'] if 'mim' in names else None
            vb = {'image': torch.cat([batch['image']] * nb, 0) if nb > 1 else batch['image']}
            if bmp is not None:
                zeros = torch.zeros_like(bmp)
                vb['image_bool_masked_pos'] = torch.cat([zeros if p == 'itc' else bmp for p in v_parts], 0)
            out = self.infer(vb, infer_mode='img_only', mask_img=bmp is not None)
            for p, piece in zip(v_parts, self._split_infer(out, [B] * nb)):
                if p == 'itc':
                    piece['img_bool_masked_pos'] = None
                    batch['_itc_img_infer'] = piece
                else:
                    batch['_mim_infer'] = piece
        # ---- L pass, then ITC (its similarities drive the ITM hard negatives)
        if 'itc' in names:
            if not unfused:
                batch['_itc_txt_infer'] = self.infer(batch, infer_mode='txt_only')
            ret.update(objectives.compute_itc(self, batch))
        # ---- VL pass
        vl_parts, ids, masks, imgs = [], [], [], []
        if 'mlm' in names:
            vl_parts.append(('mlm', B))
            ids.append(batch['text_ids_mlm'])
            masks.append(batch['text_mask'])
            imgs.append(batch['image'])
        if 'itm' in names:
            img_neg_idx, txt_neg_idx = objectives.sample_itm_negatives(batch, ret if 'itc' in names else None)
            neg = objectives.itm_negative_batch(batch, img_neg_idx, txt_neg_idx)
            vl_parts += [('itm_pos', B), ('itm_neg', 2 * B)]
            ids += [batch['text_ids'], neg['text_ids']]
            masks += [batch['text_mask'], neg['text_mask']]
            imgs += [batch['image'], neg['image']]
        if vl_parts:
            out = self.infer({'text_ids': torch.cat(ids, 0), 'text_mask': torch.cat(masks, 0), 'image': torch.cat(imgs, 0)},
                             infer_mode='img-txt')
            pieces = dict(zip([p for p, _ in vl_parts], self._split_infer(out, [n for _, n in vl_parts])))
            if 'mlm' in pieces:
                pieces['mlm']['txt_labels'] = batch['text_labels_mlm']
                batch['_mlm_infer'] = pieces['mlm']
            if 'itm_pos' in pieces:
                batch['_itm_infer'] = (pieces['itm_pos'], pieces['itm_neg'])
        if 'mlm' in names:
            ret.update(objectives.compute_mlm(self, batch))
        if 'mim' in names:
            ret.update(objectives.compute_mim(self, batch))
        if 'itm' in names:
            ret.update(objectives.compute_itm(self, batch, ret if 'itc' in names else None))
        return ret

    def forward(self, batch):
        """vlmo_module.py:395-436.  ``config.train.merge_passes = True`` (not a reference option) batches the
        backbone passes of the objectives by mode -- see _forward_merged."""
        batch = defaultdict(lambda: None, batch)
        ret = dict()
        if len(self.loss_names) == 0:
            ret.update(self.infer(batch))
            return ret
        if (getattr(self.config.train, 'merge_passes', False) and batch['image'] is not None
                and batch['text_ids'] is not None and getattr(self.config.train, 'mim_head_pos', 'img') == 'img'):
            return self._forward_merged(batch)
        if 'mlm' in self.loss_names:
            ret.update(objectives.compute_mlm(self, batch))
        if 'mim' in self.loss_names:
            ret.update(objectives.compute_mim(self, batch))
        if 'itc' in self.loss_names:
            ret.update(objectives.compute_itc(self, batch))
        if 'itm' in self.loss_names:
            itc_ret = ret if 'itc' in self.loss_names else None
            ret.update(objectives.compute_itm(self, batch, itc_ret))
        return ret

    @torch.jit.ignore
    def no_weight_decay(self):
        return {'itc_temp', 'transformer.pos_embed', 'transformer.img_cls_token'}
