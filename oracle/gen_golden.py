"""Golden-vector generator: runs the REFERENCE itself (build container only).

    python -m oracle.gen_golden            # writes tests/golden/*.npz

Imports the unmodified reference sources from /root/reference by path
(models/vlmo/vlmo.py, dall_e/encoder.py, models/modeling_discrete_vae.py),
loads the key-addressed synthetic weights of oracle/synth.py into the
reference modules and records their CPU fp32 outputs and gradients.  The
reference cannot travel to the GPU box; only the .npz data produced here and
this script are committed.

timm is a third-party dependency of the reference that is neither vendored in
/root/reference nor installed here (misc/requirements.txt lists it unpinned).
The five symbols the reference imports from it are provided by a stand-in that
restates their published semantics: Mlp, PatchEmbed,
DropPath, trunc_normal_, ModelEmaV2.  Parity of those five is therefore
"unpinned by reference tests" and anchored on the reference's call sites
(vlmo.py:141-157, 231-237, 132-133).
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   'tests', 'golden')


def _install_timm_standin():
    import transformers  # noqa: F401  (must be imported before the stand-in exists)
    from transformers.models.bert import modeling_bert  # noqa: F401

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None,
                     act_layer=nn.GELU, drop=0.):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.drop1 = nn.Dropout(drop)
            self.fc2 = nn.Linear(hidden_features, out_features)
            self.drop2 = nn.Dropout(drop)

        def forward(self, x):
            return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768,
                     norm_layer=None):
            super().__init__()
            self.img_size = (img_size, img_size)
            self.patch_size = (patch_size, patch_size)
            self.grid_size = (img_size // patch_size, img_size // patch_size)
            self.num_patches = self.grid_size[0] * self.grid_size[1]
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size,
                                  stride=patch_size)
            self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

        def forward(self, x):
            return self.norm(self.proj(x).flatten(2).transpose(1, 2))

    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0. or not self.training:
                return x
            keep = 1 - self.drop_prob
            shape = (x.shape[0],) + (1,) * (x.ndim - 1)
            return x * x.new_empty(shape).bernoulli_(keep).div_(keep)

    def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
        return nn.init.trunc_normal_(tensor, mean, std, a, b)

    class ModelEmaV2(nn.Module):
        def __init__(self, model, decay=0.9999, device=None):
            super().__init__()
            raise NotImplementedError('EMA branch is out of scope')

    timm = types.ModuleType('timm')
    models = types.ModuleType('timm.models')
    layers = types.ModuleType('timm.models.layers')
    utils = types.ModuleType('timm.utils')
    layers.Mlp, layers.PatchEmbed = Mlp, PatchEmbed
    layers.DropPath, layers.trunc_normal_ = DropPath, trunc_normal_
    utils.ModelEmaV2 = ModelEmaV2
    timm.models, timm.utils, models.layers = models, utils, layers
    sys.modules.update({'timm': timm, 'timm.models': models,
                        'timm.models.layers': layers, 'timm.utils': utils})


def _ref_vlmo(mc):
    from functools import partial
    from models.vlmo.vlmo import VLMO, LayerNorm
    norm_layer = partial(LayerNorm, eps=1e-12, export=True)  # vlmo_module.py:21-23
    m = VLMO(img_size=mc.img_size, patch_size=mc.patch_size, in_chans=mc.in_chans,
             num_classes=mc.num_classes, embed_dim=mc.embed_dim, depth=mc.depth,
             num_heads=mc.num_heads, mlp_ratio=int(mc.mlp_ratio),
             qkv_bias=mc.qkv_bias, qk_scale=None, drop_rate=0.0,
             attn_drop_rate=0.0, drop_path_rate=0.0, norm_layer=norm_layer,
             init_values=mc.init_values, vocab_size=mc.vocab_size,
             max_text_len=mc.max_text_len, fusion_layer=mc.fusion_layer)
    return m.eval()


def grad_probe(key, shape):
    """Fixed pseudo-random projection vector for gradient fingerprints."""
    from oracle.synth import _normal
    return _normal(777, 'probe:' + key, shape)


def out_weights(mode, shape):
    from oracle.synth import _normal
    return _normal(4242, 'R:' + mode, shape)


def run_backbone_case(name, preset, B, with_grads=True, full_out=True, seed=0, full_grad_max=16384, **model_over):
    from oracle import synth
    cfg = synth.make_config(preset, **model_over)
    mc = cfg.model
    all_experts = [('v', 'l', 'vl')] * mc.depth
    sd = synth.synth_backbone_state_dict(mc, seed, all_experts)
    model = _ref_vlmo(mc)
    missing = model.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    batch = synth.synth_batch(mc, B, seed=1234)
    P = synth.num_img_tokens(mc)
    img_mask = torch.ones(B, P, dtype=torch.int64)
    rec = {}
    modes = {
        'vl': dict(img=batch['image'], txt=batch['text_ids'], img_attn_masks=img_mask,
                   txt_attn_masks=batch['text_mask']),
        'v': dict(img=batch['image'], img_attn_masks=img_mask),
        'l': dict(txt=batch['text_ids'], txt_attn_masks=batch['text_mask']),
        'vl_mim': dict(img=batch['image'], txt=batch['text_ids'], img_attn_masks=img_mask,
                       txt_attn_masks=batch['text_mask'],
                       bool_masked_pos=batch['image_bool_masked_pos'].flatten(1)),
    }
    for mode, kw in modes.items():
        model.zero_grad(set_to_none=True)
        x, m = model.forward_features(**kw)
        xd = x.detach()
        if full_out:
            rec[f'{mode}.out'] = xd.numpy().astype(np.float32)
        else:
            rec[f'{mode}.out_cls'] = xd[:, 0].numpy().astype(np.float32)
            rec[f'{mode}.out_rows'] = xd[:, ::17].numpy().astype(np.float32)
        rec[f'{mode}.out_sum'] = np.float64(xd.double().sum().item())
        rec[f'{mode}.out_abs'] = np.float64(xd.double().abs().sum().item())
        rec[f'{mode}.mask'] = m.numpy()
        rec[f'{mode}.pooled'] = model.pooler(xd).detach().numpy()
        if with_grads and mode in ('vl', 'v', 'l', 'vl_mim'):
            R = out_weights(mode, x.shape)
            (x * R).sum().backward()
            for k, p in model.named_parameters():
                if p.grad is None:
                    continue
                g = p.grad.detach()
                rec[f'{mode}.grad_norm.{k}'] = np.float64(g.double().norm().item())
                rec[f'{mode}.grad_probe.{k}'] = np.float64(
                    (g.double() * grad_probe(k, g.shape).double()).sum().item())
                if g.numel() <= full_grad_max:
                    rec[f'{mode}.grad.{k}'] = g.numpy().astype(np.float32)
    # forward_interval (objectives.py:556-567 'fusion' MIM head position)
    xi = model.forward_interval(x=batch['image'], attn_masks=None, route='v', need_embed=True,
                                bool_masked_pos=batch['image_bool_masked_pos'].flatten(1),
                                in_layer=0, out_layer=mc.fusion_layer, need_norm=True)
    rec['interval_v.out'] = xi.detach().numpy().astype(np.float32) if full_out else \
        xi.detach()[:, ::17].numpy().astype(np.float32)
    rec['meta.B'] = np.int64(B)
    np.savez_compressed(os.path.join(OUT, f'{name}.npz'), **rec)
    print(f'wrote {name}.npz with {len(rec)} arrays')


def run_dvae_case(name, B, res, seed=0, full_logits=True, **enc_kw):
    from oracle import synth
    from dall_e.encoder import Encoder
    from models.modeling_discrete_vae import Dalle_VAE
    enc = Encoder(**enc_kw).eval()
    sd = synth.synth_dvae_state_dict(seed, **enc_kw)
    r = enc.load_state_dict(sd, strict=True)
    assert not r.missing_keys and not r.unexpected_keys
    vae = Dalle_VAE(res)            # modeling_discrete_vae.py:224-236 without the pickles
    vae.encoder = enc
    g = torch.Generator().manual_seed(99)
    x = 0.8 * torch.rand(B, 3, res, res, generator=g) + 0.1
    with torch.no_grad():
        logits = enc(x)
        ids = vae.get_codebook_indices(x)
    top2 = logits.topk(2, dim=1).values
    rec = {'ids': ids.numpy(), 'top2_gap': (top2[:, 0] - top2[:, 1]).numpy().astype(np.float32),
           'logits_sum': np.float64(logits.double().sum().item()),
           'logits_abs': np.float64(logits.double().abs().sum().item()),
           'logits_max': logits.amax(dim=1).numpy().astype(np.float32)}
    if full_logits:
        rec['logits'] = logits.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, f'{name}.npz'), **rec)
    print(f'wrote {name}.npz')


def run_dvae_pickle_case(name, **enc_kw):
    """The wire format of the OpenAI dVAE weights (dall_e/__init__.py:12-21: ``torch.load`` of a pickled
    ``dall_e.encoder.Encoder`` MODULE, not a state dict): the reference's own Encoder, pickled by torch.save exactly
    as the published encoder.pkl was, with every parameter's storage emptied so that the fixture carries the object
    graph (class paths, attribute names and values, module tree) and no megabytes of weights.  The test fills the
    parameters from oracle/synth.py's key-addressed synthetic state dict."""
    from dall_e.encoder import Encoder
    enc = Encoder(**enc_kw).eval()
    with torch.no_grad():
        for p in enc.parameters():
            p.set_(torch.empty(0))
    path = os.path.join(OUT, f'{name}.pkl')
    torch.save(enc, path)
    print(f'wrote {name}.pkl ({os.path.getsize(path)} bytes)')


def run_module_case(name, preset, B, seed=0, compact_logits=False):
    """Full VlmoModule.forward(batch) with [mlm, mim, itc, itm] (vlmo_module.py:395-436).
    Two call-argument level accommodations, arithmetic untouched:
    the dVAE pickles are absent, so objectives.create_d_vae is pointed at a Dalle_VAE whose
    encoder is a seeded dall_e Encoder with synthetic weights; torch.multinomial is wrapped to
    RECORD the hard-negative indices compute_itm draws (objectives.py:268-275)."""
    from oracle import synth
    import models.vlmo.objectives as ref_obj
    from models.build import build_model
    from models.modeling_discrete_vae import Dalle_VAE
    from dall_e.encoder import Encoder
    losses = ['mlm', 'mim', 'itc', 'itm']
    cfg = synth.make_config(preset, loss_names=losses)
    mc = cfg.model
    mc.mlp_ratio = int(mc.mlp_ratio)
    enc_kw = dict(n_hid=256, vocab_size=mc.img_vocab_size)

    def fake_create(weight_path, d_vae_type, image_size, device):
        vae = Dalle_VAE(image_size)
        vae.encoder = Encoder(**enc_kw).eval()
        vae.encoder.load_state_dict(synth.synth_dvae_state_dict(seed, **enc_kw), strict=True)
        return vae
    ref_obj.create_d_vae = fake_create
    model = build_model(cfg).eval()
    sd = {'transformer.' + k: v for k, v in synth.synth_backbone_state_dict(mc, seed).items()}
    sd.update(synth.synth_head_state_dict(mc, seed, losses))
    r = model.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys, r.unexpected_keys
    assert all(k.startswith('d_vae.') or k == 'mlm_head.decoder.weight' for k in r.missing_keys), r.missing_keys
    batch = synth.synth_batch(mc, B, seed=1234)
    drawn = []
    real_multinomial = torch.multinomial

    def rec_multinomial(w, n, *a, **k):
        out = real_multinomial(w, n, *a, **k)
        drawn.append(int(out.item()))
        return out
    torch.multinomial = rec_multinomial
    try:
        torch.manual_seed(1)
        ret = model(dict(batch))
    finally:
        torch.multinomial = real_multinomial
    rec = {'itm_img_neg_idx': np.array(drawn[:B]), 'itm_txt_neg_idx': np.array(drawn[B:2 * B])}
    # the reference tokenizer's top-2 logit gap at every masked patch, aligned with ret.mim_labels
    # (objectives.py:532-540): a visual-token id may only differ from the reference's where this gap is a near-tie
    with torch.no_grad():
        dl = model.d_vae.encoder(batch['image4dalle'])
    t2 = dl.topk(2, dim=1).values
    pos = batch['image_bool_masked_pos'].flatten(1).to(torch.bool)
    rec['mim_label_top2_gap'] = (t2[:, 0] - t2[:, 1]).flatten(1)[pos].numpy().astype(np.float32)
    total = 0
    for k, v in ret.items():
        if torch.is_tensor(v) and compact_logits and k in ('mlm_logits', 'mim_logits'):
            # full-vocabulary logits are MBs: keep every 61st column, the row log-sum-exp and the arg-max
            lg = v.detach().float()
            rec['ret.' + k + '_sub'] = lg[:, ::61].numpy()
            rec['ret.' + k + '_lse'] = torch.logsumexp(lg, 1).numpy()
            rec['ret.' + k + '_argmax'] = lg.argmax(1).numpy()
        elif torch.is_tensor(v):
            rec['ret.' + k] = v.detach().float().numpy() if v.is_floating_point() else v.detach().numpy()
        else:
            rec['ret.' + k] = np.float64(v)
        if 'task_loss' in k:
            total = total + v
    model.zero_grad(set_to_none=True)
    total.backward()
    for k, p in model.named_parameters():
        if p.grad is not None and not k.startswith('d_vae.'):
            rec['grad_norm.' + k] = np.float64(p.grad.double().norm().item())
            rec['grad_probe.' + k] = np.float64((p.grad.double() * grad_probe(k, p.shape).double()).sum().item())
    rec['meta.B'] = np.int64(B)
    np.savez_compressed(os.path.join(OUT, f'{name}.npz'), **rec)
    print(f'wrote {name}.npz with {len(rec)} arrays; losses',
          {k: float(v) for k, v in ret.items() if 'task_loss' in k})


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, REF)
    _install_timm_standin()
    cases = {
        'backbone_mini': lambda: run_backbone_case('backbone_mini', 'mini', B=3),
        'backbone_small': lambda: run_backbone_case('backbone_small', 'small', B=2),
        # the optional branches of the reference: no q/v bias (vlmo.py:57-62), no layer-scale (vlmo.py:185-192)
        'backbone_mini_plain': lambda: run_backbone_case('backbone_mini_plain', 'mini', B=3, qkv_bias=False, init_values=None),
        'backbone_debug': lambda: run_backbone_case('backbone_debug', 'debug', B=2),
        'backbone_base_b2': lambda: run_backbone_case('backbone_base_b2', 'base', B=2, full_out=False),
        # VLMo-Large (conf/model/vlmo_large.yaml:14-28: d=1024, L=24, h=16, F=12); the synthetic layer-scale stays
        # 0.5 as in every fixture (the YAML's init_values 1e-5 would make every residual branch invisible)
        'backbone_large_b2': lambda: run_backbone_case('backbone_large_b2', 'large', B=2, full_out=False, full_grad_max=1024),
        'module_mini': lambda: run_module_case('module_mini', 'mini', B=4),
        'module_base_b2': lambda: run_module_case('module_base_b2', 'base', B=2, compact_logits=True),
        # BASELINE.json configs[4]'s model: VLMo-Large + the full objective + the in-loop dVAE tokenizer
        'module_large_b2': lambda: run_module_case('module_large_b2', 'large', B=2, compact_logits=True),
        'dvae_tiny': lambda: run_dvae_case('dvae_tiny', B=2, res=32, n_hid=64, vocab_size=512),
        'dvae_small': lambda: run_dvae_case('dvae_small', B=2, res=32, n_hid=256, vocab_size=1024),
        'dvae_full_b2': lambda: run_dvae_case('dvae_full_b2', B=2, res=112, full_logits=False),
        'dvae_encoder_pickle': lambda: run_dvae_pickle_case('dvae_encoder_pickle', n_hid=64, vocab_size=512,
                                                            n_blk_per_group=2),
    }
    want = sys.argv[1:] or list(cases)
    for name in want:
        cases[name]()


if __name__ == '__main__':
    main()
