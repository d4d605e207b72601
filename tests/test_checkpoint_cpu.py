"""Checkpoint wire format, host logic only: save -> auto-resume round trip in the reference's file
layout (utils/utils.py:479-612), VLMo legacy / BEiT key remaps and position-embedding interpolation
(vlmo_module.py:187-319)."""
import copy
import types

import torch

from exploremultimodal_amd import checkpoint, optim
from exploremultimodal_amd.build import build_model
from oracle import synth

NS = types.SimpleNamespace


def _cfg(tmp, **over):
    cfg = synth.make_config('mini', loss_names=['mlm', 'mim', 'itc', 'itm'], **over)
    cfg.train.auto_resume, cfg.train.resume, cfg.train.epochs, cfg.train.start_epoch = True, '', 10, 0
    cfg.tag, cfg.exp_dir, cfg.output_dir = 'unit', str(tmp), str(tmp / 'run0')
    return cfg


class _Sched:
    def __init__(self):
        self.last = 0

    def state_dict(self):
        return {'last': self.last}

    def load_state_dict(self, sd):
        self.last = sd['last']


def test_save_then_auto_resume_round_trip(tmp_path):
    cfg = _cfg(tmp_path)
    torch.manual_seed(0)
    model = build_model(cfg)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-3)
    for p in opt.param_groups[0]['params'][:5]:
        p.grad = torch.ones_like(p)
    opt.step()
    sched, scaler = _Sched(), optim.NativeScalerWithGradNormCount()
    sched.last = 123
    for ep in (2, 3):
        name = checkpoint.save_model(cfg, ep, model, model, opt, sched, scaler)
    assert name == 'checkpoint-3.pth' and (tmp_path / 'run0' / name).exists()
    raw = torch.load(tmp_path / 'run0' / name, weights_only=False)
    assert set(raw) == {'model', 'optimizer', 'lr_scheduler', 'epoch', 'scaler', 'cfg'}
    assert 'transformer.blocks.0.mlp.v.fc1.weight' in raw['model'] and 'transformer.blocks.0.mlp.vl.fc1.weight' not in raw['model']

    cfg2 = _cfg(tmp_path)
    torch.manual_seed(1)
    model2 = build_model(cfg2)
    opt2 = torch.optim.AdamW([p for p in model2.parameters() if p.requires_grad], lr=1e-3)
    sched2 = _Sched()
    match = checkpoint.auto_load_model(cfg2, model2, model2, opt2, sched2, optim.NativeScalerWithGradNormCount())
    assert cfg2.train.resume.endswith('checkpoint-3.pth') and cfg2.train.start_epoch == 4 and sched2.last == 123
    assert not match.missing_keys and not match.unexpected_keys
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    assert len(opt2.state_dict()['state']) == 5
    # a different tag loads weights only
    cfg3 = _cfg(tmp_path)
    cfg3.tag = 'other'
    sched3 = _Sched()
    checkpoint.auto_load_model(cfg3, build_model(cfg3), model2, opt2, sched3, optim.NativeScalerWithGradNormCount())
    assert cfg3.train.start_epoch == 0 and sched3.last == 0
    # remove_models keeps the named epochs
    cfg.dist = NS(rank=0)
    checkpoint.remove_models(cfg, 3, 3)
    assert not (tmp_path / 'run0' / 'checkpoint-2.pth').exists() and (tmp_path / 'run0' / 'checkpoint-3.pth').exists()


def test_load_from_ckpt_vlmo_legacy_keys_and_beit_layout(tmp_path):
    cfg = _cfg(tmp_path)
    torch.manual_seed(0)
    src = build_model(cfg)
    sd = copy.deepcopy(src.state_dict())
    legacy = {k.replace('.mlp.v.', '.mlp.v_mlp.').replace('.mlp.l.', '.mlp.l_mlp.').replace('.mlp.vl.', '.mlp.vl_mlp.'): v
              for k, v in sd.items()}
    assert any('.mlp.v_mlp.' in k for k in legacy)
    torch.manual_seed(1)
    dst = build_model(cfg)
    match, is_beit = dst.load_from_ckpt(legacy)
    assert not is_beit and not match.missing_keys and not match.unexpected_keys
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert torch.equal(a, b), k

    # BEiT layout: un-prefixed backbone, one FFN per block, cls_token / mask_token / lm_head
    beit = {}
    for k, v in src.transformer.state_dict().items():
        if '.mlp.l.' in k or '.mlp.vl.' in k or k.startswith('txt_embeddings') or k.startswith('token_type') or k.startswith('pooler'):
            continue
        beit[k.replace('.mlp.v.', '.mlp.').replace('img_cls_token', 'cls_token').replace('img_mask_token', 'mask_token')] = v.clone()
    beit['lm_head.weight'] = src.mim_head.fc.weight.detach().clone() + 1.0
    beit['lm_head.bias'] = src.mim_head.fc.bias.detach().clone() - 1.0
    torch.manual_seed(2)
    dst = build_model(cfg)
    match, is_beit = dst.load_from_ckpt(beit)
    assert is_beit
    assert all(('.mlp.l.' in k or '.mlp.vl.' in k or 'txt_embeddings' in k or 'token_type' in k or 'pooler' in k)
               for k in match.missing_keys), match.missing_keys
    assert torch.equal(dst.transformer.blocks[0].mlp['v'].fc1.weight, src.transformer.blocks[0].mlp['v'].fc1.weight)
    assert torch.equal(dst.transformer.img_cls_token, src.transformer.img_cls_token)
    assert torch.equal(dst.mim_head.fc.weight, src.mim_head.fc.weight + 1.0)


def test_position_embedding_interpolation(tmp_path):
    small = build_model(_cfg(tmp_path, img_size=32))          # 2x2 patches at patch 16
    big = build_model(_cfg(tmp_path))                          # mini preset: 64 -> 4x4
    sd = copy.deepcopy(small.state_dict())
    pe = sd['transformer.pos_embed']
    out = big.interpolate_pos_embedding(sd)['transformer.pos_embed']
    assert out.shape == big.transformer.pos_embed.shape
    assert torch.equal(out[:, :1], pe[:, :1])
    d = pe.shape[-1]
    ref = torch.nn.functional.interpolate(pe[:, 1:].reshape(1, 2, 2, d).permute(0, 3, 1, 2), size=(4, 4), mode='bicubic',
                                          align_corners=False).permute(0, 2, 3, 1).flatten(1, 2)
    assert torch.equal(out[:, 1:], ref)
    # text positions are cut to max_text_len
    sd['transformer.txt_embeddings.position_embeddings.weight'] = torch.randn(40, d)
    out = big.interpolate_pos_embedding(sd)
    assert out['transformer.txt_embeddings.position_embeddings.weight'].shape[0] == big.transformer.max_text_len
