"""VlmoModule.forward(batch) with [mlm, mim, itc, itm] through the HIP engine vs the
reference's own run of the same module (tests/golden/module_mini.npz).  Tolerances: bf16
backbone -> losses within 2e-2 absolute, logits within 5e-2, gradient norms within 5 %."""
import os

import numpy as np
import pytest
import torch

from oracle import synth
from oracle.gen_golden import grad_probe

pytestmark = pytest.mark.gpu
DEV = 'cuda'
LOSSES = ['mlm', 'mim', 'itc', 'itm']


def _build(preset='mini'):
    from exploremultimodal_amd.build import build_model
    cfg = synth.make_config(preset, loss_names=LOSSES)
    mc = cfg.model
    model = build_model(cfg)
    sd = {'transformer.' + k: v for k, v in synth.synth_backbone_state_dict(mc, 0).items()}
    sd.update(synth.synth_head_state_dict(mc, 0, LOSSES))
    r = model.load_state_dict(sd, strict=False)
    assert not r.unexpected_keys
    assert all(k.startswith('d_vae.') or k == 'mlm_head.decoder.weight' for k in r.missing_keys), r.missing_keys
    model.d_vae.encoder.load_state_dict(synth.synth_dvae_state_dict(0, n_hid=256, vocab_size=mc.img_vocab_size))
    assert model.mlm_head.decoder.weight is model.transformer.txt_embeddings.word_embeddings.weight
    assert not any(p.requires_grad for p in model.d_vae.parameters())
    return model.to(DEV).eval(), cfg


def test_build_model_contract():
    from exploremultimodal_amd.build import build_model
    cfg = synth.make_config('mini', loss_names=LOSSES)
    m = build_model(cfg)
    assert m.no_weight_decay() == {'itc_temp', 'transformer.pos_embed', 'transformer.img_cls_token'}
    assert m.transformer.patch_size == 16
    # _freeze_params: no 'vl' expert below the fusion layer (vlmo_module.py:165-167)
    assert 'vl' not in m.transformer.blocks[0].mlp and 'vl' in m.transformer.blocks[1].mlp
    cfg.model.type = 'other'
    with pytest.raises(NotImplementedError):
        build_model(cfg)
    with pytest.raises(AssertionError):
        m.infer({}, infer_mode='bogus')


@pytest.mark.parametrize('fused_ce', [False, True, None])
@pytest.mark.parametrize('name,preset', [('module_mini', 'mini'), ('module_base_b2', 'base'), ('module_large_b2', 'large')])
def test_module_forward_matches_reference(golden_dir, name, preset, fused_ce):
    """module_base_b2: the full objective at the VLMo-Base shape (BASELINE.json configs[4]'s compute_mim + in-loop
    dVAE tokenizer on 112x112 inputs, 8192-way visual vocabulary, tied 30522-way MLM decoder), batch 2; its
    full-vocabulary logits are pinned by every 61st column, the row log-sum-exp and the arg-max.  module_large_b2: the
    same at VLMo-Large (configs[4]'s model).  fused_ce: the two vocabulary heads through the HIP cross-entropy path
    (heads.LinearCrossEntropyFn: no logits in HBM) -- losses, accuracies and every parameter gradient against the
    same reference fixture; the logits keys are then present with value None (documented deviation).  fused_ce None = the
    DEFAULT configuration: the HIP loss path with the logits of the reference's output dict computed beside it (no graph)."""
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    B = int(g['meta.B'])
    model, cfg = _build(preset)
    if fused_ce is None:
        assert getattr(cfg.train, 'fused_ce', None) is None        # what build_model(cfg) gives without any option
    else:
        cfg.train.fused_ce = fused_ce
    batch = {k: v.to(DEV) for k, v in synth.synth_batch(cfg.model, B, seed=1234).items()}
    batch['itm_neg_idx'] = (torch.from_numpy(g['itm_img_neg_idx']).to(DEV), torch.from_numpy(g['itm_txt_neg_idx']).to(DEV))
    ret = model(batch)
    if fused_ce is None:
        assert ret['mlm_logits'] is not None and not ret['mlm_logits'].requires_grad      # logits on the side, loss through HIP
        assert ret['mlm_task_loss'].grad_fn is not None and 'LinearCrossEntropyFn' in type(ret['mlm_task_loss'].grad_fn).__name__
    for k in g.files:
        if k.startswith('ret.'):
            kk = k[4:]
            for suf in ('_sub', '_lse', '_argmax'):
                if kk.endswith('logits' + suf):
                    kk = kk[:-len(suf)]
            assert kk in ret, f'missing output key {kk}'
    rep = {}
    for ln in LOSSES:
        got, ref = float(ret[f'{ln}_task_loss']), float(g[f'ret.{ln}_task_loss'])
        rep[ln] = (got, ref)
        assert abs(got - ref) <= 2e-2 + 2e-3 * abs(ref), (ln, got, ref)
    print(rep)
    # visual-token labels from the in-loop dVAE (objectives.py:532-540): >= 99 % equal to the reference's, and every
    # differing one sits where the reference's own top-2 logits are a near-tie (same rule as tests/test_dvae_gpu.py)
    lab_rows = ret['mim_labels'].cpu().numpy() == g['ret.mim_labels']
    lab_ok = lab_rows.mean()
    gaps = g['mim_label_top2_gap']
    print('mim label agreement', lab_ok, 'max reference top-2 gap among mismatches', gaps[~lab_rows].max() if (~lab_rows).any() else 0)
    assert lab_ok >= 0.99, lab_ok
    assert (gaps[~lab_rows] < 3e-2).all()
    np.testing.assert_array_equal(ret['mlm_labels'].cpu().numpy(), g['ret.mlm_labels'])
    for ln in ('mlm', 'mim'):
        assert int(ret[f'{ln}_count']) == int(g[f'ret.{ln}_count'])
        # accuracy = arg-max == label: the fused path takes the arg-max inside the GEMM epilogue
        ref_am = g[f'ret.{ln}_logits_argmax'] if f'ret.{ln}_logits_argmax' in g.files else g[f'ret.{ln}_logits'].argmax(1)
        ref_acc = float((ref_am == g[f'ret.{ln}_labels']).mean())
        assert abs(float(ret[f'{ln}_mean_acc']) - ref_acc) <= 2.0 / max(1, int(g[f'ret.{ln}_count'])), (ln, float(ret[f'{ln}_mean_acc']), ref_acc)
    for k in ('mlm_logits', 'sim_i2t', 'itm_logits', 'mim_logits'):
        if ret[k] is None:
            assert fused_ce and k in ('mlm_logits', 'mim_logits')
            continue
        got = ret[k].detach().float().cpu()
        if 'ret.' + k in g.files:
            err = np.abs(got.numpy() - g['ret.' + k]).max()
            assert err <= 5e-2, (k, err)
        else:
            rows = lab_rows if k == 'mim_logits' else slice(None)
            err = np.abs(got[:, ::61].numpy()[rows] - g['ret.' + k + '_sub'][rows]).max()
            lse = np.abs(torch.logsumexp(got, 1).numpy()[rows] - g['ret.' + k + '_lse'][rows]).max()
            assert err <= 6e-2 and lse <= 3e-2, (k, err, lse)
    total = sum(v for k, v in ret.items() if 'task_loss' in k)
    total.backward()
    # Error scale of a gradient: its own reference norm, but not less than 5 % of the largest norm in its FAMILY (the same
    # parameter in every block and of every modality expert).  bf16 rounding noise of a gradient is proportional to the
    # terms that are summed, which have the same size across a family; at batch 2 the text-only experts above the fusion
    # layer and the ITC text projection are fed by the ITC text pass alone and their column sums nearly cancel (reference
    # norms of 4e-4 ... 2e-3 against 2e-2 ... 1e-1 in their families): held to their own norm they show 15 - 45 %.
    import re
    fam_of = lambda k: re.sub(r'\.(v|l|vl)\.', '.X.', re.sub(r'blocks\.\d+\.', 'blocks.N.', k))
    fam = {}
    for k in (f[len('grad_norm.'):] for f in g.files if f.startswith('grad_norm.')):
        fam[fam_of(k)] = max(fam.get(fam_of(k), 0.0), float(g['grad_norm.' + k]))
    gmax = max(fam.values())
    rels = []
    for k, p in model.named_parameters():
        if 'grad_norm.' + k not in g.files:
            continue
        gn = float(g['grad_norm.' + k])
        assert p.grad is not None, k
        gr = p.grad.detach().float().cpu()
        pr = (gr.double() * grad_probe(k, gr.shape).double()).sum().item()
        scale = max(gn, 0.05 * fam[fam_of(k)], 1e-3 * gmax) + 1e-12     # + the noise floor of the whole backward pass
        rels.append((max(abs(gr.norm().item() - gn), abs(pr - float(g['grad_probe.' + k]))) / scale, k, gn))
    rels.sort(reverse=True)
    print('worst grads', [(round(r, 4), k, f'{gn:.3g}') for r, k, gn in rels[:8]])
    # 2-layer shape: 6 %; the 12-layer Base shape at batch 2 (bf16 operands, four summed losses): 8 %; 24-layer Large: 10 %
    tol = 6e-2 if cfg.model.depth <= 3 else (8e-2 if cfg.model.depth <= 12 else 1e-1)
    bad = [(round(r, 4), k) for r, k, _ in rels if r > tol]
    assert not bad, bad


def test_module_training_step_runs():
    """train mode: dropout on, hard negatives sampled on device, finite losses and grads."""
    model, cfg = _build()
    cfg.model.drop_rate = 0.1
    model.train()
    batch = {k: v.to(DEV) for k, v in synth.synth_batch(cfg.model, 6, seed=9).items()}
    ret = model(batch)
    total = sum(v for k, v in ret.items() if 'task_loss' in k)
    assert torch.isfinite(total)
    total.backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    assert ret['itm_logits'].shape == (18, 2)


def test_merged_passes_equal_pass_by_pass(golden_dir):
    """config.train.merge_passes: V / L / VL passes batched by mode give the same outputs as the reference's
    pass-by-pass order (rows are independent; forward values bit-identical, gradients to summation order)."""
    g = np.load(os.path.join(golden_dir, 'module_mini.npz'))
    B = int(g['meta.B'])
    outs = []
    for merged in (False, True):
        model, cfg = _build()
        cfg.train.merge_passes = merged
        batch = {k: v.to(DEV) for k, v in synth.synth_batch(cfg.model, B, seed=1234).items()}
        batch['itm_neg_idx'] = (torch.from_numpy(g['itm_img_neg_idx']).to(DEV), torch.from_numpy(g['itm_txt_neg_idx']).to(DEV))
        ret = model(batch)
        total = sum(v for k, v in ret.items() if 'task_loss' in k)
        total.backward()
        outs.append((ret, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    (r0, g0), (r1, g1) = outs
    assert set(r0) == set(r1)
    for k in r0:
        if torch.is_tensor(r0[k]):
            a, b = r0[k].detach().float(), r1[k].detach().float()
            assert a.shape == b.shape, k
            assert torch.allclose(a, b, rtol=0, atol=1e-6), (k, (a - b).abs().max().item())
        else:
            assert r0[k] == r1[k], k
    assert set(g0) == set(g1)
    for k in g0:
        denom = g0[k].norm().item() + 1e-12
        assert (g0[k] - g1[k]).norm().item() / denom <= 2e-2, (k, (g0[k] - g1[k]).norm().item() / denom)


@pytest.mark.parametrize('merged', [False, True])
def test_row_indices_from_the_host_equal_the_boolean_gathers(golden_dir, merged):
    """objectives.attach_row_indices (what prefetch.DataLoaderX adds to a batch before its upload): the MLM / MIM heads
    gather by index instead of by boolean mask -- identical outputs, counts and gradients, and no host synchronisation
    left in the forward of the four objectives (torch.cuda.set_sync_debug_mode('error'))."""
    from exploremultimodal_amd.objectives import attach_row_indices
    g = np.load(os.path.join(golden_dir, 'module_mini.npz'))
    B = int(g['meta.B'])
    outs = []
    for indexed in (False, True):
        model, cfg = _build()
        cfg.train.merge_passes = merged
        host = synth.synth_batch(cfg.model, B, seed=1234)
        if indexed:
            attach_row_indices(host)
            lab = host['text_labels_mlm']
            assert torch.equal(host['_mlm_rows'], (lab.reshape(-1) != -100).nonzero().reshape(-1))
            bm = host['image_bool_masked_pos'].reshape(B, -1) != 0
            assert torch.equal(host['_mim_rows'], bm.reshape(-1).nonzero().reshape(-1))
            full = torch.cat([torch.zeros(B, 1, dtype=torch.bool), bm], 1)          # CLS token first
            assert torch.equal(host['_mim_tok_rows'], full.reshape(-1).nonzero().reshape(-1))
        batch = {k: v.to(DEV) for k, v in host.items()}
        batch['itm_neg_idx'] = (torch.from_numpy(g['itm_img_neg_idx']).to(DEV), torch.from_numpy(g['itm_txt_neg_idx']).to(DEV))
        if indexed:
            model(dict(batch))                 # first call: allocations, weight shadows
            torch.cuda.synchronize()
            torch.cuda.set_sync_debug_mode('error')
        try:
            ret = model(dict(batch))
        finally:
            torch.cuda.set_sync_debug_mode('default')
        total = sum(v for k, v in ret.items() if 'task_loss' in k)
        total.backward()
        outs.append((ret, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    (r0, g0), (r1, g1) = outs
    assert set(r0) == set(r1)
    for k in r0:
        if torch.is_tensor(r0[k]):
            assert torch.equal(r0[k].detach(), r1[k].detach()), k
        else:
            assert r0[k] == r1[k], k
    for k in g0:
        assert (g0[k] - g1[k]).norm().item() <= 1e-3 * (g0[k].norm().item() + 1e-12), k


def test_itc_global_reduce_branch_single_rank():
    """compute_itc with config.train.global_reduce (GatherLayer over RCCL, objectives.py:99-108) at world size 1 equals
    the in-batch branch: same similarities, loss and feature gradients."""
    import torch.distributed as dist
    from exploremultimodal_amd import objectives
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29547')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        outs = []
        for glob in (False, True):
            model, cfg = _build()
            cfg.train.global_reduce = glob
            batch = {k: v.to(DEV) for k, v in synth.synth_batch(cfg.model, 4, seed=21).items()}
            ret = objectives.compute_itc(model, batch)
            ret['itc_task_loss'].backward()
            outs.append((ret, model.itc_head.dense['v'].weight.grad.clone()))
        (r0, g0), (r1, g1) = outs
        assert torch.equal(r0['sim_i2t'], r1['sim_i2t']) and torch.equal(r0['sim_t2i'], r1['sim_t2i'])
        assert torch.equal(r0['itc_task_loss'], r1['itc_task_loss'])
        torch.testing.assert_close(g0, g1, rtol=1e-4, atol=1e-6)      # two matmuls instead of one + transpose
    finally:
        dist.destroy_process_group()


def test_objectives_with_nothing_masked_return_python_zero():
    """objectives.py:67-68, 581-582: with no masked text token / image patch the losses are the python float 0. (not a
    tensor) and the accuracy counters are 0; the other outputs keep their keys."""
    model, cfg = _build()
    batch = {k: v.to(DEV) for k, v in synth.synth_batch(cfg.model, 3, seed=5).items()}
    batch['text_labels_mlm'] = torch.full_like(batch['text_labels_mlm'], -100)
    batch['text_ids_mlm'] = batch['text_ids'].clone()
    batch['image_bool_masked_pos'] = torch.zeros_like(batch['image_bool_masked_pos'])
    batch['itm_neg_idx'] = (torch.tensor([1, 2, 0], device=DEV), torch.tensor([2, 0, 1], device=DEV))
    ret = model(batch)
    assert ret['mlm_task_loss'] == 0. and not torch.is_tensor(ret['mlm_task_loss']) and ret['mlm_count'] == 0
    assert ret['mim_task_loss'] == 0. and not torch.is_tensor(ret['mim_task_loss']) and ret['mim_count'] == 0
    assert ret['mlm_logits'].shape[0] == 0 and ret['mim_logits'].shape[0] == 0
    assert torch.isfinite(ret['itc_task_loss']) and torch.isfinite(ret['itm_task_loss'])
    total = sum(v for k, v in ret.items() if 'task_loss' in k)
    total.backward()


@pytest.mark.parametrize('fused_ce', [False, True])
@pytest.mark.parametrize('amp_dtype', [torch.float16, torch.bfloat16])
def test_module_under_autocast_like_the_reference_loop(golden_dir, amp_dtype, fused_ce):
    """The reference's train step calls the model INSIDE torch.cuda.amp.autocast() and steps through its loss scaler with
    clip 5.0 (train/pretrain/multimodal.py:276-333, utils/utils.py:343-364).  Under autocast the torch-op heads (pooler,
    MLM transform, ITC / ITM linears) run in half precision around the engine's autograd Functions, which keep their own
    precision plan (fp32 residual stream, bf16 GEMM operands): losses stay within the fixture tolerance of the fp32
    reference run, every gradient is finite and has its parameter's dtype, and the scaler's step changes the weights."""
    from exploremultimodal_amd import optim
    g = np.load(os.path.join(golden_dir, 'module_base_b2.npz'))
    B = int(g['meta.B'])
    model, cfg = _build('base')
    cfg.train.fused_ce = fused_ce
    batch = {k: v.to(DEV) for k, v in synth.synth_batch(cfg.model, B, seed=1234).items()}
    batch['itm_neg_idx'] = (torch.from_numpy(g['itm_img_neg_idx']).to(DEV), torch.from_numpy(g['itm_txt_neg_idx']).to(DEV))
    with torch.autocast('cuda', dtype=amp_dtype):
        ret = model(batch)
    for ln in LOSSES:
        got, ref = float(ret[f'{ln}_task_loss']), float(g[f'ret.{ln}_task_loss'])
        # half-precision heads on top of the bf16 backbone: fixture tolerance x 1.5
        assert abs(got - ref) <= 3e-2 + 3e-3 * abs(ref), (ln, got, ref)
    total = sum(v for k, v in ret.items() if 'task_loss' in k)
    assert total.dtype == torch.float32
    params = [p for p in model.parameters() if p.requires_grad]
    before = [p.detach().clone() for p in params[:8]]
    opt = optim.FusedAdam([{'params': params, 'lr': 1e-4, 'weight_decay': 0.01}], betas=(0.9, 0.98), eps=1e-8)
    scaler = optim.NativeScalerWithGradNormCount()
    norm = scaler(total, opt, clip_grad=5.0, parameters=params, update_grad=True)
    torch.cuda.synchronize()
    assert torch.isfinite(torch.as_tensor(norm)).all() and float(norm) > 0
    seen = 0
    for n_, p in model.named_parameters():
        if p.grad is None:
            continue
        seen += 1
        assert p.grad.dtype == p.dtype, (n_, p.grad.dtype)
        assert torch.isfinite(p.grad).all(), n_
    assert seen > 100
    assert any(not torch.equal(a, p.detach()) for a, p in zip(before, params[:8]))
