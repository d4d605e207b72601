"""Fused vocabulary-head loss (heads.LinearCrossEntropyFn: VLMO_EPI_CE / vlmo_ce_reduce / VLMO_EPI_CE_BWD) against
plain PyTorch fp32 F.linear + F.cross_entropy on the same bf16-rounded operands (heads.py:86-112,
objectives.py:57-68,571-582).  Tolerances: loss 1e-3 + 5e-4 relative (bf16 operands, fp32 accumulation and soft-max);
gradients 2 % of their norm (the recomputed soft-max gradient is rounded to bf16 before the two gradient GEMMs)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('n,d,V,ignore', [(27, 128, 1000, 0), (300, 768, 8192, 0), (146, 256, 30522, 5), (1, 64, 70, 0)])
def test_linear_cross_entropy_matches_torch(n, d, V, ignore):
    from exploremultimodal_amd.heads import LinearCrossEntropyFn, _PaddedShadows
    g = torch.Generator().manual_seed(n + V)
    x = (torch.randn(n, d, generator=g)).to(DEV).bfloat16().float().requires_grad_(True)
    W = (torch.randn(V, d, generator=g) * 0.05).to(DEV).bfloat16().float().requires_grad_(True)
    b = (torch.randn(V, generator=g) * 0.1).to(DEV).requires_grad_(True)
    labels = torch.randint(0, V, (n,), generator=g).to(DEV)
    labels[:ignore] = -100
    loss, pred = LinearCrossEntropyFn.apply(x, W, b, labels, -100, _PaddedShadows())
    (loss * 3.0).backward()
    got = [t.grad.clone() for t in (x, W, b)]
    for t in (x, W, b):
        t.grad = None
    logits = F.linear(x, W, b)
    ref = F.cross_entropy(logits, labels, ignore_index=-100)
    (ref * 3.0).backward()
    assert abs(loss.item() - ref.item()) <= 1e-3 + 5e-4 * abs(ref.item()), (loss.item(), ref.item())
    agree = (pred.long() == logits.argmax(1)).float().mean().item()
    assert agree >= 0.98, agree        # near-ties may flip under bf16 rounding of nothing: operands are already bf16
    for name, a, t in zip('xWb', got, (x, W, b)):
        rel = (a - t.grad).norm().item() / (t.grad.norm().item() + 1e-12)
        assert rel <= 2e-2, (name, rel)
    # ignored rows receive no input gradient
    if ignore:
        assert float(got[0][:ignore].abs().max()) == 0.0


def test_fused_ce_in_the_objectives_equals_the_logits_path():
    """config.train.fused_ce: compute_mlm / compute_mim through heads.loss_and_pred give the same losses, accuracies
    and (to bf16 rounding) parameter gradients as the reference-shaped path that materialises the logits."""
    from exploremultimodal_amd.build import build_model
    from oracle import synth
    outs = []
    for fused in (False, True):
        cfg = synth.make_config('mini', loss_names=['mlm', 'mim'])
        cfg.train.fused_ce = fused
        torch.manual_seed(0)
        model = build_model(cfg)
        sd = {'transformer.' + k: v for k, v in synth.synth_backbone_state_dict(cfg.model, 0).items()}
        sd.update(synth.synth_head_state_dict(cfg.model, 0, ['mlm', 'mim']))
        model.load_state_dict(sd, strict=False)
        model.d_vae.encoder.load_state_dict(synth.synth_dvae_state_dict(0, n_hid=256, vocab_size=cfg.model.img_vocab_size))
        model = model.to(DEV).eval()
        batch = {k: v.to(DEV) for k, v in synth.synth_batch(cfg.model, 4, seed=1234).items()}
        ret = model(batch)
        total = ret['mlm_task_loss'] + ret['mim_task_loss']
        total.backward()
        outs.append((ret, {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    (r0, g0), (r1, g1) = outs
    assert r1['mlm_logits'] is None and r1['mim_logits'] is None and r0['mlm_logits'] is not None
    for k in ('mlm_task_loss', 'mim_task_loss'):
        a, b = float(r0[k].detach()), float(r1[k].detach())
        assert abs(a - b) <= 5e-3, (k, a, b)
    assert r0['mlm_count'] == r1['mlm_count'] and r0['mim_count'] == r1['mim_count']
    assert abs(float(r0['mim_mean_acc']) - float(r1['mim_mean_acc'])) <= 0.02
    assert set(g0) == set(g1)
    for k in g0:
        rel = (g0[k] - g1[k]).norm().item() / (g0[k].norm().item() + 1e-12)
        assert rel <= 3e-2, (k, rel)
