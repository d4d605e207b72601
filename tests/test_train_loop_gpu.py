"""End-to-end loop in the shape of train/pretrain/multimodal.py:233-333 on synthetic data: DataLoaderX hand-off ->
VlmoModule.forward([mlm, mim, itc, itm]) -> sum of task losses -> NativeScalerWithGradNormCount (backward, fused
clip, fused AdamW) with the cosine schedule -> save_model / auto_load_model resume."""
import os
import types

import pytest
import torch
from torch.utils.data import Dataset

from exploremultimodal_amd import checkpoint, optim
from exploremultimodal_amd.build import build_model
from exploremultimodal_amd.prefetch import DataLoaderX
from oracle import synth

pytestmark = pytest.mark.gpu
DEV = 'cuda'
NS = types.SimpleNamespace


class _Synthetic(Dataset):
    def __init__(self, mc, n):
        self.b = synth.synth_batch(mc, n, seed=11)
        self.n = n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        return {k: v[i] for k, v in self.b.items()}


def _setup(tmp_path, tag='loop'):
    cfg = synth.make_config('mini', loss_names=['mlm', 'mim', 'itc', 'itm'], drop_rate=0.0, attn_drop_rate=0.0,
                            drop_path_rate=0.0)
    cfg.train.merge_passes = True
    cfg.train.auto_resume, cfg.train.resume, cfg.train.epochs, cfg.train.start_epoch = True, '', 4, 0
    cfg.tag, cfg.exp_dir, cfg.output_dir = tag, str(tmp_path), str(tmp_path / 'run')
    torch.manual_seed(0)
    model = build_model(cfg).to(DEV).train()
    tcfg = NS(opt=NS(name='fusedadamw', eps=1e-8, betas=[0.9, 0.98], momentum=0.9), weight_decay=0.01, base_lr=2e-3,
              lr_mult_head=1, lr_mult_fusion=1)
    opt = optim.create_optimizer(tcfg, model)
    return cfg, model, opt


class _Sched:
    """The reference assigns lr per iteration from the cosine table (multimodal.py:253-262)."""

    def __init__(self, opt, table):
        self.opt, self.table, self.it = opt, table, 0
        self.base = [g['lr'] for g in opt.param_groups]

    def step(self):
        f = self.table[min(self.it, len(self.table) - 1)]
        for g, b in zip(self.opt.param_groups, self.base):
            g['lr'] = b * f
        self.it += 1

    def state_dict(self):
        return {'it': self.it}

    def load_state_dict(self, sd):
        self.it = sd['it']


@pytest.mark.parametrize('with_reducer', [False, True])
def test_synthetic_pretraining_loop_learns_and_resumes(tmp_path, with_reducer):
    cfg, model, opt = _setup(tmp_path)
    reducer = None
    if with_reducer:        # the data-parallel path at world size 1: engine sink buckets + RCCL + fused optimizer
        import os
        import torch.distributed as dist
        from exploremultimodal_amd.dp import GradReducer
        os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29549')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
        reducer = GradReducer(model)
    try:
        _run_loop(tmp_path, cfg, model, opt, reducer)
    finally:
        if reducer is not None:
            import torch.distributed as dist
            reducer.close()
            dist.destroy_process_group()


def _run_loop(tmp_path, cfg, model, opt, reducer):
    loader = DataLoaderX(0, max_prefetch=2, dataset=_Synthetic(cfg.model, 8), batch_size=4, shuffle=False)
    sched = _Sched(opt, [0.0, 1.0, 1.0, 0.94, 0.775, 0.55, 0.325, 0.16])   # the caller's per-iteration lr table (warm-up + half cosine)
    scaler = optim.NativeScalerWithGradNormCount(reducer)
    losses = []
    for epoch in range(3):
        for batch in loader:
            sched.step()
            ret = model(batch)
            loss = sum(v for k, v in ret.items() if 'task_loss' in k)
            norm = scaler(loss, opt, clip_grad=5.0, parameters=model.parameters())
            opt.zero_grad(set_to_none=True)
            assert torch.isfinite(norm).item()
            losses.append(loss.item())
        checkpoint.save_model(cfg, epoch, model, model, opt, sched, scaler)
    loader.shutdown()
    assert sum(losses[-2:]) < sum(losses[:2]), losses
    # resume into a freshly initialised model / optimizer
    cfg2, model2, opt2 = _setup(tmp_path)
    sched2 = _Sched(opt2, sched.table)
    match = checkpoint.auto_load_model(cfg2, model2, model2, opt2, sched2, optim.NativeScalerWithGradNormCount())
    assert cfg2.train.start_epoch == 3 and sched2.it == sched.it
    assert all(k.startswith('d_vae.') for k in match.missing_keys) and not match.unexpected_keys
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        if not k.startswith('d_vae.'):
            assert torch.equal(a, b), k
    s1, s2 = opt.state_dict()['state'], opt2.state_dict()['state']
    assert set(s1) == set(s2) and all(torch.equal(s1[i]['exp_avg'].cpu(), s2[i]['exp_avg'].cpu()) for i in s1)
    # one more step from the resumed state behaves like one more step of the original
    batch = {k: v.to(DEV) for k, v in synth.synth_batch(cfg.model, 4, seed=11).items()}
    batch['itm_neg_idx'] = (torch.tensor([1, 0, 3, 2], device=DEV), torch.tensor([2, 3, 0, 1], device=DEV))
    out = []
    for m in (model, model2):
        ret = m(dict(batch))
        out.append({k: float(v) for k, v in ret.items() if 'task_loss' in k})
    for k in out[0]:
        assert abs(out[0][k] - out[1][k]) <= 1e-5 * max(1.0, abs(out[0][k])), (k, out)


def test_zero2_step_equals_replicated_step_rccl_single_rank():
    """ZeRO-2 (conf/ds_stage/l2.yaml) on RCCL at world size 1: GradReducer(reduce_scatter=True) + zero.ZeroAdam
    (parameters re-homed into flat buffers that mirror the gradient buckets, moments for the rank's slice only, HIP
    multi-tensor step on the slice, parameter all-gather) against optim.FusedAdam on the same model: after three
    training steps with dropout off, every parameter agrees (tolerance: the reducer's bf16 gradient exchange rounds
    the gradients to 8 bits, so this compares against FusedAdam fed the SAME reducer in all-reduce mode)."""
    import torch.distributed as dist
    from exploremultimodal_amd import optim
    from exploremultimodal_amd.dp import GradReducer
    from exploremultimodal_amd.zero import ZeroAdam
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29547')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    reds = []
    try:
        cfg = synth.make_config('mini', loss_names=['mlm', 'itc'])     # no sampled negatives: both runs see the same graph
        results, grads = {}, {}
        for mode in ('zero2', 'replicated'):
            torch.manual_seed(0)
            model = build_model(cfg).to(DEV).train()
            red = GradReducer(model, reduce_scatter=(mode == 'zero2'), comm_dtype=torch.float32)
            reds.append(red)
            skip = model.no_weight_decay()
            groups = optim.get_parameter_groups(model, base_lr=1e-3, lr_mult_head=5, lr_mult_fusion=2, weight_decay=0.05,
                                                skip_list=skip)
            opt = ZeroAdam(red, groups, betas=(0.9, 0.98), eps=1e-6) if mode == 'zero2' else \
                optim.FusedAdam(groups, betas=(0.9, 0.98), eps=1e-6)
            batch = {k: v.to(DEV) for k, v in synth.synth_batch(cfg.model, 4, seed=3).items()}
            named = list(model.named_parameters())
            inits = {n: p.detach().clone() for n, p in named}
            for step in range(3):
                for p in model.parameters():
                    p.grad = None
                torch.manual_seed(100 + step)
                ret = model(dict(batch))
                loss = sum(v for k, v in ret.items() if 'task_loss' in k)
                red.prepare(loss)
                loss.backward()
                red.finish()
                torch.cuda.synchronize()
                # Two runs of a step do not reproduce bit for bit (fp32 atomics in the weight-gradient / column-sum /
                # embedding kernels: ~1e-6 of the terms summed), and Adam turns the sign of a near-zero gradient into a
                # full-size update.  So the runs are compared where that noise is harmless -- the reduced GRADIENTS, element
                # by element within 5e-3 of the tensor's largest (bf16 rounding flips downstream of an fp32 atomic) -- and the optimizers then step on IDENTICAL gradients
                # (the zero2 run's), which makes every parameter comparable element by element at 1e-5 of its update.
                if mode == 'zero2':
                    grads[step] = {n: p.grad.detach().clone() for n, p in named if p.grad is not None}
                else:
                    assert set(grads[step]) == {n for n, p in named if p.grad is not None}
                    for n, p in named:
                        if p.grad is None:
                            continue
                        ga, gb = grads[step][n], p.grad.detach()
                        tol = 5e-3 * gb.abs().max().item() + 1e-12
                        assert (ga - gb).abs().max().item() <= tol, (step, n, (ga - gb).abs().max().item(), tol)
                        p.grad.copy_(ga)
                opt.step(clip_grad=1.0)
            torch.cuda.synchronize()
            results[mode] = {n: p.detach().clone() for n, p in named}
            red.close()
        for n in results['zero2']:
            a, b, w0 = results['zero2'][n], results['replicated'][n], inits[n]
            upd = (b - w0).abs().max().item()
            err = (a - b).abs().max().item()
            ulp = 2.0 ** -23 * b.abs().max().item()              # the two clip coefficients differ in their last bit: one ulp of w
            assert err <= 1e-5 * upd + 2 * ulp + 1e-9, (n, err, upd, ulp)
    finally:
        for r in reds:
            r.close()
        dist.destroy_process_group()
