"""Optimizer side of the VLMo training step on MI355X.

Mirrors, name for name, what the reference's training loop uses:

* ``get_parameter_groups`` / ``create_optimizer`` -- utils/optim_factory.py:22-90, 93-199: parameters are split by NAME
  into bottom / fusion / head layers (learning-rate multipliers ``lr_mult_fusion`` / ``lr_mult_head``) times decay /
  no_decay (1-D tensors, ``.bias`` and the model's ``no_weight_decay()`` set get weight decay 0).
* ``FusedAdam`` -- the reference default ``fusedadamw`` = apex ``FusedAdam(adam_w_mode=True)``
  (optim_factory.py:185-186).  Here ONE HIP launch updates every parameter tensor of every group
  (``vlmo_mt_adam``); an optional fused global-norm clip (``vlmo_mt_grad_norm``) replaces
  ``torch.nn.utils.clip_grad_norm_`` without the per-tensor norm kernels or a host sync.
* ``NativeScalerWithGradNormCount`` -- utils/utils.py:337-371: ``scaler(loss, optimizer, clip_grad, parameters)``
  = backward, unscale, clip, step; returns the gradient norm.  bf16 needs no loss scaling, so the scale is 1 and
  the "skip the step on non-finite gradients" behaviour of ``GradScaler.step`` is kept inside the kernel.

State layout follows ``torch.optim.AdamW`` (``state[p] = {'step', 'exp_avg', 'exp_avg_sq'}``), so optimizer
state dicts written by the reference's ``save_model`` (utils/utils.py:479-520) load unchanged.
There is no CPU fallback: stepping parameters that are not on a GPU raises.
"""
import json
import math

import torch

from . import hip

HEAD_NAMES = ('mlm_head', 'itc_head', 'itm_head', 'mim_head', 'vqa_classifier', 'vqa_last', 'nlvr2_classifier',
              'snli_classifier')
CHUNK = 1 << 16        # elements per workgroup of the multi-tensor kernels


def get_parameter_groups(model, base_lr, lr_mult_head, lr_mult_fusion, weight_decay=1e-5, skip_list=(), logger=None):
    """3 x 2 groups by parameter name (optim_factory.py:22-90).  Groups are created in first-seen order, each
    ``{'params', 'weight_decay', 'lr'}``; a second list with the parameter NAMES is logged like the reference does."""
    fusion_layer = model.config.model.fusion_layer
    depth = model.config.model.depth
    fusion_names = [f'blocks.{i}' for i in range(fusion_layer, depth)] + ['pooler']
    groups, names = {}, {}
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        no_decay = param.dim() <= 1 or name.endswith('.bias') or name in skip_list
        wd = 0. if no_decay else weight_decay
        # NB substring match, as in the reference: 'blocks.1' also matches 'blocks.10', 'blocks.11'
        if any(h in name for h in HEAD_NAMES):
            part, lr = 'head_layer', base_lr * lr_mult_head
        elif any(f in name for f in fusion_names):
            part, lr = 'fusion_layer', base_lr * lr_mult_fusion
        else:
            part, lr = 'bottom_layer', base_lr
        key = f"{part}_{'no_decay' if no_decay else 'decay'}"
        if key not in groups:
            groups[key] = {'params': [], 'weight_decay': wd, 'lr': lr}
            names[key] = {'params': [], 'weight_decay': wd, 'lr': lr}
        groups[key]['params'].append(param)
        names[key]['params'].append(name)
    if logger is not None:
        logger.info(f'\nParam groups = {json.dumps(names, indent=2)}')
    return list(groups.values())


def create_optimizer(cfg, model, skip_list=None, logger=None):
    """cfg = the reference's ``config.train`` node: ``cfg.opt.{name,eps,betas,momentum}``, ``cfg.weight_decay``,
    ``cfg.base_lr``, ``cfg.lr_mult_head``, ``cfg.lr_mult_fusion`` (conf/train/pretrain_mum.yaml:28-36,75-82).
    adam / adamw / fusedadam / fusedadamw map to the HIP ``FusedAdam``; sgd / momentum / nesterov to torch.optim.SGD;
    the timm / apex extras of the reference factory (optim_factory.py:141-190) are not provided."""
    opt_lower = cfg.opt.name.lower()
    skip = skip_list or {}
    if hasattr(model, 'no_weight_decay'):
        skip = model.no_weight_decay()
    parameters = get_parameter_groups(model, base_lr=cfg.base_lr, lr_mult_head=cfg.lr_mult_head,
                                      lr_mult_fusion=cfg.lr_mult_fusion, weight_decay=cfg.weight_decay,
                                      skip_list=skip, logger=logger)
    opt_args = dict(lr=cfg.base_lr, weight_decay=0.)
    split = opt_lower.split('_')
    kind = split[-1]
    if len(split) > 1:
        raise NotImplementedError(f'optimizer wrapper {split[0]!r} (timm Lookahead) is not provided')
    if kind in ('sgd', 'nesterov', 'momentum', 'fusedsgd', 'fusedmomentum'):
        return torch.optim.SGD(parameters, momentum=cfg.opt.momentum, nesterov=kind in ('sgd', 'nesterov', 'fusedsgd'),
                               **opt_args)
    if kind in ('adam', 'adamw', 'fusedadam', 'fusedadamw'):
        return FusedAdam(parameters, eps=cfg.opt.eps, betas=tuple(cfg.opt.betas), adam_w_mode=kind.endswith('w'),
                         **opt_args)
    raise NotImplementedError(f'optimizer {cfg.opt.name!r}: only adam(w) / fusedadam(w) / sgd variants are provided')


class FusedAdam(torch.optim.Optimizer):
    """Adam / AdamW whose whole step is one multi-tensor HIP launch (plus two for the optional clip).

    ``step(clip_grad=None, grad_scale=1.0)``: with ``clip_grad`` the gradients are scaled by
    ``min(1, clip_grad / (norm + 1e-6))`` inside the update (``.grad`` itself is left untouched) and the global
    norm (a 0-dim device tensor, no host sync) is returned.  Parameters without ``.grad`` are skipped."""

    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-8, adam_w_mode=True,
                 weight_decay=0., amsgrad=False):
        if amsgrad:
            raise RuntimeError('FusedAdam does not support the AMSGrad variant.')
        defaults = dict(lr=lr, bias_correction=bias_correction, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.adam_w_mode = 1 if adam_w_mode else 0
        self._tabs = {}          # signature -> [device buffers, host staging, TensorList, turn] (a few at most)
        self.last_ctl = None

    # ---- tables ------------------------------------------------------------------------------------------
    def _collect(self):
        items = []
        for gi, group in enumerate(self.param_groups):
            for p in group['params']:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError('FusedAdam: parameters must live on a GPU (there is no CPU fallback)')
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse:
                    raise RuntimeError('FusedAdam: fp32 dense parameters and gradients only')
                if not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError('FusedAdam: parameters and gradients must be contiguous')
                st = self.state[p]
                if 'exp_avg' not in st:
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if 'step' not in st:
                    # apex FusedAdam (the reference's 'fusedadamw') keeps ONE step counter per param group and only
                    # the moments per parameter: a state dict saved by it resumes from the group's counter
                    st['step'] = torch.tensor(float(group.get('step', 0)))
                items.append((gi, p, st))
        return items

    def _tables(self, items):
        dev = items[0][1].device
        sig = (dev, tuple((id(p), p.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()) for _, p, st in items))
        nt = len(items)
        tab = self._tabs.get(sig)
        if tab is None:
            chunk_tensor, chunk_start = [], []
            for t, (_, p, _) in enumerate(items):
                for off in range(0, p.numel(), CHUNK):
                    chunk_tensor.append(t)
                    chunk_start.append(off)
            nc = len(chunk_tensor)
            # device tables: int64 [p | g | m | v | numel | chunk_start], float32 [lr | wd], int32 chunk_tensor
            const_i = torch.empty(5 * nt + nc, dtype=torch.int64)
            for t, (_, p, st) in enumerate(items):
                const_i[t] = p.data_ptr()
                const_i[2 * nt + t] = st['exp_avg'].data_ptr()
                const_i[3 * nt + t] = st['exp_avg_sq'].data_ptr()
                const_i[4 * nt + t] = p.numel()
            const_i[5 * nt:] = torch.tensor(chunk_start, dtype=torch.int64)
            dev_i = const_i.to(dev)
            dev_f = torch.empty(2 * nt, dtype=torch.float32, device=dev)
            dev_c = torch.tensor(chunk_tensor, dtype=torch.int32).to(dev)
            partial = torch.empty(max(nc, 1), dtype=torch.float32, device=dev)
            ctl = torch.zeros(4, dtype=torch.float32, device=dev)
            tl = hip.TensorList()
            base = dev_i.data_ptr()
            tl.p, tl.g, tl.m, tl.v = base, base + 8 * nt, base + 16 * nt, base + 24 * nt
            tl.numel, tl.chunk_start = base + 32 * nt, base + 40 * nt
            tl.lr, tl.wd = dev_f.data_ptr(), dev_f.data_ptr() + 4 * nt
            tl.chunk_tensor = dev_c.data_ptr()
            tl.n_chunks, tl.chunk = nc, CHUNK
            # per-step host staging (gradient addresses, lr, wd): pinned, rotated, each guarded by an event so a
            # host running steps ahead of the GPU never overwrites a buffer whose upload has not happened yet
            stage = [(torch.empty(nt, dtype=torch.int64).pin_memory(), torch.empty(2 * nt, dtype=torch.float32).pin_memory(),
                      torch.cuda.Event()) for _ in range(4)]
            if len(self._tabs) >= 8:        # parameter sets change rarely (frozen / unused parameters): keep a few
                self._tabs.pop(next(iter(self._tabs)))
            tab = self._tabs[sig] = [(dev_i, dev_f, dev_c, partial, ctl), stage, tl, 0]
        (dev_i, dev_f, _, partial, ctl), stage, tl, turn = tab
        host_g, host_f, ev = stage[turn % len(stage)]
        tab[3] = turn + 1
        ev.synchronize()
        for t, (gi, p, _) in enumerate(items):
            host_g[t] = p.grad.data_ptr()
            g = self.param_groups[gi]
            host_f[t] = g['lr']
            host_f[nt + t] = g['weight_decay']
        dev_i[nt:2 * nt].copy_(host_g, non_blocking=True)
        dev_f.copy_(host_f, non_blocking=True)
        ev.record()
        return tl, partial, ctl

    # ---- step --------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None, clip_grad=None, grad_scale=1.0):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        items = self._collect()
        if not items:
            return loss
        groups = {gi for gi, _, _ in items}
        ref = self.param_groups[min(groups)]
        for gi in groups:
            g = self.param_groups[gi]
            if (g['betas'], g['eps'], g['bias_correction']) != (ref['betas'], ref['eps'], ref['bias_correction']):
                raise RuntimeError('FusedAdam: betas / eps / bias_correction must be the same in every group')
        # per-parameter step counters (torch.optim.AdamW semantics): a parameter that had no gradient in some step
        # (an objective skipped, find_unused_parameters=True in the reference's DDP wrap) falls behind the others.
        # The bias corrections are per launch, so parameters are bucketed by step: one launch in the usual case.
        by_step = {}
        for it in items:
            by_step.setdefault(int(it[2]['step']), []).append(it)
        with torch.cuda.device(items[0][1].device):
            tl, partial, ctl = self._tables(items)
            use_ctl = clip_grad is not None or grad_scale != 1.0
            if use_ctl:      # the global norm covers every parameter of this step, whatever its counter
                hip.mt_grad_norm(tl, 1.0 / grad_scale, clip_grad if clip_grad is not None else 0.0, partial, ctl)
            b1, b2 = ref['betas']
            for step0, its in by_step.items():
                step = step0 + 1
                tl_b = tl if len(by_step) == 1 else self._tables(its)[0]
                a = hip.AdamArgs()
                a.beta1, a.beta2, a.eps = b1, b2, ref['eps']
                if ref['bias_correction']:
                    a.inv_bc1, a.inv_bc2 = 1.0 / (1.0 - b1 ** step), 1.0 / (1.0 - b2 ** step)
                else:
                    a.inv_bc1 = a.inv_bc2 = 1.0
                a.adam_w_mode = self.adam_w_mode
                hip.mt_adam(tl_b, a, ctl if use_ctl else None)
        # the kernel wrote the parameters and moments behind autograd's back: bump their version counters, which is
        # what invalidates the engine's cached bf16 weight shadows (engine.ShadowCache) and any saved-tensor checks
        touched = [p for _, p, _ in items]
        for _, _, st in items:
            touched += [st['exp_avg'], st['exp_avg_sq']]
        torch.autograd.graph.increment_version(touched)
        for _, _, st in items:
            st['step'] += 1        # host-side counters (GradScaler-style skips are not counted back)
        self.last_ctl = ctl if use_ctl else None
        if clip_grad is not None:
            return ctl[0]
        return loss


class NativeScalerWithGradNormCount:
    """``loss_scaler(loss, optimizer, clip_grad=..., parameters=..., update_grad=...)`` of the reference's loop
    (utils/utils.py:337-371; call site train/pretrain/multimodal.py:296-304).  bf16 arithmetic keeps fp32's exponent
    range, so the loss scale is fixed at 1; with the HIP ``FusedAdam`` the unscale + clip + step are three launches
    and the returned norm is a device tensor."""
    state_dict_key = 'amp_scaler'

    def __init__(self, reducer=None):
        self.reducer = reducer       # exploremultimodal_amd.dp.GradReducer or None

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True):
        if self.reducer is not None:
            self.reducer.prepare(loss)
        from . import engine
        # an ordinary accumulating backward(): a micro-step may add straight into the gradient views the parameters hold
        with engine.accumulate_into_grad(not create_graph):
            loss.backward(create_graph=create_graph)
        if self.reducer is not None:
            self.reducer.finish(accumulate=not update_grad)
        if not update_grad:
            return None
        if isinstance(optimizer, FusedAdam):
            norm = optimizer.step(clip_grad=clip_grad if clip_grad is not None else float('inf'))
        else:
            assert parameters is not None
            norm = torch.nn.utils.clip_grad_norm_(parameters, clip_grad if clip_grad is not None else float('inf'))
            optimizer.step()
        return norm

    def state_dict(self):
        return {'scale': 1.0, 'growth_factor': 2.0, 'backoff_factor': 0.5, 'growth_interval': 2000, '_growth_tracker': 0}

    def load_state_dict(self, state_dict):
        pass
