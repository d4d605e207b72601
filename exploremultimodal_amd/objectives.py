"""Pretraining objectives of VlmoModule.forward (models/vlmo/objectives.py:12-314, 532-607):
same function names, arguments and returned dict keys.  The backbone passes they trigger run
on the HIP engine; the loss arithmetic on the gathered rows is stock torch.

`compute_itm` draws its hard negatives with torch.multinomial like the reference
(objectives.py:266-275) but in two batched calls instead of 2*B `.item()` host syncs; a caller
that needs bit-reproducible negatives passes `batch['itm_neg_idx'] = (img_neg_idx, txt_neg_idx)`.
"""
import torch
import torch.nn.functional as F

from .dvae import create_d_vae  # noqa: F401  (objectives.create_d_vae in the reference)


class GatherLayer(torch.autograd.Function):
    """All-gather that keeps gradients (objectives.py:392-426): ``GatherLayer.apply(feat, group, rank)`` returns the
    features of every rank concatenated in rank order; backward hands each rank the SUM over ranks of the gradient
    rows that belong to it.  One ``all_gather_into_tensor`` forward; backward is one ``reduce_scatter_tensor`` (RCCL:
    1/world of the traffic of the reference's all_reduce + slice; same result) or all_reduce + slice where the
    backend has no reduce-scatter (gloo)."""

    @staticmethod
    def forward(ctx, tensor, group, rank):
        import torch.distributed as dist
        ctx.batch_size = tensor.shape[0]
        ctx.group = group
        ctx.rank = rank
        world = dist.get_world_size(group)
        tensor = tensor.contiguous()
        out = tensor.new_empty((world * tensor.shape[0],) + tuple(tensor.shape[1:]))
        dist.all_gather_into_tensor(out, tensor, group=group)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        import torch.distributed as dist
        grad_output = grad_output.contiguous()
        bs = ctx.batch_size
        if dist.get_backend(ctx.group) == 'nccl':
            grad_input = grad_output.new_empty((bs,) + tuple(grad_output.shape[1:]))
            dist.reduce_scatter_tensor(grad_input, grad_output, op=dist.ReduceOp.SUM, group=ctx.group)
            return grad_input, None, None
        grad_input = grad_output.clone()
        dist.all_reduce(grad_input, op=dist.ReduceOp.SUM, group=ctx.group)
        return grad_input[ctx.rank * bs:(ctx.rank + 1) * bs], None, None


def compute_accuracy(logits, target, all_valid=False):
    """objectives.py:24-37 -> (mean accuracy over labels != -100, their count).  ``all_valid``: the caller built the
    targets itself and none is -100 (ITC: arange, ITM: ones / zeros) -- same values without the three host
    synchronisations of the masked form (``int(keep.sum())`` and two boolean gathers)."""
    if all_valid:
        n = target.numel()
        if n == 0:
            return torch.tensor(0, device=target.device), 0
        return (logits.argmax(dim=-1) == target).float().mean(), n
    keep = target != -100
    n = int(keep.sum())
    if n == 0:
        return torch.tensor(0, device=target.device), 0
    return (logits.argmax(dim=-1)[keep] == target[keep]).float().mean(), n


def attach_row_indices(batch):
    """Input hand-off companion: on the HOST copy of a batch, before its upload, list the rows the two
    vocabulary heads will gather -- ``_mlm_rows`` = flat indices into [B * T] of the positions whose MLM label is not -100
    (objectives.py:52-56), ``_mim_rows`` = flat indices into [B * patches] of the masked patches and ``_mim_tok_rows`` = the
    same positions in the [B * (patches + 1)] token rows (CLS first) (objectives.py:542-570).  With them compute_mlm /
    compute_mim gather by index (``index_select``: the row count is known on the host) instead of by boolean mask, which
    has to ask the device for its count: four host synchronisations per step, two more in the backward.  Same rows, same
    order (ascending) as the boolean form.  A batch without these keys takes the reference's boolean path.  The keys
    describe THIS batch's masks: whoever edits ``text_labels_mlm`` / ``image_bool_masked_pos`` afterwards drops them."""
    lab = batch.get('text_labels_mlm')
    if torch.is_tensor(lab) and not lab.is_cuda:
        batch['_mlm_rows'] = (lab.reshape(-1) != -100).nonzero(as_tuple=False).reshape(-1)
    bm = batch.get('image_bool_masked_pos')
    if torch.is_tensor(bm) and not bm.is_cuda:
        flat = bm.reshape(bm.shape[0], -1) != 0
        rows = flat.reshape(-1).nonzero(as_tuple=False).reshape(-1)
        patches = flat.shape[1]
        batch['_mim_rows'] = rows
        batch['_mim_tok_rows'] = rows + torch.div(rows, patches, rounding_mode='floor') + 1
    return batch


def _vocab_head_loss(model, head, feats, labels, vocab):
    """Loss / logits / accuracy of a vocabulary head on gathered rows (objectives.py:57-68, 571-582).

    ``config.train.fused_ce``: unset (default) or True = the loss and its gradient go through the HIP path that keeps the
    [rows, vocabulary] logits out of HBM (heads.LinearCrossEntropyFn) whenever the features live on the GPU; False = the
    reference's ``F.cross_entropy(head(feats))``.  ``config.train.return_logits`` (default True: the reference's output
    dict carries `mlm_logits` / `mim_logits`, objectives.py:70-77, 584-590): with the HIP loss the logits for the dict are
    computed on the side WITHOUT a graph (nothing in the reference's loop differentiates them; a caller that does sets
    ``fused_ce = False``); False, or ``fused_ce = True`` without an explicit ``return_logits``, returns None for them.
    With no rows the loss is the python float 0."""
    n = labels.numel()
    tr = model.config.train
    fc, rl = getattr(tr, 'fused_ce', None), getattr(tr, 'return_logits', None)
    fused = (fc is None or bool(fc)) and feats.is_cuda
    want_logits = (not fused) or (bool(rl) if rl is not None else fc is None)
    if n == 0:
        return 0., (head(feats) if want_logits else None), torch.tensor(0, device=labels.device), 0
    if fused:
        loss, pred = head.loss_and_pred(feats, labels)
        logits = None
        if want_logits:
            with torch.no_grad():
                logits = head(feats)
        return loss, logits, (pred.to(labels.dtype) == labels).float().mean(), n
    logits = head(feats)
    acc, cnt = compute_accuracy(logits, labels)
    return F.cross_entropy(logits.view(-1, vocab), labels.view(-1), ignore_index=-100), logits, acc, cnt


def compute_mlm(model, batch):
    """objectives.py:40-78: MLM head on the text positions whose label is not -100."""
    infer = batch.get('_mlm_infer')
    if infer is None:
        mode = 'img-txt' if any('image' in k for k in batch.keys()) else 'txt_only'
        infer = model.infer(batch, infer_mode=mode, mask_txt=True, mask_img=False)
    labels_all = infer['txt_labels']
    rows = batch.get('_mlm_rows')
    if rows is not None and rows.device == labels_all.device:
        tf = infer['txt_feats']
        feats = tf.reshape(-1, tf.shape[-1]).index_select(0, rows)      # [n_masked, d], no host synchronisation
        labels = labels_all.reshape(-1).index_select(0, rows)
    else:
        picked = labels_all != -100
        feats = infer['txt_feats'][picked].contiguous()          # [n_masked, d]
        labels = labels_all[picked]
    loss, logits, acc, cnt = _vocab_head_loss(model, model.mlm_head, feats, labels, model.config.model.vocab_size)
    return {'mlm_task_loss': loss, 'mlm_logits': logits, 'mlm_labels': labels, 'mlm_ids': infer['txt_ids'],
            'mlm_mean_acc': acc, 'mlm_count': cnt}


def compute_itc(model, batch):
    """objectives.py:81-236: the in-batch branch and the ``global_reduce`` branch (negatives gathered from every
    rank with GatherLayer, objectives.py:99-108); the momentum / queue branches (off in
    conf/train/pretrain_mum.yaml:39-42) are out of scope."""
    with torch.no_grad():
        # in place (same values as the reference's re-assignment of .data): the parameter keeps its storage, which the
        # flat-buffer optimizers (FusedAdam's cached tables, zero.ZeroAdam's re-homed parameters) rely on
        model.itc_temp.data.clamp_(0, 4.6052)
    temp = model.itc_temp.exp()
    if model.transformer_m is not None:
        raise NotImplementedError('the momentum ITC branch is out of scope (SURVEY.md 8f)')
    img_infer = batch.get('_itc_img_infer') or model.infer(batch, infer_mode='img_only')
    txt_infer = batch.get('_itc_txt_infer') or model.infer(batch, infer_mode='txt_only')
    i_feat = model.itc_head(img_infer['co_feats'][:, 0], 'v')
    t_feat = model.itc_head(txt_infer['co_feats'][:, 0], 'l')
    bs = i_feat.size(0)
    sim_targets = torch.arange(bs, device=i_feat.device)
    if model.config.train.global_reduce:
        import torch.distributed as dist
        rank = dist.get_rank()
        # own rows rolled to the front, so that column j < bs is this rank's pair j (targets stay arange(bs))
        i_feats = torch.roll(GatherLayer.apply(i_feat, None, rank), -bs * rank, 0)
        t_feats = torch.roll(GatherLayer.apply(t_feat, None, rank), -bs * rank, 0)
        sim_i2t = i_feat @ t_feats.t() * temp
        sim_t2i = t_feat @ i_feats.t() * temp
    else:
        sim_i2t = i_feat @ t_feat.t() * temp
        sim_t2i = sim_i2t.t()
    i2t_loss = F.cross_entropy(sim_i2t, sim_targets)
    t2i_loss = F.cross_entropy(sim_t2i, sim_targets)
    itc_i2t_mean_acc, itc_i2t_count = compute_accuracy(sim_i2t[:, :bs], sim_targets, all_valid=True)
    itc_t2i_mean_acc, itc_t2i_count = compute_accuracy(sim_t2i[:, :bs], sim_targets, all_valid=True)
    return {'itc_task_loss': (i2t_loss + t2i_loss) / 2, 'i2t_Loss': i2t_loss, 't2i_Loss': t2i_loss,
            'sim_i2t': sim_i2t, 'sim_t2i': sim_t2i, 'itc_temp': temp.data,
            'itc_i2t_mean_acc': itc_i2t_mean_acc, 'itc_i2t_count': itc_i2t_count,
            'itc_t2i_mean_acc': itc_t2i_mean_acc, 'itc_t2i_count': itc_t2i_count}


def sample_itm_negatives(batch, sim_dict=None):
    """Hard-negative indices of compute_itm (objectives.py:251-275) -> (img_neg_idx, txt_neg_idx), both [B]."""
    img = batch['image']
    bs = img.size(0)
    with torch.no_grad():
        if batch.get('itm_neg_idx') is not None:
            return batch['itm_neg_idx']
        if sim_dict is not None:
            weights_i2t = F.softmax(sim_dict['sim_i2t'][:, :bs].float(), dim=1) + 1e-5
            weights_t2i = F.softmax(sim_dict['sim_t2i'][:, :bs].float(), dim=1) + 1e-5
        else:
            weights_i2t = F.softmax(torch.randn([bs, bs], device=img.device), dim=1) + 1e-5
            weights_t2i = F.softmax(torch.randn([bs, bs], device=img.device), dim=1) + 1e-5
        weights_i2t.fill_diagonal_(0)
        weights_t2i.fill_diagonal_(0)
        img_neg_idx = torch.multinomial(weights_t2i, 1).squeeze(1)     # one draw per row, no host sync
        txt_neg_idx = torch.multinomial(weights_i2t, 1).squeeze(1)
    return img_neg_idx, txt_neg_idx


def itm_negative_batch(batch, img_neg_idx, txt_neg_idx):
    """The 2B-pair negative batch of compute_itm (objectives.py:277-293): (negative image, text), (image, negative text)."""
    txt_ids, txt_mask, img = batch['text_ids'], batch['text_mask'], batch['image']
    return {'text_ids': torch.cat([txt_ids, txt_ids[txt_neg_idx]], dim=0),
            'text_mask': torch.cat([txt_mask, txt_mask[txt_neg_idx]], dim=0),
            'image': torch.cat([img[img_neg_idx], img], dim=0)}


def compute_itm(model, batch, sim_dict=None):
    """objectives.py:239-314.  ``batch['_itm_infer'] = (output_pos, output_neg)`` (set by the merged-pass forward of
    VlmoModule) supplies the two backbone results instead of running them here."""
    bs = batch['image'].size(0)
    if batch.get('_itm_infer') is not None:
        output_pos, output_neg = batch['_itm_infer']
    else:
        output_pos = model.infer(batch, infer_mode='img-txt')
        img_neg_idx, txt_neg_idx = sample_itm_negatives(batch, sim_dict)
        output_neg = model.infer(itm_negative_batch(batch, img_neg_idx, txt_neg_idx), infer_mode='img-txt')
    cls_feat = torch.cat([output_pos['cls_feats'], output_neg['cls_feats']], dim=0)
    itm_logits = model.itm_head(cls_feat)
    itm_labels = torch.cat([torch.ones(1 * bs, dtype=torch.long, device=itm_logits.device),
                            torch.zeros(2 * bs, dtype=torch.long, device=itm_logits.device)], dim=0)
    itm_loss = F.cross_entropy(itm_logits, itm_labels)
    itm_mean_acc, itm_count = compute_accuracy(itm_logits, itm_labels, all_valid=True)
    return {'itm_task_loss': itm_loss, 'itm_logits': itm_logits, 'itm_labels': itm_labels,
            'itm_mean_acc': itm_mean_acc, 'itm_count': itm_count}


def compute_mim(module, batch):
    """objectives.py:532-592."""
    with torch.no_grad():
        input_ids = module.d_vae.get_codebook_indices(batch['image4dalle']).flatten(1)
        batch['image_bool_masked_pos'] = batch['image_bool_masked_pos'].flatten(1).to(torch.bool)
        bool_masked_pos = batch['image_bool_masked_pos']
        rows, tok_rows = batch.get('_mim_rows'), batch.get('_mim_tok_rows')
        by_index = rows is not None and tok_rows is not None and rows.device == input_ids.device
        mim_labels = input_ids.reshape(-1).index_select(0, rows) if by_index else input_ids[bool_masked_pos]
    pos = module.config.train.mim_head_pos
    if batch.get('_mim_infer') is not None:
        infer = batch['_mim_infer']
    elif pos in ['img']:
        infer = module.infer(batch, infer_mode='img_only', mask_txt=False, mask_img=True)
    elif pos in ['mum']:
        infer = module.infer(batch, infer_mode='img-txt', mask_txt=False, mask_img=True)
    elif pos in ['fusion']:
        img_feats = module.transformer.forward_interval(
            x=batch['image'], attn_masks=None, route='v', need_embed=True, bool_masked_pos=bool_masked_pos,
            in_layer=0, out_layer=module.transformer.fusion_layer, need_norm=True)
        infer = {'img_feats': img_feats}
    else:
        raise KeyError(f'unknown mim_head_pos {pos!r}')
    if by_index and infer['img_feats'].shape[1] == bool_masked_pos.shape[1] + 1:
        xf = infer['img_feats']
        feats = xf.reshape(-1, xf.shape[-1]).index_select(0, tok_rows)       # the same rows, CLS skipped by the index
    else:
        feats = infer['img_feats'][:, 1:][bool_masked_pos].contiguous()      # [n_masked, d] (patch tokens only)
    loss, logits, acc, cnt = _vocab_head_loss(module, module.mim_head, feats, mim_labels,
                                              module.config.model.img_vocab_size)
    return {'mim_task_loss': loss, 'mim_logits': logits, 'mim_labels': mim_labels, 'mim_mean_acc': acc,
            'mim_count': cnt}
