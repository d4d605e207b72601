"""Checkpoint wire format of the reference's training loops: ``checkpoint-<epoch>.pth`` =
``{'model', 'optimizer', 'lr_scheduler', 'epoch', 'scaler', 'cfg'}`` (utils/utils.py:479-520), auto-resume from
the newest ``<exp_dir>/*/checkpoint-N.pth`` and loading through ``VlmoModule.load_from_ckpt`` (VLMo or BEiT key
layout, position-embedding interpolation) -- utils/utils.py:534-612.  Files written by either side load in the
other: parameter names are the reference's and the optimizer state is ``torch.optim.AdamW``-shaped.
The DeepSpeed (``.ds``) branches of the reference are not provided: sharded state is this framework's own
``exploremultimodal_amd.dp`` reducer, which keeps ordinary replicated ``.pth`` checkpoints."""
import glob
import os
import shutil
from pathlib import Path

import torch


def _is_master():
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def save_on_master(obj, path):
    if _is_master():
        torch.save(obj, path)


def save_model(cfg, epoch, model, model_without_ddp, optimizer, lr_scheduler, loss_scaler, model_ema=None):
    """utils/utils.py:479-520 (torch.amp branch).  Returns the file name."""
    if loss_scaler is None:
        raise NotImplementedError('DeepSpeed checkpoints (.ds) are not provided; pass the loss scaler object')
    if model_ema is not None:
        raise NotImplementedError('EMA twins (vlmo_ema) are out of scope')
    output_dir = Path(cfg.output_dir)
    ckpt_name = f'checkpoint-{epoch}.pth'
    to_save = {
        'model': model_without_ddp.state_dict(),
        'optimizer': optimizer.state_dict(),
        'lr_scheduler': lr_scheduler.state_dict(),
        'epoch': epoch,
        'scaler': loss_scaler.state_dict(),
        'cfg': cfg,
    }
    if _is_master():
        output_dir.mkdir(parents=True, exist_ok=True)
    save_on_master(to_save, output_dir / ckpt_name)
    return ckpt_name


def remove_models(cfg, epoch, best_epoch):
    """utils/utils.py:523-532: keep only the checkpoints of ``epoch`` and ``best_epoch``."""
    if cfg.dist.rank == 0:
        for ckpt in glob.glob(os.path.join(Path(cfg.output_dir), 'checkpoint-*')):
            t = ckpt.split('-')[-1].split('.')[0]
            if t.isdigit() and t not in [str(epoch), str(best_epoch)]:
                shutil.rmtree(ckpt) if os.path.isdir(ckpt) else os.remove(ckpt)


def latest_checkpoint(exp_dir, pattern='checkpoint-%d.pth'):
    """Newest ``<exp_dir>/*/checkpoint-N.pth`` by N -> (N, path) or (-1, '')."""
    best, best_path = -1, ''
    for ckpt in glob.glob(os.path.join(exp_dir, '*', pattern.replace('%d', '*'))):
        t = ckpt.split('-')[-1].split('.')[0]
        if t.isdigit() and int(t) > best:
            best, best_path = int(t), ckpt
    return best, best_path


def auto_load_model(cfg, model, model_without_ddp, optimizer, lr_scheduler, loss_scaler, model_ema=None, logger=None):
    """utils/utils.py:534-612 (torch.amp branch): resolve ``cfg.train.resume`` (auto-resume picks the newest
    checkpoint under ``cfg.exp_dir``), load the weights through ``load_from_ckpt`` and -- when the checkpoint was
    written by the same ``(train.phase, tag)`` -- the optimizer / schedule / scaler state and ``start_epoch``."""
    import logging
    logger = logger or logging.getLogger(__name__)
    if loss_scaler is None:
        raise NotImplementedError('DeepSpeed checkpoints (.ds) are not provided; pass the loss scaler object')
    if cfg.train.auto_resume and len(cfg.train.resume) == 0:
        n, path = latest_checkpoint(Path(cfg.exp_dir))
        if n >= 0 and len(path) > 0:
            cfg.train.resume = path
        logger.warning(f'Auto resume checkpoint: {cfg.train.resume}')
    if not cfg.train.resume:
        logger.info('No ckpt or BEiT, start training from scratch...')
        return None
    if cfg.train.resume.startswith('https'):
        raise NotImplementedError('remote checkpoints are not fetched; download the file and pass its path')
    # the reference pickles its whole config object into the file: trusted local files only
    ckpt = torch.load(cfg.train.resume, map_location='cpu', weights_only=False)
    match, is_beit = model_without_ddp.load_from_ckpt(ckpt['model'])
    if is_beit:
        logger.warning(f'Initialized BEiT pretrained => {cfg.train.resume}')
    else:
        logger.info(f'Resume checkpoint ==> {cfg.train.resume}')
    if len(match.missing_keys) > 0:
        logger.warning(f'Weights not initialized from pretrained model: {match.missing_keys}')
    if len(match.unexpected_keys) > 0:
        logger.warning(f'Weights from pretrained model not used: {match.unexpected_keys}')
    if 'cfg' in ckpt:
        ckpt_cfg = ckpt['cfg']
        if (cfg.train.phase, cfg.tag) == (ckpt_cfg.train.phase, ckpt_cfg.tag):
            if 'optimizer' in ckpt and 'lr_scheduler' in ckpt and 'epoch' in ckpt:
                cfg.train.start_epoch = ckpt['epoch'] + 1
                optimizer.load_state_dict(ckpt['optimizer'])
                if cfg.train.start_epoch < ckpt_cfg.train.epochs:
                    lr_scheduler.load_state_dict(ckpt['lr_scheduler'])
                if 'scaler' in ckpt:
                    loss_scaler.load_state_dict(ckpt['scaler'])
                logger.info('Load states with optim & sched!')
    return match
