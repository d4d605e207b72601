"""Lists the host<->device synchronisation points of one four-loss step (torch.cuda.set_sync_debug_mode('warn')):
file:line of the innermost frame inside this package for every warning, with counts.
usage: python tools/sync_probe.py [--batch 32] [--no-merge]"""
import argparse
import collections
import os
import sys
import traceback
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import synth
from exploremultimodal_amd.build import build_model

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--no-merge', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda:0')
cfg = synth.make_config('base', loss_names=['mlm', 'mim', 'itc', 'itm'], drop_rate=0.1, attn_drop_rate=0.1, drop_path_rate=0.1)
cfg.train.merge_passes = not args.no_merge
cfg.train.fused_ce = True
torch.manual_seed(0)
model = build_model(cfg).to(dev).train()
from exploremultimodal_amd.objectives import attach_row_indices
ap2 = synth.synth_batch(cfg.model, args.batch, seed=1234, mim=True)
if not os.environ.get('NO_ROW_INDICES'):
    attach_row_indices(ap2)
batch = {k: v.to(dev) for k, v in ap2.items()}


def step():
    for p in model.parameters():
        p.grad = None
    ret = model(dict(batch))
    loss = sum(v for k, v in ret.items() if 'task_loss' in k)
    loss.backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
sites = collections.Counter()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def showwarning(message, category, filename, lineno, file=None, line=None):
    if 'synchronizing' not in str(message) or 'prototype feature' in str(message):
        return
    where = None
    for fr in traceback.extract_stack()[:-1]:
        if fr.filename.startswith(root) and 'sync_probe' not in fr.filename:
            where = f'{os.path.relpath(fr.filename, root)}:{fr.lineno} {fr.line}'
    sites[where or f'{filename}:{lineno}'] += 1


warnings.showwarning = showwarning
warnings.simplefilter('always')
torch.cuda.set_sync_debug_mode('warn')
step()
torch.cuda.set_sync_debug_mode('default')
torch.cuda.synchronize()
for k, v in sites.most_common():
    print(f'{v:3d} x {k}')
print('total', sum(sites.values()))
