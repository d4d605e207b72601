"""Host-side enqueue cost of the step's pieces (no GPU sync inside the timed pieces)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from exploremultimodal_amd import engine, hip
from exploremultimodal_amd import synth
dev = torch.device('cuda', 0)
model, mc = bench.build_model('base', dev)
model.train()
B = 64
batch = synth.synth_batch(mc, B, seed=1, mim=False)
P = synth.num_img_tokens(mc)
img, ids, tmask = batch['image'].to(dev), batch['text_ids'].to(dev), batch['text_mask'].to(dev)
imask = torch.ones(B, P, dtype=torch.int64, device=dev)
R = torch.randn(B, mc.max_text_len + P, mc.embed_dim, device=dev) / 1e4
def step(timing=None):
    for p in model.parameters(): p.grad = None
    t0 = time.perf_counter()
    x, _ = model.forward_features(img=img, txt=ids, img_attn_masks=imask, txt_attn_masks=tmask)
    t1 = time.perf_counter()
    loss = (x * R).sum()
    loss.backward()
    t2 = time.perf_counter()
    if timing is not None: timing.append((t1 - t0, t2 - t1))
for _ in range(3): step()
torch.cuda.synchronize()
tm = []
for _ in range(5):
    step(tm); torch.cuda.synchronize()
print('fwd enqueue ms', [round(a*1e3,2) for a,_ in tm], 'bwd enqueue ms', [round(b*1e3,2) for _,b in tm])
# micro: one block forward call host cost
blk = model.blocks[8]
plan = engine.Plan(B, mc.max_text_len, P, dev, tmask, imask)
x = torch.randn(plan.M, mc.embed_dim, device=dev)
torch.cuda.synchronize()
with torch.no_grad():
    for _ in range(3): blk.run(x, plan, ['vl'], [(0, plan.M)], True, model._shadows, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): blk.run(x, plan, ['vl'], [(0, plan.M)], True, model._shadows, 1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
print('block.run host us (train mode, no grad):', (t1 - t0) / 20 * 1e6)
model.eval()
with torch.no_grad():
    t0 = time.perf_counter()
    for _ in range(20): blk.run(x, plan, ['vl'], [(0, plan.M)], True, model._shadows, 1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
print('block.run host us (eval):', (t1 - t0) / 20 * 1e6)
t0 = time.perf_counter()
for _ in range(20): engine.Plan(B, mc.max_text_len, P, dev, tmask, imask)
t1 = time.perf_counter()
print('Plan() host us:', (t1 - t0) / 20 * 1e6)
