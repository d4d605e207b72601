"""Does a step run faster as two half-batches on two streams (HBM-bound and MFMA-bound kernels of the two chains beside
each other, launch gaps of one chain filled by the other)?  Compares one fwd+bwd of B pairs on one stream with two
fwd+bwd of B/2 pairs enqueued on two streams.  python tools/two_stream_probe.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from exploremultimodal_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device('cuda', 0)
model, mc = bench.build_model('base', dev, drop=0.1)
model.train()
P = synth.num_img_tokens(mc)


def inputs(b, seed):
    bt = synth.synth_batch(mc, b, seed=seed)
    return dict(img=bt['image'].to(dev), txt=bt['text_ids'].to(dev), txt_attn_masks=bt['text_mask'].to(dev),
                img_attn_masks=torch.ones(b, P, dtype=torch.int64, device=dev)), \
        torch.randn(b, mc.max_text_len + P, mc.embed_dim, device=dev) / (b * 1000.0)


full = inputs(B, 1)
halves = [inputs(B // 2, 2), inputs(B // 2, 3)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def one(kw, R):
    x, _ = model.forward_features(**kw)
    (x * R).sum().backward()


def step_full():
    for p in model.parameters():
        p.grad = None
    one(*full)


def step_halves(interleave):
    for p in model.parameters():
        p.grad = None
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    if interleave:      # forward of both, then backward of both: the two chains stay in the same phase
        outs = []
        for s, (kw, R) in zip(streams, halves):
            with torch.cuda.stream(s):
                x, _ = model.forward_features(**kw)
                outs.append((x * R).sum())
        for s, l in zip(streams, outs):
            with torch.cuda.stream(s):
                l.backward()
    else:
        for s, (kw, R) in zip(streams, halves):
            with torch.cuda.stream(s):
                one(kw, R)
    for s in streams:
        cur.wait_stream(s)


def timeit(fn, n=30, w=8):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f'one stream, B={B}: {timeit(step_full):.2f} ms/step', flush=True)
print(f'two streams, 2 x B={B // 2}, chain after chain: {timeit(lambda: step_halves(False)):.2f} ms/step', flush=True)
print(f'two streams, 2 x B={B // 2}, forward both then backward both: {timeit(lambda: step_halves(True)):.2f} ms/step', flush=True)
print(f'one stream, B={B}: {timeit(step_full):.2f} ms/step', flush=True)
