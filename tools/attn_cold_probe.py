"""Attention backward at the fused-layer shape (64 sequences of 261 tokens, 12 heads) with operands hot (back-to-back
repetitions), cold (512 MB written between repetitions: nothing of qkv / ctx / dctx left in the L2s or the Infinity
Cache) and cold + touched (qkv and ctx read once by a plain reduction right before the launch, as a prefetch would).
usage: python tools/attn_cold_probe.py [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip
dev = 'cuda'
B, H, d = 64, 12, 768
lens = [int(a) for a in sys.argv[1:]] or [261, 197]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def run(fn, pre, reps=12):
    ts = []
    for i in range(reps + 2):
        pre()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        if i >= 2: ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


for N in lens:
    M = B * N
    qkv = torch.randn(M, 3 * d, device=dev).bfloat16()
    seg = torch.tensor([[b * N, N, 0, 0] for b in range(B)], dtype=torch.int32, device=dev)
    km = torch.ones(M, dtype=torch.int32, device=dev)
    ctx = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B * H, ((N + 31) // 32) * 32, device=dev)
    dctx = torch.randn(M, d, device=dev).bfloat16()
    dqkv = torch.empty(M, 3 * d, device=dev, dtype=torch.bfloat16)
    dp = hip.drop_params(0.1, True)
    hip.attn_fwd(qkv, seg, B, km, ctx, lse, H, d, N, 0.125, drop=dp, seed=1)
    bwd = lambda: hip.attn_bwd(qkv, ctx, dctx, lse, seg, B, km, dqkv, H, d, N, 0.125, drop=dp, seed=1)
    fwd = lambda: hip.attn_fwd(qkv, seg, B, km, ctx, lse, H, d, N, 0.125, drop=dp, seed=1)
    qv, cv = qkv.view(torch.int32), ctx.view(torch.int32)

    def cold():
        flush.fill_(1)

    def cold_dctx_hot():           # what the step looks like: dctx was just written by the projection's input gradient
        flush.fill_(1); dctx.add_(0)

    def touched():
        flush.fill_(1); dctx.add_(0); qv.sum(); cv.sum()
    print(f'N={N}: bwd hot {run(bwd, lambda: None):6.1f} us | cold {run(bwd, cold):6.1f} | cold, dctx hot {run(bwd, cold_dctx_hot):6.1f} '
          f'| cold + qkv, ctx touched {run(bwd, touched):6.1f} || fwd hot {run(fwd, lambda: None):6.1f} cold {run(fwd, cold):6.1f}', flush=True)
