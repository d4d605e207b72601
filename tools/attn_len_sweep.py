"""Attention forward / backward time against the sequence length at the VLMo-Base B=64 shape (which kernel takes which
length: single-pass backward up to 256 tokens, two-phase above; VLMO_ATTN_BWD=two_phase forces the latter).
usage: python tools/attn_len_sweep.py [len ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip
dev = 'cuda'
B, H, d = 64, 12, 768
lens = [int(a) for a in sys.argv[1:]] or [197, 224, 256, 261, 288]


def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


for N in lens:
    M = B * N
    qkv = torch.randn(M, 3 * d, device=dev).bfloat16()
    seg = torch.tensor([[b * N, N, 0, 0] for b in range(B)], dtype=torch.int32, device=dev)
    km = torch.ones(M, dtype=torch.int32, device=dev)
    ctx = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    npad = ((N + 31) // 32) * 32
    lse = torch.empty(B * H, npad, device=dev)
    dctx = torch.randn(M, d, device=dev).bfloat16()
    dqkv = torch.empty(M, 3 * d, device=dev, dtype=torch.bfloat16)
    dp = hip.drop_params(0.1, True)
    tf = timeit(lambda: hip.attn_fwd(qkv, seg, B, km, ctx, lse, H, d, N, 0.125, drop=dp, seed=1))
    tb = timeit(lambda: hip.attn_bwd(qkv, ctx, dctx, lse, seg, B, km, dqkv, H, d, N, 0.125, drop=dp, seed=1))
    fl = 4.0 * B * H * N * N * 64
    print(f'N={N}: fwd {tf*1e6:7.1f} us {fl/tf/1e12:6.1f} TF/s | bwd {tb*1e6:7.1f} us {2.5*fl/tb/1e12:6.1f} TF/s', flush=True)
