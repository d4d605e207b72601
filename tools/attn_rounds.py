"""Forward attention time against the number of workgroups (one per sequence and head): how much of a launch is the
last, partly filled dispatch round.  python tools/attn_rounds.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip
dev = 'cuda'
H, d = 12, 768
N = int(sys.argv[1]) if len(sys.argv) > 1 else 197
def timeit(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
for B in (11, 21, 32, 42, 43, 53, 64, 85, 86, 128):
    M = B * N
    qkv = torch.randn(M, 3 * d, device=dev).bfloat16()
    seg = torch.tensor([[0, 0, b * N, N] for b in range(B)], dtype=torch.int32, device=dev)
    km = torch.ones(M, dtype=torch.int32, device=dev)
    ctx = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B * H, ((N + 31) // 32) * 32, device=dev)
    for drop in (0.0, 0.1):
        dp = hip.drop_params(drop, True)
        t = timeit(lambda: hip.attn_fwd(qkv, seg, B, km, ctx, lse, H, d, N, 0.125, drop=dp, seed=1))
        print(f'N={N} B={B:3d} workgroups={B * H:5d} ({B * H / 512:.2f} rounds of 512) drop={drop}: {t:6.1f} us', flush=True)
