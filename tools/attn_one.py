"""One attention-backward shape in a loop (for rocprofv3 --pmc passes): python tools/attn_one.py [lenA lenB reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip
dev = 'cuda'
B, H, d = 64, 12, 768
lenA, lenB, reps = (int(x) for x in (sys.argv[1:4] + ['0', '197', '10'][len(sys.argv) - 1:]))
N = lenA + lenB
M = B * N
qkv = torch.randn(M, 3 * d, device=dev).bfloat16()
seg = torch.tensor([[b * lenA, lenA, B * lenA + b * lenB, lenB] for b in range(B)], dtype=torch.int32, device=dev)
km = torch.ones(M, dtype=torch.int32, device=dev)
ctx = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
lse = torch.empty(B * H, ((N + 31) // 32) * 32, device=dev)
dctx = torch.randn(M, d, device=dev).bfloat16()
dqkv = torch.empty(M, 3 * d, device=dev, dtype=torch.bfloat16)
dp = hip.drop_params(0.1, True)
hip.attn_fwd(qkv, seg, B, km, ctx, lse, H, d, N, 0.125, drop=dp, seed=1)
for _ in range(reps):
    hip.attn_bwd(qkv, ctx, dctx, lse, seg, B, km, dqkv, H, d, N, 0.125, drop=dp, seed=1)
torch.cuda.synchronize()
