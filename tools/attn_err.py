"""Forward attention error against an fp64 soft-max on the same bf16 inputs (no dropout).
python tools/attn_err.py   (VLMO_ATTN_FWD=chunked selects the chunked kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip
dev = 'cuda'
B, H, d = 8, 12, 768
for N, sc in ((261, 1.0), (261, 4.0), (197, 1.0), (64, 2.0)):
    torch.manual_seed(0)
    M = B * N
    qkv = (torch.randn(M, 3 * d, device=dev) * sc).bfloat16()
    seg = torch.tensor([[0, 0, b * N, N] for b in range(B)], dtype=torch.int32, device=dev)
    km = torch.ones(M, dtype=torch.int32, device=dev)
    ctx = torch.empty(M, d, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(B * H, ((N + 31) // 32) * 32, device=dev)
    hip.attn_fwd(qkv, seg, B, km, ctx, lse, H, d, N, 0.125, drop=hip.drop_params(0.0, False), seed=1)
    q, k, v = (t.reshape(B, N, H, 64).permute(0, 2, 1, 3).double() for t in qkv.split(d, dim=1))
    p = torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(M, d)
    err = (ctx.double() - ref).abs()
    rel = err / (ref.abs() + 1e-2)
    l = torch.logsumexp(q @ k.transpose(-1, -2) * 0.125, -1).reshape(B * H, N)
    print(f'N={N} scale={sc}: max abs {err.max():.3e} mean abs {err.mean():.3e} max rel {rel.max():.3e} | lse max err {(lse[:, :N].double() - l).abs().max():.2e}')
