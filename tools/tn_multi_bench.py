"""Micro-benchmark of vlmo_gemm_tn_multi on the weight gradients of `nblk` VLMo-Base blocks (random data).
VLMO_TN_SPLITS=n forces the token-dimension split.  usage: python tools/tn_multi_bench.py [nblk] [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip
nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 2
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16704
dev = 'cuda'
d, hid = 768, 3072
probs = []
fl = 0
for b in range(nblk):
    dz2 = torch.randn(M, d, device=dev).bfloat16(); h = torch.randn(M, hid, device=dev).bfloat16()
    du = torch.randn(M, hid, device=dev).bfloat16(); y2 = torch.randn(M, d, device=dev).bfloat16()
    dz1 = torch.randn(M, d, device=dev).bfloat16(); ctx = torch.randn(M, d, device=dev).bfloat16()
    dqkv = torch.randn(M, 3 * d, device=dev).bfloat16(); y1 = torch.randn(M, d, device=dev).bfloat16()
    for A, B, n1, n2 in ((dz2, h, d, hid), (du, y2, hid, d), (dz1, ctx, d, d), (dqkv, y1, 3 * d, d)):
        probs.append((A, B, torch.zeros(n1, n2, device=dev), M, n1, n2, True))
        fl += 2 * M * n1 * n2
for _ in range(3):
    hip.gemm_tn_multi(probs)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 10
a.record()
for _ in range(reps):
    hip.gemm_tn_multi(probs)
b.record()
torch.cuda.synchronize()
t = a.elapsed_time(b) / reps * 1e-3
print(f'tn_multi nblk={nblk} M={M} splits={os.environ.get("VLMO_TN_SPLITS", "auto")}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s', flush=True)
