"""In-kernel segment timing of the 256x256 ping-pong weight-gradient kernel (diagnostic build: vlmo_gemm_tn with
splits = 3000 + s): cycles per segment kind per K-tile and the in-kernel clock.  usage: python tools/tn_stamp_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from exploremultimodal_amd import hip

M, dev = 16704, 'cuda'
for name, N1, N2 in (('wgrad_fc1', 3072, 768), ('wgrad_fc2', 768, 3072), ('wgrad_qkv', 2304, 768)):
    A = torch.randn(M, N1, device=dev).bfloat16()
    B = torch.randn(M, N2, device=dev).bfloat16()
    C = torch.zeros(N1, N2, device=dev)
    tiles = ((N1 + 255) // 256) * ((N2 + 255) // 256)
    splits = max(1, 256 // tiles)
    for _ in range(100):
        hip.gemm_tn(A, B, C, M, N1, N2, splits=2000 + splits)
    buf = torch.zeros(tiles * splits * 8 * 8 + 64, dtype=torch.int64, device=dev)
    rc = hip.lib().vlmo_gemm_tn(hip._dt(A), hip._p(A), A.stride(0), hip._p(B), B.stride(0), hip._p(C), C.stride(0), M, N1, N2,
                                1.0, 3000 + splits, hip._p(buf), buf.numel() * 8, hip._stream())
    hip._check(rc, 'vlmo_gemm_tn(probe)')
    torch.cuda.synchronize()
    r = buf[:tiles * splits * 64].cpu().numpy().reshape(-1, 8, 8)
    r = r[r[:, 0, 6] > 0]
    for grp, gname in ((slice(0, 4), 'wm=0'), (slice(4, 8), 'wm=1')):
        nk = r[:, grp, 6].astype(np.float64)
        seg = np.median((r[:, grp, 0:4] / nk[..., None]).reshape(-1, 4), axis=0)
        tot = np.median(r[:, grp, 4] / nk)
        clk = np.median(r[:, grp, 4] / np.maximum(r[:, grp, 5], 1)) * 0.1
        print(f'{name} splits {splits} {gname}: cycles per K-tile: read0 {seg[0]:6.0f}  mfma0 {seg[1]:6.0f}  read1 {seg[2]:6.0f}  mfma1 {seg[3]:6.0f}  '
              f'total {tot:6.0f}  in-kernel clock {clk:.2f} GHz', flush=True)
