"""In-kernel segment timing of the 16x16x32 NT kernel (diagnostic build, tiles 908 / 909 = schedule 0 / 1 with s_memtime
stamps): cycles per segment kind (read k-half 0 | MFMA | read k-half 1 | MFMA, each up to its closing barrier) per K-tile
and the in-kernel clock.  usage: python tools/nt16_probe.py [N] [K]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from exploremultimodal_amd import hip

M = 16704
N = int(sys.argv[1]) if len(sys.argv) > 1 else 768
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
dev = 'cuda'
A = torch.randn(M, K, device=dev).bfloat16()
B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
bias = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
tiles = ((M + 255) // 256) * ((N + 255) // 256)
for tile in (908, 909):
    buf = torch.zeros(tiles * 8 * 8, dtype=torch.int64, device=dev)
    for _ in range(200):        # sustained load before the measured launch (clock)
        hip.gemm_nt(hip.EPI_BIAS, A, B, M, N, K, out, bias=bias, tile=tile - 800)
    hip.gemm_nt(hip.EPI_BIAS, A, B, M, N, K, out, bias=bias, tile=tile, colpart=buf.view(torch.float32))
    torch.cuda.synchronize()
    r = buf.cpu().numpy().reshape(tiles, 8, 8)
    r = r[r[:, 0, 6] > 0]
    nk = r[0, 0, 6]
    for grp, name in ((slice(0, 4), 'wm=0'), (slice(4, 8), 'wm=1')):
        seg = np.median(r[:, grp, 0:4].reshape(-1, 4), axis=0) / nk
        tot = np.median(r[:, grp, 4]) / nk
        clk = np.median(r[:, grp, 4] / np.maximum(r[:, grp, 5], 1)) * 100e6 / 1e9
        pro = np.median(r[:, grp, 7])
        print(f'      prologue (kernel start -> K loop) {pro:7.0f} cycles = {pro / clk / 1e3:5.2f} us; K loop {np.median(r[:, grp, 4]):8.0f} cycles = '
              f'{np.median(r[:, grp, 4]) / clk / 1e3:6.2f} us')
        print(f'sched {tile - 908} {name}: cycles per K-tile: read0 {seg[0]:6.0f}  mfma0 {seg[1]:6.0f}  read1 {seg[2]:6.0f}  mfma1 {seg[3]:6.0f}  '
              f'total {tot:6.0f}  ({nk} K-tiles)  in-kernel clock {clk:.2f} GHz', flush=True)
