"""Per-tile fixed cost of the NT kernels: time of C = A.B^T over K at fixed M x N (bias epilogue), so that
t(K) = rounds * (overhead + K/64 * per_ktile).  usage: python tools/overhead_probe.py [M] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
dev = 'cuda'


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


bias = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for tile in [int(t) for t in os.environ.get('PROBE_TILES', '3,0').split(',')]:
    for K in (64, 128, 256, 512, 768, 1536, 3072):
        A = torch.randn(M, K, device=dev).bfloat16()
        B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        t = timeit(lambda: hip.gemm_nt(hip.EPI_BIAS, A, B, M, N, K, out, bias=bias, tile=tile))
        bm = 16 * (tile - 300) if tile >= 309 else (128 if tile == 0 else 256)       # 309..320: the 16x16x32 kernels
        bn = 128 if tile == 0 else 256
        tiles = ((M + bm - 1) // bm) * ((N + bn - 1) // bn)
        slots = 512 if tile == 0 else 256
        print(f'tile={tile} M={M} N={N} K={K:5d}: {t:7.1f} us  tiles {tiles} = {tiles / slots:.2f} rounds  '
              f'{t / (tiles / slots):6.1f} us per round  {2 * M * N * K / t / 1e6:7.1f} TF/s', flush=True)
