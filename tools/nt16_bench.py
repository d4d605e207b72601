"""A/B of the NT tile variants at the VLMo-Base B=64 shapes (random data), interleaved rounds in ONE process:
every (shape, tile) is timed `rounds` times, 25 back-to-back launches each, variants alternating; median and minimum.
usage: python tools/nt16_bench.py [--M 16704] [--rounds 7] [--shapes fc1,qkv,...]"""
import argparse
import os
import statistics
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument('--M', type=int, default=16704)
ap.add_argument('--rounds', type=int, default=7)
ap.add_argument('--reps', type=int, default=25)
ap.add_argument('--shapes', default='')
ap.add_argument('--drop', type=float, default=0.1)
ap.add_argument('--tiles', default='', help='override the candidate list, e.g. 107,207')
ap.add_argument('--d', type=int, default=768, help='model width (N and K of the shapes scale with it)')
args = ap.parse_args()
dev, M = 'cuda', args.M

SHAPES = {                      # name: (epilogue, N, K, candidate tiles)
    'fc1': (hip.EPI_BIAS_GELU, 3072, 768, [318, 317, 4]),
    'qkv': (hip.EPI_BIAS, 2304, 768, [320, 319, 4]),
    'dgrad_fc2': (hip.EPI_DGELU, 3072, 768, [318, 317, 0]),
    'fc2': (hip.EPI_RESID, 768, 3072, [314, 313, 3]),
    'proj': (hip.EPI_RESID, 768, 768, [314, 313, 3]),
    'dgrad_fc1': (hip.EPI_BIAS, 768, 3072, [314, 313, 3]),
    'dgrad_qkv': (hip.EPI_BIAS, 768, 2304, [314, 313, 3]),
    'dgrad_proj': (hip.EPI_BIAS, 768, 768, [314, 313, 3]),
}
if args.d != 768:               # another width (VLMo-Large: --d 1024 --M 8352): same GEMMs, N / K scaled
    SHAPES = {k: (e, N * args.d // 768, K * args.d // 768, t) for k, (e, N, K, t) in SHAPES.items()}
names = [n for n in args.shapes.split(',') if n] or list(SHAPES)
drop = hip.drop_params(args.drop, True)


def run(fn, reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


# warm the clocks
w = torch.randn(8192, 8192, device=dev).bfloat16()
for _ in range(20):
    w @ w
torch.cuda.synchronize()
for name in names:
    epi, N, K, tiles = SHAPES[name]
    if args.tiles:
        tiles = [int(t) for t in args.tiles.split(',')]
    A = torch.randn(M, K, device=dev).bfloat16()
    B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == hip.EPI_RESID else torch.bfloat16)
    out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(bias=bias)
    if epi == hip.EPI_RESID:
        kw.update(out2=out2, resid=torch.randn(M, N, device=dev), gamma=torch.ones(N, device=dev), drop=drop, seed=3)
    if epi == hip.EPI_BIAS_GELU:
        kw.update(out2=out2, drop=drop, seed=3)
    if epi == hip.EPI_DGELU:
        kw.update(aux=torch.randn(M, N, device=dev).bfloat16(), drop=drop, seed=3,
                  colpart=torch.empty((M + 15) // 16 + 2, N, device=dev))
    fns = {t: (lambda t=t: hip.gemm_nt(epi, A, B, M, N, K, out, tile=t, **kw)) for t in tiles}
    for f in fns.values():
        run(f, 5)
    res = {t: [] for t in tiles}
    for _ in range(args.rounds):
        for t in tiles:
            res[t].append(run(fns[t], args.reps))
    base = statistics.median(res[tiles[0]])
    for t in tiles:
        med, mn = statistics.median(res[t]), min(res[t])
        print(f'{name:10s} N={N:5d} K={K:5d} tile={t:3d}: median {med:7.1f} us  min {mn:7.1f} us  {2 * M * N * K / med / 1e6:7.1f} TF/s  '
              f'x{base / med:5.3f} vs tile {tiles[0]}', flush=True)
