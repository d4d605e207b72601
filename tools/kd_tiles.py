"""The K = d GEMMs of the step (fc1, qkv, dgrad_fc2) under each tile the library offers, back to back after a long
warm-up (single-kernel timings swing with the clock state: take differences, not absolutes).
python tools/kd_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip
M, dev = 16704, 'cuda'
shapes = {'qkv': (hip.EPI_BIAS, 2304, 768), 'fc1': (hip.EPI_BIAS_GELU, 3072, 768), 'dgrad_fc2': (hip.EPI_DGELU, 3072, 768)}


def timeit(fn, n=200):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, (epi, N, K) in shapes.items():
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(bias=bias)
    if epi == hip.EPI_BIAS_GELU: kw.update(out2=out2, drop=hip.drop_params(0.1, True), seed=3)
    if epi == hip.EPI_DGELU: kw.update(aux=torch.randn(M, N, device=dev).bfloat16(), drop=hip.drop_params(0.1, True), seed=3)
    for tile in (-1, 0, 3, 4, 8):
        try:
            us = timeit(lambda: hip.gemm_nt(epi, A, B, M, N, K, out, tile=tile, **kw))
            print(f'{name:10s} tile {tile:2d}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TFLOP/s', flush=True)
        except Exception as e:
            print(f'{name:10s} tile {tile:2d}: {str(e)[:80]}')
