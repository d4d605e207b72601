"""Micro-benchmark of the GEMM kernels at the VLMo-Base B=64 shapes (random data).
usage: python tools/gemm_bench.py [--tiles 0,1] [--reps 20]"""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip

ap = argparse.ArgumentParser()
ap.add_argument('--tiles', default='0,3')
ap.add_argument('--reps', type=int, default=20)
ap.add_argument('--M', type=int, default=16704)
args = ap.parse_args()
dev = 'cuda'
M = args.M


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


shapes = [('fc1_bias', hip.EPI_BIAS, 3072, 768), ('qkv', hip.EPI_BIAS, 2304, 768), ('proj', hip.EPI_RESID, 768, 768), ('fc1', hip.EPI_BIAS_GELU, 3072, 768),
          ('fc2', hip.EPI_RESID, 768, 3072), ('dgrad_fc2', hip.EPI_DGELU, 3072, 768), ('dgrad_fc1', hip.EPI_BIAS, 768, 3072),
          ('dgrad_qkv', hip.EPI_BIAS, 768, 2304), ('dgrad_proj', hip.EPI_BIAS, 768, 768)]
for name, epi, N, K in shapes:
    A = torch.randn(M, K, device=dev).bfloat16()
    B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == hip.EPI_RESID else torch.bfloat16)
    out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    resid = torch.randn(M, N, device=dev) if epi == hip.EPI_RESID else None
    aux = torch.randn(M, N, device=dev).bfloat16() if epi == hip.EPI_DGELU else None
    gamma = torch.ones(N, device=dev)
    for tile in [int(t) for t in args.tiles.split(',')]:
        kw = dict(bias=bias, tile=tile)
        if epi == hip.EPI_RESID:
            kw.update(out2=out2, resid=resid, gamma=gamma)
        if epi == hip.EPI_BIAS_GELU:
            kw.update(out2=out2)
        if epi == hip.EPI_DGELU:
            kw.update(aux=aux)
        t = timeit(lambda: hip.gemm_nt(epi, A, B, M, N, K, out, **kw), args.reps)
        print(f'nt {name:11s} M={M} N={N:5d} K={K:5d} tile={tile}: {t*1e6:8.1f} us  {2*M*N*K/t/1e12:7.1f} TF/s', flush=True)
for name, N1, N2 in [('wgrad_fc1', 3072, 768), ('wgrad_fc2', 768, 3072), ('wgrad_qkv', 2304, 768), ('wgrad_proj', 768, 768)]:
    A = torch.randn(M, N1, device=dev).bfloat16()
    B = torch.randn(M, N2, device=dev).bfloat16()
    C = torch.zeros(N1, N2, device=dev)
    for splits in (0, 1000):
        for slab in (True,):
            t = timeit(lambda: hip.gemm_tn(A, B, C, M, N1, N2, splits=splits, slab=slab), args.reps)
            print(f'tn {name:11s} M={M} N1={N1:5d} N2={N2:5d} splits={splits:2d} slab={int(slab)}: {t*1e6:8.1f} us  {2*M*N1*N2/t/1e12:7.1f} TF/s', flush=True)
