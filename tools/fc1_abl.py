import os, sys
sys.path.insert(0, '/root/repo')
import torch
from exploremultimodal_amd import hip
M, N, K, dev = 16704, 3072, 768, 'cuda'
def timeit(fn, n=200):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
bias = torch.randn(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16); out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for M_ in (16704, 16384):
    print('M =', M_)
    print('  bias+gelu, dropout 0.1 : %.1f us' % timeit(lambda: hip.gemm_nt(hip.EPI_BIAS_GELU, A, B, M_, N, K, out, bias=bias, out2=out2, drop=hip.drop_params(0.1, True), seed=3)))
    print('  bias+gelu, no dropout  : %.1f us' % timeit(lambda: hip.gemm_nt(hip.EPI_BIAS_GELU, A, B, M_, N, K, out, bias=bias, out2=out2)))
    print('  bias only (one output) : %.1f us' % timeit(lambda: hip.gemm_nt(hip.EPI_BIAS, A, B, M_, N, K, out, bias=bias, tile=4)))
    print('  bias only, 256x256     : %.1f us' % timeit(lambda: hip.gemm_nt(hip.EPI_BIAS, A, B, M_, N, K, out, bias=bias, tile=3)))
