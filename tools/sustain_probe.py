"""Does a GEMM's rate depend on how long the chip has been under MFMA load (boost vs sustained clock)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip
dev = 'cuda'
M = 16704


def timeit(fn, reps):
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


A = torch.randn(M, 3072, device=dev).bfloat16()
B = (torch.randn(768, 3072, device=dev) * 0.05).bfloat16()
out = torch.empty(M, 768, device=dev, dtype=torch.bfloat16)
fn = lambda: hip.gemm_nt(hip.EPI_BIAS, A, B, M, 768, 3072, out, tile=3)
fl = 2 * M * 768 * 3072
for _ in range(3):
    fn()
import time
for reps in (10, 10, 50, 200, 1000, 10, 10):
    t = timeit(fn, reps)
    print(f'nt dgrad_fc1 tile3 reps={reps:5d}: {t*1e6:7.1f} us {fl/t/1e12:7.1f} TF/s', flush=True)
time.sleep(1.0)
for reps in (10, 1000):
    t = timeit(fn, reps)
    print(f'after 1 s idle: reps={reps:5d}: {t*1e6:7.1f} us {fl/t/1e12:7.1f} TF/s', flush=True)
A2 = torch.randn(M, 3072, device=dev).bfloat16(); B2 = torch.randn(M, 768, device=dev).bfloat16()
C = torch.zeros(3072, 768, device=dev)
fn2 = lambda: hip.gemm_tn(A2, B2, C, M, 3072, 768)
for reps in (10, 10, 200, 1000):
    t = timeit(fn2, reps)
    print(f'tn wgrad_fc1 (split+slab) reps={reps:5d}: {t*1e6:7.1f} us {fl/t/1e12:7.1f} TF/s', flush=True)
