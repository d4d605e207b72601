"""Is the weight-gradient kernel's rate set by where its operands come from (Infinity Cache vs HBM)?
Times one wgrad problem (fc1: 3072x768, M tokens) through the old split+slab path and through
vlmo_gemm_tn_multi, re-using ONE operand set every rep (Infinity-Cache resident, 128 MB) or cycling through
NSETS sets (HBM streaming)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip
dev = 'cuda'
M, N1, N2 = 16704, 3072, 768
NSETS = 6
sets = [(torch.randn(M, N1, device=dev).bfloat16(), torch.randn(M, N2, device=dev).bfloat16()) for _ in range(NSETS)]
C = torch.zeros(N1, N2, device=dev)


def timeit(fn, reps=12):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(reps):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


fl = 2 * M * N1 * N2
for name, nsets in (('one operand set (Infinity-Cache resident)', 1), (f'{NSETS} operand sets in turn (HBM)', NSETS)):
    t = timeit(lambda i: hip.gemm_tn(sets[i % nsets][0], sets[i % nsets][1], C, M, N1, N2))
    print(f'old split+slab  {name}: {t*1e6:7.1f} us {fl/t/1e12:7.1f} TF/s', flush=True)
    t = timeit(lambda i: hip.gemm_tn_multi([(sets[i % nsets][0], sets[i % nsets][1], C, M, N1, N2, True)]))
    print(f'tn_multi splits={os.environ.get("VLMO_TN_SPLITS","auto")} {name}: {t*1e6:7.1f} us {fl/t/1e12:7.1f} TF/s', flush=True)
