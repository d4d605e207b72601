"""HBM write / copy ceilings as seen by simple streaming kernels (for judging the GEMM epilogue bursts)."""
import torch
dev = 'cuda'


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


for mb in (26, 51, 103, 205, 410):
    n = mb * (1 << 20) // 2
    x = torch.empty(n, dtype=torch.bfloat16, device=dev)
    y = torch.empty(n, dtype=torch.bfloat16, device=dev)
    tf = timeit(lambda: x.fill_(1.0))
    tc = timeit(lambda: y.copy_(x))
    print(f'{mb:4d} MB: fill {tf*1e6:7.1f} us = {n*2/tf/1e12:5.2f} TB/s write | copy {tc*1e6:7.1f} us = {2*n*2/tc/1e12:5.2f} TB/s r+w', flush=True)
