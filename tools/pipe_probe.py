"""Which streams share a command-processor pipe with the main stream?  A long kernel (many dispatch rounds) runs on the
main stream; a tiny kernel is launched on each candidate stream right behind it.  A candidate whose hardware queue sits
on the main queue's pipe starts only when the long kernel's last workgroup has been dispatched.
Run on the GPU box: python tools/pipe_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

dev = torch.device('cuda', 0)
big = torch.empty(1 << 28, device=dev)            # 1 GiB fp32: fill = ~0.3 ms, many dispatch rounds
small = [torch.empty(256, device=dev) for _ in range(16)]
streams = [torch.cuda.Stream() for _ in range(12)]
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else (0, -1)
streams += [torch.cuda.Stream(priority=1) for _ in range(2)]       # low priority on ROCm: positive
for s in streams:                                  # bind queues in creation order
    with torch.cuda.stream(s):
        small[0].zero_()
torch.cuda.synchronize()


def long_kernel():
    for _ in range(4):
        big.mul_(1.0001)


for rep in range(2):
    for i, s in enumerate(streams):
        torch.cuda.synchronize()
        m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        m0.record()
        long_kernel()
        m1.record()
        with torch.cuda.stream(s):
            small[i].zero_()
            e1.record()
        torch.cuda.synchronize()
        print(f'rep {rep} stream {i:2d}: tiny kernel done {m0.elapsed_time(e1) * 1e3:8.1f} us after the long one started '
              f'(long: {m0.elapsed_time(m1) * 1e3:8.1f} us)', flush=True)
