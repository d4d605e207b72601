"""Host and device cost of one collective at world size 1: torch.distributed (nccl = RCCL) against the library's own
communicator (vlmo_comm_*).  Run on the GPU box: python tools/comm_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from exploremultimodal_amd import hip

os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577')
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
comm = hip.comm_init(hip.comm_unique_id(), 0, 1)
x = torch.randn(7_000_000, device='cuda').bfloat16()
shard = torch.empty_like(x)
side = torch.cuda.Stream()


def timeit(name, fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        host = (time.perf_counter() - t0) / n
        e1.record()
    torch.cuda.synchronize()
    print(f'{name:28s} host {host * 1e6:8.1f} us/call   device {e0.elapsed_time(e1) / n * 1e3:8.1f} us/call', flush=True)


def t_ar():
    with torch.cuda.stream(side):
        dist.all_reduce(x, async_op=True).wait()


def n_ar():
    with torch.cuda.stream(side):
        hip.comm_all_reduce(comm, x)


def t_rs():
    with torch.cuda.stream(side):
        dist.reduce_scatter_tensor(shard, x, async_op=True).wait()


def n_rs():
    with torch.cuda.stream(side):
        hip.comm_reduce_scatter(comm, shard, x)


timeit('torch all_reduce', t_ar)
timeit('native all_reduce', n_ar)
timeit('torch reduce_scatter', t_rs)
timeit('native reduce_scatter', n_rs)
# does the enqueue block the host while the stream still has work in front of it?
a = torch.randn(8192, 8192, device='cuda', dtype=torch.bfloat16)
for name, fn in (('torch', t_ar), ('native', n_ar)):
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for _ in range(40):
            a @ a                      # ~40 x 1.2 ms queued in front
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{name}: enqueue behind a busy stream {1e3 * (t1 - t0):.2f} ms host, drain {1e3 * (t2 - t1):.2f} ms', flush=True)
import ctypes
print('rccl copies mapped:', sorted({l.split()[-1] for l in open('/proc/self/maps') if 'rccl' in l}))
torch.cuda.synchronize()
hip.comm_destroy(comm)
dist.destroy_process_group()
