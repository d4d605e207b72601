"""dVAE encoder throughput (images/s, TFLOP/s) on synthetic 112x112 inputs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
import torch
from exploremultimodal_amd.dvae import Encoder
from exploremultimodal_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
enc = Encoder().cuda()
x = (0.8 * torch.rand(B, 3, 112, 112) + 0.1).cuda()
with torch.no_grad():
    for _ in range(2): ids = enc.codebook_indices(x)
    torch.cuda.synchronize()
    hip.profile_start()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n): ids = enc.codebook_indices(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
prof = hip.profile_stop()
print(f'B={B}: {dt*1e3:.2f} ms per batch, {B/dt:.0f} images/s, {52.119e9*B/dt/1e12:.1f} TFLOP/s')
for k, (s, f, c) in sorted(prof.items(), key=lambda kv: -kv[1][0]):
    print(f'  {k:40s} {s/n*1e3:7.2f} ms/batch  {f/s/1e12:7.1f} TF/s  {c//n} launches')
