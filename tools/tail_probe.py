import os, sys
sys.path.insert(0, '/root/repo')
import torch
from exploremultimodal_amd import hip
dev='cuda'
M,N,K=802816,256,64
t=torch.randn(M,K,device=dev).half(); w=(torch.randn(N,K,device=dev)/8).half(); b=torch.randn(N,device=dev)
raw=torch.randn(M,N,device=dev).half(); out=torch.empty(M,N,device=dev,dtype=torch.float16)
def timeit(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,bb=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    bb.record(); torch.cuda.synchronize()
    return a.elapsed_time(bb)/n*1e3
for tile in (-1,0,3,4,8):
    try:
        us=timeit(lambda: hip.gemm_nt(hip.EPI_DUAL,t,w,M,N,K,out,bias=b,resid=raw,beta=1/64,tile=tile))
        print('tile',tile,'%.1f us'%us, '%.2f TB/s'%((M*K*2+2*M*N*2)/us/1e6))
    except Exception as e:
        print('tile',tile,'failed',str(e)[:100])
us=timeit(lambda: torch.add(raw, out, alpha=0.5, out=out))
print('torch add (2 reads 1 write of the big maps) %.1f us'%us)
