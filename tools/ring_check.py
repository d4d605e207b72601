import os, sys, torch
sys.path.insert(0, os.getcwd())
from exploremultimodal_amd import hip
dev='cuda'
torch.manual_seed(0)
for (M,N,K) in ((1000,384,768),(4099,768,3072),(16704,3072,768)):
    A=torch.randn(M,K,device=dev).bfloat16(); B=(torch.randn(N,K,device=dev)*0.05).bfloat16(); bias=torch.randn(N,device=dev)
    ref=torch.empty(M,N,device=dev,dtype=torch.bfloat16); hip.gemm_nt(hip.EPI_BIAS,A,B,M,N,K,ref,bias=bias,tile=3)
    for t in [int(x) for x in os.environ.get("RING_TILES","8").split(",")]:
        out=torch.full((M,N),float('nan'),device=dev,dtype=torch.bfloat16); hip.gemm_nt(hip.EPI_BIAS,A,B,M,N,K,out,bias=bias,tile=t)
        torch.cuda.synchronize()
        print(M,N,K,'tile',t,'equal' if torch.equal(out,ref) else ('maxdiff %g' % (out.float()-ref.float()).abs().max().item()))
