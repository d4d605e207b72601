"""Run ONE gemm shape a few times (for rocprofv3 --pmc passes). usage: gemm_one.py NAME [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from exploremultimodal_amd import hip
name = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
M, dev = 16704, 'cuda'
shapes = {'qkv': (hip.EPI_BIAS, 2304, 768), 'fc1': (hip.EPI_BIAS_GELU, 3072, 768), 'fc2': (hip.EPI_RESID, 768, 3072),
          'dgrad_fc1': (hip.EPI_BIAS, 768, 3072), 'dgrad_fc2': (hip.EPI_DGELU, 3072, 768), 'proj': (hip.EPI_RESID, 768, 768)}
if name.startswith('tn_'):
    N1, N2 = {'tn_fc1': (3072, 768), 'tn_fc2': (768, 3072), 'tn_qkv': (2304, 768), 'tn_proj': (768, 768)}[name]
    A = torch.randn(M, N1, device=dev).bfloat16(); B = torch.randn(M, N2, device=dev).bfloat16()
    C = torch.zeros(N1, N2, device=dev)
    for _ in range(reps): hip.gemm_tn(A, B, C, M, N1, N2)
else:
    epi, N, K = shapes[name]
    A = torch.randn(M, K, device=dev).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if epi == hip.EPI_RESID else torch.bfloat16)
    out2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    kw = dict(bias=bias)
    if epi == hip.EPI_RESID: kw.update(out2=out2, resid=torch.randn(M, N, device=dev), gamma=torch.ones(N, device=dev))
    if epi == hip.EPI_BIAS_GELU: kw.update(out2=out2)
    if epi == hip.EPI_DGELU: kw.update(aux=torch.randn(M, N, device=dev).bfloat16())
    for _ in range(reps): hip.gemm_nt(epi, A, B, M, N, K, out, **kw)
torch.cuda.synchronize()
