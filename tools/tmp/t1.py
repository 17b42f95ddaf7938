import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
from mmnas_amd import ops
import mmnas_amd._lib as L
K, m, N = [int(a) for a in sys.argv[1:4]]
x = torch.randn(K, m, device='cuda'); y = torch.randn(K, N, device='cuda'); c = torch.zeros(m, N, device='cuda')
ops.gemm(L.GEMM_TN, [dict(M=m, A=[x], B=[y], C=c)], N, K, m, N, N, accumulate=True)
torch.cuda.synchronize()
ref = x.double().t() @ y.double()
print(K, m, N, float((c.double() - ref).abs().max() / ref.abs().max()), flush=True)
