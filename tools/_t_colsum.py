import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from henbun_amd import hip_ops as H
rng=np.random.RandomState(0)
for dt in (torch.float32, torch.float64):
    for K,M,N in ((32768,256,32),(32768,64,256),(32768,16,64),(1000,20,12),(4096,64,64)):
        A=torch.as_tensor(rng.randn(K,M),dtype=dt).cuda(); B=torch.as_tensor(rng.randn(K,N),dtype=dt).cuda()
        C,cs=H.matmul_colsum(A,B)
        torch.cuda.synchronize()
        Cr=(A.double().T@B.double()); csr=B.double().sum(0)
        print(dt,K,M,N,"C err %.2e colsum err %.2e"%(float((C.double()-Cr).abs().max()/Cr.abs().max()), float((cs.double()-csr).abs().max()/csr.abs().max())))
