"""Run each hot kernel a few times at cfg-2 sizes (for rocprofv3 --kernel-trace --stats)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from henbun_amd import hip_ops as H

def main():
    dt = torch.float32 if "--f64" not in sys.argv else torch.float64
    n, M = (8192, 512)
    if "--big" in sys.argv: n, M = 16384, 1024
    rng = np.random.RandomState(0)
    x = torch.as_tensor(rng.uniform(0, M / 2, (n, 1)), dtype=dt).cuda()
    z = torch.as_tensor(np.linspace(0, M / 2, M)[:, None], dtype=dt).cuda()
    ell = torch.ones(1, dtype=dt).cuda()
    u = torch.as_tensor(rng.randn(1, M), dtype=dt).cuda()
    eps = torch.as_tensor(rng.randn(n), dtype=dt).cuda()
    fbar = torch.as_tensor(rng.randn(1, n), dtype=dt).cuda()
    K = H.gram_fwd(z, z, ell)
    Kj = H.matutil(K, H.MATUTIL_ADD_EYE, alpha=1e-3)
    L, info = H.cholesky(Kj)
    W = H.trinv(L)
    f, A, v, e = H.sgp_fwd(x, z, ell, W, u, eps_in=eps)
    outs = (torch.empty_like(A), torch.empty_like(W), torch.empty_like(u), torch.empty_like(z), torch.empty_like(ell), None)
    for _ in range(10):
        H.gram_fwd(z, z, ell, out=K)
        H.cholesky(Kj, out=L, info=info)
        H.trinv(L, out=W)
        H.sgp_fwd(x, z, ell, W, u, eps_in=eps, out=(f, A, v, e))
        H.sgp_bwd(x, z, ell, W, u, eps, A, v, fbar, out=outs)
    torch.cuda.synchronize()
    print("done", info.item())

if __name__ == "__main__":
    main()
