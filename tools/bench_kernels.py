"""Micro-benchmark of the hot kernels at the cfg-2 sizes (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from henbun_amd import hip_ops as H

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us

def main():
    print(H.device_info())
    for dt in (torch.float32, torch.float64):
        for (n, M) in ((8192, 512), (16384, 1024), (1024, 512)):
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
            t_gram = timeit(lambda: H.gram_fwd(z, z, ell))
            t_chol = timeit(lambda: H.cholesky(Kj, out=L, info=info))
            t_trinv = timeit(lambda: H.trinv(L, out=W))
            t_fwd = timeit(lambda: H.sgp_fwd(x, z, ell, W, u, eps_in=eps, out=(f, A, v, e)))
            outs = (torch.empty_like(A), torch.empty_like(W), torch.empty_like(u), torch.empty_like(z), torch.empty_like(ell), None)
            t_bwd = timeit(lambda: H.sgp_bwd(x, z, ell, W, u, eps, A, v, fbar, out=outs))
            t_mm = timeit(lambda: H.matmul(W, W, transA=True))
            fl = M * M * n
            print("%s n=%d M=%d info=%d | gram %.1f chol %.1f trinv %.1f sgp_fwd %.1f (%.1f TF) sgp_bwd %.1f (%.1f TF) mm(MxMxM) %.1f (%.1f TF) us" % (
                str(dt)[6:], n, M, info.item(), t_gram, t_chol, t_trinv, t_fwd, fl / t_fwd * 1e-6, t_bwd, 2 * fl / t_bwd * 1e-6,
                t_mm, 2 * M ** 3 / t_mm * 1e-6), flush=True)

if __name__ == "__main__":
    main()
