import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from henbun_amd import hip_ops as H
dt = torch.float32
n, M = 8192, 512
rng = np.random.RandomState(0)
x = torch.as_tensor(rng.uniform(0, M / 2, (n, 1)), dtype=dt).cuda()
z = torch.as_tensor(np.linspace(0, M / 2, M)[:, None], dtype=dt).cuda()
ell = torch.ones(1, dtype=dt).cuda(); u = torch.as_tensor(rng.randn(1, M), dtype=dt).cuda()
eps = torch.as_tensor(rng.randn(n), dtype=dt).cuda(); fbar = torch.as_tensor(rng.randn(1, n), dtype=dt).cuda()
K = H.gram_fwd(z, z, ell); Kj = H.matutil(K, H.MATUTIL_ADD_EYE, alpha=1e-3)
L, info = H.cholesky(Kj); W = H.trinv(L)
f, A, v, e = H.sgp_fwd(x, z, ell, W, u, eps_in=eps)
g = H.CapturedGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g.begin()
    for _ in range(20):
        H.sgp_fwd(x, z, ell, W, u, eps_in=eps, out=(f, A, v, e))
    g.end()
    for rep in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(50): g.launch()
        e1.record(s); s.synchronize()
        print("graph of 20 sgp_fwd: %.1f us per sgp_fwd (rep %d)" % (e0.elapsed_time(e1) * 1e3 / 1000, rep), flush=True)
