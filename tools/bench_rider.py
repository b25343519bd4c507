"""Cholesky + inverse chain with and without the forward riders, against the separate strip kernel (cfg-2 size)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from henbun_amd import hip_ops as H
M, n = 512, 8192
z = torch.as_tensor(np.linspace(0, M / 2.0, M)[:, None], dtype=torch.float32).cuda()
ell = torch.ones(1, device="cuda")
x = torch.as_tensor(np.random.RandomState(0).uniform(0, M / 2.0, (n, 1)), dtype=torch.float32).cuda()
u = torch.randn(1, M, device="cuda"); eps = torch.randn(n, device="cuda")
K = H.gram_fwd(z, z, ell, diag_add=1e-4)
frag = torch.zeros(2 * M * M, device="cuda")
af = torch.zeros(H.sgp_frag_elems(1, n, M), device="cuda")
sws = torch.zeros(H.sgp_rider_ws_elems(n, M, 1), device="cuda")
L = torch.empty_like(K); W = torch.empty_like(K); info = torch.zeros(1, dtype=torch.int32, device="cuda")
fo = (torch.empty(1, n, device="cuda"), torch.empty(1, device="cuda"), torch.empty(n, device="cuda"), torch.empty(n, device="cuda"))
def t(fn, reps=200):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
a = t(lambda: H.cholesky_inverse(K, out=L, inv=W, info=info, frag=frag))
b = t(lambda: H.sgp_fwd(x, z, ell, W, u, eps_in=eps, wfrag=frag, a_frag=af, skip_a=True, out=(fo[0], None, fo[2], fo[3])))
c = t(lambda: H.cholesky_inverse_sgp(K, x, z, ell, u, af, sws, out=L, inv=W, info=info, frag=frag))
d = t(lambda: H.sgp_finish(sws, n, M, 1, 1, eps_in=eps, out=(fo[0], fo[2], fo[3])))
print("chain alone %.1f us + strip forward & finish %.1f us = %.1f us;  chain with riders %.1f us + finish %.1f us = %.1f us" % (a, b, a + b, c, d, c + d))
