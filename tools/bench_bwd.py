"""Time hb_sgp_bwd (fragment-major path) at cfg-2 size for several slab counts of the Lbar contraction
(HB_LBAR_FORCE_S is read once per process: run one value per process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from henbun_amd import hip_ops as H
rng = np.random.RandomState(0)
E, M, n = 1, 512, 8192
z = torch.as_tensor(np.linspace(0, M / 2.0, M)[:, None], dtype=torch.float32).cuda()
ell = torch.ones(1, device="cuda")
x = torch.as_tensor(rng.uniform(0, M / 2.0, (n, 1)), dtype=torch.float32).cuda()
K = H.gram_fwd(z, z, ell, diag_add=1e-4)
frag = torch.zeros(5 * M * M, dtype=torch.float32, device="cuda")
L, W, info = H.cholesky_inverse(K, frag=frag, frag_bf16x3=True)
u = torch.randn(1, M, device="cuda"); eps = torch.randn(n, device="cuda"); fbar = torch.randn(1, n, device="cuda")
af = torch.zeros(H.sgp_frag_elems(E, n, M), device="cuda")
f, A, v, _ = H.sgp_fwd(x, z, ell, W, u, eps_in=eps, wfrag=frag, a_frag=af)
kf = torch.zeros_like(af)
out = H.sgp_bwd(x, z, ell, W, u, eps, None, v, fbar, wfrag=frag, a_frag=af, kbar_frag=kf)
outs = (None,) + tuple(out[:4]) + (None,)
def run(): H.sgp_bwd(x, z, ell, W, u, eps, None, v, fbar, wfrag=frag, a_frag=af, kbar_frag=kf, out=outs)
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100): run()
e1.record(); torch.cuda.synchronize()
print("HB_LBAR_FORCE_S=%s: sgp_bwd (kbar strip + finish + lbar + lbar finish) %.1f us" % (os.environ.get("HB_LBAR_FORCE_S", "-"), e0.elapsed_time(e1) * 10), flush=True)
