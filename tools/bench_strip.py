"""Time the forward M^2 n contraction (hb_sgp_A) in its variants at cfg-2 and cfg-5 sizes:
row-major W (first strip form / tiled), fragment-major fp32 (second strip form), bf16x3.  HIP events, 100 launches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from henbun_amd import hip_ops as H

def t(fn, iters=100):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

rng = np.random.RandomState(0)
for name, E, M, n in (("cfg2", 1, 512, 8192), ("cfg5", 8, 512, 65536)):
    z = np.broadcast_to(np.linspace(0, M / 2.0, M)[None, :, None], (E, M, 1)).copy()
    ell = torch.ones(E, 1, device="cuda")
    zz = torch.as_tensor(z, dtype=torch.float32).cuda()
    x = torch.as_tensor(rng.uniform(0, M / 2.0, (n, 1)), dtype=torch.float32).cuda()
    K = H.gram_fwd(zz, zz, ell, diag_add=1e-4).reshape(E, M, M)
    frag = torch.zeros(5 * E * M * M, dtype=torch.float32, device="cuda")
    L, W, info = H.cholesky_inverse(K, frag=frag, frag_bf16x3=True)
    if E == 1:
        zz, ell, W = zz[0], ell[0], W.reshape(M, M)
    A = torch.empty((E, M, n) if E > 1 else (M, n), dtype=torch.float32, device="cuda")
    fl = E * float(M) * M * n
    for lab, kw in (("row-major W", {}), ("fragment-major fp32", dict(wfrag=frag)), ("bf16x3", dict(wfrag=frag, prec=H.PREC_BF16X3))):
        us = t(lambda: H.sgp_A(x, zz, ell, W, out=A, **kw))
        print("%s  %-20s %8.1f us  %6.1f TFLOP/s (M^2 n flops)" % (name, lab, us, fl / us * 1e-6), flush=True)
