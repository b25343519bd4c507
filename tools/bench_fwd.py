"""Time the training-path forward contraction hb_sgp_fwd (fragment-major W in, fragment-major A + column partials out,
finishing pass included) at cfg-2 and cfg-5 sizes.  HIP events, 50 launches.  Extra argv entries NAME=VALUE are
hb_debug_set switches (integers) to compare against the default (each timed in the same process)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from henbun_amd import hip_ops as H


def t(fn, iters=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


switches = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]
only = [a for a in sys.argv[1:] if "=" not in a]   # e.g. `cfg5`: that size only
rng = np.random.RandomState(0)
for name, E, M, n in (("cfg2", 1, 512, 8192), ("cfg5", 8, 512, 65536)):
    if only and name not in only:
        continue
    z = np.broadcast_to(np.linspace(0, M / 2.0, M)[None, :, None], (E, M, 1)).copy()
    ell = torch.ones(E, 1, device="cuda")
    zz = torch.as_tensor(z, dtype=torch.float32).cuda()
    x = torch.as_tensor(rng.uniform(0, M / 2.0, (n, 1)), dtype=torch.float32).cuda()
    K = H.gram_fwd(zz, zz, ell, diag_add=1e-4).reshape(E, M, M)
    frag = torch.zeros(5 * E * M * M, dtype=torch.float32, device="cuda")
    L, W, info = H.cholesky_inverse(K, frag=frag, frag_bf16x3=True)
    u = torch.randn(E, 1, M, device="cuda"); eps = torch.randn(E, n, device="cuda")
    fbar = torch.randn(E, 1, n, device="cuda")
    if E == 1:
        zz, ell, W, u, eps, fbar = zz[0], ell[0], W.reshape(M, M), u[0], eps[0], fbar[0]
    af = torch.zeros(H.sgp_frag_elems(E, n, M), device="cuda")
    kf = torch.zeros_like(af)
    fl = E * float(M) * M * n
    res = H.sgp_fwd(x, zz, ell, W, u, eps_in=eps, wfrag=frag, a_frag=af)
    v = res[2]
    out = H.sgp_bwd(x, zz, ell, W, u, eps, None, v, fbar, wfrag=frag, a_frag=af, kbar_frag=kf)
    outs = (None,) + tuple(out[:4]) + (None,)
    for lab, env in [("default", None)] + [("%s=%s" % (k, val), (k, val)) for k, val in switches]:
        if env: H.debug_set(env[0], int(env[1]))
        us = t(lambda: H.sgp_fwd(x, zz, ell, W, u, eps_in=eps, wfrag=frag, a_frag=af))
        ub = t(lambda: H.sgp_bwd(x, zz, ell, W, u, eps, None, v, fbar, wfrag=frag, a_frag=af, kbar_frag=kf, out=outs))
        if env: H.debug_clear()
        print("%s  %-22s fwd %8.1f us  %6.1f TFLOP/s   bwd %8.1f us  %6.1f TFLOP/s (2 M^2 n)" % (name, lab, us, fl / us * 1e-6, ub, 2 * fl / ub * 1e-6), flush=True)
