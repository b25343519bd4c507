"""Stability check: many cfg-2 Adam steps in fp32 (graph replay); the ELBO must stay finite and improve."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import henbun_amd as hb
from henbun_amd.models import SVGP, svgp_data
tf = hb.tf
np.random.seed(0)
X, Y, Z = svgp_data(200000, 512, 0, domain=256.0)
m = SVGP(X=X, Y=Y, Z=Z, dtype="float32")
opt = m.ELBO(); opt.compile(optimizer=tf.train.AdamOptimizer(1e-3))
e = [np.mean([opt.run(minibatch_size=8192) for _ in range(5)])]
t0 = time.perf_counter()
for block in range(10):
    opt.optimize(maxiter=2000, minibatch_size=8192)   # raises CholeskyError if a factorisation ever fails
    e.append(np.mean([opt.run(minibatch_size=8192) for _ in range(5)]))
    print("after %5d steps: ELBO %.6g  (theta finite: %s)" % (2000 * (block + 1), e[-1], bool(torch.isfinite(m._session.theta).all())), flush=True)
print("20000 steps in %.1f s; ELBO %.6g -> %.6g" % (time.perf_counter() - t0, e[0], e[-1]))
assert all(np.isfinite(e)) and e[-1] > e[0]
