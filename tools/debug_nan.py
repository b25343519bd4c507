import sys, os, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import henbun_amd as hb
from models import SVGP, svgp_data
tf = hb.tf
np.random.seed(0)
X, Y, Z = svgp_data(20000, 512, 0)
m = SVGP(X=X, Y=Y, Z=Z, dtype="float32")
opt = m.ELBO()
opt.compile(optimizer=tf.train.AdamOptimizer(1e-3))
for it in range(40):
    val, g = opt.gradients(minibatch_size=8192)
    bad = {k: int((~np.isfinite(v)).sum()) for k, v in g.items()}
    mx = {k: float(np.nanmax(np.abs(v))) for k, v in g.items()}
    print(it, "elbo %.1f" % val, "nonfinite", {k: b for k, b in bad.items() if b}, "max|g|", {k.split('.')[-1]: "%.2e" % v for k, v in mx.items()}, flush=True)
    if any(bad.values()) or not np.isfinite(val):
        plan = None
        break
    try:
        opt.optimize(maxiter=5, minibatch_size=8192)
    except Exception as e:
        print("optimize failed:", e); break
