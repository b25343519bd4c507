import sys, os, traceback, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np, torch
import henbun_amd as hb
from models import SVGP, svgp_data
tf = hb.tf
def main():
    for dtype in ("float64", "float32"):
        np.random.seed(0)
        X, Y, Z = svgp_data(20000, 512, 0)
        m = SVGP(X=X, Y=Y, Z=Z, dtype=dtype)
        opt = m.ELBO()
        opt.compile(optimizer=tf.train.AdamOptimizer(1e-3))
        t0 = time.time(); v = opt.run(minibatch_size=8192); print(dtype, "run", v, "build+run %.2fs" % (time.time() - t0), flush=True)
        t0 = time.time(); opt.optimize(maxiter=1, minibatch_size=8192); print("first step %.2fs" % (time.time() - t0), flush=True)
        torch.cuda.synchronize(); t0 = time.time()
        opt.optimize(maxiter=200, minibatch_size=8192)
        torch.cuda.synchronize(); dt = time.time() - t0
        print("%s: %.1f us/step  (%.0f steps/s), captured=%s steps_in_plan=%d" % (dtype, dt / 200 * 1e6, 200 / dt, opt.last_plan._graph is not None, len(opt.last_plan.steps)), flush=True)
        print("after:", opt.run(minibatch_size=8192), flush=True)
try:
    main()
except Exception:
    traceback.print_exc(); sys.exit(1)
