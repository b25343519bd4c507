"""Train 25 steps of the cfg-2 model (the full captured step: index draw + gather + forward + backward + Adam), restore
the saved optimiser state, train the same 25 steps again and compare the parameters bitwise -- over and over for
<seconds>.  Run several copies at once to put the GPU under contention (see tools/replay_determinism.py).
  python tools/train_determinism.py <seconds> <tag>"""
import os, sys, time, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import henbun_amd as hb
from henbun_amd.models import SVGP, svgp_data
tf = hb.tf
np.random.seed(1234)
M, n = 512, 8192
X, Y, Z = svgp_data(200000, M, seed=0, domain=0.5 * M)
m = SVGP(X=X, Y=Y, Z=Z, dtype="float32", seed=0)
opt = m.ELBO()
opt.compile(optimizer=tf.train.AdamOptimizer(1e-3))
opt.optimize(maxiter=3, minibatch_size=n)   # plans built, graph captured
path = os.path.join(tempfile.mkdtemp(), "state.npz")
opt.save_state(path)
sess = m._session
def run():
    opt.restore_state(path)
    opt.optimize(maxiter=25, minibatch_size=n)
    torch.cuda.synchronize()
    return sess.theta.detach().clone()
t_ref = run()
bad, it, t0 = 0, 0, time.time()
while time.time() - t0 < float(sys.argv[1]):
    t = run()
    if not torch.equal(t, t_ref):
        bad += 1
        if bad <= 3:
            print("run %d differs: max |d theta| %.4g, finite %s" % (it, (t - t_ref).abs().max().item(), bool(torch.isfinite(t).all())), flush=True)
    it += 1
print("%s: %d x 25 training steps, %d differ" % (sys.argv[2], it, bad), flush=True)
