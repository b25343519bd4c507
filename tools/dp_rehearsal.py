"""Rehearsal of the data-parallel step on ONE GPU: N ranks (gloo) share cuda:0.  Checks that the ranks' parameters
stay identical and that the 2-rank trajectory equals a 1-process run fed the same two minibatches' mean gradient.
  HENBUN_ONE_DEVICE=1 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dp_rehearsal.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HENBUN_ONE_DEVICE", "1")
import numpy as np, torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
import henbun_amd as hb
from henbun_amd.models import SVGP, svgp_data
tf = hb.tf
np.random.seed(1234)
M, n = int(os.environ.get('DP_M', 128)), int(os.environ.get('DP_n', 1024))
Nl = int(os.environ.get('DP_N', 20000))
X, Y, Z = svgp_data(Nl, M, seed=rank, domain=0.5 * M)
m = SVGP(X=X, Y=Y, Z=Z, dtype="float32", seed=0)
m.N = Nl * world
opt = m.ELBO()
opt.compile(optimizer=tf.train.AdamOptimizer(float(os.environ.get('DP_LR', 1e-2))))
sess = m._session
if os.environ.get("DP_TRACE"):
    for it in range(12):
        try:
            opt.optimize(maxiter=1, minibatch_size=n)
        except Exception as ex:
            print("rank %d step %d: %s" % (rank, it, ex), flush=True)
        g = opt.last_plan.gflat if opt.last_plan is not None else None
        th = sess.theta.detach()
        torch.cuda.synchronize()
        print("rank %d step %d: theta finite %s (nan count %d), gflat finite %s |g| %.4g" % (
            rank, it, bool(torch.isfinite(th).all()), int((~torch.isfinite(th)).sum()),
            None if g is None else bool(torch.isfinite(g).all()), float("nan") if g is None else float(g.norm())), flush=True)
for it in range(6):
    opt.optimize(maxiter=10, minibatch_size=n)
    th = sess.theta.detach().clone()
    ref = th.clone()
    dist.broadcast(ref, 0)
    e = opt.run(minibatch_size=n)
    print("rank %d after %3d steps: ELBO %.5g  |theta| %.6f  max|theta - theta_rank0| %.3g  finite %s" % (
        rank, 10 * (it + 1), e, float(th.norm()), float((th - ref).abs().max()), bool(torch.isfinite(th).all())), flush=True)
dist.barrier()
dist.destroy_process_group()
