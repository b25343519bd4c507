"""Is the cfg2 step CPU-enqueue-bound or GPU-bound?  Times graph replays three ways."""
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
opt.optimize(maxiter=20, minibatch_size=8192)
plan = opt.last_plan
torch.cuda.synchronize()
K = 300
# 1. enqueue-only time vs total
t0 = time.perf_counter()
for _ in range(K):
    plan.run()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("plan.run x%d: enqueue %.1f us/step, total %.1f us/step" % (K, (t1 - t0) / K * 1e6, (t2 - t0) / K * 1e6))
# 2. GPU time of single replays (events on the plan's stream)
with plan._on_stream():
    st = torch.cuda.current_stream()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.synchronize()
        e0.record(st); plan._graph.launch(); e1.record(st)
        st.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("single replay GPU time: min %.1f  median %.1f us" % (min(ts), sorted(ts)[len(ts) // 2]))
    # 3. back-to-back replays, events around the whole batch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(K):
        plan._graph.launch()
    e1.record(st)
    st.synchronize()
    print("back-to-back replays: %.1f us/step (GPU events)" % (e0.elapsed_time(e1) * 1e3 / K))
print("nodes per replay:", len(plan.steps), "plan steps")
