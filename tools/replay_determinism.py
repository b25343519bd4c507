"""Replay the cfg-2 forward+backward hipGraph with the RNG streams rewound before every replay and compare the
flat gradient bitwise: `python tools/replay_determinism.py <seconds> <tag> [cfg2|cfg2_f64|cfg3|cfg3_f64|cfg4|cfg5|ragged_f64]`.  Run two or three copies at once on one
GPU to put the kernels under contention (waves of a workgroup then drift apart by thousands of cycles): that is how
the missing-barrier hazard in the Cholesky in-panel step was found (3 % of replays went NaN; 0 after the fix).
On a mismatch the first non-finite plan buffers are listed in topological order."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import henbun_amd as hb
from henbun_amd.models import SVGP, Amortised, ExpertsGPR, svgp_data
tf = hb.tf
np.random.seed(1234)
cfg = sys.argv[3] if len(sys.argv) > 3 else "cfg2"   # cfg2 | cfg2_f64 | cfg3 | cfg3_f64 | cfg4 | cfg5 | ragged_f64
rng0 = np.random.RandomState(0)
if cfg in ("cfg2", "cfg2_f64"):
    M, n = 512, 8192
    X, Y, Z = svgp_data(200000, M, seed=0, domain=0.5 * M)
    m = SVGP(X=X, Y=Y, Z=Z, q_shape="diagonal", dtype="float64" if cfg.endswith("f64") else "float32", seed=0)
elif cfg in ("cfg3", "cfg3_f64"):
    M, n = (1024, 16384) if cfg == "cfg3" else (512, 4096)
    X, Y, Z = svgp_data(100000, M, seed=0, domain=0.5 * M)
    m = SVGP(X=X, Y=Y, Z=Z, q_shape="fullrank", dtype="float64" if cfg.endswith("f64") else "float32", seed=0)
    m.u.q_sqrt = 0.1 * np.eye(M) + 0.01 * np.tril(rng0.randn(M, M))
elif cfg == "ragged_f64":
    M, n = 200, 3001   # ragged sizes: 32-column Cholesky with identity padding, scalar GEMM paths
    X, Y, Z = svgp_data(20000, M, seed=0, domain=0.5 * M)
    m = SVGP(X=X, Y=Y, Z=Z, q_shape="fullrank", dtype="float64", seed=0)
    m.u.q_sqrt = 0.1 * np.eye(M) + 0.01 * np.tril(rng0.randn(M, M))
elif cfg == "cfg4":
    N, Din, H_, L_, n = 200000, 64, 256, 16, 32768
    Z0 = rng0.randn(N, L_).astype(np.float32)
    Yd = np.tanh(Z0 @ (rng0.randn(L_, Din).astype(np.float32) / np.sqrt(L_))) + 0.1 * rng0.randn(N, Din).astype(np.float32)
    m = Amortised(Y=Yd, L=L_, H=H_, dtype="float32")
elif cfg == "cfg5":
    E, M, n = 4, 512, 16384
    X, Y, Z = svgp_data(200000, M, 0, domain=512.0)
    ells = list(np.linspace(0.6, 1.2, E)) + list(np.linspace(0.8, 1.4, E))
    m = ExpertsGPR(X=X, Y=Y, Z=Z, ells=ells, dtype="float32")
else:
    raise SystemExit("unknown config " + cfg)
opt = m.ELBO()
m.initialize()
sess = m._session
# forward + backward only (the plan behind Optimizer.gradients: no Adam, so the state stays fixed between replays)
opt.compile(optimizer=tf.train.AdamOptimizer(1e-3))
opt.gradients(minibatch_size=n)
plan = opt.last_plan
states = {k: r.state.clone() for k, r in sess.rngs.items()}
def replay():
    for k, r in sess.rngs.items():
        r.state.copy_(states[k])
    torch.cuda.synchronize()
    plan.run()
    torch.cuda.synchronize()
    return torch.cat([plan._buf[t].reshape(-1).to(torch.float64) for t in plan.outputs])
g0 = replay()
print("first replay: |g| %.6g finite %s" % (g0.norm().item(), bool(torch.isfinite(g0).all())), flush=True)
bad, it, t0 = 0, 0, time.time()
while time.time() - t0 < float(sys.argv[1]):
    g = replay()
    if not torch.equal(g, g0):
        bad += 1
        if bad <= 5:
            d = (g - g0).abs()
            print("replay %d differs: max abs diff %.4g at %d, finite %s" % (it, d.max().item(), int(d.argmax()), bool(torch.isfinite(g).all())), flush=True)
            from henbun_amd import graph as G
            order = G.topo_order(plan.outputs)
            shown = 0
            for nd in order:
                for o in nd.outputs:
                    b = plan._buf.get(o)
                    if b is None or not b.is_floating_point():
                        continue
                    nf = int((~torch.isfinite(b)).sum())
                    if nf:
                        print("   first non-finite: node %s#%d out %s shape %s: %d bad of %d; inputs: %s" % (
                            nd.op, nd.id, nd.outputs.index(o), tuple(b.shape), nf, b.numel(),
                            [(t.node.op, bool(torch.isfinite(plan._buf[t]).all()) if t in plan._buf and plan._buf[t].is_floating_point() else None) for t in nd.inputs]), flush=True)
                        shown += 1
                if shown >= 3:
                    break
    it += 1
print("%s %s: %d replays, %d differ" % (sys.argv[2], cfg, it, bad), flush=True)
