"""Run the other BASELINE configs at full size (timing + sanity), on the GPU box."""
import sys, os, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import henbun_amd as hb
from henbun_amd.models import SVGP, Amortised, ExpertsGPR, svgp_data
tf = hb.tf

def timed(opt, n, steps=100, warm=10):
    opt.optimize(maxiter=warm, minibatch_size=n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.optimize(maxiter=steps, minibatch_size=n)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return dt / steps

def cfg3():
    np.random.seed(0)
    M, n, N = 1024, 16384, 200000
    X, Y, Z = svgp_data(N, M, 0, domain=512.0)
    m = SVGP(X=X, Y=Y, Z=Z, q_shape="fullrank", dtype="float32")
    rng = np.random.RandomState(0)
    m.u.q_sqrt = 0.1 * np.eye(M) + 0.01 * np.tril(rng.randn(M, M))
    opt = m.ELBO(); opt.compile(optimizer=tf.train.AdamOptimizer(1e-3))
    e0 = opt.run(minibatch_size=n)
    t = timed(opt, n, 50, 5)
    e1 = opt.run(minibatch_size=n)
    F = 3.0 * M * M * n + 3.0 * M ** 3 + 4.0 * M * n
    print("cfg3 fullrank M=1024 n=16384: %.2f ms/step (%.0f steps/s, %.1f TFLOP/s algorithmic), ELBO %.4g -> %.4g" % (t * 1e3, 1 / t, F / t * 1e-12, e0, e1), flush=True)
    prof = opt._plans[[k for k in opt._plans if k[0] == "opt"][0]].profile(iters=5)
    print({k: round(v[0], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:10]}, flush=True)

def cfg4():
    np.random.seed(0)
    rng = np.random.RandomState(0)
    N, Din, H, L, n = 400000, 64, 256, 16, 32768
    Z0 = rng.randn(N, L).astype(np.float32)
    Y = np.tanh(Z0 @ (rng.randn(L, Din).astype(np.float32) / np.sqrt(L))) + 0.1 * rng.randn(N, Din).astype(np.float32)
    m = Amortised(Y=Y, L=L, H=H, dtype="float32")
    opt = m.ELBO(); opt.compile(optimizer=tf.train.AdamOptimizer(1e-3), dp_reduce="sum")
    e0 = np.mean([opt.run(minibatch_size=n, training=False) for _ in range(3)])
    t = timed(opt, n, 100, 10)
    e1 = np.mean([opt.run(minibatch_size=n, training=False) for _ in range(3)])
    F = 3 * 2.0 * n * (Din * H + H * 2 * L + L * Din)
    print("cfg4 amortised n=32768 [64,256,32]: %.3f ms/step (%.0f steps/s, %.1f M samples/s, %.1f TFLOP/s), held-out ELBO %.4g -> %.4g" % (t * 1e3, 1 / t, n / t * 1e-6, F / t * 1e-12, e0, e1), flush=True)
    prof = opt._plans[[k for k in opt._plans if k[0] == "opt"][0]].profile(iters=5)
    print({k: round(v[0], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]}, flush=True)

def cfg5():
    np.random.seed(0)
    rng = np.random.RandomState(0)
    E, M, n, N = 4, 512, 65536, 1000000
    X, Y, Z = svgp_data(N, M, 0, domain=512.0)
    ells = list(np.linspace(0.6, 1.2, E)) + list(np.linspace(0.8, 1.4, E))
    m = ExpertsGPR(X=X, Y=Y, Z=Z, ells=ells, dtype="float32")
    opt = m.ELBO(); opt.compile(optimizer=tf.train.AdamOptimizer(1e-3))
    e0 = opt.run(minibatch_size=n)
    print("cfg5 initial ELBO", e0, flush=True)
    t = timed(opt, n, 50, 5)
    e1 = opt.run(minibatch_size=n)
    F = 2 * E * (3 * 2.0 * M * M * n)
    print("cfg5 %d experts + %d gates x M=512 n=65536: %.2f ms/step (%.0f steps/s, %.1f M samples/s, %.1f TFLOP/s algorithmic), ELBO %.4g -> %.4g" % (E, E, t * 1e3, 1 / t, n / t * 1e-6, F / t * 1e-12, e0, e1), flush=True)
    prof = opt._plans[[k for k in opt._plans if k[0] == "opt"][0]].profile(iters=5)
    print({k: round(v[0], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]}, flush=True)

which = sys.argv[1:] or ["cfg4", "cfg3", "cfg5"]
for f in [globals()[w] for w in which]:
    try:
        f()
    except Exception:
        traceback.print_exc()
