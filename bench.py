#!/usr/bin/env python3
"""ELBO-step benchmark (BASELINE.json metric: ELBO steps/sec and samples/sec at
fixed minibatch, 1/2/4/8 MI355X).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1]): sparse variational GP
regression, 1-D UnitRBF, N = 1e6 synthetic points, M = 512 inducing points,
minibatch 8192 per GPU, diagonal Normal q(u), 'diagonal' residual, fp32.
One step = device-side minibatch index draw + row gather + reparameterised
sample + ELBO forward + full backward + fused Adam (= one iteration of the
reference's Optimizer.optimize, model.py:263-267).  Inputs are resident in HBM.

With N > 1 (launched by torch.distributed.run, one rank per GPU) every rank owns
a 1/N shard of the data and draws its own 8192-row minibatch (weak scaling); the
flat gradient is all-reduced over RCCL once per step.

Rank 0 prints ONE JSON line: the contract fields plus
  "roofline"     -- the M^2 n contraction kernel (A = L^-1 K(z,x)), timed live with
                    HIP events on the plan's stream against the fp32 MFMA peak;
  "cpu_baseline" -- the CPU oracle's restatement of the reference graph
                    (torch-CPU fp32 + autograd + TF-formula Adam) on the host cores,
                    bounded sample (kind "port": TensorFlow is not available offline).
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

N_TOTAL = 1_000_000
M_INDUCING = 512
MINIBATCH = 8192
PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix), spec


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("HB_BENCH_CPU_THREADS", "16"))))


def cpu_baseline(seconds=15.0):
    """Oracle ('port') timed on the host cores: same workload, bounded number of steps."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch

    import henbun_oracle as O
    from models import svgp_data

    threads = usable_cores()
    torch.set_num_threads(threads)
    dt = torch.float32
    rng = np.random.RandomState(0)
    Ns = 100_000  # rows actually materialised for the sample (indices are drawn from it)
    X, Y, Z = svgp_data(Ns, M_INDUCING, 0, domain=0.5 * M_INDUCING)
    Xt, Yt = torch.as_tensor(X, dtype=dt), torch.as_tensor(Y, dtype=dt)
    params = {
        "z": torch.as_tensor(Z, dtype=dt), "ell_raw": torch.as_tensor(O.log1pe_backward_np(np.ones(1)), dtype=dt),
        "q_mu": torch.as_tensor(0.1 * rng.randn(1, M_INDUCING), dtype=dt),
        "q_sqrt": torch.as_tensor(0.1 * rng.randn(M_INDUCING), dtype=dt),
        "k_var_raw": torch.as_tensor(O.log1pe_backward_np(np.ones(1)), dtype=dt),
        "var_raw": torch.as_tensor(O.log1pe_backward_np(np.ones(1)), dtype=dt),
    }
    names = list(params)
    leaves = [params[k].clone() for k in names]
    adam = O.AdamTF(leaves, lr=1e-3)

    def step():
        idx = torch.as_tensor(rng.randint(0, Ns, MINIBATCH))
        u = torch.randn(M_INDUCING, dtype=dt)
        eps = torch.randn(MINIBATCH, dtype=dt)
        # jitter 0.1 (timing only): with inputs up to 256 the reference's |a|^2+|b|^2-2ab^T squared distance
        # carries ~4e-3 absolute error in fp32, so Kmm is not PD at the default 1e-5; the op sequence and
        # flop count do not depend on the jitter value.
        fn = lambda p: O.svgp_elbo(p, Xt[idx], Yt[idx], float(N_TOTAL), u, eps, jitter=0.1)
        _, g = O.grads_of(fn, dict(zip(names, leaves)))
        adam.step([-g[k] for k in names])

    step()  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 400:
            break
    sps = n / el
    return {"value": sps * MINIBATCH, "unit": "samples/s", "steps_per_sec": sps, "cores": threads, "kind": "port",
            "sample": "%d steps of the same cfg2 step (n=%d, M=%d, fp32, torch-CPU restatement of the reference "
                      "op graph + autograd + TF-formula Adam; jitter 0.1 for fp32 PD-safety of the reference distance formula) "
                      "in %.1f s" % (n, MINIBATCH, M_INDUCING, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    args = ap.parse_args()

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal switches (not used by the driver): HENBUN_DIST_BACKEND=gloo + HENBUN_ONE_DEVICE=1 run the N > 1 code
    # path with every rank on cuda:0 of a one-GPU box (RCCL refuses two ranks on one device)
    backend = os.environ.get("HENBUN_DIST_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("HENBUN_ONE_DEVICE") else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)

    import henbun_amd as hb
    from models import SVGP, svgp_data

    tf = hb.tf
    # this rank's shard of the synthetic data set (fp64 master copy, cast on upload)
    np.random.seed(1234)  # identical parameter initialisation on every rank (data below is rank-specific)
    n_local = N_TOTAL // world
    X, Y, Z = svgp_data(n_local, M_INDUCING, seed=rank, domain=0.5 * M_INDUCING)
    m = SVGP(X=X, Y=Y, Z=Z, dtype=args.dtype, seed=0)
    m.N = N_TOTAL  # the ELBO's N/n rescale uses the global data count
    opt = m.ELBO()
    with contextlib.redirect_stdout(sys.stderr):  # compile() announces itself on stdout like the reference does
        opt.compile(optimizer=tf.train.AdamOptimizer(1e-3))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    opt.optimize(maxiter=max(args.warmup, 1), minibatch_size=MINIBATCH)
    barrier()
    t0 = time.perf_counter()
    opt.optimize(maxiter=args.steps, minibatch_size=MINIBATCH)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    elbo_after = opt.run(minibatch_size=MINIBATCH)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    plan = opt.last_plan if getattr(opt.last_plan, "gflat", None) is not None else None
    steps_per_sec = args.steps / elapsed
    global_batch = MINIBATCH * world

    # ---- roofline of the dominant contraction kernel, timed live on the plan's stream
    H = m._session.H
    sess = m._session
    Mi, n = M_INDUCING, MINIBATCH
    dt = sess.torch_dtype
    xs = torch.as_tensor(X[:n], dtype=dt).cuda()
    zb = sess.param_view(m.gp.z)
    ell = H.ewise("SOFTPLUS", [sess.param_view(m.gp.kern.lengthscales)])
    K = H.gram_fwd(zb, zb, ell)
    L, info = H.cholesky(H.matutil(K, H.MATUTIL_ADD_EYE, alpha=1e-5))
    W = H.trinv(L)
    A = torch.empty(Mi, n, dtype=dt, device="cuda")
    iters = 200
    with torch.cuda.stream(sess.stream):
        for _ in range(10):
            H.sgp_A(xs, zb, ell, W, out=A)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(sess.stream)
        for _ in range(iters):
            H.sgp_A(xs, zb, ell, W, out=A)
        e1.record(sess.stream)
        sess.stream.synchronize()
    kern_us = e0.elapsed_time(e1) * 1e3 / iters
    flops = float(Mi) * Mi * n  # triangular solve A = L^-1 Kmn: M^2 n flops
    achieved = flops / (kern_us * 1e-6) * 1e-12
    peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "float32" else 78.6
    # HBM-side bytes per launch of this kernel: from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
    # WRITE_SIZE, gfx950 correction applied; profiles/r01_pmc_traffic.json) -- counters cannot be read from here
    traffic, traffic_src = None, None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as fh:
            pmc = json.load(fh)["kernels"]
        # fp32 runs the column-strip form of the kernel, fp64 the tiled form
        want = ["sgp_A_strip_kernel<1>"] if args.dtype == "float32" else ["sgp_A_kernel<double, 1, true>"]
        for name, rec in pmc.items():
            if any(w in name for w in want):
                traffic, traffic_src = rec["traffic_bytes"], "profiles/r01_pmc_traffic.json"
    except (OSError, KeyError, ValueError):
        pass
    roofline = {"kernel": "%s (A = L^-1 K(z,x), %dx%d by %d)" % ("sgp_A_strip_kernel<1>" if args.dtype == "float32" else "sgp_A_kernel<double,1,true>", Mi, Mi, n),
                "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": traffic, "traffic_source": traffic_src, "avg_kernel_us": kern_us,
                "flops_per_launch": flops,
                "algorithmic_bytes_per_launch": float(dt.itemsize) * (Mi * n + Mi * Mi / 2 + n + Mi)}

    breakdown = None
    try:
        prof = opt._plans[[k for k in opt._plans if k[0] == "opt"][0]].profile(iters=20)
        breakdown = {k: round(v[0], 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:12]}
    except Exception as e:  # profiling is informational only
        breakdown = {"error": str(e)}

    out = {
        "metric": "elbo_samples_per_sec", "value": steps_per_sec * global_batch, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.dtype == "float32" else "f64", "data": "synthetic",
        "config": {"workload": "cfg2 SVGP ELBO step: 1-D UnitRBF, N=1e6, M=512 inducing, minibatch 8192 per GPU, "
                               "diag Normal q(u), diagonal residual, Adam lr 1e-3", "global_batch": global_batch,
                   "parallelism": "dp%d" % world, "N": N_TOTAL, "M": M_INDUCING},
        "steps_per_sec": steps_per_sec,
        "flops_per_step_algorithmic": 3.0 * Mi * Mi * n + 3.0 * Mi ** 3 + 4.0 * Mi * n,
        "elbo_after": elbo_after,
        "roofline": roofline,
        "step_breakdown_us_eager_replay": breakdown,  # informational: small ops are CPU-launch bound in this replay

    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
