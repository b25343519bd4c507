#!/usr/bin/env python3
"""ELBO-step benchmark (BASELINE.json metric: ELBO steps/sec and samples/sec at
fixed minibatch, 1/2/4/8 MI355X).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2|cfg3|cfg4|cfg5] [--scaling weak|strong]

Default workload (BASELINE.json configs[1]): sparse variational GP
regression, 1-D UnitRBF, N = 1e6 synthetic points, M = 512 inducing points,
minibatch 8192 per GPU, diagonal Normal q(u), 'diagonal' residual, fp32.
One step = device-side minibatch index draw + row gather + reparameterised
sample + ELBO forward + full backward + fused Adam (= one iteration of the
reference's Optimizer.optimize, model.py:263-267).  Inputs are resident in HBM.
--config selects the other BASELINE configs at their full sizes (cfg3 full-rank q(u) M = 1024 n = 16384;
cfg4 amortised encoder [64,256,32] n = 32768; cfg5 4 experts + 4 gates x M = 512 n = 65536).

With N > 1 there is one rank per GPU.  Under torch.distributed.run (WORLD_SIZE set) this process IS a rank;
launched plainly (`python bench.py --gpus N`) it starts the N ranks itself -- fresh child processes through
`python -m torch.distributed.run` on 127.0.0.1, BEFORE anything in this process touches the GPU -- and exits
with their return code.  `--gpus` must equal the world size (anything else is an error, rc 2).  Every rank owns a 1/N
shard of the data; `--scaling weak` (default) keeps the per-GPU minibatch, `--scaling strong` keeps the GLOBAL
minibatch and gives every rank 1/N of it; with N > 1 the line always carries the OTHER mode too, measured right
after (`"also": {...}`).  The flat gradient is all-reduced over RCCL once per step (hb_allreduce_sum behind the C ABI).
`--dry-run` runs the launch plumbing only (gloo process group, shard arithmetic, one all-reduce on CPU tensors)
and prints the JSON skeleton with `n_gpus` = world size: the CPU test of the N > 1 launch path.

Rank 0 prints ONE JSON line: the contract fields plus
  "roofline"      -- the plan step with the largest share of the step time, FOUND AT RUN TIME (every step of the
                     plan is timed with HIP events on the plan's stream, back to back), priced with the work
                     model of its op against the fp32 MFMA peak or the HBM peak;
  "roofline_step" -- algorithmic flops of the whole step / measured step time / fp32 MFMA peak;
  "step_time_us"  -- median / min / max of per-step HIP-event intervals over the timed steps' replay;
  "cpu_baseline"  -- the CPU oracle's restatement of the reference graph (torch-CPU fp32 + autograd + TF-formula
                     Adam) on ALL usable host cores, bounded sample (kind "port": TensorFlow is not available).
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: Peak FP32 (matrix), spec
PEAK_F64_MFMA_TFLOPS = 78.6
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 matrix peak (the sparsity figure is twice that and is never used)
PEAK_HBM_GBS = 8000.0         # same guide: HBM3E peak BW, spec (6.29 TB/s measured copy)

CONFIGS = {
    "cfg2": dict(N=1_000_000, M=512, n=8192, desc="cfg2 SVGP ELBO step: 1-D UnitRBF, N=1e6, M=512 inducing, minibatch "
                 "8192 %s, diag Normal q(u), diagonal residual, Adam lr 1e-3"),
    "cfg3": dict(N=1_000_000, M=1024, n=16384, desc="cfg3 SVGP ELBO step: full-covariance q(u) ('fullrank'), N=1e6, "
                 "M=1024 inducing, minibatch 16384 %s, diagonal residual, Adam lr 1e-3"),
    "cfg4": dict(N=4_000_000, M=0, n=32768, desc="cfg4 amortised ELBO step: NeuralNet [64,256,32] sigmoid encoder -> "
                 "LOCAL diag q(z), L=16, linear Gaussian decoder, N=4e6, minibatch 32768 %s, Adam lr 1e-3"),
    "cfg5": dict(N=1_000_000, M=512, n=65536, desc="cfg5 mixture of 4 sparse-GP experts + 4 sparse-GP gates x M=512 "
                 "(one expert-batched SparseGP), N=1e6, minibatch 65536 %s, Adam lr 1e-3"),
}


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
    cap = os.environ.get("HB_BENCH_CPU_THREADS")
    return max(1, min(n, int(cap))) if cap else max(1, n)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, seconds=15.0):
    """Oracle ('port') timed on the host cores: the cfg2 step, bounded number of steps."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import torch

    import henbun_oracle as O
    from henbun_amd.models import svgp_data

    M, n, N = 512, 8192, 1_000_000
    threads = usable_cores()
    torch.set_num_threads(threads)
    dt = torch.float32
    rng = np.random.RandomState(0)
    Ns = 100_000  # rows actually materialised for the sample (indices are drawn from it)
    X, Y, Z = svgp_data(Ns, M, 0, domain=0.5 * M)
    Xt, Yt = torch.as_tensor(X, dtype=dt), torch.as_tensor(Y, dtype=dt)
    params = {
        "z": torch.as_tensor(Z, dtype=dt), "ell_raw": torch.as_tensor(O.log1pe_backward_np(np.ones(1)), dtype=dt),
        "q_mu": torch.as_tensor(0.1 * rng.randn(1, M), dtype=dt),
        "q_sqrt": torch.as_tensor(0.1 * rng.randn(M), dtype=dt),
        "k_var_raw": torch.as_tensor(O.log1pe_backward_np(np.ones(1)), dtype=dt),
        "var_raw": torch.as_tensor(O.log1pe_backward_np(np.ones(1)), dtype=dt),
    }
    names = list(params)
    leaves = [params[k].clone() for k in names]
    adam = O.AdamTF(leaves, lr=1e-3)

    def step():
        idx = torch.as_tensor(rng.randint(0, Ns, n))
        u = torch.randn(M, dtype=dt)
        eps = torch.randn(n, dtype=dt)
        # jitter 0.1 (timing only): with inputs up to 256 the reference's |a|^2+|b|^2-2ab^T squared distance
        # carries ~4e-3 absolute error in fp32, so Kmm is not PD at the default 1e-5; the op sequence and
        # flop count do not depend on the jitter value.
        fn = lambda p: O.svgp_elbo(p, Xt[idx], Yt[idx], float(N), u, eps, jitter=0.1)
        _, g = O.grads_of(fn, dict(zip(names, leaves)))
        adam.step([-g[k] for k in names])

    step()  # warm-up
    t0 = time.perf_counter()
    k = 0
    while True:
        step()
        k += 1
        el = time.perf_counter() - t0
        if el >= seconds or k >= 400:
            break
    sps = k / el
    return {"value": sps * n, "unit": "samples/s", "steps_per_sec": sps, "cores": threads, "cpu": cpu_model(),
            "kind": "port",
            "sample": "%d steps of the cfg2 step (n=%d, M=%d, fp32, torch-CPU restatement of the reference op graph "
                      "+ autograd + TF-formula Adam on %d threads; jitter 0.1 for fp32 PD-safety of the reference "
                      "distance formula) in %.1f s" % (k, n, M, threads, el)}


# ----------------------------------------------------------------------------------------------- workloads
def build_model(name, cfg, world, rank, dtype, n_local):
    """(model, dp_reduce, flops_per_step_algorithmic(global minibatch))."""
    import numpy as np

    from henbun_amd.models import SVGP, Amortised, ExpertsGPR, svgp_data

    np.random.seed(1234)  # identical parameter initialisation on every rank (the data below is rank-specific)
    N, M, n = cfg["N"], cfg["M"], cfg["n"]
    rows = N // world
    if name in ("cfg2", "cfg3"):
        X, Y, Z = svgp_data(rows, M, seed=rank, domain=0.5 * M)
        m = SVGP(X=X, Y=Y, Z=Z, q_shape="fullrank" if name == "cfg3" else "diagonal", dtype=dtype, seed=0)
        if name == "cfg3":
            r0 = np.random.RandomState(0)
            m.u.q_sqrt = 0.1 * np.eye(M) + 0.01 * np.tril(r0.randn(M, M))     # SURVEY 8(d) initialisation
        m.N = N
        flops = lambda nb: 3.0 * M * M * nb + 3.0 * M ** 3 + 4.0 * M * nb
        return m, "mean", flops
    if name == "cfg4":
        Din, Hd, L = 64, 256, 16
        r = np.random.RandomState(100 + rank)
        W0 = np.random.RandomState(7).randn(L, Din).astype(np.float32) / np.sqrt(L)
        Y = np.empty((rows, Din), dtype=np.float32)
        for lo in range(0, rows, 500_000):   # generated in blocks: bounded host memory
            hi = min(rows, lo + 500_000)
            Y[lo:hi] = np.tanh(r.randn(hi - lo, L).astype(np.float32) @ W0) + 0.1 * r.randn(hi - lo, Din).astype(np.float32)
        m = Amortised(Y=Y, L=L, H=Hd, dtype=dtype, seed=0)
        flops = lambda nb: 3 * 2.0 * nb * (Din * Hd + Hd * 2 * L + L * Din)
        return m, "sum", flops
    if name == "cfg5":
        E = 4
        X, Y, Z = svgp_data(rows, M, seed=rank, domain=256.0)
        Y = np.where(X < 128, np.sin(X), 0.3 * np.sin(3.0 * X)) + 0.1 * np.random.RandomState(50 + rank).randn(rows, 1)
        ells = list(np.linspace(0.6, 1.2, E)) + list(np.linspace(0.8, 1.4, E))
        m = ExpertsGPR(X=X, Y=Y, Z=Z, ells=ells, dtype=dtype, seed=0)
        m.N = N
        flops = lambda nb: 2 * E * (3.0 * M * M * nb + 3.0 * M ** 3 + 4.0 * M * nb)
        return m, "mean", flops
    raise ValueError(name)


def work_model(node, itemsize):
    """(flops, algorithmic bytes, bound) of the plan step emitted by graph node `node`, SURVEY.md 8(d) accounting:
    M x n intermediates that a fused implementation would not move are NOT counted as algorithmic bytes."""
    import numpy as np

    op = node.op
    ish = [tuple(t.shape) for t in node.inputs]
    osh = [tuple(t.shape) for t in node.outputs]
    numel = lambda sh: float(np.prod(sh)) if len(sh) else 1.0
    if op == "cholesky":
        M = ish[0][-1]
        B = numel(ish[0][:-2])
        return B * 2.0 * M ** 3 / 3.0, B * 3.0 * M * M * itemsize, "mfma"      # factor + the inverse riding along
    if op in ("sgp", "sgp_grad"):
        x, z = ish[0], ish[1]
        n, d, M = x[-2], x[-1], z[-2]
        E = numel(z[:-2])
        f = E * M * M * n * (1.0 if op == "sgp" else 2.0)
        b = E * (n * d + M * d + M * M) * itemsize                              # 8(d): nothing of size M*n
        return f, b, "mfma"
    if op == "matmul":
        a, b = ish[0], ish[1]
        o = osh[0]
        k = a[-2] if node.attrs.get("ta") else a[-1]
        f = 2.0 * numel(o) * k
        return f, (numel(a) + numel(b) + numel(o)) * itemsize, "mfma"
    if op in ("trinv",):
        M = ish[0][-1]
        return numel(ish[0][:-2]) * M ** 3 / 3.0, 2.0 * numel(ish[0]) * itemsize, "mfma"
    byts = (sum(numel(s) for s in ish) + sum(numel(s) for s in osh)) * itemsize
    return 0.0, byts, "hbm"


def time_steps_standalone(plan, torch, iters=40):
    """[(label, node, avg_us)] for every step of the plan: each step closure is launched `iters` times back to back
    on the plan's stream between two HIP events (steps with side effects on the parameters excluded)."""
    out = []
    st = plan.stream
    with torch.cuda.stream(st):
        for s in plan.steps:
            if s in plan.side_effect_steps:
                continue
            for _ in range(3):
                s()
            plan.H.side_flush()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(iters):
                s()
            plan.H.side_flush()   # a step that only RECORDS a side job (hb_side_push_*): run what it recorded here
            e1.record(st)
            st.synchronize()
            out.append((plan.step_labels.get(id(s), "other"), plan.step_nodes.get(id(s)), e0.elapsed_time(e1) * 1e3 / iters))
    return out


def _free_port():
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of this one (which has not
    imported torch, let alone touched a GPU) and return their exit code.  Rank 0's JSON line goes to our stdout."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, world, rank):
    """Launch plumbing without a GPU: what a rank does up to device bring-up, on a gloo group."""
    import torch
    import torch.distributed as dist

    from henbun_amd import parallel

    if world > 1:
        dist.init_process_group("gloo")
    cfg = CONFIGS[args.config]
    n_global = cfg["n"] * world if args.scaling == "weak" else cfg["n"]
    lo, hi = parallel.shard_rows(cfg["N"], rank, world)
    t = torch.tensor([float(rank + 1), float(hi - lo)], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t)
    ok = t[0].item() == world * (world + 1) / 2 and int(t[1].item()) == cfg["N"]
    if rank == 0:
        print(json.dumps({"metric": "elbo_samples_per_sec", "value": None, "unit": "samples/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "scaling": args.scaling, "dry_run": True,
                          "config": {"name": args.config, "global_batch": n_global, "per_gpu_batch": n_global // world,
                                     "parallelism": "dp%d" % world, "shard_rows_rank0": hi - lo},
                          "plumbing_ok": bool(ok), "rng_streams_rank0": parallel.rng_stream_ids(0)}))
    if world > 1:
        dist.destroy_process_group()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", default="float32", choices=["float32", "float64"])
    ap.add_argument("--contraction", default="native", choices=["native", "bf16x3"],
                    help="operand form of the M^2 n contractions (bf16x3: BASELINE cfg 5's reduced-precision variant in its "
                         "usable form: three-term bf16 operands, fp32 accumulation, fp32-level accuracy)")
    ap.add_argument("--tri-pack", action="store_true", help="cfg3: keep the full-rank q_sqrt as its packed lower triangle")
    ap.add_argument("--dry-run", action="store_true", help="launch plumbing only (gloo, no GPU): see the module docstring")
    ap.add_argument("--no-also", action="store_true", help="N > 1: skip the second measurement in the other scaling mode")
    args = ap.parse_args()

    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus)       # parent of the ranks: no torch import, no GPU call in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but the launcher started %d ranks (WORLD_SIZE): refusing to report a line whose "
                  "n_gpus would not be what was asked for" % (args.gpus, world), file=sys.stderr)
        return 2
    if args.dry_run:
        return dry_run(args, world, rank)

    import numpy as np
    import torch

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
    import henbun_amd as hb

    tf = hb.tf
    st = hb.settings.get_settings()
    st.numerics.contraction = args.contraction
    st.numerics.tri_pack = bool(args.tri_pack)
    hb.settings._stack.append(st)   # for the whole run (same as `with hb.settings.temp_settings(st):`)
    cfg = CONFIGS[args.config]
    n_global = cfg["n"] * world if args.scaling == "weak" else cfg["n"]
    n_local = n_global // world
    m, dp_reduce, flops_fn = build_model(args.config, cfg, world, rank, args.dtype, n_local)
    opt = m.ELBO()
    with contextlib.redirect_stdout(sys.stderr):  # compile() announces itself on stdout like the reference does
        opt.compile(optimizer=tf.train.AdamOptimizer(1e-3), dp_reduce=dp_reduce)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_steps(nl):
        """W warm-up steps, then EXACTLY K steps between barrier + synchronize; max over ranks (seconds)."""
        opt.optimize(maxiter=max(args.warmup, 1), minibatch_size=nl)
        barrier()
        t0 = time.perf_counter()
        opt.optimize(maxiter=args.steps, minibatch_size=nl)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    elapsed = timed_steps(n_local)
    plan = opt.last_plan

    # per-step HIP-event intervals (one event per step on the plan's stream; all ranks take part in the steps)
    ksteps = max(min(args.steps, 400), 20)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(ksteps + 1)]
    with torch.cuda.stream(plan.stream):
        for i in range(ksteps):
            evs[i].record(plan.stream)
            opt._run_steps(plan, 1)
        evs[ksteps].record(plan.stream)
        plan.stream.synchronize()
    opt._check_step_failure(plan)
    per = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(ksteps))
    elbo_after = opt.run(minibatch_size=n_local)
    barrier()

    # the exchange really spans `world` ranks (an RCCL communicator of another size would be a silent 1-GPU run)
    dp_ranks = 1
    if world > 1:
        from henbun_amd import parallel

        comm = parallel.Communicator.create(m._session.device)
        dp_ranks = comm.world_size if comm is not None else dist.get_world_size()
        assert dp_ranks == world, "data-parallel exchange spans %d ranks, expected %d" % (dp_ranks, world)
        assert getattr(plan, "dp_mode", "none") != "none", "N > 1 but the compiled step has no gradient exchange"

    # N > 1: the other scaling mode, same protocol, right after (strong: global minibatch fixed, n / N per rank;
    # weak: per-GPU minibatch fixed) -- north_star quotes a strong-scaling target, the contract line is weak by default
    also = None
    if world > 1 and not args.no_also:
        other = "strong" if args.scaling == "weak" else "weak"
        ng2 = cfg["n"] if other == "strong" else cfg["n"] * world
        nl2 = ng2 // world
        el2 = timed_steps(nl2)
        also = {"scaling": other, "global_batch": ng2, "per_gpu_batch": nl2, "ms_per_step": el2 / args.steps * 1e3,
                "value": args.steps / el2 * ng2, "unit": "samples/s", "steps_per_sec": args.steps / el2}

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    steps_per_sec = args.steps / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    itemsize = 4 if args.dtype == "float32" else 8
    peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "float32" else PEAK_F64_MFMA_TFLOPS
    flops_step = flops_fn(n_global)          # whole job, algorithmic
    flops_rank = flops_fn(n_local)           # what this rank's step computes

    # ---- roofline of the step that dominates, found at run time
    roofline, breakdown = None, None
    try:
        timed = time_steps_standalone(plan, torch)
        total = sum(t for _, _, t in timed)
        breakdown = {}
        for lab, _, t in timed:
            breakdown[lab] = round(breakdown.get(lab, 0.0) + t, 1)
        breakdown = dict(sorted(breakdown.items(), key=lambda kv: -kv[1])[:14])
        lab, node, us = max(timed, key=lambda r: r[2])
        f, b, bound = work_model(node, itemsize) if node is not None else (0.0, 0.0, "hbm")
        if bound == "mfma":
            achieved, pk, unit = f / (us * 1e-6) * 1e-12, peak, "TFLOP/s"
        else:
            achieved, pk, unit = b / (us * 1e-6) * 1e-9, PEAK_HBM_GBS, "GB/s"
        shapes = "" if node is None else " ".join("x".join(str(d) for d in t.shape) or "scalar" for t in node.inputs[:4])
        traffic, traffic_src = None, None
        try:
            # the newest per-round table (tools/pmc_traffic.sh -> tools/make_traffic_json.py): PMC counters of an earlier
            # run of this same command, regenerated with the kernels -- the source file is named in the line
            import glob

            tfile = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))[-1]
            with open(tfile) as fh:
                pmc = json.load(fh)
            rec = pmc.get(args.config, {}).get(lab)
            if rec:
                traffic = rec["traffic_bytes"]
                traffic_src = "profiles/%s (%s)" % (os.path.basename(tfile), rec.get("kernel", lab))
        except (OSError, KeyError, ValueError, IndexError):
            pass
        peaks_16bit = None
        if bound == "mfma" and args.contraction == "bf16x3" and lab in ("sgp", "sgp_grad"):
            # the M^2 n contractions run on the bf16 matrix pipe: SIX bf16 MFMAs (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi)
            # per fp32-equivalent product, so their own ceiling is the dense bf16 peak / 6; both fractions are reported
            pk = PEAK_BF16_MFMA_TFLOPS / 6.0
            peaks_16bit = {"bf16_dense_peak": PEAK_BF16_MFMA_TFLOPS, "divisor": 6, "frac_of_bf16x3_effective_peak": achieved / pk,
                           "frac_of_16bit_peak": achieved / PEAK_BF16_MFMA_TFLOPS, "frac_of_fp32_mfma_peak": achieved / peak}
        roofline = {"kernel": "%s [%s]" % (lab, shapes), "bound": bound, "achieved": achieved, "peak": pk, "unit": unit,
                    "frac": achieved / pk, "traffic": traffic, "traffic_source": traffic_src, "avg_kernel_us": us,
                    "share_of_step": us / total if total else None, "flops_per_launch": f,
                    "algorithmic_bytes_per_launch": b, "bf16x3": peaks_16bit,
                    "how": "largest of all plan steps, each timed stand-alone with HIP events on the plan's stream "
                           "(%d back-to-back calls); a step may be a short chain of launches" % 40}
    except Exception as e:  # never lose the headline over the diagnostics
        roofline = {"error": repr(e)}

    out = {
        "metric": "elbo_samples_per_sec", "value": steps_per_sec * n_global, "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": ("f32" if args.dtype == "float32" else "f64") + ("" if args.contraction == "native" else
                                                                 " (M^2 n contractions: bf16x3 operands, f32 accumulate)"),
        "data": "synthetic",
        "config": {"workload": cfg["desc"] % ("per GPU" if args.scaling == "weak" else "global"),
                   "global_batch": n_global, "per_gpu_batch": n_local, "parallelism": "dp%d" % world, "N": cfg["N"],
                   "M": cfg["M"], "name": args.config, "dp_exchange": getattr(plan, "dp_mode", "none"), "dp_ranks": dp_ranks,
                   "contraction": args.contraction,
                   "tri_pack": bool(args.tri_pack)},
        "steps_per_sec": steps_per_sec,
        "flops_per_step_algorithmic": flops_step,
        "elbo_after": elbo_after,
        "step_time_us": {"median": per[len(per) // 2], "min": per[0], "max": per[-1], "steps": ksteps,
                         "how": "HIP events between consecutive steps on the plan's stream"},
        "roofline": roofline,
        "roofline_step": {"bound": "mfma", "achieved": flops_rank / (ms_per_step * 1e-3) * 1e-12, "peak": peak,
                          "unit": "TFLOP/s", "frac": flops_rank / (ms_per_step * 1e-3) * 1e-12 / peak,
                          "what": "algorithmic flops of one rank's step / ms_per_step"},
        "step_breakdown_us_standalone": breakdown,
    }
    if also is not None:
        out["also"] = also
    if world == 1 and not args.no_cpu_baseline and args.config == "cfg2":
        out["cpu_baseline"] = cpu_baseline(cfg)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
