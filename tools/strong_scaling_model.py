"""Per-rank step time of a STRONG-scaled job, measured on one GPU: the step of rank 0 of a world of R is the single-rank
step at the per-rank minibatch n / R (the replicated part -- Gram, Cholesky + inverse, the VJP products, KL, Adam -- does
not shrink), plus pack + ONE all-reduce + Adam as the data-parallel tail.  The tail runs here on a one-rank RCCL
communicator (settings.runtime.force_dp), i.e. its launches are timed but the wire is not: T(R) = T_rank(n / R) + t_wire(R).

    python tools/strong_scaling_model.py cfg2 [cfg5 ...]         -> table for R = 1, 2, 4, 8

Nothing here needs a second GPU or a second process; DESIGN.md section 6 quotes the table (profiles/r04_strong_scaling_model.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import henbun_amd as hb  # noqa: E402
import bench  # noqa: E402


def timeit(opt, plan, steps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with plan._on_stream():
        e0.record(torch.cuda.current_stream())
        opt._run_steps(plan, steps)
        e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


for name in sys.argv[1:] or ["cfg2"]:
    cfg = bench.CONFIGS[name]
    steps = 200 if name in ("cfg2", "cfg4") else 40
    print("%s: global minibatch %d, per-rank step at n / R (one GPU; DP tail on a one-rank communicator)" % (name, cfg["n"]))
    print("   %3s %9s %12s %12s %10s" % ("R", "n / R", "plain us", "with tail us", "T(1)/T(R)"))
    base = None
    for R in (1, 2, 4, 8):
        n = cfg["n"] // R
        row = []
        for force in (False, True):
            st = hb.settings.get_settings()
            st.runtime.force_dp = force
            with hb.settings.temp_settings(st):
                m, dp_reduce, _ = bench.build_model(name, cfg, 1, 0, "float32", n)
                opt = m.ELBO()
                opt.compile(dp_reduce=dp_reduce)
                opt.optimize(maxiter=5, minibatch_size=n)
                plan = opt.last_plan
                ts = sorted(timeit(opt, plan, steps) for _ in range(5))
                row.append(ts[2])
            del m, opt, plan
            torch.cuda.empty_cache()
        base = base or row[0]
        print("   %3d %9d %12.1f %12.1f %10.2f" % (R, n, row[0], row[1], base / row[1] if R > 1 else 1.0))
