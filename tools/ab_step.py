"""Same-process A/B of the optimisation step under two (or more) settings: the variants' plans are built once, then timed
in INTERLEAVED rounds on one device (boxes differ by several per cent, cdna_hip_programming.md rule 24; never rank builds
by timings taken on different devices).

    python tools/ab_step.py cfg2 runtime.fold_gram=0 runtime.fold_gram=1
    python tools/ab_step.py cfg2 debug:chol_persist=0 debug:chol_persist=1      (hb_debug_set switches, set at plan build)

Each variant is a comma-separated list of `section.key=value` settings and/or `debug:key=value` switches."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import henbun_amd as hb  # noqa: E402
from henbun_amd import hip_ops as H  # noqa: E402
import bench  # noqa: E402

name = sys.argv[1]
variants = sys.argv[2:]
cfg = bench.CONFIGS[name] if hasattr(bench, "CONFIGS") else None
plans = []
for v in variants:
    st = hb.settings.get_settings()
    dbg = {}
    for item in v.split(","):
        k, val = item.split("=")
        if k.startswith("debug:"):
            dbg[k[6:]] = int(val)
        else:
            sec, key = k.split(".")
            cur = getattr(getattr(st, sec), key, None)
            setattr(getattr(st, sec), key, type(cur)(float(val)) if isinstance(cur, (bool, int, float)) else val)
    for k, val in dbg.items():
        H.debug_set(k, val)
    with hb.settings.temp_settings(st):
        m, dp_reduce, _ = bench.build_model(name, cfg, 1, 0, "float32", cfg["n"])
        opt = getattr(m, "ELBO")()
        opt.compile(dp_reduce=dp_reduce)
        opt.optimize(maxiter=5, minibatch_size=cfg["n"])
        plan = opt.last_plan
    H.debug_clear()
    plans.append((v, m, opt, plan))
torch.cuda.synchronize()


def timeit(opt, plan, steps=200):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with plan._on_stream():
        e0.record(torch.cuda.current_stream())
        opt._run_steps(plan, steps)
        e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps


res = {v: [] for v in variants}
for rnd in range(7):
    for v, m, opt, plan in plans:
        res[v].append(timeit(opt, plan))
for v in variants:
    a = np.array(res[v][1:])
    print("%-50s median %.2f us/step  min %.2f  (rounds: %s)" % (v, np.median(a), a.min(), " ".join("%.1f" % t for t in a)))
