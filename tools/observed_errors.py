"""Observed-vs-bound table of every fp32 tolerance of the GPU suite (tests/parity.py: observe).

    HB_OBSERVED_OUT=gpurun_out/observed.jsonl python -m pytest tests -m gpu -q ; python tools/observed_errors.py gpurun_out/observed.jsonl

One row per ASSERTION SITE (the name with its [shape] part removed: a site is one line of a test, run over a
parametrisation): the largest and smallest error observed there, the bound written in the test, bound / largest.
Sites whose bound is looser than 10x the largest observed value are listed at the end (bounds at or below the 1e-5
north-star bar excepted)."""
import json
import re
import sys
from collections import OrderedDict

sites = OrderedDict()
for ln in open(sys.argv[1]):
    r = json.loads(ln)
    k = re.sub(r"\[[^\]]*\]", "[]", r["name"])
    s = sites.setdefault(k, {"max": 0.0, "min": float("inf"), "tol": 0.0, "n": 0})
    s["max"], s["min"], s["tol"], s["n"] = max(s["max"], r["err"]), min(s["min"], r["err"]), max(s["tol"], r["tol"]), s["n"] + 1
loose = []
print("%-62s %4s %11s %11s %9s %7s" % ("assertion site", "runs", "largest", "smallest", "bound", "ratio"))
for k, s in sites.items():
    ratio = s["tol"] / s["max"] if s["max"] > 0 else float("inf")
    if ratio > 10 and s["tol"] > 1e-5:
        loose.append(k)
    print("%-62s %4d %11.3e %11.3e %9.1e %7.1f" % (k, s["n"], s["max"], s["min"], s["tol"], ratio))
print("%d sites, %d assertions; bound looser than 10x the largest observed value at %d site(s): %s" % (
    len(sites), sum(s["n"] for s in sites.values()), len(loose), ", ".join(loose) or "-"))
