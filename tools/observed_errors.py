"""Observed-vs-bound table of every fp32 tolerance of the GPU suite (tests/parity.py: observe).

    python tools/observed_errors.py --run gpurun_out/observed.jsonl [pytest args ...]    (default args: tests -m gpu -q)
    python tools/observed_errors.py gpurun_out/observed.jsonl                            (table of an earlier run)

`--run` runs pytest IN THIS PROCESS with tests/parity.observe wrapped by a recorder: every (name, error, bound) goes to the
file, and a bound that does not hold is recorded and reported at the end instead of stopping its test (a calibration
run sees every value).  The wrapper lives here; the suite itself has no switch that weakens an assertion.

One row per ASSERTION SITE (the name with its [shape] part removed: a site is one line of a test, run over a
parametrisation): the largest and smallest error observed there, the bound written in the test, bound / largest.
Sites whose bound is looser than 10x the largest observed value are listed at the end (bounds at or below the 1e-5
north-star bar excepted)."""
import json
import os
import re
import sys
from collections import OrderedDict

if len(sys.argv) > 1 and sys.argv[1] == "--run":
    out = sys.argv[2]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    sys.path.insert(0, os.path.join(root, "tests"))
    sys.path.insert(0, root)
    import pytest

    import parity

    _orig = parity.observe
    _broken = []
    open(out, "w").close()

    def _recording_observe(name, err, tol):
        with open(out, "a") as f:
            f.write(json.dumps({"name": name, "err": float(err), "tol": float(tol)}) + "\n")
        try:
            return _orig(name, err, tol)
        except AssertionError as e:
            _broken.append(str(e))
            return float(err)

    parity.observe = _recording_observe
    rc = pytest.main(sys.argv[3:] or [os.path.join(root, "tests"), "-m", "gpu", "-q"])
    print("pytest rc %s; %d bound(s) did not hold:" % (rc, len(_broken)))
    for b in _broken:
        print("   ", b)
    sys.argv = [sys.argv[0], out]

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
