"""Observed-vs-bound table of every fp32 tolerance of the GPU suite (tests/parity.py: observe).

    HB_OBSERVED_OUT=gpurun_out/observed.jsonl python -m pytest tests -m gpu -q ; python tools/observed_errors.py gpurun_out/observed.jsonl

Prints, per assertion name, the observed error, the bound and bound / observed; lists bounds looser than 10x."""
import json
import sys
from collections import OrderedDict

rows = OrderedDict()
for ln in open(sys.argv[1]):
    r = json.loads(ln)
    k = r["name"]
    if k in rows:
        rows[k]["err"] = max(rows[k]["err"], r["err"])
    else:
        rows[k] = dict(r)
loose = 0
print("%-72s %11s %9s %8s" % ("assertion", "observed", "bound", "ratio"))
for k, r in rows.items():
    ratio = r["tol"] / r["err"] if r["err"] > 0 else float("inf")
    flag = "  LOOSE" if ratio > 10 and not (r["tol"] <= 1e-5) else ""
    loose += bool(flag)
    print("%-72s %11.3e %9.1e %8.1f%s" % (k, r["err"], r["tol"], ratio, flag))
print("%d assertions, %d with a bound looser than 10x the observed value (bounds at the 1e-5 north-star bar excepted)" % (len(rows), loose))
