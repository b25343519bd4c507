"""Print the launch plan of a compiled optimisation step: one line per plan step (op label, emitting node, operand
shapes; fused elementwise clusters list their member ops).  `python tools/dump_plan.py [cfg2|cfg3|cfg4|cfg5] [--explain]`.
`--explain`: the planner's fusion ledger (Plan.explain) -- every fusion pass that looked at a node, whether it fired, and why
not; side candidates that no host adopted; small steps that stayed launches of their own."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import henbun_amd as hb
from henbun_amd import graph as G
import bench

explain = "--explain" in sys.argv
argv = [a for a in sys.argv[1:] if a != "--explain"]
name = argv[0] if argv else "cfg2"
cfg = dict(bench.CONFIGS[name]); cfg["N"] = min(cfg["N"], 200000)
m, dp_reduce, _ = bench.build_model(name, cfg, 1, 0, "float32", cfg["n"])
opt = m.ELBO(); opt.compile(dp_reduce=dp_reduce)
opt.optimize(1, cfg["n"])
plan = opt.last_plan
orig = G.Plan._emit_cluster
print("%d plan steps" % len(plan.steps))
for i, s in enumerate(plan.steps):
    lab = plan.step_labels.get(id(s), "other")
    node = plan.step_nodes.get(id(s))
    desc = ""
    if node is not None:
        desc = " in=" + ",".join("x".join(map(str, t.shape)) or "()" for t in node.inputs) + \
               " out=" + ",".join("x".join(map(str, t.shape)) or "()" for t in node.outputs)
        if node.op == "ew":
            desc += " f=" + node.attrs["f"]
    print("%3d %-24s%s" % (i, lab, desc))
for c in {id(c): c for c in plan._clusters.values()}.values():
    if len(c.nodes) >= 2:
        print("cluster[%d] space=%s:" % (len(c.nodes), tuple(c.space)),
              " ".join((n.attrs.get("f") or n.op) + "(" + ",".join("x".join(map(str, t.shape)) or "()" for t in n.inputs) + ")" for n in c.nodes))

if explain:
    print("\nfusion ledger (pass | node | fired | why not):")
    for pass_name, lab, fired, why in plan.explain:
        print("  %-72s %-34s %-5s %s" % (pass_name, lab, "yes" if fired else "NO", why if (not fired or why.startswith("host:")) else ""))
    for c in plan._side_cands:
        if not c["cell"]["defer"] and not c.get("dead"):
            nd = c.get("node")
            print("  %-72s %-34s %-5s %s" % ("side job rides on a host launch", nd.op if nd is not None else "-", "NO",
                                            "no host launch (persistent Cholesky, in-workgroup split-K GEMM) came after it"))
