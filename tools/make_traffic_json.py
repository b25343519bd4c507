"""Fold tools/pmc_traffic.sh outputs (per-kernel HBM-side bytes per launch) into the per-plan-step table bench.py reads
for `roofline.traffic`:  python tools/make_traffic_json.py cfg2=gpurun_out/r2b/pmc [cfg3=...] > profiles/r02_pmc_traffic.json
A plan step may be a short chain of launches; its traffic is the sum over the chain's kernels of bytes x dispatches,
divided by the dispatches of the chain's once-per-step kernel."""
import json, sys

GROUPS = {  # plan label -> (once-per-step kernel, kernel-name fragments of the chain)
    "cholesky": ("chol_persist_kernel", ["chol_", "tril_inplace_kernel"]),   # (round 3: once-per-step kernel tril_inplace_kernel)
    "sgp": ("sgp_finish_part_kernel", ["sgp_A_", "sgp_finish_part_kernel", "sgp_fwd"]),
    "sgp_grad": ("sgp_bwd_finish_kernel", ["sgp_kbar", "sgp_strip_finish", "sgp_lbar", "sgp_bwd", "sgp_rowgrad"]),
    "matmul": ("matmul_kernel<float, false, false", ["matmul_kernel", "matmul_splitk_finish", "matmul_wgk"]),
}
out = {}
for spec in sys.argv[1:]:
    cfg, path = spec.split("=")
    raw = json.load(open(path + "/pmc_traffic_raw.json"))
    out[cfg] = {}
    for label, (main, frags) in GROUPS.items():
        mains = [k for k in raw if main in k]
        if not mains:
            continue
        steps = sum(raw[k]["dispatches"] for k in mains)
        members = {k: v for k, v in raw.items() if any(f in k for f in frags)}
        tot = sum(v["traffic_bytes"] * v["dispatches"] for v in members.values())
        out[cfg][label] = {
            "traffic_bytes": tot / steps,
            "kernel": " + ".join(sorted(k.split("(")[0].replace("void ", "") for k in members)),
            "per_kernel_bytes_per_launch": {k.split("(")[0].replace("void ", ""): round(v["traffic_bytes"]) for k, v in members.items()},
            "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes over bench.py), KB -> bytes, FETCH_SIZE "
                   "doubled for gfx950 (MI355X_MICROARCH.md); chain total per optimisation step",
        }
json.dump(out, sys.stdout, indent=1)
print()
