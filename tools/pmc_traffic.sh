#!/bin/bash
# HBM-side traffic (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes) and issue/MFMA counters of every kernel of
# the bench step; writes gpurun_out/<tag>/pmc_kernels.txt and pmc_traffic.json (bytes per launch, FETCH_SIZE doubled
# for gfx950 as /opt/skills/guides/MI355X_MICROARCH.md prescribes).   tools/pmc_traffic.sh <tag> [bench.py args]
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  d=$out/pass$i
  rm -rf "$d"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$d" -o p -- python3 "$root/bench.py" --no-cpu-baseline --steps 100 --warmup 10 "$@" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; continue; }
  f=$(find "$d" -name 'p_counter_collection.csv' | head -1)
  cp "$f" "$out/counters_pass$i.csv"
  rm -rf "$d"
done
python3 - "$out" <<'PY'
import csv, sys, collections, json, os
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2, 3):
    p = os.path.join(out, "counters_pass%d.csv" % i)
    if not os.path.exists(p):
        continue
    for r in csv.DictReader(open(p)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    os.remove(p)
lines, traffic = [], {}
for k, cs in sorted(acc.items(), key=lambda kv: -len(next(iter(kv[1].values())))):
    n = max(len(v) for v in cs.values())
    if n < 50:
        continue
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    lines.append("%s  (%d dispatches)" % (k[:100], n))
    for c, v in sorted(avg.items()):
        lines.append("    %-28s avg %.5g" % (c, v))
    if "FETCH_SIZE" in avg or "WRITE_SIZE" in avg:
        # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half of wide streaming reads (guide): doubled
        fetch = 2.0 * avg.get("FETCH_SIZE", 0.0) * 1024.0
        write = avg.get("WRITE_SIZE", 0.0) * 1024.0
        traffic[k] = {"fetch_bytes_corrected": fetch, "write_bytes": write, "traffic_bytes": fetch + write, "dispatches": n}
        lines.append("    HBM-side bytes per launch: fetch (x2 gfx950) %.4g + write %.4g = %.4g" % (fetch, write, fetch + write))
    w = avg.get("SQ_WAVE_CYCLES")
    if w:
        lines.append("    of wave cycles: parked %.0f %%, issue-wait %.0f %%, issuing %.0f %%" % (
            100 * avg.get("SQ_WAIT_ANY", 0) / w, 100 * avg.get("SQ_WAIT_INST_ANY", 0) / w, 100 * avg.get("SQ_ACTIVE_INST_ANY", 0) / w))
    if avg.get("SQ_LDS_IDX_ACTIVE"):
        lines.append("    LDS bank-conflict cycles / LDS-active cycles: %.1f %%" % (100 * avg.get("SQ_LDS_BANK_CONFLICT", 0) / avg["SQ_LDS_IDX_ACTIVE"]))
open(os.path.join(out, "pmc_kernels.txt"), "w").write("\n".join(lines) + "\n")
json.dump(traffic, open(os.path.join(out, "pmc_traffic_raw.json"), "w"), indent=1)
print("\n".join(lines[:80]))
PY
