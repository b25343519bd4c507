#!/bin/bash
# per-launch durations of the Cholesky chain with riders (rocprofv3 kernel trace of tools/bench_rider.py)
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/trace_rider
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out" -o t -- python3 "$root/tools/bench_rider.py" > "$out/log.txt" 2>&1
f=$(find "$out" -name 't_kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0][:28] for r in rows]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0 for r in rows]
grid = [r.get("Grid_Size_X", r.get("Grid_Size", "?")) for r in rows]
# last occurrence of each phase: find the last 'tril' kernels and print the 9 kernels ending there
idx = [i for i, n in enumerate(names) if "tril_inplace" in n]
for which, i in (("chain alone (an early repetition)", idx[50]), ("chain with riders (a late repetition)", idx[-50])):
    print(which)
    for j in range(i - 8, i + 1):
        print("   %-30s grid %8s  %.2f us   gap before %.2f us" % (names[j], grid[j], dur[j], (int(rows[j]["Start_Timestamp"]) - int(rows[j-1]["End_Timestamp"])) / 1000.0))
PY
