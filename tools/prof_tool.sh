#!/bin/bash
# rocprofv3 --kernel-trace --stats of a python tool; prints the per-kernel average durations.
#   tools/prof_tool.sh <tag> <script.py> [args...]      (environment variables of the caller reach the tool)
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o t -- python3 "$root/$1" "${@:2}" > "$out/tool.out" 2> "$out/tool.err"
f=$(find "$out" -name 't_kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 20:
        print("%-110s calls %5d avg %8.2f us" % (r["Name"][:110], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
find "$out" -name '*.csv' -delete; find "$out" -name '*.db' -delete
