#!/bin/bash
# rocprofv3 --kernel-trace --stats of bench.py on the GPU box; the per-kernel summary lands in gpurun_out/<tag>/.
#   tools/prof_bench.sh <tag> [bench.py args...]
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o bench -- python3 "$root/bench.py" --no-cpu-baseline "$@" > "$out/bench.json" 2> "$out/bench.err"
f=$(find "$out" -name 'bench_kernel_stats.csv' | head -1)
cp "$f" "$out/kernel_stats.csv"
steps=$(python3 -c "import json,sys; d=json.load(open('$out/bench.json')); print(d['steps']+d['warmup']+d['step_time_us']['steps']+1)")
python3 "$root/tools/prof_summary.py" "$(dirname "$f")" "$steps" 40 | tee "$out/summary.txt"
find "$out" -name '*.csv' ! -name 'kernel_stats.csv' -delete; find "$out" -name '*.db' -delete
