#!/bin/bash
# rocprofv3 counter passes over a python tool; summaries (per-kernel averages) land in gpurun_out/<tag>/pmc_*.txt
#   tools/pmc_run.sh <tag> <script.py> [args...]
set -e
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  d=$out/pass$i
  rm -rf "$d"
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$d" -o p -- python3 "$root/$1" "${@:2}" > "$out/pass$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/pass$i.log"; continue; }
  f=$(find "$d" -name 'p_counter_collection.csv' | head -1)
  python3 - "$f" > "$out/pmc_pass$i.txt" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    n = max(len(v) for v in cs.values())
    if n < 20: continue
    print(k, "(%d dispatches)" % n)
    for c, v in sorted(cs.items()):
        print("    %-34s avg %.4g" % (c, sum(v) / len(v)))
PY
  cat "$out/pmc_pass$i.txt"
  rm -rf "$d"
done
