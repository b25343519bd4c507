#!/bin/bash
# Every measurement the round's profiles/ files come from, in one GPU call:  gpurun -- 'bash tools/final_measurements.sh'
# Outputs under gpurun_out/r4final/ (copy what is to be kept into profiles/r04_*: tools/collect_profiles.sh).
set -e
o=gpurun_out/r4final
mkdir -p $o
tools/_bin/stream_peak > $o/stream_peak.txt 2>&1 || true
tools/prof_bench.sh r4final/cfg2 --steps 200 --warmup 20 > /dev/null
python bench.py --steps 200 --warmup 20 > $o/bench_cfg2.json 2> $o/bench_cfg2.err
python bench.py --no-cpu-baseline --contraction bf16x3 --steps 200 --warmup 20 > $o/bench_cfg2_bf16x3.json 2>/dev/null
for c in cfg3 cfg4 cfg5; do python bench.py --no-cpu-baseline --config $c --steps 50 --warmup 5 > $o/bench_$c.json 2>/dev/null; done
python bench.py --no-cpu-baseline --config cfg5 --contraction bf16x3 --steps 50 --warmup 5 > $o/bench_cfg5_bf16x3.json 2>/dev/null
python bench.py --no-cpu-baseline --config cfg3 --tri-pack --steps 50 --warmup 5 > $o/bench_cfg3_tripack.json 2>/dev/null
tools/prof_bench.sh r4final/cfg4 --config cfg4 --steps 100 --warmup 10 > /dev/null
tools/prof_bench.sh r4final/cfg5 --config cfg5 --steps 20 --warmup 3 > /dev/null
python tools/dump_plan.py cfg5 --explain > $o/plan_cfg5.txt 2>/dev/null
python tools/dump_plan.py cfg4 --explain > $o/plan_cfg4.txt 2>/dev/null
python tools/bench_fwd.py > $o/bench_fwd_bwd.txt 2>/dev/null
python tools/bw_rows.py > $o/bw_rows.txt 2>/dev/null
python tools/dump_plan.py cfg2 --explain > $o/plan_cfg2.txt 2>/dev/null
tools/pmc_traffic.sh r4final/pmc_cfg2 > /dev/null 2>&1
tools/pmc_traffic.sh r4final/pmc_cfg4 --config cfg4 > /dev/null 2>&1
for t in chol_persist_stamps mlp_stamps xlane_cost strip_stamps; do [ -x tools/_bin/$t ] && (timeout -k 5 60 tools/_bin/$t > $o/$t.txt 2>&1 || true); done
head -c 300 $o/bench_cfg2.json
