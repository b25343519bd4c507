#!/bin/bash
# Every measurement the round's profiles/ files come from, in one GPU call:  gpurun -- 'bash tools/final_measurements.sh'
# Outputs under gpurun_out/r2final/ (copy what is to be kept into profiles/r02_*).
set -e
o=gpurun_out/r2final
mkdir -p $o
tools/prof_bench.sh r2final/cfg2 --steps 200 --warmup 20 > /dev/null
python bench.py --steps 200 --warmup 20 > $o/bench_cfg2.json 2> $o/bench_cfg2.err
python bench.py --no-cpu-baseline --contraction bf16x3 --steps 200 --warmup 20 > $o/bench_cfg2_bf16x3.json 2>/dev/null
for c in cfg3 cfg4 cfg5; do python bench.py --no-cpu-baseline --config $c --steps 50 --warmup 5 > $o/bench_$c.json 2>/dev/null; done
python bench.py --no-cpu-baseline --config cfg5 --contraction bf16x3 --steps 50 --warmup 5 > $o/bench_cfg5_bf16x3.json 2>/dev/null
python bench.py --no-cpu-baseline --config cfg3 --tri-pack --steps 50 --warmup 5 > $o/bench_cfg3_tripack.json 2>/dev/null
python tools/bw_rows.py > $o/bw_rows.txt 2>/dev/null
python tools/dump_plan.py cfg2 > $o/plan_cfg2.txt 2>/dev/null
tools/pmc_traffic.sh r2final/pmc_cfg2 > /dev/null 2>&1
tools/pmc_traffic.sh r2final/pmc_cfg3 --config cfg3 > /dev/null 2>&1
for t in chol_stamps ss rs; do [ -x tools/_bin/$t ] && (timeout -k 5 60 tools/_bin/$t > $o/$t.txt 2>&1 || true); done
tools/trace_rider.sh > $o/rider_trace.txt 2>&1 || true
python tools/jit_vs_interp.py > $o/jit_vs_interp.txt 2>/dev/null || true
head -c 300 $o/bench_cfg2.json
