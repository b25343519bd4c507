set -e
mkdir -p gpurun_out/r2b
for s in 6 8 14 16; do HB_LBAR_FORCE_S=$s python tools/bench_bwd.py 2>/dev/null | tail -1; done > gpurun_out/r2b/lbar_S.txt
python tools/bench_bwd.py 2>/dev/null | tail -1 >> gpurun_out/r2b/lbar_S.txt
cat gpurun_out/r2b/lbar_S.txt
tools/prof_bench.sh r2b/cfg2 --steps 200 --warmup 20 > /dev/null
python bench.py --steps 200 --warmup 20 > gpurun_out/r2b/bench_cfg2.json 2> gpurun_out/r2b/bench_cfg2.err
python bench.py --no-cpu-baseline --contraction bf16x3 --steps 200 --warmup 20 > gpurun_out/r2b/bench_cfg2_bf16x3.json 2>/dev/null
for c in cfg3 cfg4 cfg5; do python bench.py --no-cpu-baseline --config $c --steps 50 --warmup 5 > gpurun_out/r2b/bench_$c.json 2>/dev/null; done
python bench.py --no-cpu-baseline --config cfg5 --contraction bf16x3 --steps 50 --warmup 5 > gpurun_out/r2b/bench_cfg5_bf16x3.json 2>/dev/null
python tools/bw_rows.py > gpurun_out/r2b/bw_rows.txt 2>/dev/null
tools/pmc_traffic.sh r2b/pmc > /dev/null 2>&1
head -c 600 gpurun_out/r2b/bench_cfg2.json
