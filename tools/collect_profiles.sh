#!/bin/bash
# Copy the outputs of tools/final_measurements.sh (gpurun_out/r4final) into profiles/ under their round-4 names.
set -e
s=gpurun_out/r4final
p=profiles
cp $s/stream_peak.txt $p/r04_stream_peak.txt
cp $s/bench_cfg2.json $p/r04_bench_cfg2_1gpu.json
for c in cfg2_bf16x3 cfg3 cfg3_tripack cfg4 cfg5 cfg5_bf16x3; do cp $s/bench_$c.json $p/r04_bench_${c}_1gpu.json; done
cp $s/cfg2/kernel_stats.csv $p/r04_rocprofv3_kernel_stats_bench_cfg2.csv
cp $s/cfg2/summary.txt $p/r04_rocprofv3_summary_bench_cfg2.txt
cp $s/cfg4/kernel_stats.csv $p/r04_rocprofv3_kernel_stats_bench_cfg4.csv
cp $s/cfg4/summary.txt $p/r04_rocprofv3_summary_bench_cfg4.txt
cp $s/cfg5/kernel_stats.csv $p/r04_rocprofv3_kernel_stats_bench_cfg5.csv
cp $s/cfg5/summary.txt $p/r04_rocprofv3_summary_bench_cfg5.txt
for c in 2 4 5; do grep -v "amdgpu.ids\|^compiling\|^finished" $s/plan_cfg$c.txt > $p/r04_plan_cfg$c.txt; done
grep -v amdgpu.ids $s/bench_fwd_bwd.txt > $p/r04_contractions_fwd_bwd.txt
cp $s/bw_rows.txt $p/r04_bw_rows.txt
cp $s/pmc_cfg2/pmc_kernels.txt $p/r04_pmc_cfg2_kernels.txt
cp $s/pmc_cfg4/pmc_kernels.txt $p/r04_pmc_cfg4_kernels.txt
python tools/make_traffic_json.py cfg2=$s/pmc_cfg2 cfg4=$s/pmc_cfg4 > $p/r04_pmc_traffic.json
cp $s/chol_persist_stamps.txt $p/r04_chol_persist_stamps.txt
cp $s/mlp_stamps.txt $p/r04_mlp_bwd_stamps.txt
cp $s/strip_stamps.txt $p/r04_strip3_stamps.txt
cp $s/xlane_cost.txt $p/r04_xlane_cost.txt
ls -la $p | grep r04
