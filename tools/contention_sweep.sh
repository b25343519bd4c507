# two copies of tools/replay_determinism.py per configuration, at the same time on one GPU
for cfg in cfg2 cfg2_f64 cfg3 cfg3_f64 ragged_f64 cfg4 cfg5; do
  for p in A B; do (timeout -k 10 200 python tools/replay_determinism.py 8 proc$p $cfg > gpurun_out/sweep_${cfg}_$p.log 2>&1 &); done
  sleep 45
  for p in A B; do grep "replays\|first non-finite\|Error" gpurun_out/sweep_${cfg}_$p.log | tail -3; done
done
