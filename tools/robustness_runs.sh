#!/bin/bash
# Bitwise replay / training determinism, the same under contention (three processes on the one GPU), the two-rank
# rehearsal on one device and the long run -- on the build as it stands.  Outputs: gpurun_out/r4robust/.
o=gpurun_out/r4robust
mkdir -p $o
for c in cfg2 cfg5 cfg4 cfg3; do
  timeout -k 10 120 python tools/replay_determinism.py 12 x $c > $o/replay_$c.log 2>&1
  grep "replays\|first non-finite\|Error" $o/replay_$c.log | tail -2
done
echo "three copies at once:"
(timeout -k 10 150 python tools/replay_determinism.py 20 A cfg2 > $o/cont_A.log 2>&1 &)
(timeout -k 10 150 python tools/replay_determinism.py 20 B cfg2 > $o/cont_B.log 2>&1 &)
timeout -k 10 150 python tools/replay_determinism.py 20 C cfg5 > $o/cont_C.log 2>&1
sleep 20
for p in A B C; do grep "replays\|first non-finite\|Error" $o/cont_$p.log | tail -2; done
timeout -k 10 200 python tools/train_determinism.py 25 b > $o/train.log 2>&1
grep -i "steps\|differ\|Error" $o/train.log | tail -3
timeout -k 10 200 python tools/long_run.py > $o/long_run.log 2>&1
tail -2 $o/long_run.log
HENBUN_ONE_DEVICE=1 timeout -k 10 300 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp_rehearsal.py > $o/dp_rehearsal.log 2>&1
tail -6 $o/dp_rehearsal.log
