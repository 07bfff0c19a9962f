#!/bin/bash
# Rehearsal of the driver's N > 1 invocation on ONE GPU (round 6): `python bench.py --gpus N` -- the launcher parent never touches the GPU and
# starts N gloo ranks that time-slice the device -- for the default workload (cfg2), all eight LLFF scenes in both placements (cfg4) and the
# three-view DTU config (cfg5).  N defaults to 5: the pool's process guard admits six GPU processes at a time and torch.distributed.run's agent counts as one
# (six ranks: "7 processes had the GPU open (limit 6)", run killed), so EIGHT ranks cannot be run on these boxes; five still covers what four
# does not (an odd world: shares that cut views unevenly; cfg5: two ranks own no view; cfg4 replicas: 8 scenes over 5 ranks, three ranks hold two).  Keeps every line and the wall time of each run under gpurun_out/r6_ranks/.
#   bash tools/rehearse_ranks.sh [N]
set -u
N=${1:-5}
OUT=gpurun_out/r6_ranks
mkdir -p $OUT
export NIW_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
rc_all=0
for cfg in cfg2 cfg5 cfg4; do
  extra=""
  [ $cfg = cfg4 ] && extra="--kernel-steps 0"
  SECONDS=0
  timeout -k 10 900 python bench.py --gpus $N --config $cfg --steps 5 --warmup 2 $extra > $OUT/${cfg}_n$N.json 2> $OUT/${cfg}_n$N.err
  rc=$?
  echo "$cfg ranks=$N rc=$rc wall_s=$SECONDS" | tee -a $OUT/summary.txt
  [ $rc -ne 0 ] && { rc_all=$rc; tail -20 $OUT/${cfg}_n$N.err; break; }
done
exit $rc_all
