cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r6}_lines; TAG=${1:-r6}; rm -rf $O; mkdir -p $O
run() { name=$1; shift; python bench.py "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run cfg2 --gpus 1 --steps 20 --warmup 5
run cfg2_launched --gpus 1 --steps 20 --warmup 5 --hip-graph off --lean
run cfg1 --config cfg1 --steps 40 --lean
python bench.py --config cfg1 --steps 40 --lean --fused-step off > $O/cfg1_mirror.json 2> $O/cfg1_mirror.err; echo "cfg1_mirror rc=$?"
run cfg3 --config cfg3 --steps 50 --lean
run cfg3_launched --config cfg3 --steps 50 --lean --hip-graph off
run cfg3_serial --config cfg3 --steps 50 --lean --hip-graph off --overlap off
run cfg4 --config cfg4 --steps 10 --lean
run cfg4_horns --config cfg4-horns --steps 50 --lean
run cfg4_orchids --config cfg4-orchids --steps 50 --lean
run cfg5 --config cfg5 --steps 50 --lean
run cfg3_shard8 --config cfg3 --shard-of 8 --steps 100 --lean
run cfg3_shard8_launched --config cfg3 --shard-of 8 --steps 100 --lean --hip-graph off
run cfg2_shard8 --config cfg2 --shard-of 8 --steps 50 --lean
run cfg2_shard8_launched --config cfg2 --shard-of 8 --steps 50 --lean --hip-graph off
run cfg2_weak8 --config cfg2 --shard-of 8 --scaling weak --steps 20 --lean --hip-graph off
run cfg3_rccl1 --config cfg3 --force-dist --steps 50 --lean
run cfg2_rccl1_split --config cfg2 --force-dist --steps 20 --lean
run cfg2_rccl1_flat --config cfg2 --force-dist --steps 20 --lean --split-exchange off
run cfg4_replicas_of8 --config cfg4 --placement replicas --shard-of 8 --steps 50 --lean
run cfg4_shard_of8 --config cfg4 --placement shard --shard-of 8 --steps 10 --lean
run cfg2_bf16x3 --precision bf16x3 --steps 20 --no-cpu-baseline --no-torch-baseline --no-forward-only --no-composite-scan
run cfg2_bf16 --precision bf16 --steps 20 --no-cpu-baseline --no-torch-baseline --no-forward-only --no-composite-scan
# the driver's N > 1 invocation rehearsed with FIVE gloo ranks on this one GPU (tools/rehearse_ranks.sh: the process guard admits no more)
for cfg in cfg2 cfg5 cfg4; do
  extra=""; [ $cfg = cfg4 ] && extra="--kernel-steps 0"
  SECONDS=0
  NIW_DIST_BACKEND=gloo timeout -k 10 900 python bench.py --gpus 5 --config $cfg --steps 5 --warmup 2 $extra > $O/n5_gloo_$cfg.json 2> $O/n5_gloo_$cfg.err; echo "n5 $cfg rc=$? wall_s=$SECONDS"
done
python - $O <<'PY'
import json, sys, glob, os
out = {}
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    lines = [x for x in open(f).read().splitlines() if x.startswith("{")]
    if lines:
        out[os.path.basename(f)[:-5]] = json.loads(lines[-1])
json.dump(out, open(sys.argv[1] + "/bench_lines.json", "w"), indent=1)
for k, l in out.items():
    print(f"{k:24s} {l['ms_per_step']:9.4f} ms  {l['value']/1e6:7.2f} M/s  frac {l['frac_of_train_roofline']:.4f}  graph {l['hip_graph']}  comm {l.get('comm_ms')} exposed {l.get('comm_exposed_ms')}  strong {(l.get('strong') or {}).get('ms_per_step')}  replicas {(l.get('replicas') or {}).get('ms_per_step')}")
PY
rocprofv3 --kernel-trace -d $O/t3 -o t3 --output-format csv -- python3 bench.py --config cfg3 --lean --steps 10 --hip-graph off --kernel-steps 0 > $O/t3.log 2>&1
rocprofv3 --kernel-trace -d $O/t3s8 -o t3s8 --output-format csv -- python3 bench.py --config cfg3 --lean --steps 10 --shard-of 8 --hip-graph off --kernel-steps 0 > $O/t3s8.log 2>&1
rocprofv3 --kernel-trace -d $O/t2 -o t2 --output-format csv -- python3 bench.py --config cfg2 --lean --steps 6 --hip-graph off --kernel-steps 0 > $O/t2.log 2>&1
rocprofv3 --kernel-trace -d $O/t2s8 -o t2s8 --output-format csv -- python3 bench.py --config cfg2 --lean --steps 10 --shard-of 8 --hip-graph off --kernel-steps 0 > $O/t2s8.log 2>&1
python tools/iteration_timeline.py $O/t2s8/t2s8_kernel_trace.csv > $O/timeline_cfg2_shard8.txt
python tools/iteration_timeline.py $O/t3/t3_kernel_trace.csv > $O/timeline_cfg3.txt
python tools/iteration_timeline.py $O/t3s8/t3s8_kernel_trace.csv > $O/timeline_cfg3_shard8.txt
python tools/iteration_timeline.py $O/t2/t2_kernel_trace.csv > $O/timeline_cfg2.txt
rm -rf $O/t3 $O/t3s8 $O/t2 $O/t2s8
