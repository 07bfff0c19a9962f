#!/bin/bash
# Refresh the rocprofv3 evidence under profiles/ in one go (run on the GPU box from the repo root):
#   bash tools/collect_profiles.sh r5        -> gpurun_out/profiles_r5/*  (copy what should be judged into profiles/)
# Kernel-trace statistics per config, the three PMC passes over the MLP kernels (separate runs, --kernel-trace only beside --pmc),
# the two composite traffic passes, and the un-profiled microbenchmarks.  The profiled program itself follows `--`.
set -u
tag=${1:-r6}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
W=/tmp/niw_prof_$$; rm -rf $W; mkdir -p $W
stats() {   # name, bench flags...
  local name=$1; shift
  # the driver's flags (--gpus 1 --steps 20 --warmup 5) with --lean --kernel-steps 0: the side measurements of the full line (eval render,
  # PSNR-parity run at toy sizes, torch baseline) launch the same kernels at other sizes and would pollute the per-kernel averages
  rocprofv3 --kernel-trace --stats --output-format csv -d $W/$name -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --lean --kernel-steps 0 "$@" > $W/$name.log 2>&1
  local f=$(find $W/$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" > $OUT/${tag}_kernel_stats_$name.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.DictWriter(sys.stdout, fieldnames=list(rows[0].keys()))
w.writeheader()
for r in rows:
    r["Name"] = r["Name"][:150]
    w.writerow(r)
PY
  echo "stats $name: $(tail -1 $W/$name.log | cut -c1-160)"
}
stats cfg2 --config cfg2
stats cfg3 --config cfg3
stats cfg5 --config cfg5
stats cfg3_shard8 --config cfg3 --shard-of 8
stats cfg2_bf16x3 --config cfg2 --precision bf16x3
stats cfg2_bf16 --config cfg2 --precision bf16
for c in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  d=${c%%:*}; ctr=${c#*:}
  rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $W/pmc/$d -o pm -- python3 $ROOT/tools/mlp_bench.py --iters 2 --sizes 4086x192 > $W/pmc_$d.log 2>&1
  echo "pmc $d: $(ls $W/pmc/$d 2>/dev/null | head -3 | tr '\n' ' ')"
done
python3 $ROOT/tools/mlp_traffic.py $W/pmc $OUT/${tag} > $W/mlp_traffic.log 2>&1 || tail -5 $W/mlp_traffic.log
# the same three passes over the opt-in fast-precision kernels (their roofline is HBM as much as MFMA: bench.py reports both ceilings)
for prec in bf16x3 bf16; do
  for c in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "mfma:SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    d=${c%%:*}; ctr=${c#*:}
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $W/pmc_$prec/$d -o pm -- python3 $ROOT/tools/mlp_bench.py --iters 2 --sizes 4086x192 --precision $prec > $W/pmc_${prec}_$d.log 2>&1
  done
  python3 $ROOT/tools/mlp_traffic.py $W/pmc_$prec $OUT/${tag}_fast_$prec $prec > $W/mlp_traffic_$prec.log 2>&1 || tail -5 $W/mlp_traffic_$prec.log
  echo "pmc $prec: $(ls $OUT | grep fast_$prec | tr '\n' ' ')"
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $W/pmc/cfetch -o pm -- python3 $ROOT/tools/composite_bench.py --iters 3 --sizes full > $W/cfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $W/pmc/cwrite -o pm -- python3 $ROOT/tools/composite_bench.py --iters 3 --sizes full > $W/cwrite.log 2>&1
python3 $ROOT/tools/composite_traffic.py $W/pmc $OUT/${tag}_composite_traffic.json > $W/ctraffic.log 2>&1 || tail -5 $W/ctraffic.log
python3 - $ROOT $OUT/${tag}_microbench.json <<'PY'
import json, subprocess, sys
root, out = sys.argv[1], sys.argv[2]
doc = {}
for name, cmd in (("mlp_bench", ["tools/mlp_bench.py"]), ("warp_bench", ["tools/warp_bench.py"]), ("composite_bench", ["tools/composite_bench.py"])):
    r = subprocess.run([sys.executable] + [f"{root}/{cmd[0]}"] + cmd[1:], capture_output=True, text=True, cwd=root)
    doc[name] = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
json.dump(doc, open(out, "w"), indent=1)
print("microbench", {k: len(v) for k, v in doc.items()})
PY
ls -la $OUT
