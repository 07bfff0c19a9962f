"""Runs bench.py for every BASELINE config (plus the 1/8-shard strong-scaling proxies and the eager variant) as child processes
and collects their JSON lines into one file:  python tools/bench_all.py gpurun_out/bench_lines.json [--steps 20 --warmup 5]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = {
    "cfg2": ["--config", "cfg2"],
    "cfg3": ["--config", "cfg3", "--lean"],
    "cfg3_eager": ["--config", "cfg3", "--lean", "--hip-graph", "off"],
    "cfg4": ["--config", "cfg4", "--lean"],
    "cfg4-horns": ["--config", "cfg4-horns", "--lean"],
    "cfg4-orchids": ["--config", "cfg4-orchids", "--lean"],
    "cfg5": ["--config", "cfg5", "--lean"],
    "cfg3_shard8": ["--config", "cfg3", "--shard-of", "8", "--lean", "--steps", "100"],
    "cfg3_shard8_eager": ["--config", "cfg3", "--shard-of", "8", "--lean", "--steps", "100", "--hip-graph", "off"],
    "cfg2_shard8": ["--config", "cfg2", "--shard-of", "8", "--lean", "--steps", "50"],
    # the opt-in fast-precision modes (separate lines, never the headline)
    "cfg2_bf16x3": ["--config", "cfg2", "--precision", "bf16x3"],
    "cfg2_bf16": ["--config", "cfg2", "--precision", "bf16"],
    "cfg3_bf16x3": ["--config", "cfg3", "--lean", "--precision", "bf16x3"],
    "cfg5_bf16x3": ["--config", "cfg5", "--lean", "--precision", "bf16x3"],
    # one rank through RCCL (a forced one-rank process group): the collective path on hardware
    "cfg3_rccl_1rank": ["--config", "cfg3", "--lean", "--force-dist"],
}


def main():
    out_path = sys.argv[1]
    extra = sys.argv[2:] or ["--steps", "20", "--warmup", "5"]
    only = os.environ.get("NIW_BENCH_ONLY")
    lines = {}
    for name, flags in RUNS.items():
        if only and name not in only.split(","):
            continue
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + extra + flags
        r = subprocess.run(cmd, capture_output=True, text=True)
        js = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode or not js:
            lines[name] = dict(error=r.returncode, stderr=r.stderr[-2000:])
            print(name, "FAILED", r.stderr[-500:], flush=True)
        else:
            lines[name] = json.loads(js[-1])
            d = lines[name]
            print(f"{name:20s} {d['ms_per_step']:8.3f} ms/step {d['value'] / 1e6:7.2f} M/s frac {d['frac_of_train_roofline']:.3f} loss {d['loss']:.4f} graph {d['hip_graph']}", flush=True)
        with open(out_path, "w") as f:
            json.dump(lines, f, indent=1)


if __name__ == "__main__":
    main()
