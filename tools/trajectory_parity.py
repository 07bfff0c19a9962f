#!/usr/bin/env python3
"""Long-horizon parity evidence (VERDICT r2 item 8 i; diagnostics): the demo scene trained for `--steps` chained iterations on the HIP
engine and on the oracle (run on the same GPU through torch's kernels) with identical weights and identical random draws, plus HIP
runs that differ only in the draws, whose spread is the yardstick for "the same trajectory" (oracle/parity.long_trajectories).

    python tools/trajectory_parity.py [--steps 1000] [--out gpurun_out/trajectory_parity.json]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--spread-runs", type=int, default=3)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--size", type=int, nargs=2, default=[48, 64], help="image size H W (300 400: the BASELINE image scale)")
    ap.add_argument("--rays", type=int, default=256, help="rays per view and step")
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16"], help="field-MLP arithmetic of the HIP engine (the oracle stays fp32)")
    ap.add_argument("--out", default="gpurun_out/trajectory_parity.json")
    a = ap.parse_args()
    from oracle import parity
    t0 = time.perf_counter()
    kw = dict(views=a.views, size=tuple(a.size), R=a.rays, S=a.samples, precision=a.precision)
    pair = parity.long_trajectories("cuda:0", steps=a.steps, draw_seed=0, oracle=True, **kw)
    print(f"HIP + oracle, {a.steps} steps: {time.perf_counter() - t0:.1f} s", flush=True)
    spread = [parity.long_trajectories("cuda:0", steps=a.steps, draw_seed=s, oracle=False, **kw)["hip"] for s in range(1, 1 + a.spread_runs)]
    n = len(pair["it"])
    rows = []
    for i in range(n):
        ps = [pair["hip"]["psnr"][i]] + [r["psnr"][i] for r in spread]
        rr = [pair["hip"]["rel_rot"][i]] + [r["rel_rot"][i] for r in spread]
        rows.append(dict(it=pair["it"][i], psnr_hip=round(pair["hip"]["psnr"][i], 3), psnr_oracle=round(pair["oracle"]["psnr"][i], 3),
                         psnr_draw_spread=round(max(ps) - min(ps), 3), rel_rot_hip=round(pair["hip"]["rel_rot"][i], 3),
                         rel_rot_oracle=round(pair["oracle"]["rel_rot"][i], 3), rel_rot_draw_spread=round(max(rr) - min(rr), 3)))
    tail = rows[len(rows) // 2:]
    summary = dict(steps=a.steps, precision=a.precision, shape=f"{a.views} views x {a.rays} rays x {a.samples} samples on {a.size[0]}x{a.size[1]} images", max_abs_psnr_diff_db=round(max(abs(r["psnr_hip"] - r["psnr_oracle"]) for r in rows), 3),
                   max_psnr_draw_spread_db=round(max(r["psnr_draw_spread"] for r in rows), 3),
                   second_half_mean_abs_psnr_diff_db=round(sum(abs(r["psnr_hip"] - r["psnr_oracle"]) for r in tail) / len(tail), 3),
                   second_half_mean_psnr_draw_spread_db=round(sum(r["psnr_draw_spread"] for r in tail) / len(tail), 3),
                   final=rows[-1], seconds=round(time.perf_counter() - t0, 1))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(dict(summary=summary, rows=rows), f, indent=1)
    print(json.dumps(summary))
