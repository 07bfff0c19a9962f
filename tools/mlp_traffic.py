#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of tools/mlp_bench.py (csv output) into profiles/<tag>_traffic.json and profiles/<tag>_mfma_util.json.

    cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d OUT/<fetch|write|mfma> -o pm -- python3 tools/mlp_bench.py --iters 2 --sizes 4086x192
    done
    python3 tools/mlp_traffic.py OUT profiles/r2

Units and gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3 sections): FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE doubled (128-byte
requests tallied at 64 bytes); effective clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES
(64 per v_mfma_f32_32x32x2_f32) / (4 SIMDs x 256 CUs x cycles)."""
import collections
import csv
import json
import re
import sys

SAMPLES = 4086 * 192
NAMES = {"mlp_fwd_kernel<true>": "mlp_fwd_train", "mlp_fwd_kernel<false>": "mlp_fwd", "mlp_bwd_dx_kernel": "mlp_bwd_dx",
         "dw_gemm_kernel<4, 2, 2, 4": "mlp_bwd_dw_wide_batch", "dw_gemm_kernel<8, 1, 1, 2": "mlp_bwd_dw_skinny_batch",
         "dw_gemm_kernel<4, 2, 1, 5": "mlp_bwd_dw_colour", "dw_gemm_kernel<4, 1, 1, 9": "mlp_bwd_dw_colour", "dw_reduce_kernel": "mlp_bwd_dw_reduce",
         "dw_heads_kernel": "mlp_bwd_dw_heads",
         # the opt-in fast-precision kernels (tools/mlp_bench.py --precision bf16x3 | bf16)
         "mlp_fwd_fast_kernel<3, true>": "mlp_fwd_train", "mlp_fwd_fast_kernel<3, false>": "mlp_fwd", "mlp_bwd_dx_fast_kernel<3>": "mlp_bwd_dx",
         "mlp_fwd_fast_kernel<1, true>": "mlp_fwd_train", "mlp_fwd_fast_kernel<1, false>": "mlp_fwd", "mlp_bwd_dx_fast_kernel<1>": "mlp_bwd_dx",
         "dw_gemm_fast_kernel<4, 2, 2, 4": "mlp_bwd_dw_wide_batch", "dw_gemm_fast_kernel<8, 1, 1, 2": "mlp_bwd_dw_skinny_batch",
         "dw_gemm_fast_kernel<4, 2, 1, 5": "mlp_bwd_dw_colour", "dw_gemm_half_kernel<4, 2, 2, 4": "mlp_bwd_dw_wide_batch",
         "dw_gemm_half_kernel<8, 1, 1, 2": "mlp_bwd_dw_skinny_batch", "dw_gemm_half_kernel<4, 2, 1, 5": "mlp_bwd_dw_colour"}
ALGO = {  # algorithmic bytes per sample: reads / writes of the workspaces, fp32
    "mlp_fwd_train": dict(read=16.0, write=9384.0), "mlp_fwd": dict(read=16.0, write=16.0),
    # dX reads: 288 B sign masks + 384 B parked skip / view gradients (written and read back) + 384 B saved encodings (for d point,
    # d direction) + d_rgb 12 + d_sigma 4 + rgb 12 + raw density 4 + depth 4.
    # dX writes (round 3: counted by rows actually written, not by the padded row count of the workspace, which over-stated them by
    # 592 B and hid a read excess behind a "combined ratio 1.0"): dY0..dY6 7*256 rows + dY7 256 rows + 1 quad (d sigma_raw) + dYrgb0 128
    # rows + dYrgb1 2 quads + stash 96 rows = 2284 rows * 4 B = 9136 B, + 32 B of parked per-sample ray gradients.
    "mlp_bwd_dx": dict(read=1092.0, write=9136.0 + 32.0), "mlp_bwd_dw_wide_batch": dict(read=7 * 2 * 256 * 4.0, write=0.0),
    # exact mode from 131 k samples: the skinny launch holds the two encoding-column pieces; the density row and the colour rows are
    # dw_heads_kernel's (h7 256 rows + hr 128 rows + the two dY quads).  Fast modes: four-piece skinny launch, no heads kernel.
    "mlp_bwd_dw_skinny_batch": dict(read=(320 + 320) * 4.0, write=0.0), "mlp_bwd_dw_heads": dict(read=(256 + 128 + 4 + 4) * 4.0, write=0.0),
    "mlp_bwd_dw_colour": dict(read=(128 + 288) * 4.0, write=0.0)}
SKINNY_FOUR_PIECES = (320 + 320 + 257 + 131) * 4.0


def per_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        m = re.search(r"(mlp_fwd_kernel<\w+>|mlp_bwd_dx_kernel|mlp_fwd_fast_kernel<\d, \w+>|mlp_bwd_dx_fast_kernel<\d>|dw_gemm(?:_fast|_half)?_kernel<\d, \d, \d, \d|"
                      r"dw_reduce_kernel|dw_heads_kernel)", r["Kernel_Name"])
        if m:
            agg[NAMES[m.group(1)]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            agg[NAMES[m.group(1)]]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


def bf16_algo():
    """NIW_PREC_BF16 keeps every row the dW pass multiplies as bf16 (niw_mlp_fast.hip kHalfWorkspace): activations / dY rows 2 bytes per
    sample instead of 4; the raw density, the sign masks and the stash rows stay fp32"""
    a = {k: dict(v) for k, v in ALGO.items()}
    act_rows = 64 + 7 * 256 + 256 + 32 + 128                                   # enc, h1..h7, feat, venc, hr
    a["mlp_fwd_train"] = dict(read=16.0, write=act_rows * 2.0 + 2 * 4.0 + 288.0)
    dy_rows = 7 * 256 + 256 + 4 + 128 + 8                                       # dY0..dY6, dY7 (+ d sigma quad), dYrgb0, dYrgb1 (2 quads)
    a["mlp_bwd_dx"] = dict(read=288.0 + 384.0 + 192.0 + 12 + 4 + 12 + 4 + 4, write=dy_rows * 2.0 + 96 * 4.0 + 32.0)
    for k in ("mlp_bwd_dw_wide_batch", "mlp_bwd_dw_skinny_batch", "mlp_bwd_dw_colour"):
        a[k] = dict(read=ALGO[k]["read"] / 2, write=0.0)
    return a


def main(out, prefix, precision="fp32"):
    global ALGO
    if precision != "fp32":
        ALGO["mlp_bwd_dw_skinny_batch"] = dict(read=SKINNY_FOUR_PIECES, write=0.0)
    if precision == "bf16":
        ALGO = bf16_algo()
    f, w, m = (per_kernel(f"{out}/{d}/pm_counter_collection.csv") for d in ("fetch", "write", "mfma"))
    traffic = {}
    for k in f:
        if k not in ALGO:
            continue
        fr, wr = f[k]["FETCH_SIZE"] * 1024 / SAMPLES, w[k]["WRITE_SIZE"] * 1024 / SAMPLES
        traffic[k] = dict(fetch_raw=round(fr, 1), fetch_corrected=round(2 * fr, 1), write=round(wr, 1), algorithmic_read=ALGO[k]["read"],
                          algorithmic_write=ALGO[k]["write"], traffic_over_algorithmic=round((2 * fr + wr) / (ALGO[k]["read"] + ALGO[k]["write"]), 3),
                          # the two directions separately: a combined ratio can hide a read excess behind an over-stated write figure
                          read_over_algorithmic=round(2 * fr / ALGO[k]["read"], 3),
                          write_over_algorithmic=round(wr / ALGO[k]["write"], 3) if ALGO[k]["write"] else None,
                          read_excess_bytes_per_sample=round(2 * fr - ALGO[k]["read"], 1))
    doc = dict(precision=precision,
               source=f"rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/mlp_bench.py --iters 2 --sizes 4086x192 --precision {precision}; "
                      "MI355X; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a coalesced stream), WRITE_SIZE as reported; KiB = 1024 B",
               samples_per_launch=SAMPLES, bytes_per_sample=traffic,
               note="read excess of the register-chained kernels = weight fragments re-fetched through the fabric: the 2.1 MB transposed / forward "
                    "image competes for the XCD's 4 MiB L2 with the kernel's own streaming stores (9.4 KB per sample; nt) -- ~13 B/sample without "
                    "stores (eval forward), a few hundred with them; served by the Infinity Cache (FETCH_SIZE counts fabric requests, cache hits "
                    "included), 3-6 % of the kernel's traffic and far from any bandwidth limit")
    json.dump(doc, open(prefix + "_traffic.json", "w"), indent=1)
    rows = []
    for k, d in m.items():
        sec, cyc = d["dur_ns"] * 1e-9, d["GRBM_GUI_ACTIVE"] / 8
        rows.append(dict(kernel=k, avg_ms=round(sec * 1e3, 3), clock_ghz=round(cyc / sec / 1e9, 3),
                         mfma_util=round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * cyc), 4),
                         frac_of_2p4ghz_peak=round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * sec * 2.4e9), 4)))
    json.dump(dict(precision=precision, source=f"rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/mlp_bench.py --iters 2 --sizes 4086x192 --precision {precision} "
                          "(profiled passes run a few per cent slower than un-profiled ones)", kernels=rows), open(prefix + "_mfma_util.json", "w"), indent=1)
    print(json.dumps(traffic, indent=1))
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "fp32")
