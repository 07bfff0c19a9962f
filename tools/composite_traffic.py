#!/usr/bin/env python3
"""Summarise the rocprofv3 PMC passes of tools/composite_bench.py (csv output) into profiles/r2_composite_traffic.json:
per compositing kernel and size, HBM bytes per launch from the counters next to the algorithmic bytes and the kernel duration
of the same dispatches.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT/cfetch -o pm -- python3 tools/composite_bench.py --iters 3 --sizes full
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d OUT/cwrite -o pm -- python3 tools/composite_bench.py --iters 3 --sizes full
    python3 tools/composite_traffic.py OUT profiles/r2_composite_traffic.json

Units and gfx950 corrections (MI355X_MICROARCH.md, HBM section): both counters are reported in KiB; FETCH_SIZE tallies the 128-byte
requests of a wide coalesced read at 64 bytes and is doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores."""
import collections
import csv
import json
import re
import sys


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        m = re.search(r"(composite_(?:fwd|bwd)(?:_scalar|_span)?_kernel(?:<\d+(?:, \w+)*>)?)", r["Kernel_Name"])
        if not m:
            continue
        a = agg[(m.group(1), int(r["Grid_Size"]))]
        a[0] += 1
        a[1] += float(r["Counter_Value"]) * 1024.0
        a[2] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return {k: (n, b / n, t / n) for k, (n, b, t) in agg.items()}


def main(out_dir, dest):
    fetch, write = per_kernel(f"{out_dir}/cfetch/pm_counter_collection.csv"), per_kernel(f"{out_dir}/cwrite/pm_counter_collection.csv")
    rows = []
    for (name, grid), (n, fb, ft) in sorted(fetch.items()):
        if (name, grid) not in write:
            continue
        _, wb, wt = write[(name, grid)]
        G = int(re.search(r"<(\d+)", name).group(1)) if "<" in name else 64
        # the bench's full-image sizes are 120,000 rays x {64, 128, 192}: span kernels (round 5) carry the quads per lane Q = S / 64 as their
        # first template argument, the one-quad kernels of rounds 2-4 the lanes per ray G = S / 4 rounded up to a power of two
        S = 64 * G if "_span" in name else {16: 64, 32: 128, 64: 192}.get(G)
        N = 120000
        bwd = "bwd" in name
        algo_read = N * S * 20 + N * (24 if bwd else 12)
        algo_write = N * S * (16 if bwd else 4) + N * (12 if bwd else 20)
        dur = 0.5 * (ft + wt)
        rows.append(dict(kernel=name, n_rays=N, samples=S, launches_profiled=n,
                         fetch_size_bytes=round(fb), fetch_corrected_bytes=round(2 * fb), write_size_bytes=round(wb),
                         algorithmic_read_bytes=algo_read, algorithmic_write_bytes=algo_write,
                         traffic_over_algorithmic=round((2 * fb + wb) / (algo_read + algo_write), 4),
                         avg_duration_us=round(dur / 1e3, 1),
                         hbm_gbps_from_counters=round((2 * fb + wb) / dur, 1), algorithmic_gbps=round((algo_read + algo_write) / dur, 1),
                         frac_of_6290_achievable=round((algo_read + algo_write) / dur / 6290.0, 4), frac_of_8000_peak=round((algo_read + algo_write) / dur / 8000.0, 4)))
    doc = dict(source="rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/composite_bench.py --iters 3 --sizes full",
               corrections="counters in KiB; FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE as reported",
               note="durations are the profiled dispatches' own End-Start timestamps (profiled passes clock a little lower than un-profiled ones)",
               kernels=rows)
    with open(dest, "w") as f:
        json.dump(doc, f, indent=1)
    for r in rows:
        print(r["kernel"], r["samples"], r["avg_duration_us"], "us", r["algorithmic_gbps"], "GB/s algorithmic", r["hbm_gbps_from_counters"], "GB/s counters",
              "x%.3f" % r["traffic_over_algorithmic"])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
