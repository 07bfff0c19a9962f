#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (rocpd sqlite output) of `bench.py` into profiles/r1_traffic.json and
profiles/r1_mfma_util.json.  Passes (each its own run, `--kernel-trace` only, as the guide prescribes):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d OUT/fetch -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d OUT/write -o pmc -- python3 bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d OUT/mfma -o pmc -- python3 bench.py ...
    python tools/pmc_summary.py OUT

gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in KiB... here the
values are taken as reported in units of 1 KiB?  No: rocprofv3 reports both in kilobytes (1024 B); FETCH_SIZE under-counts a
coalesced stream by 2x on gfx950 and is doubled; WRITE_SIZE is used as reported.
"""
import collections
import glob
import json
import os
import sqlite3
import sys

FINE = 784512          # samples of the fine-network launches of the bench (18 x 227 x 192)


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    d = collections.defaultdict(lambda: [0, 0.0, 0.0, 0])
    for name, val, dur, grid in c.execute("select kernel_name, value, duration, grid_size from counters_collection where counter_name=?", (counter,)):
        n = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        k = (n, grid)
        d[k][0] += 1; d[k][1] += val; d[k][2] += dur
    return d


def find(out, sub):
    f = glob.glob(os.path.join(out, sub, "**", "*.db"), recursive=True)
    return f[0] if f else None


def main(out):
    res = {}
    fdb, wdb, mdb = find(out, "fetch"), find(out, "write"), find(out, "mfma")
    if fdb and wdb:
        fetch, write = per_kernel(fdb, "FETCH_SIZE"), per_kernel(wdb, "WRITE_SIZE")
        rows = {}
        for (n, grid), (cnt, val, dur) in ((k, v[:3]) for k, v in fetch.items()):
            if not n.startswith(("mlp_", "dw_gemm", "composite")):
                continue
            w = write.get((n, grid), [1, 0.0, 0.0])
            rows[f"{n}@grid{grid}"] = dict(launches=cnt, fetch_raw_kb_per_launch=val / cnt, fetch_corrected_kb_per_launch=2 * val / cnt,
                                            write_kb_per_launch=w[1] / max(w[0], 1))
        res["traffic"] = rows
    if mdb:
        busy, gui = per_kernel(mdb, "SQ_VALU_MFMA_BUSY_CYCLES"), per_kernel(mdb, "GRBM_GUI_ACTIVE")
        agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
        for (n, grid), (cnt, val, dur, _) in busy.items():
            g = gui[(n, grid)]
            a = agg[n]
            a[0] += cnt; a[1] += val; a[2] += g[1]; a[3] += dur
        rows = []
        for n, (cnt, mf, gu, dur) in sorted(agg.items(), key=lambda kv: -kv[1][3]):
            if not n.startswith(("mlp_", "dw_", "warp_", "composite")):
                continue
            cyc, sec = gu / 8, dur * 1e-9
            rows.append(dict(kernel=n, launches=cnt, total_ms=round(sec * 1e3, 3), clock_ghz=round(cyc / sec / 1e9, 3),
                             mfma_util=round(mf / (4 * 256 * cyc), 4), frac_of_2p4ghz_peak=round(mf / (4 * 256 * sec * 2.4e9), 4)))
        res["mfma"] = rows
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
