"""Static check for the gfx950 store-data hazard found in round 3 (tools/store_war_hazard.hip):
a buffer_store_dwordx3/x4 whose soffset is an SGPR, followed IMMEDIATELY by a vector instruction that rewrites one of its data
registers, stores the new value when two or more waves share a SIMD.  LLVM inserts the wait state only for stores with an immediate
soffset, so hipcc can emit the pair.  This script compiles every csrc/*.hip to assembly and lists the kernels that contain it, with the
occupancy their register count allows; kernels that can never have a second wave on their SIMD are safe.

    python tools/check_store_hazard.py            -> prints one line per (kernel, sites, occupancy), exit code 1 if an unsafe one exists
"""
import concurrent.futures
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neural_invertible_warp_amd", "csrc")
# kernels launched with an LDS request that admits ONE workgroup per CU, if any (none at present: the fast-precision kernels, which had the
# pair, now pass their row offsets through the descriptor and store with soffset 0, which the compiler pads)
ONE_WORKGROUP_PER_CU_BY_LDS = ()


def compile_to_asm(src, out):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", f"-I{CSRC}", "-mllvm",
           "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def scan(path):
    """-> {kernel: [sites, occupancy]}"""
    lines = open(path).read().split("\n")
    cur, occ, hits = None, {}, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
        m = re.match(r"\s*; Occupancy: (\d+)", l)
        if m and cur:
            occ[cur] = int(m.group(1))
        m = re.match(r"\s+buffer_store_dwordx[34] v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\], (s\d+|m0)\b", l)
        if not (m and cur):
            continue
        lo, hi = int(m.group(1)), int(m.group(2))
        j = i + 1
        while j < len(lines) and (not lines[j].strip() or lines[j].strip().startswith(";")):
            j += 1
        nxt = lines[j].strip() if j < len(lines) else ""
        if not nxt.startswith("v_") or nxt.startswith("v_cmp") or nxt.startswith("v_mfma"):
            continue                                   # compares write SGPRs; an MFMA's result lands many cycles later
        w = re.match(r"v_\w+\s+v\[(\d+):(\d+)\]", nxt)
        written = range(int(w.group(1)), int(w.group(2)) + 1) if w else None
        if written is None:
            w = re.match(r"v_\w+\s+v(\d+)\b", nxt)
            written = [int(w.group(1))] if w else []
        if any(lo <= r <= hi for r in written):
            hits[cur] = hits.get(cur, 0) + 1
    return {k: [n, occ.get(k)] for k, n in hits.items()}


def main():
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    unsafe = 0
    with tempfile.TemporaryDirectory() as tmp, concurrent.futures.ThreadPoolExecutor(max_workers=4) as pool:
        outs = list(pool.map(lambda s: compile_to_asm(s, os.path.join(tmp, os.path.basename(s) + ".s")), srcs))
        for src, out in zip(srcs, outs):
            for kernel, (sites, occupancy) in scan(out).items():
                pinned = any(name in kernel for name in ONE_WORKGROUP_PER_CU_BY_LDS)
                safe = occupancy == 1 or pinned
                unsafe += 0 if safe else 1
                print(f"{os.path.basename(src)}: {kernel[:90]}: {sites} site(s), register occupancy {occupancy}"
                      f"{' (one workgroup per CU by its LDS request)' if pinned and occupancy != 1 else ''}: {'ok' if safe else 'UNSAFE'}")
    return 1 if unsafe else 0


if __name__ == "__main__":
    sys.exit(main())
