"""Static check for the gfx950 store-data hazard found in round 3 (tools/store_war_hazard.hip, HISTORY.md):
a buffer_store_dwordx3/x4 whose soffset is an SGPR, followed IMMEDIATELY by a vector instruction that rewrites one of its data
registers, stores the new value when two or more waves share a SIMD.  LLVM inserts the wait state only for stores with an immediate
soffset, so hipcc can emit the pair.  This script compiles every csrc/*.hip to assembly and lists the kernels that contain it, with the
occupancy their register count allows (rounds 3-5 took kernels that can never have a second wave of their own on a SIMD to be safe; round 6
measured that a co-resident wave of ANOTHER kernel triggers the hazard too, so every adjacent site is now an error).  Round 4: the scan covers
LLVM's 2-wait-state window (an `s_nop N` counts N + 1), separating ADJACENT sites (the measured failure) from sites one instruction
further (measured safe: profiles/r4_store_hazard.jsonl, 0 of 67 M for x2 / x3 / x4), and `self_test()` proves that the scan fires on
the microbenchmark's own assembly.

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


# (any addressing form: a vector offset, `off`, or an index / offset pair -- the hazard is about the DATA registers and the SGPR soffset)
STORE = re.compile(r"\s+buffer_store_dwordx([234]) v\[(\d+):(\d+)\], (?:v\d+|v\[\d+:\d+\]|off), s\[\d+:\d+\], (s\d+|m0)\b")
WINDOW = 2          # wait states LLVM's model of the VALU-writes-store-data hazard uses on gfx940+ (an `s_nop N` counts N + 1)


def _written(instr):
    """vector registers an instruction writes (its first operand), [] for anything that is not a vector-register write"""
    if not instr.startswith("v_") or instr.startswith("v_cmp") or instr.startswith("v_mfma"):
        return []                                  # compares write SGPRs / VCC; an MFMA's result lands many cycles later
    w = re.match(r"v_\w+\s+v\[(\d+):(\d+)\]", instr)
    if w:
        return list(range(int(w.group(1)), int(w.group(2)) + 1))
    w = re.match(r"v_\w+\s+v(\d+)\b", instr)
    return [int(w.group(1))] if w else []


def scan(path, widths=(3, 4)):
    """-> {kernel: [adjacent sites, sites inside the window but not adjacent, occupancy]}.
    adjacent: a vector write of the store's data registers is the NEXT issued instruction -- the case that corrupts memory when waves
    share a SIMD (tools/store_war_hazard.hip, profiles/r4_store_hazard.jsonl: dwordx3 and dwordx4; dwordx2 never; nothing with one
    instruction or one s_nop between).  windowed: such a write within LLVM's WINDOW wait states but not adjacent -- measured safe on
    gfx950, listed so that a change of either the compiler or the measurement shows up."""
    lines = open(path).read().split("\n")
    cur, occ, hits = None, {}, {}
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
        m = re.match(r"\s*; Occupancy: (\d+)", l)
        if m and cur:
            occ[cur] = int(m.group(1))
        m = STORE.match(l)
        if not (m and cur) or int(m.group(1)) not in widths:
            continue
        lo, hi = int(m.group(2)), int(m.group(3))
        waited, j, first = 0, i + 1, True
        while j < len(lines) and waited < WINDOW:
            nxt = lines[j].strip()
            j += 1
            if not nxt or nxt.startswith(";") or nxt.endswith(":") or nxt.startswith("."):
                continue
            if any(lo <= r <= hi for r in _written(nxt)):
                rec = hits.setdefault(cur, [0, 0])
                rec[0 if first else 1] += 1
                break
            nop = re.match(r"s_nop (\d+)", nxt)
            waited += int(nop.group(1)) + 1 if nop else 1
            first = False
    return {k: [a, b, occ.get(k)] for k, (a, b) in hits.items()}


def self_test():
    """the checker must FIRE on the microbenchmark that demonstrates the hazard: the kernels of tools/store_war_hazard.hip whose clobber
    is the next instruction (template arguments <width, 0, waves>) are adjacent sites for dwordx3 / dwordx4, those with one unrelated
    instruction between are windowed sites, those with two instructions or an s_nop 0 ... s_nop between are clean or windowed -> bool"""
    with tempfile.TemporaryDirectory() as tmp:
        out = compile_to_asm(os.path.join(ROOT, "tools", "store_war_hazard.hip"), os.path.join(tmp, "hazard.s"))
        found = scan(out, widths=(2, 3, 4))
    by = {}
    for kernel, (adjacent, windowed, _) in found.items():
        m = re.search(r"ILi(\d)ELi(\d)ELi(\d)E", kernel)              # k<WIDTH, GAP, WAVES>
        if m:
            by[(int(m.group(1)), int(m.group(2)), int(m.group(3)))] = (adjacent, windowed)
    ok = True
    for width in (2, 3, 4):
        ok = ok and by.get((width, 0, 2), (0, 0))[0] >= 1 and by.get((width, 0, 1), (0, 0))[0] >= 1       # next instruction: adjacent
        ok = ok and by.get((width, 1, 2), (0, 0)) == (0, 1)                                                # one between: windowed only
        ok = ok and by.get((width, 2, 2), (0, 0)) == (0, 0) and by.get((width, 3, 2), (0, 0))[0] == 0      # two between / s_nop: not adjacent
    return ok


def main():
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    unsafe = 0
    with tempfile.TemporaryDirectory() as tmp, concurrent.futures.ThreadPoolExecutor(max_workers=4) as pool:
        outs = list(pool.map(lambda s: compile_to_asm(s, os.path.join(tmp, os.path.basename(s) + ".s")), srcs))
        for src, out in zip(srcs, outs):
            for kernel, (adjacent, windowed, occupancy) in scan(out).items():
                pinned = any(name in kernel for name in ONE_WORKGROUP_PER_CU_BY_LDS)
                # round 6: one wave per SIMD of THIS kernel is no protection -- a wave of another kernel (a second stream, another process)
                # on the SIMD triggers the hazard just the same (tools/store_war_hazard_foreign.hip): no adjacent site anywhere
                safe = adjacent == 0
                unsafe += 0 if safe else 1
                print(f"{os.path.basename(src)}: {kernel[:90]}: {adjacent} adjacent site(s), {windowed} more inside the {WINDOW}-wait-state window (measured safe), "
                      f"register occupancy {occupancy}{' (one workgroup per CU by its LDS request)' if pinned and occupancy != 1 else ''}: {'ok' if safe else 'UNSAFE'}")
    return 1 if unsafe else 0


if __name__ == "__main__":
    sys.exit(main())
