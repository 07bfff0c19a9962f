#!/usr/bin/env python3
"""What happens INSIDE a launch of the field-MLP kernels (round 6): per-wave s_memrealtime stamps of the diagnostic build
(csrc/niw_trace.h; `make -C neural_invertible_warp_amd/csrc VARIANT=trace EXTRA=-DNIW_TRACE`, selected with NIW_LIB_PATH) turned into

    dispatch skew          entry time of every workgroup after the launch's first wave (median / p90 / max; per XCD)
    time to first MFMA     entry -> end of the prologue (encodings / colour head / first slice in LDS)
    per-layer durations    median and max over the waves (multi-round launches: round 1 against rounds >= 3 on the same SIMD)
    exposed tail           how long a SIMD sits idle between its own last store and the launch's last store (mean / max)
    span vs launch         last store - first entry, against the device-event time per launch of the same back-to-back train

for the training forward, the dX chain and the NT-GEMM launches of the weight gradient, at ONE round of workgroups (252 x 128 samples:
a rank's 1/8 share of cfg3) and at eight rounds.  One line of JSON per (size, kernel); --out collects them.

    NIW_LIB_PATH=$PWD/neural_invertible_warp_amd/libniw_hip_trace.so python tools/launch_trace.py --out gpurun_out/r6_launch_trace.json
"""
import argparse
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SLOTS, TICK_US, FLOP = 16, 0.01, 2 * 527872

STAGES = {
    "fwd_train": ["prologue (points, encodings)", "layer 0", "layer 1", "layer 2", "layer 3", "layer 4", "layer 5", "layer 6", "layer 7 + density", "colour layers"],
    "bwd_dx": ["prologue (colour head)", "colour layers^T", "layer 7^T", "layer 6^T", "layer 5^T", "layer 4^T", "layer 3^T", "layer 2^T", "layer 1^T",
               "layer 0^T + encodings^T"],
    "dw": ["prologue (first slice to LDS)", "reduction over the samples", "partial tile store"],
}
DW_KINDS = {4224: "wide 256x256", 4212: "quadrant 128x128", 8112: "skinny 256x64", 4215: "colour 128x320"}


def stats(v):
    import numpy as np
    v = np.asarray(v, dtype=np.float64)
    if v.size == 0:
        return None
    return dict(min=round(float(v.min()), 2), median=round(float(np.median(v)), 2), p90=round(float(np.percentile(v, 90)), 2), max=round(float(v.max()), 2),
                mean=round(float(v.mean()), 2))


def analyse(buf, n_stages, waves_per_wg):
    """buf: [waves, 16] uint64 stamps (0 = never written) -> dict of microsecond statistics"""
    import numpy as np
    t = buf[:, :n_stages + 1].astype(np.int64)
    live = (t[:, 0] > 0) & (t[:, n_stages] > 0)
    if not live.any():
        return None
    t, hw = t[live], buf[live, 15]
    t0 = t[:, 0].min()
    us = (t - t0) * TICK_US
    entry, end = us[:, 0], us[:, n_stages]
    cyc = (buf[live, 14].astype(np.int64) - buf[live, 13].astype(np.int64)).astype(np.float64)
    ghz = cyc / np.maximum((t[:, n_stages] - t[:, 0]).astype(np.float64), 1.0) * 0.1          # cycles per 10 ns tick -> GHz
    span = float(end.max())
    xcc = ((hw >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
    cu = ((hw >> np.uint64(8)) & np.uint64(0xf)).astype(np.int64)
    sh = ((hw >> np.uint64(12)) & np.uint64(0x1)).astype(np.int64)
    se = ((hw >> np.uint64(13)) & np.uint64(0x7)).astype(np.int64)
    simd = ((hw >> np.uint64(4)) & np.uint64(0x3)).astype(np.int64)
    slot = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd               # the SIMD a wave ran on
    # round of a wave = how many waves of this launch ran on its SIMD before it
    order = np.lexsort((entry, slot))
    rnd = np.zeros(len(entry), dtype=np.int64)
    prev, k = None, 0
    for i in order:
        k = k + 1 if slot[i] == prev else 0
        prev = slot[i]
        rnd[i] = k
    dur = np.diff(us, axis=1)                                             # [waves, n_stages]
    slots = sorted(set(slot.tolist()))
    last_end = np.array([end[slot == s].max() for s in slots])
    out = dict(waves=int(live.sum()), workgroups=int(live.sum() // waves_per_wg), simds_used=len(slots), cus_used=len(set((slot // 4).tolist())),
               rounds=int(rnd.max() + 1), span_us=round(span, 2),
               dispatch_skew_us=stats(entry[rnd == 0]),
               dispatch_skew_by_xcd_us={str(x): round(float(np.median(entry[(rnd == 0) & (xcc == x)])), 2) for x in sorted(set(xcc.tolist()))},
               wave_duration_us=stats(end - entry),
               wave_duration_by_xcd_us={str(x): round(float(np.median((end - entry)[xcc == x])), 2) for x in sorted(set(xcc.tolist()))},
               exposed_tail_us=stats(span - last_end),
               shader_clock_ghz=stats(ghz), shader_clock_by_xcd_ghz={str(x): round(float(np.median(ghz[xcc == x])), 3) for x in sorted(set(xcc.tolist()))},
               stage_us={})
    first, late = rnd == 0, rnd >= 2
    for j in range(n_stages):
        out["stage_us"][j] = dict(all=stats(dur[:, j]), round_1=stats(dur[first, j]), rounds_3_on=stats(dur[late, j]) if late.any() else None)
    out["sum_of_stage_medians_us"] = round(sum(o["all"]["median"] for o in out["stage_us"].values()), 2)
    # gaps between consecutive waves of one SIMD (multi-round launches): end of one -> entry of the next
    if rnd.max() > 0:
        gaps = []
        for s in slots:
            idx = np.where(slot == s)[0]
            idx = idx[np.argsort(entry[idx])]
            gaps += [entry[b] - end[a] for a, b in zip(idx[:-1], idx[1:])]
        out["gap_between_waves_of_a_simd_us"] = stats(gaps)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="252x128,2016x128")
    ap.add_argument("--train", type=int, default=6, help="launches per back-to-back train (the last one's stamps are read)")
    ap.add_argument("--trains", default=None, help="comma-separated train lengths: every (size, kernel) is traced once per length (does a one-round "
                                                   "launch run differently after 1.5 ms of load than after 30 ms?)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import numpy as np
    import torch
    from neural_invertible_warp_amd import _lib, ops
    from oracle import niw_oracle as O
    lib = _lib.load()
    setters = {}
    for unit in ("fwd", "bwd", "dw"):
        try:
            fn = getattr(lib, f"niw_trace_set_{unit}")
        except AttributeError:
            raise SystemExit("this library has no trace entry points: build `make -C neural_invertible_warp_amd/csrc VARIANT=trace EXTRA=-DNIW_TRACE` and "
                             "set NIW_LIB_PATH to neural_invertible_warp_amd/libniw_hip_trace.so")
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int]
        setters[unit] = fn
    dev, P = "cuda:0", ops._p
    p = O.make_nerf_params(1)
    flat = torch.cat([p[f"{n}.{k}"].reshape(-1) for n, _, _ in O.nerf_layer_shapes() for k in ("weight", "bias")]).to(dev)
    st8 = ops.FieldState(flat)
    packed = st8.packed()
    results = []
    for spec in args.sizes.split(","):
        N, S = (int(x) for x in spec.split("x"))
        M, mpad = N * S, lib.niw_mlp_padded_rows(N, S)
        center = torch.randn(N, 3, device=dev) * 0.1
        ray = torch.randn(N, 3, device=dev)
        depth = (torch.rand(N, S, device=dev).sort(dim=1).values * 4 + 0.5).contiguous()
        rgb, sigma = torch.empty(N, S, 3, device=dev), torch.empty(N, S, device=dev)
        save, gradws = torch.empty(ops.SAVE_ROWS * mpad, device=dev), torch.empty(ops.GRAD_ROWS * mpad, device=dev)
        partial = torch.empty(lib.niw_mlp_bwd_workspace_floats(N, S), device=dev)
        d_params = torch.empty(ops.NERF_PARAM_FLOATS, device=dev)
        d_rgb, d_sigma = torch.randn(N, S, 3, device=dev), torch.randn(N, S, device=dev)
        dc, dr = torch.zeros(N, 3, device=dev), torch.zeros(N, 3, device=dev)
        b3, bv, st = ops._farr([1.0] * 10, 10), ops._farr([1.0] * 4, 4), ops._stream()
        fns = dict(
            fwd_train=("fwd", 0, lambda: _lib.call("niw_mlp_fwd", P(packed), P(center), P(ray), P(depth), None, N, S, b3, bv, None, 1, 0, P(rgb), P(sigma), P(save), st)),
            bwd_dx=("bwd", 0, lambda: _lib.call("niw_mlp_bwd_dx", P(packed), P(center), P(ray), P(depth), N, S, 1, 0, P(rgb), P(d_rgb), P(d_sigma), P(save), P(gradws), P(dc), P(dr), st)))
        dw = lambda: _lib.call("niw_mlp_bwd_dw", P(save), P(gradws), N, S, 0, P(partial), P(d_params), st)
        for kind in DW_KINDS:
            fns[f"dw[{DW_KINDS[kind]}]"] = ("dw", kind, dw)
        n_waves = 8 * 2048                      # more than any launch here has (wave index = linear block x waves per block)
        trace = torch.zeros(n_waves, SLOTS, dtype=torch.int64, device=dev)
        trains = [int(x) for x in args.trains.split(",")] if args.trains else [args.train]
        for name, (unit, kind, fn), train in [(n, f, tl) for n, f in fns.items() for tl in trains]:
            for _ in range(8):
                fn()
            torch.cuda.synchronize()
            trace.zero_()
            torch.cuda.synchronize()
            assert setters[unit](trace.data_ptr(), n_waves, kind) == 0
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(train):
                fn()
            b.record()
            torch.cuda.synchronize()
            assert setters[unit](None, 0, 0) == 0
            per_launch_us = a.elapsed_time(b) / train * 1e3
            buf = trace.cpu().numpy().view(np.uint64)
            stages = STAGES["dw" if unit == "dw" else name]
            res = analyse(buf, len(stages), 8 if unit == "dw" else 4)
            if res is None:
                continue                         # (this NT-GEMM shape is not launched at this size)
            res["stage_us"] = {stages[j]: v for j, v in res["stage_us"].items()}
            if name == "fwd_train":
                ok = (buf[:, 0] > 0) & (buf[:, 11] > 0)
                res["prologue_until_inputs_loaded_us"] = stats((buf[ok, 11].astype(np.int64) - buf[ok, 0].astype(np.int64)) * TICK_US)
                ok = ok & (buf[:, 12] > 0)
                res["prologue_encodings_us"] = stats((buf[ok, 12].astype(np.int64) - buf[ok, 11].astype(np.int64)) * TICK_US)
                res["prologue_encoding_stores_us"] = stats((buf[ok, 1].astype(np.int64) - buf[ok, 12].astype(np.int64)) * TICK_US)
            line = dict(kernel=name, rays=N, samples=S, mlp_evals=M, train=train, event_us_per_launch=round(per_launch_us, 2),
                        outside_the_kernel_us=round(per_launch_us - res["span_us"], 2) if unit != "dw" else None,
                        mfma_floor_us=round(M * FLOP / 157.3e12 * 1e6, 2) if unit != "dw" else None, **res)
            results.append(line)
            print(json.dumps(line), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(dict(tick="s_memrealtime, 10 ns", lib=os.environ.get("NIW_LIB_PATH"), device=torch.cuda.get_device_name(0), lines=results), f, indent=1)


if __name__ == "__main__":
    main()
