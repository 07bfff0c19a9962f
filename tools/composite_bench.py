"""Compositing scan alone (niw_composite_fwd / niw_composite_bwd through the C ABI) at the sizes that matter:
a full 300x400 image (120,000 rays x 64 / 128 / 192 samples: the eval path, 0.15-0.56 GB per launch) and the
cfg2 / cfg3 training launches.  Prints one JSON line per size with the achieved ALGORITHMIC GB/s
(bytes the operation must move / device-event time) against the 6.29 TB/s streaming-copy rate of MI355X.

    python3 tools/composite_bench.py [--iters 20] [--sizes full|train|all] [--out FILE]

PMC passes for profiles/r2_composite_traffic.json (separate runs, kernel trace only):
    rocprofv3 --kernel-trace --pmc FETCH_SIZE  -d gpurun_out/cfetch -- python3 tools/composite_bench.py --iters 3 --sizes full
    rocprofv3 --kernel-trace --pmc WRITE_SIZE  -d gpurun_out/cwrite -- python3 tools/composite_bench.py --iters 3 --sizes full
NIW_LIB_PATH selects a diagnostic build (e.g. -DNIW_COMPOSITE_NO_LDS)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FULL = [(120000, 64), (120000, 128), (120000, 192)]
TRAIN = [(4086, 64), (4086, 192), (2034, 128), (2046, 128)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--sizes", default="all")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    from neural_invertible_warp_amd import ops
    dev = "cuda:0"
    sizes = FULL if args.sizes == "full" else TRAIN if args.sizes == "train" else FULL + TRAIN
    lines = []
    from neural_invertible_warp_amd import _lib
    P = ops._p
    for N, S in sizes:
        gen = torch.Generator(device=dev).manual_seed(N + S)
        ray = torch.randn(N, 3, device=dev, generator=gen)
        rgb_s = torch.rand(N, S, 3, device=dev, generator=gen)
        sig = torch.rand(N, S, device=dev, generator=gen) * 2
        dep = (torch.rand(N, S, device=dev, generator=gen) * 0.9 / S + torch.arange(S, device=dev) / S + 1.0).contiguous()
        g_rgb = torch.randn(N, 3, device=dev, generator=gen)
        rgb, depth, opa, prob = torch.empty(N, 3, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, S, device=dev)
        d_rgb_s, d_sig, d_ray = torch.empty_like(rgb_s), torch.empty_like(sig), torch.empty_like(ray)
        st = ops._stream()
        fwd = lambda: _lib.call("niw_composite_fwd", P(ray), P(rgb_s), P(sig), P(dep), N, S, 0, 0.0, P(rgb), P(depth), P(opa), P(prob), st)
        bwd = lambda: _lib.call("niw_composite_bwd", P(ray), P(rgb_s), P(sig), P(dep), N, S, 0, 0.0, P(g_rgb), None, None, None, P(d_rgb_s), P(d_sig), P(d_ray), st)

        def timed(fn):
            # `iters` launches back to back between two device events: kernel time + the ~1.5 us kernel boundary
            fn(); fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            return [a.elapsed_time(b) / args.iters]

        t_f, t_b = timed(fwd), timed(bwd)
        bytes_f = N * S * 24 + N * 32            # rgb 12 + sigma 4 + depth 4 in, prob 4 out per sample; ray 12 in + 20 out per ray
        bytes_b = N * S * 36 + N * 36            # 20 in + 16 out per sample (no d_prob); ray 12 + d_rgb 12 in, d_ray 12 out per ray
        med = lambda v: sorted(v)[len(v) // 2]
        line = dict(n_rays=N, samples=S, mbytes_fwd=round(bytes_f / 1e6, 2), mbytes_bwd=round(bytes_b / 1e6, 2),
                    fwd_us=round(med(t_f) * 1e3, 1), bwd_us=round(med(t_b) * 1e3, 1),
                    fwd_gbps=round(bytes_f / med(t_f) / 1e6, 1), bwd_gbps=round(bytes_b / med(t_b) / 1e6, 1),
                    fwd_frac_of_6290=round(bytes_f / med(t_f) / 1e6 / 6290, 3), bwd_frac_of_6290=round(bytes_b / med(t_b) / 1e6 / 6290, 3),
                    lib=os.environ.get("NIW_LIB_PATH", "product"))
        print(json.dumps(line), flush=True)
        lines.append(line)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(lines, f, indent=1)


if __name__ == "__main__":
    main()
