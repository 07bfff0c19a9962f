"""Field-MLP kernels alone (niw_mlp_fwd with / without activation saves, niw_mlp_bwd_dx, niw_mlp_bwd_dw) through the C ABI at
sample counts from one 256-workgroup round (32,768 samples = a 1/8 shard of the 2048-ray batch) to the cfg2 fine pass
(784,512): device-event time of `iters` back-to-back launches into pre-allocated workspaces, TFLOP/s against the fp32-MFMA
peak.  NIW_LIB_PATH selects a diagnostic build."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLOP = 2 * 527872

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--sizes", default="252x128,504x128,1008x128,2034x128,4086x64,4086x192")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16x3", "bf16"], help="arithmetic of the field MLP (include/niw.h enum niw_precision)")
    args = ap.parse_args()
    import torch
    from neural_invertible_warp_amd import _lib, ops
    from oracle import niw_oracle as O
    dev = "cuda:0"; P = ops._p; lib = _lib.load()
    p = O.make_nerf_params(1)
    flat = torch.cat([p[f"{n}.{k}"].reshape(-1) for n, _, _ in O.nerf_layer_shapes() for k in ("weight", "bias")]).to(dev)
    st8 = ops.FieldState(flat, precision=args.precision); packed = st8.packed()
    prec = ops.PREC[args.precision]
    bprec = prec if args.precision in ops.DX_PRECISIONS else 0; wprec = prec if args.precision in ops.DW_PRECISIONS else 0
    bpacked = packed if bprec == prec else st8.packed_fp32()       # backward kernels that do not exist in a fast mode run exact
    for spec in args.sizes.split(","):
        N, S = (int(x) for x in spec.split("x"))
        M = N * S; mpad = lib.niw_mlp_padded_rows(N, S)
        center = torch.randn(N, 3, device=dev) * 0.1; ray = torch.randn(N, 3, device=dev)
        depth = (torch.rand(N, S, device=dev).sort(dim=1).values * 4 + 0.5).contiguous()
        rgb = torch.empty(N, S, 3, device=dev); sigma = torch.empty(N, S, device=dev)
        save = torch.empty(ops.SAVE_ROWS * mpad, device=dev); gradws = torch.empty(ops.GRAD_ROWS * mpad, device=dev)
        partial = torch.empty(lib.niw_mlp_bwd_workspace_floats(N, S), device=dev); d_params = torch.empty(ops.NERF_PARAM_FLOATS, device=dev)
        d_rgb = torch.randn(N, S, 3, device=dev); d_sigma = torch.randn(N, S, device=dev)
        dc = torch.zeros(N, 3, device=dev); dr = torch.zeros(N, 3, device=dev)
        b3 = ops._farr([1.0] * 10, 10); bv = ops._farr([1.0] * 4, 4); st = ops._stream()
        fns = dict(
            fwd_eval=lambda: _lib.call("niw_mlp_fwd", P(packed), P(center), P(ray), P(depth), None, N, S, b3, bv, None, 1, prec, P(rgb), P(sigma), None, st),
            fwd_train=lambda: _lib.call("niw_mlp_fwd", P(packed), P(center), P(ray), P(depth), None, N, S, b3, bv, None, 1, prec, P(rgb), P(sigma), P(save), st),
            bwd_dx=lambda: _lib.call("niw_mlp_bwd_dx", P(bpacked), P(center), P(ray), P(depth), N, S, 1, bprec, P(rgb), P(d_rgb), P(d_sigma), P(save), P(gradws), P(dc), P(dr), st),
            bwd_dw=lambda: _lib.call("niw_mlp_bwd_dw", P(save), P(gradws), N, S, wprec, P(partial), P(d_params), st))
        line = dict(precision=args.precision, dx_precision="fp32" if bprec == 0 else args.precision, dw_precision="fp32" if wprec == 0 else args.precision, rays=N, samples=S, mlp_evals=M, workgroups=int(mpad // 128), lib=os.environ.get("NIW_LIB_PATH", "product"))
        for name, fn in fns.items():
            for _ in range(8): fn()      # (the library's second stream -- niw_mlp_bwd_dw's heads kernel -- pays one-time runtime set-up in its first few uses)
            # ... and then at least 30 ms of back-to-back launches: after an idle period the chip runs its first milliseconds at ~2.0 GHz instead
            # of 2.39 (in-kernel s_memtime / s_memrealtime stamps, profiles/r6_launch_trace.json: a train of 6 one-round launches 2.04 GHz, of 80
            # 2.39 GHz), which rounds 2-5 of this tool read as a "fixed cost" of the small launches (10 x 0.29 ms measured right after a sync)
            torch.cuda.synchronize(); t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.03:
                for _ in range(10): fn()
                torch.cuda.synchronize()
            for _ in range(10): fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters): fn()
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / args.iters
            line[name] = dict(us=round(ms * 1e3, 1), tflops=round(M * FLOP / ms / 1e9, 1), frac=round(M * FLOP / ms / 1e9 / 157.3, 3))
        print(json.dumps(line), flush=True)

if __name__ == "__main__":
    main()
