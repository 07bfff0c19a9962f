"""EXPERIMENT: the 16-sample-wave forward (experiments/niw_mlp16.hip) against the product forward: same outputs, device-event time.
Build:  make -C neural_invertible_warp_amd/csrc VARIANT=v16 EXPERIMENTS=niw_mlp16.hip;  run with NIW_LIB_PATH=neural_invertible_warp_amd/libniw_hip_v16.so.
Result (MI355X): HISTORY.md (round 3, full size) and profiles/r4_v16_small_launches.json (round 4, a rank's 1/8 share)."""
import argparse, ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
FLOP = 2 * 527872

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--sizes", default="7x5,252x128,2034x128,4086x64,4086x192")
    args = ap.parse_args()
    import torch
    from neural_invertible_warp_amd import _lib, ops
    from oracle import niw_oracle as O
    dev = "cuda:0"; P = ops._p; lib = _lib.load()
    vp, i64, ci = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    lib.niw_mlp16_packed_floats.restype = i64
    lib.niw_mlp16_pack_weights.argtypes = [vp, vp, vp]
    lib.niw_mlp16_fwd.argtypes = [vp, vp, vp, vp, vp, i64, ci, vp, vp, vp, ci, vp, vp, vp, vp]
    p = O.make_nerf_params(1)
    flat = torch.cat([p[f"{n}.{k}"].reshape(-1) for n, _, _ in O.nerf_layer_shapes() for k in ("weight", "bias")]).to(dev)
    st8 = ops.FieldState(flat); packed = st8.packed()
    packed16 = torch.empty(lib.niw_mlp16_packed_floats(), device=dev)
    st = ops._stream()
    assert lib.niw_mlp16_pack_weights(P(flat), P(packed16), st) == 0
    for spec in args.sizes.split(","):
        N, S = (int(x) for x in spec.split("x"))
        M = N * S; mpad = lib.niw_mlp_padded_rows(N, S)
        center = torch.randn(N, 3, device=dev) * 0.1; ray = torch.randn(N, 3, device=dev)
        depth = (torch.rand(N, S, device=dev).sort(dim=1).values * 4 + 0.5).contiguous()
        rgb = torch.empty(N, S, 3, device=dev); sigma = torch.empty(N, S, device=dev)
        rgb2 = torch.empty(N, S, 3, device=dev); sigma2 = torch.empty(N, S, device=dev)
        save = torch.zeros(ops.SAVE_ROWS * mpad, device=dev); save2 = torch.zeros(ops.SAVE_ROWS * mpad, device=dev)
        b3 = ops._farr([1.0, 1.0, 1.0, 0.7, 0.2, 0.0, 0.0, 0.0, 0.0, 0.0], 10); bv = ops._farr([1.0] * 4, 4)
        ref = lambda sv: _lib.call("niw_mlp_fwd", P(packed), P(center), P(ray), P(depth), None, N, S, b3, bv, None, 1, 0, P(rgb), P(sigma), sv, st)
        new = lambda sv: lib.niw_mlp16_fwd(P(packed16), P(center), P(ray), P(depth), None, N, S, b3, bv, None, 1, P(rgb2), P(sigma2), sv, st)
        ref(P(save)); rc = new(P(save2)); torch.cuda.synchronize()
        assert rc == 0
        rows = 64 + 8 * 256 + 32 + 128              # activations (then the density pre-activation; the sign records differ in format)
        a = save[: rows * mpad].view(-1, mpad, 4)[:, :M]; b = save2[: rows * mpad].view(-1, mpad, 4)[:, :M]
        sg = (save[rows * mpad: rows * mpad + M] - save2[rows * mpad: rows * mpad + M]).abs().max()
        line = dict(rays=N, samples=S, rgb_err=float((rgb - rgb2).abs().max()), sigma_rel=float(((sigma - sigma2).abs() / (sigma.abs() + 1e-3)).max()),
                    save_err=float((a - b).abs().max()), sigma_raw_err=float(sg))
        for name, fn in dict(ref_eval=lambda: ref(None), v16_eval=lambda: new(None), ref_train=lambda: ref(P(save)), v16_train=lambda: new(P(save2))).items():
            fn(); fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters): fn()
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            line[name] = dict(us=round(ms * 1e3, 1), frac=round(M * FLOP / ms / 1e9 / 157.3, 3))
        print(json.dumps(line), flush=True)

if __name__ == "__main__":
    main()
