"""NVP warp kernels alone (niw_warp_fwd / niw_warp_bwd incl. its parameter-gradient GEMMs) at the training shapes:
cfg3 (18 views x 226 points), cfg2 (18 x 454), cfg4 extremes (56 x 72, 23 x 178), cfg5 (3 x 1364), 1/8 shards.
Device-event time of `iters` back-to-back calls through the C ABI.  NIW_LIB_PATH selects a diagnostic build."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=20); args = ap.parse_args()
    import torch
    from neural_invertible_warp_amd import _lib, ops
    from oracle import niw_oracle as O
    dev = "cuda:0"
    P = ops._p
    for B, N in ((18, 226), (18, 454), (56, 72), (23, 178), (3, 1364), (18, 30)):
        torch.manual_seed(0)
        w_emb = torch.randn(ops.WARP_WEMB_FLOATS, device=dev) * 0.1
        view_b = torch.randn(B, 3, 2, 128, device=dev) * 0.1
        w_head = torch.randn(ops.WARP_WHEAD_FLOATS, device=dev) * 0.02
        pts = torch.randn(B, N, 3, device=dev)
        out = torch.empty_like(pts); d_out = torch.randn_like(pts); xin = torch.empty(B, N, 3, 3, device=dev)
        ws = torch.empty(_lib.load().niw_warp_bwd_workspace_floats(B, N), device=dev)
        dwe, dvb, dwh, dp = torch.empty_like(w_emb), torch.empty_like(view_b), torch.empty_like(w_head), torch.empty_like(pts)
        cw = ops._farr([1.0] * 6, 6); iw = ops._farr([0.3, 0.6, 1, 1, 1, 1], 6)
        st = ops._stream()
        fwd = lambda: _lib.call("niw_warp_fwd", P(w_emb), P(view_b), P(w_head), P(pts), B, N, cw, iw, None, 0, None, None, 0, P(out), P(xin), st)
        bwd = lambda: _lib.call("niw_warp_bwd", P(w_emb), P(view_b), P(w_head), P(pts), B, N, cw, iw, None, 0, None, None, P(xin), P(d_out), P(ws), P(dwe), P(dvb), P(dwh), P(dp), st)
        def timed(fn):
            fn(); fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters): fn()
            b.record(); torch.cuda.synchronize()
            return a.elapsed_time(b) / args.iters * 1e3
        print(json.dumps(dict(views=B, points=N, fwd_us=round(timed(fwd), 1), bwd_group_us=round(timed(bwd), 1), lib=os.environ.get("NIW_LIB_PATH", "product"))), flush=True)

if __name__ == "__main__":
    main()
