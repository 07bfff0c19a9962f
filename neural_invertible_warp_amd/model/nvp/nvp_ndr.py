"""Mirror of the reference's DeformNetwork (model/nvp/nvp_ndr.py:229-572) as constructed by
the INN models (model/barf_inn_llff.py:54-55, model/pose_models/inn.py:23-27): same
constructor arguments, parameter names (lin{b}_a_0.weight_g/.weight_v/.bias, lin{b}_a_1.*,
lin{b}_b_0.*, lin{b}_b_1.*, lin{b}_c.*), initialisation and forward / inverse signatures.

Split of the work: the per-parameter / per-view preprocessing (weight norm, code projection,
latent half of the first layers: O(parameters), [B,128]-sized tensors) is one fused launch
(niw_warp_prep_fwd, backward niw_warp_prep_bwd) over the flat parameter buffer; everything per
point (embedding, coupling blocks, rotations; forward, inverse and backward) runs in niw_warp_*.
"""
import math

import numpy as np
import torch

from ... import ops
from ..._lib import NiwError

_HID, _LAT, _NF = 128, 128, 6


class _WNLinear(torch.nn.Module):
    """Linear layer under old-style nn.utils.weight_norm (dim=0): parameters weight_g [out,1],
    weight_v [out,in], bias [out] (reference nvp_ndr.py:291-292)."""

    def __init__(self, k_in, k_out, ori_in):
        super().__init__()
        v = torch.zeros(k_out, k_in)
        torch.nn.init.normal_(v[:, :ori_in], 0.0, np.sqrt(2) / np.sqrt(k_out))      # nvp_ndr.py:278-282
        # registration order of the reference: nn.Linear's bias first, then what weight_norm adds
        self.bias = torch.nn.Parameter(torch.zeros(k_out))
        self.weight_g = torch.nn.Parameter(v.norm(dim=1, keepdim=True))
        self.weight_v = torch.nn.Parameter(v)

    def weight(self):
        return self.weight_v * (self.weight_g / self.weight_v.norm(dim=1, keepdim=True))


def _zero_linear(k_in, k_out):
    lin = torch.nn.Linear(k_in, k_out)
    torch.nn.init.constant_(lin.bias, 0.0)
    torch.nn.init.constant_(lin.weight, 0.0)
    return lin


def anneal_window(alpha_ratio, n_freq=_NF):
    """reference model/nvp/embedder.py:47-49"""
    return [(1.0 - math.cos(math.pi * max(min(alpha_ratio * n_freq - i, 1.0), 0.0))) * 0.5 for i in range(n_freq)]


class DeformNetwork(torch.nn.Module):

    def __init__(self, d_feature, d_in, d_out_1, d_out_2, n_blocks, d_hidden, n_layers, skip_in=(4,), multires=0,
                 weight_norm=True, actfn="softplus", reference_exact=True):
        super().__init__()
        if not (d_feature == _LAT and d_in == 3 and d_out_1 == 1 and d_out_2 == 3 and n_blocks == 3 and d_hidden == _HID
                and n_layers == 1 and len(skip_in) == 0 and multires == _NF and weight_norm and actfn == "softplus"):
            raise NiwError("DeformNetwork: libniw_hip.so implements the configuration the reference models build "
                           "(d_feature=128, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3, d_hidden=128, n_layers=1, skip_in=[], "
                           "multires=6, weight_norm=True, actfn='softplus')")
        self.n_blocks, self.skip_in = n_blocks, skip_in
        self.reference_exact = reference_exact        # reproduce the dim-1 slicing of embedder.py:47
        ea, eb = 2 * (1 + 2 * multires), 1 + 2 * multires
        for b in range(n_blocks):
            setattr(self, f"lin{b}_a_0", _WNLinear(ea + d_feature, d_hidden, ori_in=2))
            setattr(self, f"lin{b}_a_1", _zero_linear(d_hidden, d_out_1))
        for b in range(n_blocks):
            setattr(self, f"lin{b}_b_0", _WNLinear(eb + d_feature, d_hidden, ori_in=1))
            setattr(self, f"lin{b}_b_1", _zero_linear(d_hidden, d_out_2))
        for b in range(n_blocks):
            setattr(self, f"lin{b}_c", _zero_linear(d_feature, d_feature))
        self._ea, self._eb = ea, eb
        self.window_dev = None      # device tensor [12] of annealing windows read by the kernels at run time (engine.StepConstants)
        self.grad_sink = None       # (parameter-gradient buffer, code-gradient buffer) written instead of .grad (ops.warp_prepare)
        self.code_rows = None       # (first, end) rows of the latent table the next forward's `deformation_code` holds; None: all of it

    # ------------------------------------------------------------------ operand preparation
    def _ensure_flat(self):
        """The kernels read the parameters from ONE flat buffer in parameters() order (layout documented in
        csrc/niw_warp_prep.hip); the Parameters are views of it.  (Re-)flatten after construction, .to() /
        .cuda() or any external re-assignment of parameter storage."""
        ps = list(self.parameters())
        flat = getattr(self, "_flat", None)
        if flat is not None:
            off, ok = 0, True
            for p in ps:
                if p.data_ptr() != flat.data_ptr() + 4 * off or p.device != flat.device:
                    ok = False
                    break
                off += p.numel()
            if ok:
                return flat
        flat = torch.cat([p.detach().reshape(-1).float() for p in ps]).contiguous()
        assert flat.numel() == ops.WARP_PARAM_FLOATS
        off = 0
        for p in ps:
            p.data = flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        self._flat = flat
        return flat

    @property
    def flat_params(self):
        return self._ensure_flat()

    def _operands(self, code):
        """-> w_emb [3*(128*26+128*13)], view_b [B,3,2,128], w_head [3*516] (layout of include/niw.h):
        weight norm (nvp_ndr.py:291-292), code projection (:381) and the latent half of the first layers,
        fused in niw_warp_prep_fwd / _bwd."""
        flat = self._ensure_flat()
        sink, rows = self.grad_sink, self.code_rows
        if sink is not None and rows is not None and torch.is_grad_enabled():
            # `code` is rows [rows[0], rows[1]) of the latent table whose gradient buffer the sink holds (a rank's window of views under
            # ray sharding, ...parallel.ViewWindow): the kernel writes those rows, the other views get no gradient from this rank
            full, a, b = sink[1], rows[0] * _LAT, rows[1] * _LAT
            full[:a].zero_()
            full[b:].zero_()
            sink = (sink[0], full[a:b])
        return ops.warp_prepare(flat, list(self.parameters()), code, grad_sink=sink)

    def _anneal(self, alpha_ratio):
        """-> (chan_w[6], index_window[6] | None).  reference_exact: the window multiplies whole points
        (2i+1)d..(2i+3)d-1 along dim 1 (embedder.py:47 on 4-D input); the kernel derives the per-point scale from the
        point index, so the six values travel by value with the launch."""
        w = anneal_window(alpha_ratio)
        return ([1.0] * _NF, w) if self.reference_exact else (w, None)

    def _apply_warp(self, deformation_code, input_pts, alpha_ratio, inverse):
        if input_pts.dim() != 4 or input_pts.shape[2] != 1 or input_pts.shape[3] != 3:
            raise NiwError(f"DeformNetwork: input_pts must be [B,N,1,3], got {tuple(input_pts.shape)}")
        B, P = input_pts.shape[:2]
        if deformation_code.shape != (B, _LAT):
            raise NiwError(f"DeformNetwork: deformation_code must be [{B},{_LAT}], got {tuple(deformation_code.shape)}")
        w_emb, view_b, w_head = self._operands(deformation_code)
        chan_w, index_window = self._anneal(float(alpha_ratio))
        out = ops.warp_points(w_emb, view_b, w_head, input_pts.reshape(B, P, 3), chan_w, index_window, inverse=inverse,
                              window_dev=self.window_dev, use_index_window=self.reference_exact)
        return out.view(B, P, 1, 3)

    def forward(self, deformation_code, input_pts, alpha_ratio=0):
        """reference nvp_ndr.py:365-468: code [B,128], input_pts [B,N,1,3] -> [B,N,1,3]"""
        return self._apply_warp(deformation_code, input_pts, alpha_ratio, inverse=False)

    def inverse(self, deformation_code, input_pts, alpha_ratio):
        """reference nvp_ndr.py:471-567 (gradient-free: only debug helpers call it)"""
        with torch.no_grad():
            return self._apply_warp(deformation_code, input_pts, alpha_ratio, inverse=True)


# ---------------------------------------------------------------------------------------------------------------
# Helpers shared by the two users of the warp: the LLFF graph (barf_inn_llff.Graph.attach_warp / get_pose) and the DTU
# pose network (pose_models.inn.INNPoseParams).  Both build the same network, anneal its embedding the same way and
# push [grid ; centre] points through it per view.
# ---------------------------------------------------------------------------------------------------------------

def build_warp_network(opt, latent_dim):
    """The DeformNetwork every INN model instantiates (barf_inn_llff.py:54-55, pose_models/inn.py:23-27)."""
    nvp = opt.inn.real_nvp
    return DeformNetwork(d_feature=latent_dim, d_in=3, d_out_1=1, d_out_2=3, n_blocks=3, d_hidden=nvp.d_hidden, n_layers=1, skip_in=[],
                         multires=nvp.multires, weight_norm=True, actfn=opt.inn.actfn).to(opt.device)


def embedding_anneal_ratio(opt, it):
    """alpha of the warp's annealed embedder: it / max_pe_iter clamped to [0,1] under c2f, else 1 (barf_inn_llff.py:351-354)"""
    nvp = opt.inn.real_nvp
    if nvp.c2f == True:  # noqa: E712  (the reference compares with == True; yaml may hold None)
        return max(min(it / nvp.max_pe_iter, 1), 0)
    return 1


class _SplitRays(torch.autograd.Function):
    """warped [B,2n,3] = [grid ; centre] -> (ray = grid - centre, centre, grid), contiguous.  One node instead of two slices and a
    subtraction: its backward writes d warped = [d grid + d ray ; d centre - d ray] in two launches, where autograd's own chain was
    a negation, two zero-filled [B,2n,3] buffers, two slice copies and their sum."""

    @staticmethod
    def forward(ctx, warped, n):
        grid, center = warped[:, :n].contiguous(), warped[:, n:].contiguous()
        ctx.n, ctx.shape = n, warped.shape
        ctx.set_materialize_grads(False)
        return grid - center, center, grid

    @staticmethod
    def backward(ctx, d_ray, d_center, d_grid):
        n = ctx.n
        out = torch.empty(ctx.shape, device=(d_ray if d_ray is not None else d_center if d_center is not None else d_grid).device)
        g, c = out[:, :n], out[:, n:]
        if d_ray is None:
            g.zero_() if d_grid is None else g.copy_(d_grid)
            c.zero_() if d_center is None else c.copy_(d_center)
        else:
            g.copy_(d_ray) if d_grid is None else torch.add(d_grid, d_ray, out=g)
            torch.neg(d_ray, out=c) if d_center is None else torch.sub(d_center, d_ray, out=c)
        return out, None


def warp_grid_and_center(net, code, grid, center, alpha_ratio):
    """grid, center [B,R,3] (gradient-free inputs) -> (ray, center_3D, grid_3D), each [B,R,3]: the points of every view go
    through the view's warp as ONE batch [grid ; centre], rays are re-formed from the warped end points.  The two stacked
    [B,2R,3] tensors (the warp's input and output) ride along on the results for the alignment loss (`stacked_points`)."""
    n = grid.shape[1]
    stacked_in = torch.cat([grid, center], dim=1)
    warped = net.forward(code, stacked_in.unsqueeze(2), alpha_ratio=alpha_ratio).squeeze(2)
    ray, center_3D, grid_3D = _SplitRays.apply(warped, n)
    grid_3D._niw_stack = center_3D._niw_stack = (warped, stacked_in)
    return ray, center_3D, grid_3D


def stacked_points(grid_3D, center_3D):
    """-> (warped [B,2R,3], un-warped [B,2R,3]) when `grid_3D` / `center_3D` are the pair `warp_grid_and_center` returned (the
    alignment loss then needs no concatenation, and its gradient no slice scatter), else None"""
    st = getattr(grid_3D, "_niw_stack", None)
    return st if st is not None and getattr(center_3D, "_niw_stack", None) is st else None
