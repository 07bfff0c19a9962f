"""Mirror of the reference's model/nerf.py classes that sit on the render path: `NeRF`
(nerf.py:367-483) and `Graph` (nerf.py:243-365), same method names, argument meaning and
state-dict keys; the arithmetic is delegated to libniw_hip.so through ..ops.
"""
import numpy as np
import torch

from .. import camera, ops
from .._lib import NiwError
from ..util import edict

_SUPPORTED_ARCH = dict(layers_feat=[None, 256, 256, 256, 256, 256, 256, 256, 256], layers_rgb=[None, 128, 3], skip=[4])


def _slice_rays(opt):
    """Rays per slice of a full-image render.  The reference slices by `nerf.rand_rays` (nerf.py:325) to bound its memory;
    `nerf.eval_slice_rays` overrides it (the HIP path takes up to 2^24 samples per launch: a 300x400 image in one go)."""
    n = opt.nerf.get("eval_slice_rays") or opt.nerf.rand_rays
    s = opt.nerf.sample_intvs + (opt.nerf.sample_intvs_fine or 0 if opt.nerf.fine_sampling else 0)
    return max(1, min(int(n), ((1 << 24) - 128) // max(s, 1)))


def _layer_dims(layers):
    return list(zip(layers[:-1], layers[1:]))


class _BaseGraph(torch.nn.Module):
    """reference model/base.py:191-211"""

    def __init__(self, opt):
        super().__init__()

    def L1_loss(self, pred, label=0):
        return (pred.contiguous() - label).abs().mean()

    def MSE_loss(self, pred, label=0):
        return ((pred.contiguous() - label) ** 2).mean()


class NeRF(torch.nn.Module):

    def __init__(self, opt):
        super().__init__()
        self.define_network(opt)

    # ------------------------------------------------------------------ parameters
    def define_network(self, opt):
        """reference nerf.py:373-402.  The kernels are built for the one architecture every
        reference config uses; anything else is refused loudly."""
        arch = opt.arch
        if (list(arch.layers_feat) != _SUPPORTED_ARCH["layers_feat"] or list(arch.layers_rgb) != _SUPPORTED_ARCH["layers_rgb"]
                or list(arch.skip) != _SUPPORTED_ARCH["skip"] or not arch.posenc or arch.posenc.L_3D != ops.L3D
                or arch.posenc.L_view != ops.LVIEW or not opt.nerf.view_dep):
            raise NiwError("NeRF: only the 8x256 / skip[4] / rgb[128,3] / L_3D=10 / L_view=4 view-dependent architecture "
                           "of the reference configs is built into libniw_hip.so")
        if arch.density_activ not in ops.ACT:
            raise NiwError(f"NeRF: density activation {arch.density_activ!r} not supported (relu, softplus)")
        input_3D_dim = 3 + 6 * arch.posenc.L_3D
        input_view_dim = 3 + 6 * arch.posenc.L_view
        device = torch.device(opt.device)
        flat = torch.zeros(ops.NERF_PARAM_FLOATS, device=device, dtype=torch.float32)
        self.mlp_feat = torch.nn.ModuleList()
        self.mlp_rgb = torch.nn.ModuleList()
        self.total_param = 0
        off = 0
        L = _layer_dims(arch.layers_feat)
        specs = []
        for li, (k_in, k_out) in enumerate(L):
            if li == 0: k_in = input_3D_dim
            if li in arch.skip: k_in += input_3D_dim
            if li == len(L) - 1: k_out += 1
            specs.append((self.mlp_feat, k_in, k_out, "first" if li == len(L) - 1 else None))
        Lr = _layer_dims(arch.layers_rgb)
        for li, (k_in, k_out) in enumerate(Lr):
            if li == 0: k_in = arch.layers_feat[-1] + input_view_dim
            specs.append((self.mlp_rgb, k_in, k_out, "all" if li == len(Lr) - 1 else None))
        for mlist, k_in, k_out, out in specs:
            lin = torch.nn.Module()
            lin.in_features, lin.out_features = k_in, k_out
            lin.weight = torch.nn.Parameter(flat[off:off + k_in * k_out].view(k_out, k_in))
            off += k_in * k_out
            lin.bias = torch.nn.Parameter(flat[off:off + k_out])
            off += k_out
            if arch.tf_init:
                self.tensorflow_init_weights(opt, lin, out=out)
            else:
                torch.nn.init.kaiming_uniform_(lin.weight, a=5 ** 0.5)
                torch.nn.init.uniform_(lin.bias, -1 / k_in ** 0.5, 1 / k_in ** 0.5)
            mlist.append(lin)
            self.total_param += lin.weight.numel()
        assert off == ops.NERF_PARAM_FLOATS
        self._state = ops.FieldState(flat)
        self.progress_host = None

    def tensorflow_init_weights(self, opt, linear, out=None):
        """reference nerf.py:404-414"""
        relu_gain = torch.nn.init.calculate_gain("relu")
        with torch.no_grad():
            if out == "all":
                torch.nn.init.xavier_uniform_(linear.weight)
            elif out == "first":
                torch.nn.init.xavier_uniform_(linear.weight[:1])
                torch.nn.init.xavier_uniform_(linear.weight[1:], gain=relu_gain)
            else:
                torch.nn.init.xavier_uniform_(linear.weight, gain=relu_gain)
            torch.nn.init.zeros_(linear.bias)

    def field_parameters(self):
        """The 20 weight/bias Parameters in state-dict order (views of one flat buffer)."""
        out = []
        for lin in list(self.mlp_feat) + list(self.mlp_rgb):
            out += [lin.weight, lin.bias]
        return out

    @property
    def flat_params(self):
        self._ensure_flat()
        return self._state.flat

    def _ensure_flat(self):
        """Parameters must alias the flat buffer the kernels read; re-flatten after .to()/.cuda()
        or any external re-assignment of parameter storage."""
        ps = self.field_parameters()
        flat = self._state.flat
        off, ok = 0, True
        for p in ps:
            if p.data_ptr() != flat.data_ptr() + 4 * off or p.device != flat.device:
                ok = False
                break
            off += p.numel()
        if ok:
            return
        new = torch.cat([p.detach().reshape(-1).float() for p in ps]).contiguous()
        off = 0
        for p in ps:
            p.data = new[off:off + p.numel()].view(p.shape)
            off += p.numel()
        self._state = ops.FieldState(new)

    # ------------------------------------------------------------------ encoding weights
    def band_weights(self, opt, L):
        """BARF coarse-to-fine band mask (model/barf_inn_llff.py:427-442); all ones here."""
        return [1.0] * L

    def positional_encoding(self, opt, input, L):
        """reference nerf.py:476-483.  Interface parity only: the render path computes the
        encoding inside niw_mlp_fwd in MFMA slot order and never materialises it."""
        shape = input.shape
        freq = 2 ** torch.arange(L, dtype=torch.float32, device=input.device) * np.pi
        spectrum = input[..., None] * freq
        enc = torch.stack([spectrum.sin(), spectrum.cos()], dim=-2)
        w = torch.tensor(self.band_weights(opt, L), dtype=torch.float32, device=input.device)
        return (enc * w).view(*shape[:-1], -1)

    # ------------------------------------------------------------------ evaluation
    def _run(self, opt, center, ray, depth, mode):
        self._ensure_flat()
        noise = None
        if opt.nerf.density_noise_reg and mode == "train":
            noise = torch.randn(depth.shape, device=depth.device) * opt.nerf.density_noise_reg    # nerf.py:428-429
        return ops.field_mlp(self._state, self.field_parameters(), center, ray, depth,
                             self.band_weights(opt, ops.L3D), self.band_weights(opt, ops.LVIEW), opt.arch.density_activ, noise)

    def forward(self, opt, points_3D, ray_unit=None, mode=None):
        """reference nerf.py:416-447: points [...,3], unit view directions [...,3] -> rgb [...,3],
        density [...].  Evaluated by the same fused kernel as forward_samples with one sample per
        ray at depth 0 (x = c + 0 * v is exact)."""
        assert ray_unit is not None
        shape = points_3D.shape[:-1]
        pts = points_3D.reshape(-1, 3)
        dirs = ray_unit.expand_as(points_3D).reshape(-1, 3)
        rgb, density = self._run(opt, pts, dirs, torch.zeros(pts.shape[0], 1, device=pts.device), mode)
        return rgb.view(*shape, 3), density.view(*shape)

    def forward_samples(self, opt, center, ray, depth_samples, mode=None):
        """reference nerf.py:449-456: center, ray [B,R,3], depth_samples [B,R,S,1] ->
        rgb [B,R,S,3], density [B,R,S]."""
        B, R, S = depth_samples.shape[:3]
        rgb, density = self._run(opt, center.reshape(-1, 3), ray.reshape(-1, 3), depth_samples.reshape(B * R, S), mode)
        return rgb.view(B, R, S, 3), density.view(B, R, S)

    def composite(self, opt, ray, rgb_samples, density_samples, depth_samples):
        """reference nerf.py:458-474 -> rgb [B,R,3], depth [B,R,1], opacity [B,R,1], prob [B,R,S,1]."""
        B, R, S = density_samples.shape
        bg = opt.data.bgcolor if opt.nerf.setbg_opaque else None
        rgb, depth, opacity, prob = ops.composite(ray.reshape(-1, 3), rgb_samples.reshape(B * R, S, 3),
                                                  density_samples.reshape(B * R, S), depth_samples.reshape(B * R, S), bg)
        return rgb.view(B, R, 3), depth.view(B, R, 1), opacity.view(B, R, 1), prob.view(B, R, S, 1)


class Graph(_BaseGraph):

    def __init__(self, opt):
        super().__init__(opt)
        self.nerf = NeRF(opt)
        if opt.nerf.fine_sampling:
            self.nerf_fine = NeRF(opt)

    def forward(self, opt, var, mode=None):
        """reference nerf.py:251-274"""
        batch_size = len(var.idx)
        pose = self.get_pose(opt, var, mode=mode)
        if opt.nerf.rand_rays and mode in ["train", "test-optim"]:
            var.ray_idx = torch.randperm(opt.H * opt.W, device=opt.device)[:opt.nerf.rand_rays // batch_size]
            ret = self.render(opt, pose, intr=var.intr, ray_idx=var.ray_idx, mode=mode)
        else:
            ret = self.render_by_slices(opt, pose, intr=var.intr, mode=mode) if opt.nerf.rand_rays else \
                self.render(opt, pose, intr=var.intr, mode=mode)
        var.update(ret)
        return var

    def compute_loss(self, opt, var, mode=None):
        """reference nerf.py:276-288"""
        loss = edict()
        ray_idx = var.ray_idx if (opt.nerf.rand_rays and mode in ["train", "test-optim"]) else None
        n_norm = getattr(opt, "loss_norm_elements", None)
        if opt.loss_weight.render is not None:
            loss.render = ops.mse_gather(var.rgb, var.image, ray_idx, n_norm)
        if opt.loss_weight.render_fine is not None:
            assert opt.nerf.fine_sampling
            loss.render_fine = ops.mse_gather(var.rgb_fine, var.image, ray_idx, n_norm)
        return loss

    def get_pose(self, opt, var, mode=None):
        return var.pose

    def _render_rays(self, opt, center, ray, mode=None, depth_range=None):
        """Shared tail of render / render_local (reference nerf.py:304-319)."""
        batch_size = ray.shape[0]
        depth_samples = self.sample_depth(opt, batch_size, num_rays=ray.shape[1]) if depth_range is None else \
            self.sample_depth(opt, batch_size, num_rays=ray.shape[1], depth_range=depth_range)
        rgb_samples, density_samples = self.nerf.forward_samples(opt, center, ray, depth_samples, mode=mode)
        rgb, depth, opacity, prob = self.nerf.composite(opt, ray, rgb_samples, density_samples, depth_samples)
        ret = edict(rgb=rgb, depth=depth, opacity=opacity)
        if opt.nerf.fine_sampling:
            with torch.no_grad():
                B, R, S = prob.shape[:3]
                _, merged = ops.sample_pdf_merge(prob.reshape(B * R, S), depth_samples.reshape(B * R, S),
                                                 opt.nerf.sample_intvs_fine, opt.nerf.depth.range)
                depth_samples = merged.view(B, R, -1, 1)
            rgb_samples, density_samples = self.nerf_fine.forward_samples(opt, center, ray, depth_samples, mode=mode)
            rgb_fine, depth_fine, opacity_fine, _ = self.nerf_fine.composite(opt, ray, rgb_samples, density_samples, depth_samples)
            ret.update(rgb_fine=rgb_fine, depth_fine=depth_fine, opacity_fine=opacity_fine)
        return ret

    def render(self, opt, pose, intr=None, ray_idx=None, mode=None):
        """reference nerf.py:293-319 (the NaN retry loop of :296 guarded a CUDA bug and forced a
        host sync per call; the HIP ray generator is deterministic, so it is not reproduced)."""
        center, ray = camera.get_center_and_ray(opt, pose, intr=intr, ray_idx=ray_idx)
        if opt.camera.ndc:
            center, ray = camera.convert_NDC(opt, center, ray, intr=intr)
        return self._render_rays(opt, center, ray, mode=mode)

    def _hold_weights(self):
        """context: pack the networks' weights once for a loop of gradient-free renders (ops.FieldState.hold)"""
        import contextlib
        stack = contextlib.ExitStack()
        if not torch.is_grad_enabled():
            for net in (self.nerf, getattr(self, "nerf_fine", None)):
                if net is not None:
                    net._ensure_flat()
                    stack.enter_context(net._state.hold())
        return stack

    def render_by_slices(self, opt, pose, intr=None, mode=None):
        """reference nerf.py:321-332"""
        ret_all = edict(rgb=[], depth=[], opacity=[])
        if opt.nerf.fine_sampling:
            ret_all.update(rgb_fine=[], depth_fine=[], opacity_fine=[])
        with self._hold_weights():
            step = _slice_rays(opt)
            for c in range(0, opt.H * opt.W, step):
                ray_idx = torch.arange(c, min(c + step, opt.H * opt.W), device=opt.device)
                ret = self.render(opt, pose, intr=intr, ray_idx=ray_idx, mode=mode)
                for k in ret: ret_all[k].append(ret[k])
        for k in ret_all: ret_all[k] = torch.cat(ret_all[k], dim=1)
        return ret_all

    def sample_depth(self, opt, batch_size, num_rays=None, depth_range=None):
        """reference nerf.py:334-344 -> [B,R,S,1]; the stratified draw is torch.rand as in the
        reference (nerf.py:337), the arithmetic is niw_sample_stratified."""
        rng = opt.nerf.depth.range if depth_range is None else depth_range
        num_rays = num_rays or opt.H * opt.W
        S = opt.nerf.sample_intvs
        u = torch.rand(batch_size, num_rays, S, 1, device=opt.device) if opt.nerf.sample_stratified else None
        d = ops.sample_stratified(None if u is None else u.view(-1, S), batch_size * num_rays, S, rng, opt.nerf.depth.param, opt.device)
        return d.view(batch_size, num_rays, S, 1)

    def sample_depth_from_pdf(self, opt, pdf):
        """reference nerf.py:346-365: pdf [B,R,S] -> [B,R,Sf,1]"""
        B, R, S = pdf.shape
        dummy = torch.zeros(B * R, S, device=pdf.device)
        fine, _ = ops.sample_pdf_merge(pdf.reshape(B * R, S), dummy, opt.nerf.sample_intvs_fine, opt.nerf.depth.range)
        return fine.view(B, R, -1, 1)


class Model:
    """The reference's engine interface for the vanilla model (model/nerf.py:20-160, driven by train.py:21-32), ground-truth
    poses: torch.optim.Adam with one param group per network and ExponentialLR exactly as the reference builds them
    (the parameters are ordinary nn.Parameters, views of the flat buffers the kernels read); checkpoints are the
    reference's model.ckpt dict.  Logging back-ends are outside the path."""

    def __init__(self, opt):
        import os
        self.opt = opt
        os.makedirs(opt.output_path, exist_ok=True)
        self.it = self.iter_start = 0

    def load_dataset(self, opt, eval_split="val"):
        import importlib
        import os
        name = opt.data.dataset
        root = opt.data.get("root") or "data/{}".format(name)
        if name != "synthetic" and not os.path.isdir("{}/{}".format(root, opt.data.scene)):
            print("[niw] dataset {}/{} not found: using the procedural scene".format(root, opt.data.scene))
            name = "synthetic"
        data = importlib.import_module("neural_invertible_warp_amd.data.{}".format(name))
        self.train_data = data.Dataset(opt, split="train", subset=opt.data.get("train_sub"))
        self.test_data = data.Dataset(opt, split="test" if opt.data.get("val_on_test") else eval_split, subset=opt.data.get("val_sub"))
        for d in (self.train_data, self.test_data):
            d.prefetch_all_data(opt)
            d.all = edict({k: v.to(opt.device) for k, v in d.all.items()})

    def build_networks(self, opt):
        torch.manual_seed(opt.seed or 0)
        self.graph = Graph(opt).to(opt.device)

    def setup_optimizer(self, opt):
        optimizer = getattr(torch.optim, opt.optim.algo)
        self.optim = optimizer([dict(params=self.graph.nerf.parameters(), lr=opt.optim.lr)])
        if opt.nerf.fine_sampling:
            self.optim.add_param_group(dict(params=self.graph.nerf_fine.parameters(), lr=opt.optim.lr))
        gamma = (opt.optim.lr_end / opt.optim.lr) ** (1. / opt.max_iter) if opt.optim.get("lr_end") else 1.0
        self.sched = torch.optim.lr_scheduler.ExponentialLR(self.optim, gamma=gamma)

    def restore_checkpoint(self, opt):
        import os
        self.epoch_start = self.iter_start = 0
        name = None
        if opt.resume:
            name = "{0}/model.ckpt".format(opt.output_path) if opt.resume is True else "{0}/model/{1}.ckpt".format(opt.output_path, opt.resume)
        elif opt.load is not None:
            name = opt.load
        if name is None:
            return
        ck = torch.load(name, map_location=opt.device, weights_only=False)
        for child_name, child in self.graph.named_children():
            sd = {".".join(k.split(".")[1:]): v for k, v in ck["graph"].items() if k.startswith(child_name + ".")}
            if sd:
                child.load_state_dict(sd)
        if opt.resume:
            self.optim.load_state_dict(ck["optim"])
            self.sched.load_state_dict(ck["sched"])
            self.epoch_start, self.iter_start = ck["epoch"] or 0, ck["iter"] or 0

    def setup_visualizer(self, opt):
        pass

    def train(self, opt):
        self.graph.train()
        var = self.train_data.all
        self.it = self.iter_start
        if self.iter_start == 0:
            self.validate(opt, 0)
        while self.it < opt.max_iter:
            loss = self.train_iteration(opt, var, None)
            self.sched.step()
            if self.it % opt.freq.scalar == 0:
                print("[train it {}] {}".format(self.it, " ".join("{}={:.5f}".format(k, float(v.detach())) for k, v in loss.items())), flush=True)
            if self.it % opt.freq.val == 0:
                self.validate(opt, self.it)
            if self.it % opt.freq.ckpt == 0:
                self.save_checkpoint(opt, ep=None, it=self.it)
        return self

    def summarize_loss(self, opt, var, loss):
        total = 0.
        for key in loss:
            if opt.loss_weight[key] is not None:
                total = total + 10 ** float(opt.loss_weight[key]) * loss[key]
        loss.update(all=total)
        return loss

    def train_iteration(self, opt, var, loader=None):
        self.optim.zero_grad(set_to_none=True)
        var = self.graph.forward(opt, edict(var), mode="train")
        loss = self.summarize_loss(opt, var, self.graph.compute_loss(opt, var, mode="train"))
        loss.all.backward()
        self.optim.step()
        self.it += 1
        return loss

    @torch.no_grad()
    def validate(self, opt, ep=None):
        self.graph.eval()
        psnr = []
        allv = self.test_data.all
        for i in range(len(self.test_data)):
            var = self.graph.forward(opt, edict({k: v[i:i + 1] for k, v in allv.items()}), mode="val")
            rgb = var.rgb_fine if opt.nerf.fine_sampling else var.rgb
            rgb_map = rgb.view(-1, opt.H, opt.W, 3).permute(0, 3, 1, 2)
            psnr.append(-10 * self.graph.MSE_loss(rgb_map, var.image).log10().item())
        self.graph.train()
        out = edict(psnr=sum(psnr) / max(len(psnr), 1))
        print("[val it {}] PSNR {:.2f}".format(ep, out.psnr), flush=True)
        return out

    def save_checkpoint(self, opt, ep=0, it=0, latest=False):
        import os
        import shutil
        os.makedirs("{0}/model".format(opt.output_path), exist_ok=True)
        ck = dict(epoch=ep, iter=it, graph=self.graph.state_dict(), optim=self.optim.state_dict(), sched=self.sched.state_dict())
        torch.save(ck, "{0}/model.ckpt".format(opt.output_path))
        if not latest:
            shutil.copy("{0}/model.ckpt".format(opt.output_path), "{0}/model/{1}.ckpt".format(opt.output_path, ep or it))
