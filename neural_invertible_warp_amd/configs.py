"""Option trees (`opt`) of the configurations named in BASELINE.json, with the key names of the
reference's yaml files (options/base.yaml, nerf_llff_repr.yaml, nerf_inn_llff.yaml,
barf_inn_llff.yaml, barf_inn_dtu.yaml) restricted to what the render path reads.
"""
import copy

from .util import edict

_ARCH = dict(layers_feat=[None, 256, 256, 256, 256, 256, 256, 256, 256], layers_rgb=[None, 128, 3], skip=[4],
             posenc=dict(L_3D=10, L_view=4), density_activ="softplus", tf_init=True)

_BASE = dict(
    model=None, seed=0, device="cuda:0", gpu=0, cpu=False, H=300, W=400, max_iter=200000, barf_c2f=None,
    group="0_test", name="debug", output_root="output", resume=False, load=None, batch_size=None,
    freq=dict(scalar=200, vis=1000, val=2000, ckpt=5000),
    arch=_ARCH,
    nerf=dict(view_dep=True, depth=dict(param="inverse", range=[1, 0]), sample_intvs=128, sample_stratified=True,
              fine_sampling=False, sample_intvs_fine=None, rand_rays=2048, density_noise_reg=None, setbg_opaque=None),
    data=dict(dataset="llff", root=None, scene="fern", image_size=[300, 400], bgcolor=None, center_crop=None, val_ratio=0.1,
              train_sub=None, val_sub=None, val_on_test=False),
    camera=dict(model="perspective", ndc=False),
    loss_weight=dict(render=0, render_fine=None, global_alignment=None),
    optim=dict(algo="Adam", lr=1e-3, lr_end=1e-4, lr_pose=5e-4, lr_pose_end=1e-8, test_photo=True, test_iter=100),
    inn=dict(real_nvp=dict(c2f=True, max_pe_iter=100000, d_hidden=128, multires=6, latent_dim=128), actfn="softplus"),
    warp_latent=dict(enc_type="l2fbarf", embed_dim=128),
)


def _mk(**over):
    d = copy.deepcopy(_BASE)

    def merge(dst, src):
        for k, v in src.items():
            if isinstance(v, dict) and isinstance(dst.get(k), dict):
                merge(dst[k], v)
            else:
                dst[k] = v

    merge(d, over)
    return edict(d)


def cfg1_nerf_llff_repr(device="cuda:0"):
    """options/nerf_llff_repr.yaml at 300x400: relu density, noise 1, metric depth [0,1], 1024 rays x (64 + 128)."""
    return _mk(model="nerf", device=device, max_iter=500000,
               arch=dict(density_activ="relu"),
               nerf=dict(depth=dict(param="metric", range=[0, 1]), sample_intvs=64, fine_sampling=True, sample_intvs_fine=128,
                         rand_rays=1024, density_noise_reg=1),
               loss_weight=dict(render=0, render_fine=0), optim=dict(lr=5e-4, lr_end=5e-5))


def cfg2_nerf_inn_llff_hier(device="cuda:0"):
    """nerf_inn_llff.yaml hyper-parameters with 4096 rays x (64 coarse + 128 fine) hierarchical samples
    (BASELINE.json configs[1]), rendered from NVP-warped rays (barf_inn_llff get_pose)."""
    return _mk(model="barf_inn_llff", device=device, barf_c2f=[0.1, 0.5],
               nerf=dict(sample_intvs=64, fine_sampling=True, sample_intvs_fine=128, rand_rays=4096),
               loss_weight=dict(render=0, render_fine=0))


def cfg3_barf_inn_llff(device="cuda:0", global_alignment=4):
    """scripts/train_llff.sh:1 -- barf_inn_llff.yaml, --barf_c2f=[0.1,0.5], 2048 rays x 128 samples."""
    return _mk(model="barf_inn_llff", device=device, barf_c2f=[0.1, 0.5],
               loss_weight=dict(render=0, global_alignment=global_alignment))


LLFF_TRAIN_VIEWS = dict(fern=18, flower=31, fortress=38, horns=56, leaves=24, orchids=23, room=37, trex=50)


def cfg5_barf_inn_dtu(device="cuda:0"):
    """scripts/train_dtu.sh:6 -- barf_inn_dtu.yaml, 3 sparse views, metric depth [1.2, 5.2]."""
    return _mk(model="barf_inn_dtu", device=device, barf_c2f=[0.1, 0.5],
               nerf=dict(depth=dict(param="metric", range=[1.2, 5.2])),
               data=dict(dataset="dtu", scene="scan65"),
               pose=dict(parameterization="inn", init="noisy_gt", noise=0.15, n_first_fixed_poses=0),
               loss_weight=dict(render=0, global_alignment=3))


# yaml name (the reference's --yaml argument) -> builder, for options.set() when no yaml directory is given
BY_YAML = dict(nerf_llff_repr=cfg1_nerf_llff_repr, nerf_inn_llff=cfg2_nerf_inn_llff_hier, barf_inn_llff=cfg3_barf_inn_llff,
               barf_inn_dtu=cfg5_barf_inn_dtu)
