"""Option trees (`opt`).

`BY_YAML` restates, key for key on everything the render path and the engine read, the reference's yaml files as `options.load_options`
resolves them through their `_parent_` chains (options/nerf_llff_repr.yaml, nerf_inn_llff.yaml, barf_inn_llff.yaml,
barf_inn_dtu.yaml): `--yaml=<name>` without `--options_dir` therefore trains the same problem as the reference command line
(tests/test_options.py diffs the two wherever a reference checkout is present).  The BASELINE.json benchmark configurations are
separate builders (`cfg1 .. cfg5`): a yaml tree plus exactly the overrides of the launch line they stand for.
"""
import copy

from .util import edict

_ARCH = dict(layers_feat=[None, 256, 256, 256, 256, 256, 256, 256, 256], layers_rgb=[None, 128, 3], skip=[4],
             posenc=dict(L_3D=10, L_view=4), density_activ="softplus", tf_init=True)

# options/base.yaml + nerf_inn_llff.yaml (the parent of every INN configuration)
_NERF_INN_LLFF = dict(
    model=None, yaml=None, seed=0, device="cuda:0", gpu=0, cpu=False, H=480, W=640, max_iter=200000,
    group="0_test", name="debug", output_root="output", resume=False, load=None, batch_size=None, max_epoch=None,
    freq=dict(scalar=200, vis=2000, val=2000, ckpt=5000),
    arch=_ARCH,
    nerf=dict(view_dep=True, depth=dict(param="inverse", range=[1, 0]), sample_intvs=128, sample_stratified=True,
              fine_sampling=False, sample_intvs_fine=None, rand_rays=2048, density_noise_reg=None, setbg_opaque=None),
    data=dict(dataset="llff", root=None, scene="fern", image_size=[480, 640], center_crop=None, val_ratio=0.1,
              train_sub=None, val_sub=None, val_on_test=False, preload=True, augment={}, num_workers=4, llffhold=8),
    camera=dict(model="perspective", ndc=False),
    loss_weight=dict(render=0, render_fine=None, global_alignment=None),
    optim=dict(algo="Adam", lr=1e-3, lr_end=1e-4, sched=dict(type="ExponentialLR", gamma=None)),
)

# options/barf_inn_llff.yaml on top
_BARF_INN = dict(
    barf_c2f=None,
    camera=dict(noise_type="barf", noise_barf=None, noise_l2g_r=None, noise_l2g_t=None),
    optim=dict(lr_pose=5e-4, lr_pose_end=1e-8, lr_feature=1e-3, sched_pose=dict(type="ExponentialLR", gamma=None, step_size=None),
               warmup_pose=None, test_photo=True, test_iter=100),
    inn=dict(real_nvp=dict(c2f=True, max_pe_iter=100000, d_hidden=128, multires=6), actfn="softplus", optimize=dict(enabled=True)),
    warp_latent=dict(enc_type="l2fbarf", optimize=dict(enabled=True), embed_dim=128),
)

# options/nerf_inn_dtu.yaml + barf_inn_dtu.yaml
_DTU = dict(
    barf_c2f=None,
    nerf=dict(depth=dict(param="metric", range=[1, 0])),      # metric sampling takes the range from the data (var.depth_range)
    data=dict(dataset="dtu", scene="scan82", image_size=[300, 400],
              dtu=dict(split_type=None, dtuhold=8, train_sub=None, val_sub=None, crop_ratio=None, crop=None, resize_by="max", resize=None,
                       resize_factor=None, mask_img=False, light_cond=3, max_images=49, increase_depth_range_by_x_percent=0)),
    camera=dict(noise=None),
    optim=dict(lr_pose=5e-4, lr_pose_end=1e-8, sched_pose=dict(type="ExponentialLR", gamma=None), warmup_pose=None, test_photo=True,
               test_iter=100),
    freq=dict(early_termination=100000),
    inn=dict(real_nvp=dict(c2f=True, max_pe_iter=100000, d_hidden=128, multires=6, latent_dim=128), actfn="softplus"),
    pose=dict(parameterization="inn", init="noisy_gt", noise=0.15, n_first_fixed_poses=0, optimize_relative_poses=False,
              dtu_reconstruction=False),
)


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _tree(*layers, **over):
    d = {}
    for layer in layers + (over,):
        _merge(d, layer)
    opt = edict(d)
    opt.H, opt.W = opt.data.image_size
    return opt


# ------------------------------------------------------------------------------------------ the reference's yaml files
def yaml_nerf_llff_repr(device="cuda:0"):
    """options/nerf_llff_repr.yaml: vanilla NeRF reproduction on LLFF (relu density + noise, metric depth [0,1], 1024 rays x (64 + 128))"""
    opt = _tree(_NERF_INN_LLFF, device=device, max_iter=500000, freq=dict(vis=1000),
                arch=dict(density_activ="relu"),
                nerf=dict(depth=dict(param="metric", range=[0, 1]), sample_intvs=64, fine_sampling=True, sample_intvs_fine=128,
                          rand_rays=1024, density_noise_reg=1),
                optim=dict(lr=5e-4, lr_end=5e-5))
    opt.loss_weight = edict(render=0, render_fine=0)            # this yaml descends from base.yaml directly: no alignment term
    return opt


def yaml_nerf_inn_llff(device="cuda:0"):
    """options/nerf_inn_llff.yaml"""
    return _tree(_NERF_INN_LLFF, device=device)


def yaml_barf_inn_llff(device="cuda:0"):
    """options/barf_inn_llff.yaml"""
    return _tree(_NERF_INN_LLFF, _BARF_INN, device=device)


def yaml_barf_inn_dtu(device="cuda:0"):
    """options/barf_inn_dtu.yaml"""
    return _tree(_NERF_INN_LLFF, _DTU, device=device)


# yaml name (the reference's --yaml argument) -> builder, for options.set() when no yaml directory is given
BY_YAML = dict(nerf_llff_repr=yaml_nerf_llff_repr, nerf_inn_llff=yaml_nerf_inn_llff, barf_inn_llff=yaml_barf_inn_llff,
               barf_inn_dtu=yaml_barf_inn_dtu)

# ------------------------------------------------------------------------------------------ BASELINE.json configurations
_BENCH_IMAGE = dict(data=dict(image_size=[300, 400]))        # "300x400 LLFF scene" of BASELINE.json north_star


def cfg1_nerf_llff_repr(device="cuda:0"):
    """configs[0]: options/nerf_llff_repr.yaml at 300x400 (--model=nerf --yaml=nerf_llff_repr --data.image_size=[300,400])"""
    return _tree(yaml_nerf_llff_repr(device), _BENCH_IMAGE, model="nerf")


def cfg2_nerf_inn_llff_hier(device="cuda:0"):
    """configs[1]: nerf_inn_llff.yaml hyper-parameters with 4096 rays x (64 coarse + 128 fine) hierarchical samples, rendered from
    NVP-warped rays (barf_inn_llff get_pose) with the c2f encoding of the shipped scripts"""
    return _tree(yaml_barf_inn_llff(device), _BENCH_IMAGE, model="barf_inn_llff", barf_c2f=[0.1, 0.5],
                 nerf=dict(sample_intvs=64, fine_sampling=True, sample_intvs_fine=128, rand_rays=4096),
                 loss_weight=dict(render=0, render_fine=0))


def cfg3_barf_inn_llff(device="cuda:0", global_alignment=4):
    """configs[2]: scripts/train_llff.sh:1 -- --model=barf_inn_llff --yaml=barf_inn_llff --barf_c2f=[0.1,0.5]
    --loss_weight.global_alignment=4 (2048 rays x 128 samples)"""
    return _tree(yaml_barf_inn_llff(device), _BENCH_IMAGE, model="barf_inn_llff", barf_c2f=[0.1, 0.5],
                 loss_weight=dict(global_alignment=global_alignment))


# train views of the 8 LLFF scenes (images minus the held-out 10 %, data/llff.py:32-33): configs[3]
LLFF_TRAIN_VIEWS = dict(fern=18, flower=31, fortress=38, horns=56, leaves=24, orchids=23, room=37, trex=50)


def cfg5_barf_inn_dtu(device="cuda:0"):
    """configs[4]: scripts/train_dtu.sh:6 -- --model=barf_inn_dtu --yaml=barf_inn_dtu --barf_c2f=[0.1,0.5] --data.scene=scan65
    --data.dtu.split_type=pixelnerf --data.dtu.train_sub=3 --loss_weight.global_alignment=3"""
    return _tree(yaml_barf_inn_dtu(device), model="barf_inn_dtu", barf_c2f=[0.1, 0.5],
                 data=dict(scene="scan65", dtu=dict(split_type="pixelnerf", train_sub=3)),
                 loss_weight=dict(global_alignment=3))
