"""Option handling with the reference's command-line syntax and `_parent_` yaml inheritance (options.py:13-92)."""
import pytest

from neural_invertible_warp_amd import options


def test_parse_arguments_syntax():
    o = options.parse_arguments(["--model=barf_inn_llff", "--barf_c2f=[0.1,0.5]", "--loss_weight.global_alignment=4", "--data.root=",
                                 "--resume", "--cpu!", "--optim.lr=1.e-3", "--name=a=b"])
    assert o.model == "barf_inn_llff" and o.barf_c2f == [0.1, 0.5] and o.loss_weight.global_alignment == 4
    assert o.data.root is None and o.resume is True and o.cpu is False and o.optim.lr == 1e-3 and o.name == "a=b"
    with pytest.raises(AssertionError):
        options.parse_arguments(["model=x"])


def test_builtin_tree_and_overrides(tmp_path):
    opt = options.set(options.parse_arguments(["--model=barf_inn_llff", "--yaml=barf_inn_llff", "--cpu", f"--output_root={tmp_path}",
                                               "--data.image_size=[30,40]", "--nerf.rand_rays=512", "--seed=3", "--name=run"]))
    assert opt.device == "cpu" and (opt.H, opt.W) == (30, 40) and opt.nerf.rand_rays == 512
    assert opt.nerf.sample_intvs == 128 and opt.inn.real_nvp.multires == 6          # untouched defaults of the tree
    assert opt.output_path == f"{tmp_path}/0_test/run_seed3"
    options.save_options_file(opt)
    with pytest.raises(KeyError):
        options.set(options.parse_arguments(["--model=x", "--yaml=unknown_yaml"]))


def test_yaml_directory_with_parent_inheritance(tmp_path):
    d = tmp_path / "options"
    d.mkdir()
    (d / "base.yaml").write_text("group: 0_test\nname: debug\nseed: 0\ngpu: 0\ncpu: true\noutput_root: %s\n"
                                 "data: {image_size: [10, 12], scene: a}\noptim: {lr: 1.e-3, algo: Adam}\n" % tmp_path)
    (d / "mid.yaml").write_text("_parent_: options/base.yaml\noptim: {lr: 5.e-4}\nnerf: {rand_rays: 64}\n")
    (d / "leaf.yaml").write_text("_parent_: options/mid.yaml\ndata: {scene: b}\n")
    opt = options.set(options.parse_arguments(["--model=m", "--yaml=leaf", f"--options_dir={d}", "--nerf.rand_rays=32"]))
    assert opt.optim.lr == 5e-4 and opt.optim.algo == "Adam" and opt.data.scene == "b" and opt.data.image_size == [10, 12]
    assert opt.nerf.rand_rays == 32 and opt.model == "m" and (opt.H, opt.W) == (10, 12)
