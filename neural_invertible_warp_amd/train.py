"""Training entry point: the reference's command line and engine call sequence (train.py:9-32).

    python -m neural_invertible_warp_amd.train --model=barf_inn_llff --yaml=barf_inn_llff \\
        --barf_c2f=[0.1,0.5] --loss_weight.global_alignment=4 --data.scene=fern [--max_iter=N] [--options_dir=DIR]

Multi-GPU: `python -m torch.distributed.run --nproc-per-node N -m neural_invertible_warp_amd.train ...` (rays are sharded over the
ranks, gradients all-reduced; parallel.py).  The engine class is looked up by model name, like the reference does, and driven
through the same stages in the same order.
"""
import contextlib
import importlib
import sys

import torch

from . import options

# the reference engine's stages, in call order; every `Model` class of this package implements them with (opt) arguments
STAGES = ("load_dataset", "build_networks", "setup_optimizer", "restore_checkpoint", "setup_visualizer", "train")


def build_options(argv):
    opt = options.set(options.parse_arguments(argv))
    options.save_options_file(opt)
    return opt


def run(opt):
    """-> the engine object after training (its .graph / .it / .test_data are what callers inspect)"""
    engine_cls = importlib.import_module("{}.model.{}".format(__package__, opt.model)).Model
    on_gpu = opt.device != "cpu"
    with (torch.cuda.device(opt.device) if on_gpu else contextlib.nullcontext()):
        engine = engine_cls(opt)
        for stage in STAGES:
            getattr(engine, stage)(opt)
    return engine


def main(argv=None):
    return run(build_options(sys.argv[1:] if argv is None else argv))


if __name__ == "__main__":
    main()
