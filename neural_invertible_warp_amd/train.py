"""Training entry point with the reference's command line and call sequence (train.py:9-32):

    python -m neural_invertible_warp_amd.train --model=barf_inn_llff --yaml=barf_inn_llff \\
        --barf_c2f=[0.1,0.5] --loss_weight.global_alignment=4 --data.scene=fern [--max_iter=N] [--options_dir=DIR]

Multi-GPU: `python -m torch.distributed.run --nproc-per-node N -m neural_invertible_warp_amd.train ...` (rays are
sharded over the ranks, gradients all-reduced; parallel.py).  Model modules are located by name like the reference does.
"""
import importlib
import sys

import torch

from . import options


def main(argv=None):
    opt_cmd = options.parse_arguments(sys.argv[1:] if argv is None else argv)
    opt = options.set(opt_cmd)
    options.save_options_file(opt)
    ctx = torch.cuda.device(opt.device) if opt.device != "cpu" else torch.device("cpu")
    with ctx:
        model = importlib.import_module("neural_invertible_warp_amd.model.{}".format(opt.model))
        m = model.Model(opt)
        m.load_dataset(opt)
        m.build_networks(opt)
        m.setup_optimizer(opt)
        m.restore_checkpoint(opt)
        m.setup_visualizer(opt)
        m.train(opt)
    return m


if __name__ == "__main__":
    main()
