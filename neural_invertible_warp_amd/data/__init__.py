"""Datasets feeding the render path: `llff` parses the public LLFF layout
(`poses_bounds.npy` + `images/`) exactly as the reference's data/llff.py does; `synthetic` is the
procedural stand-in used where no dataset is mounted (bench.py, tests, the GPU box)."""
