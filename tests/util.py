"""Shared helpers for the test-suite."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def t(a):
    return torch.from_numpy(np.asarray(a, dtype=np.float32))


def check_grad_summary(grad, g, key, rtol=1e-3):
    """Gradient fixtures hold the L2 norm and a strided sample (see make_golden.gsum).
    Tolerance: rtol * max|g| absolute on the sample, rtol relative on the norm."""
    assert grad is not None, key
    f = grad.detach().reshape(-1).double().cpu()
    norm, sample, stride = float(g[key + ".norm"]), g[key + ".sample"], int(g[key + ".stride"])
    mine = f[::stride].numpy()
    scale = max(float(np.abs(sample).max()), 1e-12)
    err = float(np.abs(mine - sample).max())
    assert err <= rtol * scale + 1e-7, f"{key}: sample err {err:.3e} scale {scale:.3e}"
    assert abs(float(f.norm()) - norm) <= rtol * max(norm, 1e-12) + 1e-7, f"{key}: norm {float(f.norm())} vs {norm}"
