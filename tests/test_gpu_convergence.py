"""Whole-pipeline sanity on a geometrically consistent scene (tools/teacher_student_demo.py): images of an analytic
blob scene rendered from perturbed poses; barf_inn_llff trains from identity poses through engine.INNTrainer (warp,
c2f encoding, MLP, compositing, global-alignment loss, fused Adam).  Per-step parity is covered elsewhere; this
checks that hundreds of chained steps actually fit the scene and that, with the alignment-loss weight of the reference's
training script (10^4), the learnt poses move TOWARDS the ground truth (gauge-free measure: relative rotations between all pairs
of views; full recovery needs the reference's 200k-iteration schedule and real scenes, and is not asserted)."""
import math

import pytest

pytestmark = pytest.mark.gpu


def test_trains_a_consistent_scene():
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import teacher_student_demo as demo
    hist = demo.run(steps=800, views=8, size=(40, 56), log_every=100, quiet=True, ga=4)
    psnr, rel = [h[1] for h in hist], [h[4] for h in hist]
    assert all(math.isfinite(p) for p in psnr)
    assert psnr[0] < 16 and psnr[-1] > 30 and psnr[-1] - psnr[0] > 15, psnr          # 12 dB -> ~40 dB on MI355X
    assert all(math.isfinite(h[3]) for h in hist[1:])                                # pose evaluation runs on the trained model
    assert rel[-1] < rel[0] - 0.7, rel                                               # 5.9 deg (identity poses) -> ~4 deg
