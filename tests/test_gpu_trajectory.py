"""Long-horizon parity, run by the driver (round 4; rounds 2-3 kept it as a builder-run tool, tools/trajectory_parity.py): 240 chained
train iterations of barf_inn_llff WITH the alignment term on the demo scene, HIP engine vs the oracle (autograd + torch.optim.Adam on the
same GPU through torch's kernels) from identical weights with identical pixel and stratified draws (oracle/parity.long_trajectories).

fp32 trajectories of a non-convex optimisation separate chaotically, so "the same trajectory" is measured against the spread between
HIP runs that differ ONLY in their random draws: the HIP-vs-oracle PSNR gap must stay an order of magnitude inside that spread, and small
in absolute terms.  Needs a GPU (~40 s)."""
import pytest

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_240_step_training_trajectory_tracks_the_oracle():
    from oracle import parity
    steps = 240
    kw = dict(steps=steps, views=8, size=(48, 64), R=256, S=64, log_every=20)
    pair = parity.long_trajectories(DEV, draw_seed=0, oracle=True, **kw)
    others = [parity.long_trajectories(DEV, draw_seed=s, oracle=False, **kw)["hip"] for s in (1, 2)]
    n = len(pair["it"])
    assert pair["it"][-1] == steps - 1 and n >= 12
    gap = [abs(a - b) for a, b in zip(pair["hip"]["psnr"], pair["oracle"]["psnr"])]
    spread = [max(x) - min(x) for x in zip(pair["hip"]["psnr"], *[o["psnr"] for o in others])]
    rot_gap = [abs(a - b) for a, b in zip(pair["hip"]["rel_rot"], pair["oracle"]["rel_rot"])]
    half = n // 2
    mean_gap, mean_spread = sum(gap[half:]) / (n - half), sum(spread[half:]) / (n - half)
    print(f"{steps} steps: PSNR hip {pair['hip']['psnr'][-1]:.3f} dB oracle {pair['oracle']['psnr'][-1]:.3f} dB; |dPSNR| max {max(gap):.3f} dB, second-half mean "
          f"{mean_gap:.3f} dB; spread between HIP runs that differ in their draws: second-half mean {mean_spread:.3f} dB; "
          f"relative rotation error hip {pair['hip']['rel_rot'][-1]:.3f} deg oracle {pair['oracle']['rel_rot'][-1]:.3f} deg (max gap {max(rot_gap):.3f})")
    assert pair["hip"]["psnr"][-1] > pair["hip"]["psnr"][0] + 3.0, "the run must actually train"
    assert mean_gap <= 0.25 and max(gap) <= 0.6, (mean_gap, max(gap))
    assert mean_gap <= 0.25 * mean_spread, (mean_gap, mean_spread)
    assert max(rot_gap) <= 0.15, rot_gap


def test_160_step_trajectory_with_the_fine_pass_tracks_the_oracle():
    """The cfg2 shape (round 5): coarse pass, inverse-CDF resampling, FINE network on the merged positions, both photometric losses, one
    Adam over both networks (reference model/nerf.py:34-38, 293-319; rays from the warp, model/nerf_inn_llff.py:493-573).  cfg2's warp
    gradients agree with float64 only to a percent in a single step -- ANY fp32 evaluation does (profiles/r4_fp64_parity.txt) -- so one
    step can only catch errors above a percent there; a chained run can: an error of that size in the warp's gradients would steer the
    poses, and with them the fine image, away from the oracle's within a few dozen steps.  Same criterion as the run above: the
    HIP-vs-oracle gap of the fine image's PSNR inside a quarter of the spread between HIP runs that differ only in their draws."""
    from oracle import parity
    steps = 160
    kw = dict(steps=steps, views=6, size=(48, 64), R=192, S=32, Sf=32, log_every=10)
    pair = parity.long_trajectories(DEV, draw_seed=0, oracle=True, **kw)
    others = [parity.long_trajectories(DEV, draw_seed=s, oracle=False, **kw)["hip"] for s in (1, 2)]
    n = len(pair["it"])
    assert pair["it"][-1] == steps - 1 and n >= 12
    half = n // 2
    for key in ("psnr", "psnr_coarse"):
        gap = [abs(a - b) for a, b in zip(pair["hip"][key], pair["oracle"][key])]
        spread = [max(x) - min(x) for x in zip(pair["hip"][key], *[o[key] for o in others])]
        mean_gap, mean_spread = sum(gap[half:]) / (n - half), sum(spread[half:]) / (n - half)
        print(f"{steps} steps, {key}: hip {pair['hip'][key][-1]:.3f} dB oracle {pair['oracle'][key][-1]:.3f} dB; |d| max {max(gap):.3f} dB, second-half mean "
              f"{mean_gap:.3f} dB; spread between HIP runs that differ in their draws: second-half mean {mean_spread:.3f} dB")
        assert mean_gap <= 0.25 and max(gap) <= 0.6, (key, mean_gap, max(gap))
        assert mean_gap <= 0.25 * mean_spread, (key, mean_gap, mean_spread)
    rot_gap = [abs(a - b) for a, b in zip(pair["hip"]["rel_rot"], pair["oracle"]["rel_rot"])]
    print(f"relative rotation error hip {pair['hip']['rel_rot'][-1]:.3f} deg oracle {pair['oracle']['rel_rot'][-1]:.3f} deg (max gap {max(rot_gap):.3f})")
    assert pair["hip"]["psnr"][-1] > pair["hip"]["psnr"][0] + 3.0, "the run must actually train"
    assert max(rot_gap) <= 0.15, rot_gap
