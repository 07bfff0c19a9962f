"""The driver's N > 1 invocation rehearsed at the widest width a one-GPU box allows (round 6).  The pool's process guard admits six GPU
processes at a time and this test process is one of them, so the rehearsal here runs FOUR gloo ranks time-slicing this GPU (the driver's
own N = 4 command line); tools/rehearse_ranks.sh runs five outside pytest and keeps the lines (profiles/r6_bench_lines.json).  Checked: every
rank is seen, the ranks' parameters agree after the timed steps (bench.py exits non-zero otherwise), losses are finite, the line carries
the self-diagnosing fields (split_exchange on / off, launched vs replayed, wire time of the bucket) and the run stays far inside the
driver's 600 s."""
import json
import os
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RANKS = 4


def _driver_line(extra, port, timeout=900, env_extra=None):
    """-> (line, wall seconds).  ONE logged retry when a rank reports a non-finite loss.  That happened once in ~45 runs of four or five
    ranks sharing this one GPU and was traced to the gfx950 store-data hazard firing in the training forward when a wave of another kernel
    shares its SIMD (HISTORY.md round 6; fixed in csrc/niw_mlp_fwd.hip, guarded by tests/test_store_hazard.py); the retry stays as a
    seat belt for what several processes on one device may still expose that one process per GPU does not."""
    try:
        return _driver_line_once(extra, port, timeout, env_extra)
    except AssertionError as e:
        if "training loss is" not in str(e):
            raise
        print(f"[test_gpu_ranks] a rank diverged while four processes shared the GPU; one retry.  First failure: {str(e)[-600:]}", flush=True)
        return _driver_line_once(extra, port + 40, timeout, env_extra)


def _driver_line_once(extra, port, timeout=900, env_extra=None):
    """bench.py as the four ranks torch.distributed.run would start (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), started
    directly: the agent process would be a sixth GPU process beside this one and the four ranks -- exactly the pool's limit, and a run
    that crosses it is killed whole.  The ranks' program is the same either way (bench.py reads the environment the agent would set);
    the agent itself is exercised by tests/test_gpu_sharding.py::test_bench_line_under_torchrun_with_two_ranks."""
    import torch
    torch.cuda.empty_cache()          # this process's cached blocks (earlier full-size tests) go back to the device the four ranks share
    base = dict(os.environ, NIW_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(RANKS),
                OMP_NUM_THREADS="4", **(env_extra or {}))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(RANKS), "--steps", "3", "--warmup", "2"] + extra
    t0 = time.perf_counter()
    import tempfile
    files = [(tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")) for _ in range(RANKS)]       # (no pipe can fill while a peer is waited for)
    procs = [subprocess.Popen(cmd, cwd=ROOT, env=dict(base, RANK=str(r), LOCAL_RANK=str(r)), stdout=files[r][0], stderr=files[r][1], text=True)
             for r in range(RANKS)]
    try:
        for p in procs:
            p.wait(timeout=timeout)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    wall = time.perf_counter() - t0
    outs = []
    for fo, fe in files:
        fo.seek(0); fe.seek(0)
        outs.append((fo.read(), fe.read()))
        fo.close(); fe.close()
    for p, (out, err) in zip(procs, outs):
        assert p.returncode == 0, err[-3000:]
    lines = [l for out, _ in outs for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == RANKS and line["ranks_seen"] == RANKS and line["backend"] == "gloo"
    assert wall < 600, f"{wall:.0f} s: the driver allows 600"
    return line, wall


def test_four_ranks_default_workload_line_is_self_diagnosing():
    """cfg2 (the driver's default): weak headline + strong, and the A/B legs of the weak workload"""
    import math
    line, wall = _driver_line([], 29731)
    assert line["scaling"] == "weak" and line["strong"]["scaling"] == "strong" and math.isfinite(line["loss"]) and math.isfinite(line["strong"]["loss"])
    assert line["config"]["rays_per_gpu"] == 4095 and line["strong"]["rays_per_gpu"] in (1021, 1022)
    se, hg = line["split_exchange"], line["hip_graph_ab"]
    assert se["on_ms"] > 0 and se["off_ms"] > 0 and se["auto_resolves_to"] == "on"          # weak-scaled cfg2: two rounds of workgroups and more
    assert se["on"]["comm_exposed_ms"] is not None and se["off"]["comm_ms"] is not None
    assert hg["launched_ms"] > 0 and (hg["replayed_ms"] is None and "capture_failed" in hg or hg["replayed_ms"] > 0)
    assert line["comm_bucket_bytes"] == 4 * 1228308 and abs(line["comm_wire_ms"] - 2 * 3 / 4 * 4913232 / 153e9 * 1e3) < 1e-3
    assert line["comm_ms"] is not None and line["comm_exposed_ms"] is not None
    assert 10 < line["peak_memory_gb"] < 40, line["peak_memory_gb"]            # a rank's 16.6 GB workspace (+ batch, parameters); 288 GB per GPU
    assert line["param_checksum"] is not None
    print(f"4 gloo ranks, cfg2: {wall:.0f} s wall; weak {line['ms_per_step']:.1f} ms, strong {line['strong']['ms_per_step']:.1f} ms; split exchange on/off "
          f"{se['on_ms']}/{se['off_ms']} ms; launched/replayed {hg['launched_ms']}/{hg['replayed_ms']} ms")


def test_four_ranks_dtu_three_views_leave_a_rank_without_a_view():
    """cfg5: 3 views x 682 rays over 4 ranks -- rank 3's share lies inside view 2, whose first ray is rank 2's: it owns no view, counts no
    alignment term, and still all-reduces; 8 ranks leave five such ranks (tools/rehearse_ranks.sh: five ranks, two such)"""
    import math
    from neural_invertible_warp_amd import parallel
    wins = [parallel.ViewWindow(3, 682, r, RANKS) for r in range(RANKS)]
    assert [w.own1 - w.own0 for w in wins] == [1, 1, 1, 0]
    line, wall = _driver_line(["--config", "cfg5", "--kernel-steps", "0"], 29733)
    assert math.isfinite(line["loss"]) and math.isfinite(line["strong"]["loss"])
    assert line["split_exchange"]["on_ms"] is None and line["split_exchange"]["off_ms"] > 0          # no fine network: nothing to split
    print(f"4 gloo ranks, cfg5: {wall:.0f} s wall; weak {line['ms_per_step']:.1f} ms, strong {line['strong']['ms_per_step']:.1f} ms")


def test_four_ranks_all_eight_scenes_both_placements():
    """cfg4: ray shard (one all-reduce per scene and step) and the replicas (two scenes per rank at N = 4, no exchange)"""
    from neural_invertible_warp_amd import configs
    # (--scaling strong: the eight scenes' own batches split four ways, 10 GB per rank; the weak-scaled form -- 39 GB of workspaces per
    # rank, four ranks on this one device beside whatever this test process still holds -- is tools/rehearse_ranks.sh's)
    line, wall = _driver_line(["--config", "cfg4", "--kernel-steps", "0", "--ab", "off", "--scaling", "strong"], 29735, timeout=1200)
    rep = line["replicas"]
    assert line["placement"] == "shard" and rep["comm_ms"] == 0 and line["scaling"] == "strong"
    assert {r["scene"]: r["rank"] for r in rep["scenes"]} == {sc: i % RANKS for i, sc in enumerate(configs.LLFF_TRAIN_VIEWS)}
    assert line["comm_bucket_bytes"] == 8 * 4 * (698256 - 18 * 128) + 4 * 128 * sum(configs.LLFF_TRAIN_VIEWS.values())
    print(f"4 gloo ranks, cfg4: {wall:.0f} s wall; shard {line['ms_per_step']:.1f} ms, replicas {rep['ms_per_step']:.1f} ms")


def test_a_failed_capture_on_one_rank_costs_the_field_not_the_line():
    """the replayed A/B leg: rank 2's capture "fails" (injected; a real failure leaves that rank's HIP state unusable).  Every rank must learn
    of it host-side BEFORE any rank replays -- a replay's all-reduce would wait for rank 2 for ever -- skip the leg, and leave with exit
    code 0 after rank 0 has printed the complete line, whose hip_graph_ab says what happened"""
    line, wall = _driver_line(["--config", "cfg3", "--kernel-steps", "0"], 29737, env_extra={"NIW_TEST_FAIL_CAPTURE_RANK": "2"})
    hg = line["hip_graph_ab"]
    assert hg["replayed_ms"] is None and hg["launched_ms"] > 0 and "rank(s) [2]" in hg["capture_failed"], hg
    assert line["strong"] is not None and line["split_exchange"]["off_ms"] > 0 and line["value"] > 0
    print(f"4 gloo ranks, cfg3, capture failure injected on rank 2: {wall:.0f} s wall; line complete, hip_graph_ab = {hg}")
