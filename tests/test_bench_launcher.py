"""bench.py must be able to start its own ranks: `python bench.py --gpus N` with no rank environment re-launches itself under
torch.distributed.run as a CHILD process, from a parent that has not imported torch (the GPU box refuses an exec or a
fork from a GPU-initialised process).  CPU-only checks of that decision; the ranks themselves need a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_decision():
    import bench
    assert bench.needs_launcher(8, {})                                   # the driver's `python bench.py --gpus 8`
    assert not bench.needs_launcher(1, {})                               # one GPU: run in place
    assert not bench.needs_launcher(8, {"WORLD_SIZE": "8", "RANK": "3"})  # already a rank of torch.distributed.run
    assert not bench.needs_launcher(2, {"RANK": "0"})


def test_launcher_command_line():
    import bench
    argv = ["--gpus", "4", "--steps", "7", "--config", "cfg3"]
    cmd = bench.launcher_command(4, argv, port=29876)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29876"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv


def test_parent_does_not_import_torch_and_propagates_the_exit_code(tmp_path):
    """The parent of a `--gpus 2` run: replaces torch.distributed.run by a stub module that records its arguments and exits 7; the
    parent must hand the arguments over unchanged, never import torch, and exit with the child's code."""
    stub = tmp_path / "torch" / "distributed"
    stub.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (stub / "__init__.py").write_text("")
    (stub / "run.py").write_text("import sys, json, os\n"
                                 "open(os.environ['NIW_STUB_OUT'], 'w').write(json.dumps(sys.argv[1:]))\n"
                                 "sys.exit(7)\n")
    out = tmp_path / "argv.json"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=str(tmp_path), NIW_STUB_OUT=str(out))
    probe = ("import sys, runpy\n"
             "sys.argv = ['bench.py', '--gpus', '2', '--steps', '3', '--lean']\n"
             "try:\n"
             "    runpy.run_path(%r, run_name='__main__')\n"
             "except SystemExit as e:\n"
             "    assert 'torch.cuda' not in sys.modules and 'torch._C' not in sys.modules, 'parent initialised torch'\n"
             "    sys.exit(e.code)\n" % os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 7, (r.returncode, r.stdout, r.stderr)
    import json
    argv = json.loads(out.read_text())
    assert "--nproc-per-node=2" in argv and argv[-5:] == ["--gpus", "2", "--steps", "3", "--lean"]


def test_parent_restarts_fresh_ranks_without_the_graph_when_a_capture_failed(tmp_path):
    """`--hip-graph on` under N > 1: a rank whose capture failed leaves the marker file and dies; the parent (which never touched the
    GPU) must start a SECOND set of ranks with `--hip-graph off` appended and exit with THEIR code.  The stub plays both generations."""
    stub = tmp_path / "torch" / "distributed"
    stub.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (stub / "__init__.py").write_text("")
    (stub / "run.py").write_text("import sys, json, os\n"
                                 "log = os.environ['NIW_STUB_OUT']\n"
                                 "runs = json.load(open(log)) if os.path.exists(log) else []\n"
                                 "runs.append(sys.argv[1:])\n"
                                 "json.dump(runs, open(log, 'w'))\n"
                                 "if len(runs) == 1:\n"
                                 "    open(os.environ['NIW_CAPTURE_FAILED_FILE'], 'w').write('capture failed')\n"
                                 "    sys.exit(1)\n"
                                 "sys.exit(0)\n")
    out = tmp_path / "runs.json"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=str(tmp_path), NIW_STUB_OUT=str(out))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--lean", "--hip-graph", "on"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    import json
    runs = json.loads(out.read_text())
    assert len(runs) == 2
    assert runs[0][-2:] == ["--hip-graph", "on"] and runs[1][-4:] == ["--hip-graph", "on", "--hip-graph", "off"]      # argparse: the last one wins
    port = lambda argv: argv[argv.index("--master-port") + 1]
    assert port(runs[0]) != port(runs[1])                                 # a fresh rendezvous
    assert "starting fresh ranks" in r.stderr


def test_a_failing_rank_without_the_marker_is_not_retried(tmp_path):
    stub = tmp_path / "torch" / "distributed"
    stub.mkdir(parents=True)
    (tmp_path / "torch" / "__init__.py").write_text("")
    (stub / "__init__.py").write_text("")
    (stub / "run.py").write_text("import sys, os\nopen(os.environ['NIW_STUB_OUT'], 'a').write('x')\nsys.exit(3)\n")
    out = tmp_path / "count"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PYTHONPATH=str(tmp_path), NIW_STUB_OUT=str(out))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--lean"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and out.read_text() == "x"


def test_replica_placement_deals_every_scene_to_exactly_one_rank():
    """bench.py --config cfg4 --placement replicas: scene i trains whole on rank i mod N (the reference runs the eight
    scenes as eight independent processes, scripts/train_llff.sh:1-8)"""
    import bench
    from neural_invertible_warp_amd import configs
    scenes = list(configs.LLFF_TRAIN_VIEWS)
    assert len(scenes) == 8
    for world in (1, 2, 4, 8):
        parts = [bench.scenes_of_rank(scenes, r, world) for r in range(world)]
        assert sorted(sc for p in parts for sc in p) == sorted(scenes)             # exhaustive, disjoint
        assert {len(p) for p in parts} == {8 // world}                             # balanced
    assert [bench.scenes_of_rank(scenes, r, 8) for r in range(8)] == [[sc] for sc in scenes]
    assert bench.scenes_of_rank(scenes, 0, 8) == ["fern"]                          # the single-GPU proxy `--placement replicas --shard-of 8`


def test_cpu_baseline_reports_threads_sample_and_calibration():
    """bench.py's cpu_baseline leg (the oracle timed on the host cores) on a toy shape: it must say how many cores the process may use,
    which thread count it chose from its sweep, what the sample was, and carry the build container's oracle / reference calibration"""
    import bench
    out = bench.cpu_baseline(B=3, S=8, Sf=8, H=12, W=16, budget_s=5.0)
    assert out["kind"] == "port" and out["unit"] == "ray-samples/s" and out["value"] > 0
    assert out["cores"] in {int(k) for k in out["thread_sweep"]} and out["affinity_cores"] >= 1
    assert "3 views x" in out["sample"] and len(out["seconds_per_step"]) >= 3             # 1 warm-up + >= 3 timed steps
    assert out["value"] >= out["value_median"] > 0 and out["forward_only_value"] > out["value_median"] and out["cpu_model"]
    assert 0.9 <= out["calibration"]["oracle_over_reference_time"] <= 1.1            # the port times within 10 % of the reference
    assert "r5_oracle_calibration.json" in out["calibration"]["source"]


def test_wire_time_of_the_gradient_exchange():
    """comm_wire_ms of the N > 1 line: a ring all-reduce moves 2 (N - 1) / N x the bucket over one xGMI link per neighbour (153 GB/s)"""
    import bench
    assert bench.wire_ms(4_913_232, 1) == 0.0
    assert abs(bench.wire_ms(4_913_232, 8) - 2 * 7 / 8 * 4_913_232 / 153e9 * 1e3) < 1e-12          # cfg2's bucket: 0.056 ms
    assert abs(bench.wire_ms(153e9, 2) - 1000.0) < 1e-6


def _flag_rank(rank, world, port, ok, q):
    import torch.distributed as dist
    import bench
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank, bench.host_flags_agree("t", ok, rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def test_capture_outcome_is_agreed_on_host_side_with_two_ranks():
    """the replayed A/B leg of the N > 1 line: every rank learns through the store -- no device work, no collective -- that rank 1's capture
    failed, before any rank enters a replay's all-reduce"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_flag_rank, args=(r, 2, 29877, r == 0, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {0: [1], 1: [1]}
