"""Diagnostic (round 4): host time of every train iteration from process start, through a one-rank RCCL group.
    python tools/rccl_probe.py [ev]
Prints `host_ms@ms_since_init` per call.  What it showed: launched (not replayed) iterations stall the host ONCE for 50-100 ms around the
seventh call -- the HIP runtime growing an internal pool under the cross-stream events of the second stream / the collective's stream --
and never again; bench.py therefore runs at least 16 untimed iterations in front of a launched timed region."""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29535")
import torch.distributed as dist
from neural_invertible_warp_amd import parallel
import bench
torch.cuda.set_device(0)
parallel.init_from_env(force=True)
t_init = time.perf_counter()
loads, _ = bench.build_workloads("cfg3", "cuda:0", 0, 1, "weak", 0, hip_graph=False)
tr, var0 = loads[0][:2]
if len(sys.argv) > 1: tr.comm_events = []
host = []
for i in range(80):
    a = time.perf_counter(); tr.train_iteration(type(var0)(var0)); b = time.perf_counter()
    host.append((1e3*(b-a), 1e3*(a - t_init)))
torch.cuda.synchronize()
print(" ".join(f"{h:.1f}@{t:.0f}" for h, t in host))
dist.destroy_process_group()
