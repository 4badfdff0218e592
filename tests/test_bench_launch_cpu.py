"""bench.py's own rank launcher (`python bench.py --gpus N` without torchrun), exercised on CPU: the parent stays
GPU-free, starts N child ranks with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, rank 0 prints ONE JSON line whose
n_gpus is the size of the process group (gloo here, RCCL on the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *argv], capture_output=True, text=True, timeout=300, env=e)


@pytest.mark.timeout(600)
def test_gpus2_spawns_two_ranks_and_reports_group_size():
    r = _run("--gpus", "2", "--cpu-stub", "--steps", "3", "--warmup", "0")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert sum(l.lstrip().startswith("{") for l in lines) == 1, r.stdout  # only rank 0 prints a record ...
    line = json.loads(lines[-1])  # ... and it is the LAST stdout line (library banners come before it)
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"


def test_under_a_launcher_it_is_one_rank():
    # WORLD_SIZE already set (torchrun): no second level of spawning, the process is a rank itself
    r = _run("--gpus", "1", "--cpu-stub", "--steps", "2", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_fewer_devices_than_requested_is_an_error():
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has the devices")
    r = _run("--gpus", "2", "--steps", "1")
    assert r.returncode == 2
    assert "HIP device(s) visible" in r.stderr and not r.stdout.strip()


def test_child_failure_ends_the_job():
    # a rank that dies must not leave the others waiting in a collective: the launcher terminates them
    # (the stub's last rank exits with code 3 before the rendezvous when steps < 0)
    r = _run("--gpus", "2", "--cpu-stub", "--steps", "-1")
    assert r.returncode != 0
    assert not any(l.lstrip().startswith("{") for l in r.stdout.splitlines())


def test_cpu_core_pinning_and_baseline_total_flags():
    """--cpu-cores pins a rank to its own slice of the allowed host cores before anything touches the GPU (8 ranks on a 16-core
    cgroup get two each); --baseline-total parses for the speech / stress workloads.  Run in a child: the mask is per process."""
    code = r"""
import os, sys
sys.path.insert(0, %r)
import bench
allowed = sorted(os.sched_getaffinity(0))
n = bench.pin_cpu_cores(2, 1)
now = sorted(os.sched_getaffinity(0))
assert n == min(2, len(allowed)) and len(now) == n and set(now) <= set(allowed), (allowed, now)
if len(allowed) >= 4:
    assert now == allowed[2:4]
a = bench.parse(["--config", "stress", "--baseline-total", "--gpus", "8", "--cpu-cores", "2", "--xylo-lif", "queue:3"])
assert a.baseline_total and a.cpu_cores == 2 and a.gpus == 8 and a.xylo_lif == "queue:3"
print("ok")
""" % os.path.dirname(BENCH)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


def test_schedule_defaults():
    """--schedule auto: the scan-lane schedule (four batches in flight) for the speech workload only; an explicit --streams is kept and
    handed to the child runs, a defaulted one is not."""
    sys.path.insert(0, os.path.dirname(BENCH))
    import bench

    a = bench.parse([])
    assert (a.schedule, a.streams, a.streams_given) == ("graphs", 3, False)
    a = bench.parse(["--config", "speech"])
    assert (a.schedule, a.streams, a.scan_lane_cus) == ("scan-lane", 4, 4)
    a = bench.parse(["--config", "speech", "--schedule", "graphs"])
    assert (a.schedule, a.streams) == ("graphs", 3)
    a = bench.parse(["--config", "stress", "--schedule", "scan-lane", "--streams", "5", "--scan-lane-cus", "6"])
    assert (a.schedule, a.streams, a.streams_given, a.scan_lane_cus) == ("scan-lane", 5, True, 6)
