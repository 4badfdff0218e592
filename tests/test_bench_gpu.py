"""GPU test of bench.py's contract and of its RCCL path: a FRESH child process (this test process has initialised the GPU and
never execs) runs a short bench with MICLOC_FORCE_DIST=1, i.e. init_process_group("nccl") with one rank, the barriers, the
MAX all-reduce of the timing and the device all-gather of the per-rank MAE curves -- the collective code of the N > 1 launch
(paper_plots/target_snn_localization.py:447-467 shards over trials; the gather is the sweep's one exchange step)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", **env_extra)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--repeats", "2", "--no-cpu-baseline",
                        "--no-other-configs"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in p.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    assert lines, p.stdout.decode(errors="replace")[-2000:]
    return json.loads(lines[-1])


def test_bench_rccl_one_rank_group_and_contract():
    d = _run([], {"MICLOC_FORCE_DIST": "1"})
    assert d["rccl_ranks"] == 1 and d["n_gpus"] == 1
    assert d["metric"].startswith("audio samples/sec through STHT+RZCC+SNN beamform")
    assert d["steps"] == 2 and d["warmup"] == 1 and len(d["ms_per_step_repeats"]) == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64"
    frames = d["config"]["trials_per_gpu"] * d["config"]["frames_per_trial"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]  # whole-job frames over the timed region
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["kernel"] == "beamform_ws_kernel" and 0.3 < r["frac"] < 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert len(d["mae_deg_per_snr"]) == 11 and d["mae_deg_per_snr"][0] > d["mae_deg_per_snr"][-1]
    # the same run without a process group gives the same contract and rccl_ranks == 0
    d0 = _run([], {})
    assert d0["rccl_ranks"] == 0 and d0["n_gpus"] == 1
    assert d0["mae_deg_per_snr"] == d["mae_deg_per_snr"]  # same batch, same arithmetic, gathered or not
