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
                        "--no-other-configs", "--sustained-seconds", "0.4"] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
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
    # SURVEY 8d's clock beside the driver's, inside `config`; the sustained region beside the K-step regions
    assert d["config"]["value_e2e"] == d["value_e2e"] and d["config"]["e2e_ms_per_step"] > d["ms_per_step"] * 0.9
    sus = d["sustained"]
    assert sus["seconds"] >= 0.4 and sus["steps"] >= 4 * sus["segment_steps"] and len(sus["ms_per_step_segments"]) >= 4
    assert abs(sus["ms_per_step"] - sus["seconds"] / sus["steps"] * 1e3) < 1e-9 and "value_source" in d
    assert sus["telemetry_source"].startswith("amdsmi"), sus["telemetry_source"]  # the GPU box lets an ordinary user read the clocks
    tw = sus["telemetry"]["whole"]
    assert tw and 500 <= tw["sclk_mhz_mean"] <= 2600 and 50 < tw["socket_w_mean"] < 1500
    # the exchange step: ONE all_gather_into_tensor of {doa, p_max, argmax} per trial, timed; the MAE recomputed from the gathered
    # trials on the host is the device's (micloc_doa_error_f64) -- same arithmetic, another summation order
    ex = d["exchange"]
    assert ex["collectives"] == 1 and ex["bytes_per_rank"] == 3 * 8 * d["config"]["trials_per_gpu"] and d["exchange_ms"] == ex["exchange_ms"] > 0
    assert max(abs(a - b) for a, b in zip(ex["mae_deg_per_snr_from_gathered_trials"], ex["mae_deg_per_snr_device_same_batch"])) < 1e-9
    assert len(d["config"]["device_uuid_per_rank"]) == 1 and len(d["ms_per_step_per_rank"]) == 1
    # the same run without a process group gives the same contract and rccl_ranks == 0
    d0 = _run([], {})
    assert d0["rccl_ranks"] == 0 and d0["n_gpus"] == 1
    assert d0["mae_deg_per_snr"] == d["mae_deg_per_snr"]  # same batch, same arithmetic, gathered or not


def test_bench_stress_share_of_baseline_total_under_rccl():
    """BASELINE config 5 the way the 8-GPU run starts it (--config stress --baseline-total; here 64 trials on one rank): the general
    beamforming kernel, the strong-scaling bookkeeping and the exchange step under a one-rank RCCL group."""
    d = _run(["--config", "stress", "--baseline-total", "--trials", "64", "--sustained-seconds", "0"], {"MICLOC_FORCE_DIST": "1"})
    assert d["rccl_ranks"] == 1 and d["n_gpus"] == 1 and d["scaling"] == "strong"
    c = d["config"]
    assert c["trials_per_gpu"] == 64 and c["num_mic"] == 64 and c["num_doa"] == 1440 and c["frames_per_trial"] == 9599
    assert d["roofline"]["kernel"] == "beamform_gen_kernel" and 0.3 < d["roofline"]["frac"] < 1.0
    assert d["exchange"]["collectives"] == 1 and d["exchange_ms"] > 0 and len(d["ms_per_step_per_rank"]) == 1
    assert len(c["device_uuid_per_rank"]) == 1 and "sustained" not in d


def test_bench_two_ranks_rehearsed_on_one_device():
    """The N > 1 launch rehearsed on the one GPU of this box: `bench.py --gpus 2 --share-device` goes through the
    GPU-free rank launcher (`launch_ranks`), both ranks use device 0, the process group is gloo with every collective staged through
    host memory (RCCL refuses two ranks on one device) -- and everything else is the real world > 1 code: rank r builds ITS batches
    (its DoAs, trials numbered from r B), per-rank clocks are gathered, the timing is the MAX over ranks, the exchange step gathers
    {doa, p_max, argmax} per trial from both ranks and rank 0 recomputes the MAE per SNR from the gathered trials.  Checked against
    two single-process runs that build rank 0's and rank 1's workload (--as-rank): the 2-rank MAE curves are their means."""
    d2 = _run(["--gpus", "2", "--share-device"], {})
    assert d2["n_gpus"] == 2 and d2["rccl_ranks"] == 0 and d2["backend"].startswith("gloo") and d2["config"]["shared_device"] is True
    assert len(d2["ms_per_step_per_rank"]) == 2 and all(t > 0 for t in d2["ms_per_step_per_rank"])
    assert d2["ms_per_step"] >= max(d2["ms_per_step_per_rank"]) * 0.999  # the job's time is the slowest rank's (per region: median of maxima)
    uu = d2["config"]["device_uuid_per_rank"]
    assert len(uu) == 2 and uu[0] == uu[1]  # the same physical device: a rehearsal, not a scaling measurement
    frames = d2["config"]["trials_per_gpu"] * d2["config"]["frames_per_trial"]
    assert abs(d2["value"] - 2 * frames / (d2["ms_per_step"] * 1e-3)) <= 1e-6 * d2["value"]  # whole-job frames: both ranks' batches
    ex = d2["exchange"]
    assert ex["collectives"] == 1 and ex["transport"].startswith("gloo") and len(ex["mae_deg_per_snr_from_gathered_trials"]) == 11
    assert d2["sustained"]["seconds"] >= d2["sustained"]["seconds_this_rank"] * 0.999
    singles = [_run(["--as-rank", str(r)], {"MICLOC_FORCE_DIST": "1"}) for r in (0, 1)]
    assert singles[0]["mae_deg_per_snr"] != singles[1]["mae_deg_per_snr"]  # the ranks really hold different trials
    for key, single_key in (("mae_deg_per_snr", "mae_deg_per_snr"),):
        want = [(a + b) / 2 for a, b in zip(singles[0][single_key], singles[1][single_key])]
        assert max(abs(a - b) for a, b in zip(d2[key], want)) < 1e-9, (d2[key], want)
    want = [(a + b) / 2 for a, b in zip(singles[0]["exchange"]["mae_deg_per_snr_device_same_batch"], singles[1]["exchange"]["mae_deg_per_snr_device_same_batch"])]
    assert max(abs(a - b) for a, b in zip(ex["mae_deg_per_snr_from_gathered_trials"], want)) < 1e-9
    assert ex["mae_deg_per_snr_device_same_batch"] == singles[0]["exchange"]["mae_deg_per_snr_device_same_batch"]  # rank 0's own batch


def test_other_workloads_report_a_sustained_region():
    """the speech workload (scan-lane schedule: eager launches, the serial checkpoint scans on a stream of their own) and
    the Xylo workload carry a `sustained` block too -- one uninterrupted region beside the K-step regions, with the chip's telemetry."""
    d = _run(["--config", "speech", "--trials", "6", "--sustained-seconds", "0.6"], {})
    assert d["config"]["schedule"].startswith("scan-lane") and d["config"]["hip_graphs"] is False
    sus = d["sustained"]
    assert sus["seconds"] >= 0.6 and sus["steps"] >= 4 and sus["telemetry_source"].startswith("amdsmi")
    assert 0.3 < sus["ratio_to_timed_regions"] < 1.5 and d["config"]["sustained_ms_per_step"] == sus["ms_per_step"]
    assert d["config"]["memory_gb"]["peak_reserved"] > 0.1
    x = _run(["--config", "xylo", "--trials", "110", "--sustained-seconds", "0.6"], {})
    assert x["sustained"]["seconds"] >= 0.6 and x["config"]["sustained_ms_per_step"] == x["sustained"]["ms_per_step"]
    assert x["parity"].startswith("UNPINNED")
