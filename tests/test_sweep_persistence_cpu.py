"""Per-shard persistence of the sweeps (`out_dir=`, sweep.ShardStore): a rank that dies after its first batch leaves that
batch on disk; the restarted job loads what exists, computes only the missing trials, and ends on the bits of an uninterrupted
run -- with the reference's MT19937 stream replayed across the skipped trials, also when the world size changes between the runs.
World-size-2 gloo on the CPU, the localizer injected (the C oracle: allowed in tests), ranks as real child processes so that one of
them can be killed.  Reference: paper_plots/target_snn_localization.py:447-467 (the loop), :525 / snn_localization_benchmark.py:588-592
(the scripts save their results)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

CHILD = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
rank, world, port, out_dir, res_dir, die_after = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
import torch.distributed as dist
from haghighatshoarmuir2024_amd.sweep import noisy_target_sweep
from micloc.array_geometry import CenterCircularArray
from micloc.snn_beamformer import SNNBeamformer, neuron_impulse_response
from oracle import oracle as O

if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
bfz = np.load(os.path.join(%(root)r, "tests", "golden", "bf_mat_chirp449_bipolar.npz"))
tau = 1.0 / (2 * np.pi * 2000)
beamf = SNNBeamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=48_000)
calls = []

def oracle_localizer(sig_batch, time_vec):
    if die_after >= 0 and len(calls) == die_after:
        os._exit(17)  # the rank dies: no clean-up, no goodbye to the process group
    calls.append(len(sig_batch))
    nir = neuron_impulse_response(time_vec, beamf.tau_vec)
    b, a = beamf.bandpass_filter
    pw, am = O.snn_chain_batch(sig_batch, beamf.kernel, b, a, beamf.spk_encoder.robust_width, True, nir, bfz["bf_mat"])
    return am.astype(np.int64), pw[np.arange(len(am)), am]

res = noisy_target_sweep(beamf, bfz["bf_mat"], bfz["doa_list"], snr_db_vec=[0.0, 10.0], num_sim=9, seed=5, mode="parity", rank=rank, world_size=world,
                         localizer=oracle_localizer, batch_trials=3, test_duration=20e-3, out_dir=out_dir if out_dir != "-" else None)
np.savez(os.path.join(res_dir, f"w{world}_r{rank}.npz"), **{k: v for k, v in res.items() if isinstance(v, np.ndarray)})
json.dump({"calls": calls, "persistence": res.get("persistence")}, open(os.path.join(res_dir, f"w{world}_r{rank}.json"), "w"))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
""" % {"root": ROOT}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _job(world, out_dir, res_dir, die=None, timeout=240):
    """Start `world` ranks; die = (rank, batches it finishes before dying).  Returns the exit codes; when a rank died the others are
    terminated like bench.py's launcher does (they would wait in the gather for ever)."""
    port = str(_free_port())
    procs = []
    for r in range(world):
        da = die[1] if die and die[0] == r else -1
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD, str(r), str(world), port, str(out_dir), str(res_dir), str(da)],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    codes = [None] * world
    if die:
        codes[die[0]] = procs[die[0]].wait(timeout=timeout)
        for r, p in enumerate(procs):
            if r != die[0]:
                try:
                    codes[r] = p.wait(timeout=20)  # (it may have finished its own shard and be blocked in the gather)
                except subprocess.TimeoutExpired:
                    p.terminate()
                    codes[r] = p.wait(timeout=20)
    else:
        for r, p in enumerate(procs):
            codes[r] = p.wait(timeout=timeout)
            assert codes[r] == 0, p.stderr.read().decode(errors="replace")[-3000:]
    return codes


@pytest.mark.timeout(900)
def test_killed_rank_resumes_to_the_uninterrupted_result(tmp_path):
    from oracle import oracle as O

    O.build()
    store, r_ref, r_a, r_b, r_c = (tmp_path / n for n in ("store", "ref", "a", "b", "c"))
    for d in (store, r_ref, r_a, r_b, r_c):
        d.mkdir()
    # the uninterrupted single-process run, no persistence: 18 trials
    assert _job(1, "-", r_ref) == [0]
    ref = np.load(r_ref / "w1_r0.npz")
    assert ref["argmax"].shape == (2, 9)
    # world 2, shards of 9 trials = 3 batches of 3; rank 1 dies at its second batch (one batch persisted)
    codes = _job(2, store, r_a, die=(1, 1))
    assert codes[1] == 17 and not (r_a / "w2_r1.npz").exists()
    (sub,) = [d for d in os.listdir(store)]
    files = sorted(f for f in os.listdir(store / sub) if f.startswith("trials_"))
    assert any(f.startswith("trials_00000009_00000012_3_") for f in files)  # rank 1's first batch: trials 9, 10, 11
    assert not any(f.startswith(".tmp") for f in os.listdir(store / sub))
    n_rank0 = sum(f < "trials_00000009" for f in files)
    assert 1 <= n_rank0 <= 3  # rank 0 finished some or all of its batches before it was stopped
    meta = json.load(open(store / sub / "meta.json"))
    assert meta["sweep"] == "noisy" and meta["total"] == 18 and meta["seed"] == 5 and len(meta["bf_mat"]["sha256"]) == 64
    # restart: both ranks load what exists and compute the rest; the result is the uninterrupted run's, bit for bit
    assert _job(2, store, r_b) == [0, 0]
    for r in range(2):
        got = np.load(r_b / f"w2_r{r}.npz")
        for key in ("doa", "argmax", "pmax", "err", "mae_deg"):
            np.testing.assert_array_equal(got[key], ref[key], err_msg=f"rank {r} {key}")
    info = [json.load(open(r_b / f"w2_r{r}.json")) for r in range(2)]
    assert info[1]["calls"] == [3, 3] and info[1]["persistence"]["files_written"] == 2  # rank 1 recomputed two batches, not three
    assert info[0]["calls"] == [3] * (3 - n_rank0) and info[1]["persistence"]["trials_loaded"] == 3 * (n_rank0 + 1)
    # a third run finds everything (no localizer call at all) -- with ANOTHER world size: coverage is per trial, not per shard
    assert _job(1, store, r_c) == [0]
    got = np.load(r_c / "w1_r0.npz")
    for key in ("doa", "argmax", "pmax", "err", "mae_deg"):
        np.testing.assert_array_equal(got[key], ref[key])
    info = json.load(open(r_c / "w1_r0.json"))
    assert info["calls"] == [] and info["persistence"]["trials_loaded"] == 18 and info["persistence"]["files_written"] == 0


def test_store_key_follows_the_arguments(tmp_path):
    """Another seed, another bf_mat or another SNR vector is another directory; a damaged or foreign file is ignored."""
    from haghighatshoarmuir2024_amd.sweep import ShardStore

    W = np.arange(12.0).reshape(3, 4)
    a = ShardStore(tmp_path, "noisy", 10, seed=1, mode="parity", bf_mat=W, snr=np.zeros(10))
    b = ShardStore(tmp_path, "noisy", 10, seed=2, mode="parity", bf_mat=W, snr=np.zeros(10))
    c = ShardStore(tmp_path, "noisy", 10, seed=1, mode="parity", bf_mat=W + 1e-16 * 0 + np.eye(3, 4) * 1e-12, snr=np.zeros(10))
    assert len({a.dir, b.dir, c.dir}) == 3
    a.put([4, 5, 7], [0.1, 0.2, 0.3], [3, 2, 1], [1.0, 2.0, 3.0])
    open(os.path.join(a.dir, "trials_garbage.npy"), "wb").write(b"not an array")
    np.save(os.path.join(a.dir, "trials_00000000_00000002_2.npy"), np.zeros(2))  # wrong dtype: foreign
    a2 = ShardStore(tmp_path, "noisy", 10, seed=1, mode="parity", bf_mat=W, snr=np.zeros(10))
    assert a2.dir == a.dir and a2.files_loaded == 1 and a2.trials_loaded == 3
    assert list(np.flatnonzero(a2.have)) == [4, 5, 7] and a2.covered(4, 6) and not a2.covered(4, 8)
    assert list(a2.rec["index"][[4, 5, 7]]) == [3, 2, 1] and list(a2.rec["pmax"][[4, 5, 7]]) == [1.0, 2.0, 3.0]
