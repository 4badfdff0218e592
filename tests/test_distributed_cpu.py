"""World-size-2 gloo test of the sharded Monte-Carlo sweep (CPU).  The per-trial localizer is injected; here it
is the CPU oracle (allowed in tests), so the test checks the sharding, the reference RNG replay on every rank,
and the gather, end to end against the reference's golden sweep."""
import os
import socket

import numpy as np
import pytest

from conftest import golden, ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, num_sim, out_dir):
    import sys

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from haghighatshoarmuir2024_amd.sweep import noisy_target_sweep
    from micloc.array_geometry import CenterCircularArray
    from micloc.snn_beamformer import SNNBeamformer, neuron_impulse_response
    from oracle import oracle as O

    dist.init_process_group("gloo", rank=rank, world_size=world)
    k = np.load(os.path.join(ROOT, "tests", "golden", "kat_init.npz"))
    bfz = np.load(os.path.join(ROOT, "tests", "golden", "bf_mat_chirp449_bipolar.npz"))
    z = np.load(os.path.join(ROOT, "tests", "golden", "sweep_seed0.npz"))
    tau = 1.0 / (2 * np.pi * 2000)
    beamf = SNNBeamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=48_000)

    def oracle_localizer(sig_batch, time_vec):
        nir = neuron_impulse_response(time_vec, beamf.tau_vec)
        b, a = beamf.bandpass_filter
        pw, am = O.snn_chain_batch(sig_batch, beamf.kernel, b, a, beamf.spk_encoder.robust_width, True, nir, bfz["bf_mat"])
        return am.astype(np.int64), pw[np.arange(len(am)), am]

    res = noisy_target_sweep(beamf, bfz["bf_mat"], bfz["doa_list"], snr_db_vec=z["snr_db_vec"][:1], num_sim=num_sim, seed=int(z["seed"]),
                             mode="parity", rank=rank, world_size=world, localizer=oracle_localizer)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_sharded_sweep_world2_matches_reference(tmp_path):
    import torch.multiprocessing as mp

    from oracle import oracle as O

    O.build()
    num_sim = 24
    port = _free_port()
    mp.spawn(_worker, args=(2, port, num_sim, str(tmp_path)), nprocs=2, join=True)
    z = golden("sweep_seed0.npz")
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    for key in ("doa", "argmax", "err", "pmax", "mae_deg"):
        np.testing.assert_array_equal(r0[key], r1[key])  # every rank ends with the full result
    # the first num_sim trials of SNR 0 are the first num_sim draws of the reference's stream
    np.testing.assert_array_equal(r0["doa"][0], z["doa"][0, :num_sim])
    np.testing.assert_array_equal(r0["argmax"][0], z["argmax"][0, :num_sim])
    np.testing.assert_allclose(r0["err"][0], z["err"][0, :num_sim], rtol=0, atol=1e-12)
    np.testing.assert_allclose(r0["pmax"][0], z["pmax"][0, :num_sim], rtol=1e-10)


def test_gather_shards_uneven_world3(tmp_path):
    """gather of ragged shards (7 items over 3 ranks) with gloo."""
    import torch.multiprocessing as mp

    port = _free_port()
    mp.spawn(_gather_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    for r in range(3):
        got = np.load(tmp_path / f"g{r}.npz")
        np.testing.assert_array_equal(got["a"], np.arange(7) * 10)
        np.testing.assert_array_equal(got["b"], np.arange(7) * 0.5)
        np.testing.assert_array_equal(got["c"], np.arange(7, dtype=np.int32) - 3)
        np.testing.assert_array_equal(got["d"], (np.arange(7) % 3).astype(np.int8))
        assert got["c"].dtype == np.int32 and got["d"].dtype == np.int8


def _gather_worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from haghighatshoarmuir2024_amd.sweep import gather_shards, shard_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(7, rank, world)
    stats = {}
    full = gather_shards({"a": np.arange(lo, hi, dtype=np.int64) * 10, "b": np.arange(lo, hi, dtype=np.float64) * 0.5,
                          "c": np.arange(lo, hi, dtype=np.int32) - 3, "d": (np.arange(lo, hi) % 3).astype(np.int8)}, 7, rank, world, stats=stats)
    # ONE collective for all four arrays: a record of 3 items per array (the widest shard), each array padded to 8 bytes
    assert stats["collectives"] == 1 and stats["bytes_per_rank"] == 24 + 24 + 16 + 8 and stats["exchange_ms"] > 0
    np.savez(os.path.join(out_dir, f"g{rank}.npz"), **full)
    dist.barrier()
    dist.destroy_process_group()


def _design_worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from haghighatshoarmuir2024_amd.sweep import sharded_design

    dist.init_process_group("gloo", rank=rank, world_size=world)
    doa_list = np.linspace(-np.pi, np.pi, 11)
    calls = []

    def design_fn(doas):  # stands in for design_from_template: one unit-norm column per DoA
        calls.append(len(doas))
        cols = np.stack([np.cos(np.arange(1, 7) * d) for d in doas], axis=1)
        return cols / np.linalg.norm(cols, axis=0, keepdims=True)

    W = sharded_design(design_fn, doa_list, rank, world)
    np.savez(os.path.join(out_dir, f"d{rank}.npz"), W=W, calls=np.asarray(calls))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_design_world3(tmp_path):
    """bf_mat columns designed on 3 ranks (11 DoAs: shards of 4, 4, 3) and assembled by one all-gather."""
    import torch.multiprocessing as mp

    from haghighatshoarmuir2024_amd.sweep import sharded_design

    port = _free_port()
    mp.spawn(_design_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    doa_list = np.linspace(-np.pi, np.pi, 11)
    ref = np.stack([np.cos(np.arange(1, 7) * d) for d in doa_list], axis=1)
    ref /= np.linalg.norm(ref, axis=0, keepdims=True)
    sizes = []
    for r in range(3):
        got = np.load(tmp_path / f"d{r}.npz")
        np.testing.assert_array_equal(got["W"], ref)
        sizes.append(int(got["calls"][0]))
    assert sizes == [4, 4, 3]
    # single process: the plain call
    np.testing.assert_array_equal(sharded_design(lambda d: ref[:, : len(d)], doa_list), ref)
    with pytest.raises(ValueError):
        sharded_design(lambda d: ref[:, :2], doa_list)


def _speech_worker(rank, world, port, out_dir):
    import sys

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _speech_like_sweep(rank, world)
    np.savez(os.path.join(out_dir, f"s{world}_{rank}.npz"), **{k: v for k, v in res.items() if isinstance(v, np.ndarray)})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _speech_like_sweep(rank, world):
    """speech_target_sweep (no bandwidth correction, source resampled from 16 kHz) on a short stand-in utterance, with the
    CPU oracle as the localizer."""
    from haghighatshoarmuir2024_amd.sweep import speech_source, speech_target_sweep
    from micloc.array_geometry import CenterCircularArray
    from micloc.snn_beamformer import SNNBeamformer, neuron_impulse_response
    from oracle import oracle as O

    bfz = np.load(os.path.join(ROOT, "tests", "golden", "bf_mat_chirp449_bipolar.npz"))
    tau = 1.0 / (2 * np.pi * 2000)
    beamf = SNNBeamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=48_000)
    pcm = np.load(os.path.join(ROOT, "tests", "golden", "speech_trial.npz"))
    src = speech_source(48_000, pcm16=pcm["pcm16"][20000:20800], rate=int(pcm["rate"]))  # 50 ms of the utterance

    def oracle_localizer(sig_batch, time_vec):
        nir = neuron_impulse_response(time_vec, beamf.tau_vec)
        b, a = beamf.bandpass_filter
        pw, am = O.snn_chain_batch(sig_batch, beamf.kernel, b, a, beamf.spk_encoder.robust_width, True, nir, bfz["bf_mat"])
        return am.astype(np.int64), pw[np.arange(len(am)), am]

    return speech_target_sweep(beamf, bfz["bf_mat"], bfz["doa_list"], src, snr_db_vec=[0.0, 10.0, 20.0], num_sim=3, seed=21, mode="parity",
                               rank=rank, world_size=world, localizer=oracle_localizer, batch_trials=2)


@pytest.mark.timeout(600)
def test_speech_sweep_sharding_invariant_world2(tmp_path):
    """The speech harness (9 trials over 2 ranks, batches of 2) gives every rank the single-process result: the reference's
    RNG stream stays aligned across shards and batch flushes."""
    import torch.multiprocessing as mp

    from oracle import oracle as O

    O.build()
    one = _speech_like_sweep(0, 1)
    assert one["argmax"].shape == (3, 3) and np.all(np.diff(one["snr_db_vec"]) > 0)
    port = _free_port()
    mp.spawn(_speech_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in range(2):
        got = np.load(tmp_path / f"s2_{r}.npz")
        for key in ("doa", "argmax", "err", "pmax", "mae_deg"):
            np.testing.assert_array_equal(got[key], one[key], err_msg=f"rank {r} {key}")
