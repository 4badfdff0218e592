"""Streaming with exact state hand-off (streaming.StreamingLocalizer, micloc_stream_encode_tile_f64 / micloc_stream_localize_tile_f64): a recording pushed tile by
tile gives the spikes, power and arg-max of the one-shot call bit for bit, for any tiling -- including the reference's own
golden trials and the speech-length trial."""
import hashlib

import numpy as np
import pytest

from conftest import campaign_seeds, golden
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _beamformer(bipolar=True):
    from micloc.array_geometry import CenterCircularArray
    from micloc.snn_beamformer import SNNBeamformer

    tau = 1 / (2 * np.pi * 2000)
    return SNNBeamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=bipolar, fs=48_000)


def _stream(bf, W, x, tiles, wrap=True, **kw):
    from haghighatshoarmuir2024_amd.streaming import StreamingLocalizer

    B, T, M = x.shape
    L2 = len(bf.kernel) // 2
    s = StreamingLocalizer(bf, W, B, T, wrap_tail=x[:, T - L2 :, :] if wrap else None, keep_raster=True, **kw)
    t = 0
    for n in tiles:
        s.push(x[:, t : t + n, :])
        t += n
    assert t == T
    out = s.finish(want_spikes=True)
    out["status"] = s.status()
    return out


@pytest.mark.parametrize("bipolar", [True, False])
def test_stream_equals_one_shot_for_any_tiling(cfg2, bipolar):
    z = golden("trials_cfg2.npz")
    rng = np.random.RandomState(8)
    x = np.concatenate([z["sig_in"], rng.randn(5, 4799, 7)])
    bf = _beamformer(bipolar)
    W = cfg2["bf_mat"]
    one = bf.localize_batch(W, x, return_spikes=True)
    ref_spikes = one["spikes"].cpu().numpy()
    if bipolar:
        np.testing.assert_array_equal(ref_spikes[:3], z["spikes"])  # == the reference's own spikes
    T = x.shape[1]
    tilings = [[T], [2400, T - 2400], [16] * 20 + [T - 320], [1600, 1600, T - 3200], [4784, 15], [256] * 18 + [T - 256 * 18]]
    rt = np.random.RandomState(1)
    cuts = np.sort(rt.choice(np.arange(1, T // 16), size=12, replace=False)) * 16
    tilings.append(list(np.diff(np.concatenate([[0], cuts, [T]]))))
    for tiles in tilings:
        out = _stream(bf, W, x, tiles)
        np.testing.assert_array_equal(out["spikes"].cpu().numpy(), ref_spikes, err_msg=f"tiles={tiles[:4]}...")
        np.testing.assert_array_equal(out["power"].cpu().numpy(), one["power"].cpu().numpy())
        np.testing.assert_array_equal(out["argmax"].cpu().numpy(), one["argmax"].cpu().numpy())
    # without the wrap-around rows only the encoder's transient differs: a causal stream cannot know the end of the recording
    out = _stream(bf, W, x, [1600, 1600, T - 3200], wrap=False)
    assert not np.array_equal(out["spikes"].cpu().numpy(), ref_spikes)


def test_stream_plateaus_and_errors(cfg2):
    """Digital silence across tile boundaries (plateaus of the running sum, clusters open at a boundary), API errors."""
    from haghighatshoarmuir2024_amd import _lib
    from haghighatshoarmuir2024_amd.streaming import StreamingLocalizer

    bf = _beamformer()
    rng = np.random.RandomState(3)
    T = 3200
    x = rng.randn(2, T, 7)
    x[0, 700:2300, :] = 0.0
    x[1, :1000, 2] = 0.0
    one = bf.localize_batch(cfg2["bf_mat"], x, return_spikes=True)
    for tiles in ([800] * 4, [1600, 1600], [704, 16, 16, 2464]):
        out = _stream(bf, cfg2["bf_mat"], x, tiles)
        np.testing.assert_array_equal(out["spikes"].cpu().numpy(), one["spikes"].cpu().numpy())
        np.testing.assert_array_equal(out["power"].cpu().numpy(), one["power"].cpu().numpy())
    s = StreamingLocalizer(bf, cfg2["bf_mat"], 2, T)
    with pytest.raises(ValueError):
        s.push(x[:, :100, :])  # not a multiple of 16 and not the last tile
    s.push(x[:, :1600, :])
    with pytest.raises(_lib.MiclocError):
        s.finish()
    with pytest.raises(ValueError):
        s.push(x[:, :, :3])
    # out-of-band input (alternating samples: one endless cluster) overflows the candidate ring: reported, not silently wrong
    bad = np.tile(np.where(np.arange(T) % 2 == 0, 1.0, -1.0)[None, :, None], (1, 1, 7)) * (1 + 0.01 * rng.rand(1, T, 7))
    from haghighatshoarmuir2024_amd import runtime

    spk1 = runtime.rzcc_encode(bad[0], 12, True)  # the one-shot operator handles it (fallback kernel)
    assert int((spk1 != 0).sum()) > 0
    s2 = StreamingLocalizer(_beamformer_identity(), np.eye(14)[:, :3], 1, T)
    s2.push(bad)
    try:
        s2.finish()
    except _lib.MiclocError as e:
        assert "overflow" in str(e)


def _beamformer_identity():
    """A beamformer whose band contains the alternating test input (so that the band-pass does not remove it)."""
    from micloc.array_geometry import CenterCircularArray
    from micloc.snn_beamformer import SNNBeamformer

    tau = 1 / (2 * np.pi * 20000)
    return SNNBeamformer(CenterCircularArray(4.5e-2, 7), 10e-3, [12000.0, 23000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=48_000)


def test_stream_speech_length(cfg2):
    """T = 332 157 in 0.25 s tiles (the live demo's frame length): the SHA-256 of the reference's spike raster."""
    from test_speech_config import speech_trial_input

    z, t, sig = speech_trial_input(cfg2)
    bf = _beamformer()
    T = sig.shape[0]
    tiles = [12000] * (T // 12000) + ([T % 12000] if T % 12000 else [])
    out = _stream(bf, cfg2["bf_mat"], sig[None], tiles)
    spikes = out["spikes"][0].cpu().numpy()
    assert hashlib.sha256(np.ascontiguousarray(spikes).tobytes()).digest() == z["spikes_sha256"].tobytes()
    np.testing.assert_allclose(out["power"][0].cpu().numpy(), z["power"], rtol=1e-10)
    assert int(out["argmax"][0]) == int(z["argmax"])
    assert out["status"]["frames"] == T and out["status"]["lag_failures"] == 0
    # ... and bit for bit the one-shot call (its time reduction has the same, streamable order for any length)
    one = bf.localize_batch(cfg2["bf_mat"], sig[None])
    np.testing.assert_array_equal(out["power"].cpu().numpy(), one["power"].cpu().numpy())


def test_stream_incremental_live_source(cfg2):
    """The live form: the length is not known in advance (final=True comes with the last tile), no raster is kept -- memory is
    O(tile) -- and every push returns the running power / arg-max over the frames whose spikes are final.  After the last tile
    they equal the one-shot call bit for bit (the reference's live loop, localization_demo_snn.py:125-193, restarts the chain
    every 0.25 s instead and never sees more than one frame)."""
    import torch

    from haghighatshoarmuir2024_amd.streaming import StreamingLocalizer

    z = golden("trials_cfg2.npz")
    rng = np.random.RandomState(4)
    T = 48_000 + 4799  # a recording longer than the 1 s the neuron kernel is normalised over
    x = rng.randn(3, T, 7)
    x[:, :4799, :] += z["sig_in"]
    bf = _beamformer()
    W = cfg2["bf_mat"]
    one = bf.localize_batch(W, x)
    L2 = len(bf.kernel) // 2
    torch.cuda.synchronize()
    mem0 = torch.cuda.memory_allocated()
    s = StreamingLocalizer(bf, W, 3, wrap_tail=x[:, T - L2 :, :], max_tile=2400, lag_frames=1024)
    state_bytes = torch.cuda.memory_allocated() - mem0
    assert state_bytes < 3 * (2400 + 480) * 7 * 8 * 8  # a few tiles' worth, nothing that grows with the recording
    frames, t = [], 0
    while t < T:
        n = min(2400, T - t)
        alloc0 = torch.cuda.memory_allocated()
        power, argmax = s.push(torch.from_numpy(x[:, t : t + n, :]).cuda(), final=t + n == T)
        assert torch.cuda.memory_allocated() - alloc0 <= 3 * n * 7 * 8 + 4096  # (only the tile handed in by this test)
        t += n
        frames.append(s.status()["frames"])
        if frames[-1] > 0:
            assert power.shape == (3, W.shape[1]) and float(power.min()) >= 0.0 and int(argmax.max()) < W.shape[1]
    assert frames == sorted(frames) and frames[-1] == T and all(f % 256 == 0 for f in frames[:-1])
    assert all(t_ - f <= 1024 + 256 for t_, f in zip(range(2400, T, 2400), frames))  # the horizon trails the input by a few clusters
    out = s.finish()
    np.testing.assert_array_equal(out["power"].cpu().numpy(), one["power"].cpu().numpy())
    np.testing.assert_array_equal(out["argmax"].cpu().numpy(), one["argmax"].cpu().numpy())
    with pytest.raises(ValueError):
        s.finish(want_spikes=True)  # no raster was kept


def test_stream_window_lag_is_reported(cfg2):
    """A window too small for the input's longest silence (a plateau holds its cluster open) is an error, never a wrong result."""
    from haghighatshoarmuir2024_amd import _lib
    from haghighatshoarmuir2024_amd.streaming import StreamingLocalizer

    bf = _beamformer()
    rng = np.random.RandomState(5)
    T = 9600
    x = rng.randn(1, T, 7)
    x[0, 1000:7000, :] = 0.0  # digital silence: the running sum stands still, the next candidate's plateau started long ago
    one = bf.localize_batch(cfg2["bf_mat"], x)
    ok = _stream(bf, cfg2["bf_mat"], x, [800] * 12, max_tile=800, lag_frames=8192)
    np.testing.assert_array_equal(ok["power"].cpu().numpy(), one["power"].cpu().numpy())
    s = StreamingLocalizer(bf, cfg2["bf_mat"], 1, T, max_tile=800, lag_frames=256)
    for t in range(0, T, 800):
        s.push(x[:, t : t + 800, :])
    with pytest.raises(_lib.MiclocError, match="window"):
        s.finish()


@pytest.mark.parametrize("num_mic,G", [(16, 75), (40, 130)])
def test_stream_other_kernel_families(num_mic, G):
    """More than 8 microphones take the other LIF + beamforming kernels (time-stationary: 512-frame chunks; slab: more than 64
    channels): the device-side chunk range and the window's trial stride reach them too -- streamed == one-shot, bit for bit."""
    from micloc.array_geometry import CircularArray
    from micloc.snn_beamformer import SNNBeamformer
    from haghighatshoarmuir2024_amd.streaming import StreamingLocalizer

    tau = 1 / (2 * np.pi * 2000)
    bf = SNNBeamformer(CircularArray(0.1, num_mic), 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=48_000)
    rng = np.random.RandomState(num_mic)
    W = rng.randn(2 * num_mic, G)
    W /= np.linalg.norm(W, axis=0, keepdims=True)
    T = 6000
    x = rng.randn(3, T, num_mic)
    one = bf.localize_batch(W, x)
    L2 = len(bf.kernel) // 2
    for tiles in ([T], [1600, 1600, 1600, 1200], [2048, 512, 3440]):
        s = StreamingLocalizer(bf, W, 3, T, wrap_tail=x[:, T - L2 :, :], max_tile=max(tiles), lag_frames=2048)
        t = 0
        for n in tiles:
            s.push(x[:, t : t + n, :])
            t += n
        out = s.finish()
        np.testing.assert_array_equal(out["power"].cpu().numpy(), one["power"].cpu().numpy(), err_msg=f"tiles={tiles}")
        np.testing.assert_array_equal(out["argmax"].cpu().numpy(), one["argmax"].cpu().numpy())


def test_stream_push_replay_is_one_graph_per_tile(cfg2):
    """The live loop (micloc/localization_demo_snn.py:125-193: one 0.25 s frame after the other) as ONE hipGraph launch per tile: the
    stream's clock lives on the device, so every launch of a tile of n frames has the same arguments.  push_replay captures the tile
    on its second occurrence and replays it from then on -- running estimates equal to push()'s after every tile, the final result
    equal to the one-shot call bit for bit, with the window sliding and np.roll's wrap-around rows inside the graph."""
    import torch

    from haghighatshoarmuir2024_amd.streaming import StreamingLocalizer

    z = golden("trials_cfg2.npz")
    rng = np.random.RandomState(9)
    n, tiles = 2400, 11
    T = tiles * n + 799
    x = rng.randn(2, T, 7)
    x[:, :4799, :] += z["sig_in"][:2]
    bf = _beamformer()
    W = cfg2["bf_mat"]
    one = bf.localize_batch(W, x)
    L2 = len(bf.kernel) // 2
    kw = dict(wrap_tail=x[:, T - L2 :, :], max_tile=n, lag_frames=1024)
    a = StreamingLocalizer(bf, W, 2, **kw)  # eager pushes
    g = StreamingLocalizer(bf, W, 2, **kw)  # graph replays
    xd = torch.from_numpy(x).cuda()
    for k in range(tiles):
        pa, aa = a.push(xd[:, k * n : (k + 1) * n, :])
        pg, ag = g.push_replay(xd[:, k * n : (k + 1) * n, :])
        assert torch.equal(pa, pg) and torch.equal(aa, ag), k
        assert a.status() == g.status(), k
    assert list(g._graphs) == [n] and g.base > 0  # one captured graph, and the window did slide inside it
    a.push(xd[:, tiles * n :, :], final=True)
    g.push(xd[:, tiles * n :, :], final=True)  # (the ragged last tile is an ordinary push)
    oa, og = a.finish(), g.finish()
    np.testing.assert_array_equal(og["power"].cpu().numpy(), one["power"].cpu().numpy())
    np.testing.assert_array_equal(og["argmax"].cpu().numpy(), one["argmax"].cpu().numpy())
    np.testing.assert_array_equal(oa["power"].cpu().numpy(), og["power"].cpu().numpy())
    # known length: push_replay recognises the last tile by itself and issues it as a final push
    h = StreamingLocalizer(bf, W, 2, total_frames=tiles * n, wrap_tail=x[:, tiles * n - L2 : tiles * n, :], max_tile=n, lag_frames=1024)
    for k in range(tiles):
        h.push_replay(xd[:, k * n : (k + 1) * n, :])
    assert h.done and len(h._graphs) == 1
    ref = bf.localize_batch(W, x[:, : tiles * n, :])
    np.testing.assert_array_equal(h.finish()["power"].cpu().numpy(), ref["power"].cpu().numpy())


@pytest.mark.parametrize("seed", campaign_seeds("streaming", 200))
def test_stream_random_tilings_eager_and_replayed(cfg2, seed):
    """Random tile sequences (lengths from a small set, so that several lengths get their own captured graph), every tile randomly
    pushed eagerly or replayed, a window small enough to slide many times: the device clock, the in-graph slide and the wrap-around
    rows must give the one-shot result bit for bit whatever the mix.  200 seeds in the driver's run (the id says so); seeds >= 4 draw
    shorter recordings (3 000 ... 9 000 frames) so that the campaign fits the run."""
    import torch

    from haghighatshoarmuir2024_amd.streaming import StreamingLocalizer

    rng = np.random.RandomState(100 + seed)
    sizes = [int(v) for v in rng.choice([64, 160, 256, 400, 1024, 1600], size=3, replace=False)]
    T = int(rng.randint(9000, 16000)) if seed < 4 else int(rng.randint(3000, 9000))
    x = rng.randn(2, T, 7)
    x[:, : min(T, 4799), :] += golden("trials_cfg2.npz")["sig_in"][:2][:, : min(T, 4799), :]
    bf = _beamformer()
    W = cfg2["bf_mat"]
    one = bf.localize_batch(W, x)
    L2 = len(bf.kernel) // 2
    s = StreamingLocalizer(bf, W, 2, wrap_tail=x[:, T - L2 :, :], max_tile=max(sizes), lag_frames=1536, keep_raster=True, total_frames=T)
    xd = torch.from_numpy(x).cuda()
    t = 0
    while t < T:
        n = int(rng.choice(sizes))
        if t + n >= T:
            s.push(xd[:, t:, :], final=True)  # the ragged rest
            break
        (s.push_replay if rng.rand() < 0.6 else s.push)(xd[:, t : t + n, :])
        t += n
    out = s.finish(want_spikes=True)
    if seed < 4:  # (the long recordings: the window must have slid and some tile must have been replayed from a captured graph)
        assert s.base > 0 and len(s._graphs) >= 1
    np.testing.assert_array_equal(out["power"].cpu().numpy(), one["power"].cpu().numpy())
    np.testing.assert_array_equal(out["argmax"].cpu().numpy(), one["argmax"].cpu().numpy())
    ref = bf.localize_batch(W, x, return_spikes=True)
    assert torch.equal(out["spikes"], ref["spikes"])
