"""Counter-based random numbers of the throughput-mode sweep: the oracle's Philox-4x32-10 against the published
known-answer vectors (Random123 kat_vectors), the uniform / normal mappings, and the stream layout."""
import numpy as np

from oracle import oracle as O


def test_philox_known_answers():
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]),
        ([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2, [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]),
        ([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0], [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]),
    ]
    for ctr, key, want in kat:
        assert O.philox4x32_10(ctr, key) == want


def test_uniform_mapping_and_layout():
    u = O.uniform(1001, seed=(5 << 32) | 9, substream=3, lo=0.0, hi=1.0)
    assert u.min() >= 0.0 and u.max() < 1.0
    # element 2i / 2i+1 come from words (0,1) / (2,3) of counter (i, epoch, 0xFFFFFFFF, substream), key (seed lo, seed hi)
    r = O.philox4x32_10([7, 0, 0xFFFFFFFF, 3], [9, 5])
    assert u[14] == ((r[1] << 32 | r[0]) >> 11) * 2.0**-53
    assert u[15] == ((r[3] << 32 | r[2]) >> 11) * 2.0**-53
    v = O.uniform(1001, seed=(5 << 32) | 9, substream=3, lo=0.0, hi=2 * np.pi)
    np.testing.assert_array_equal(v, 0.0 + (2 * np.pi - 0.0) * u)
    assert not np.array_equal(u, O.uniform(1001, seed=(5 << 32) | 9, substream=4))
    # the epoch is a counter word of its own: (substream s, epoch e) is not (s + e, 0)
    ue = O.uniform(1001, seed=(5 << 32) | 9, substream=3, epoch=1)
    assert not np.array_equal(ue, O.uniform(1001, seed=(5 << 32) | 9, substream=4))
    r = O.philox4x32_10([7, 1, 0xFFFFFFFF, 3], [9, 5])
    assert ue[14] == ((r[1] << 32 | r[0]) >> 11) * 2.0**-53


def test_generators_do_not_share_blocks():
    """Same seed, default substream / epoch: the uniforms (DoA draws) and trial 0's normals come from different Philox
    blocks (round 2 drew both from counter (i, 0, 0, 0))."""
    seed = 77
    r_uni = O.philox4x32_10([5, 0, 0xFFFFFFFF, 0], [seed, 0])
    r_nrm = O.philox4x32_10([5, 0, 0, 0], [seed, 0])
    assert r_uni != r_nrm
    u = O.uniform(12, seed)
    assert u[10] == ((r_uni[1] << 32 | r_uni[0]) >> 11) * 2.0**-53
    z = O.normals(12, seed, 0, 0)
    u1 = (((r_nrm[1] << 32 | r_nrm[0]) >> 11) + 1) * 2.0**-53
    assert np.isclose(z[10] ** 2 + z[11] ** 2, -2 * np.log(u1), rtol=1e-12)
    # epochs of the normals are disjoint from neighbouring substreams
    assert not np.array_equal(O.normals(64, seed, 0, 3, epoch=1), O.normals(64, seed, 1, 3, epoch=0))


def test_normals_statistics_and_trial_independence():
    z = O.normals(400_001, seed=1, substream=0, trial=0)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01 and abs(np.mean(z**4) - 3) < 0.1
    z1 = O.normals(1000, seed=1, substream=0, trial=1)
    assert abs(np.corrcoef(z[:1000], z1)[0, 1]) < 0.15
    # pair i -> elements 2i (cos), 2i+1 (sin) of the same radius
    r = O.philox4x32_10([3, 0, 0, 0], [1, 0])
    u1 = (((r[1] << 32 | r[0]) >> 11) + 1) * 2.0**-53
    assert np.isclose(z[6] ** 2 + z[7] ** 2, -2 * np.log(u1), rtol=1e-12)


def test_awgn_snr():
    rng = np.random.RandomState(0)
    x = rng.randn(3, 5000, 7) * np.array([1.0, 0.1, 5.0])[:, None, None]
    y, sigma = O.awgn(x, [0.0, 10.0, -10.0], seed=3, first_trial=10)
    for b, snr in enumerate([0.0, 10.0, -10.0]):
        got = 10 * np.log10(np.mean(x[b] ** 2) / np.mean((y[b] - x[b]) ** 2))
        assert abs(got - snr) < 0.2
    y2, _ = O.awgn(x[1:], [10.0, -10.0], seed=3, first_trial=11)
    np.testing.assert_array_equal(y2, y[1:])  # numbered by global trial: independent of the batch split
