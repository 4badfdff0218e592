"""The C-ABI's threading contract ("re-entrant per hipStream_t, no global state besides a per-device handle"; the
reference itself is single-threaded, ref:micloc/snn_beamformer.py:283-370 has no counterpart): two host threads, each with its own
plan, stream and workspace, drive micloc_snn_pipeline_stages_f64 at the same time on different inputs (ctypes releases the GIL for
the duration of every call, so the library IS entered concurrently); the results equal the serial run bit for bit, and
micloc_last_hip_error is per thread."""
import ctypes
import threading

import numpy as np
import pytest

from conftest import golden
from oracle import oracle as O

pytestmark = pytest.mark.gpu

NCALL = 50


@pytest.fixture(scope="module")
def torch():
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch


def _make(cfg2, which, torch):
    """Plan + NCALL different input batches of one thread (thread 1: the reference's bipolar config at G = 449; thread 2: a
    unipolar plan with another band, another robust width and a random bf_mat at G = 96 -- other kernels, other table shapes)."""
    from haghighatshoarmuir2024_amd.runtime import Plan

    z = golden("trials_cfg2.npz")
    rng = np.random.RandomState(100 + which)
    if which == 0:
        p = Plan(7, cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True)
        p.set_neuron_kernel(cfg2["nir"])
        p.set_bf_mat(cfg2["bf_mat"])
        T = 4799
    else:
        b, a = O.bandpass(48_000, [1500.0, 3000.0])
        p = Plan(7, cfg2["kernel"], b, a, 8, False)
        p.set_neuron_kernel(cfg2["nir"][:20])
        W = rng.randn(14, 96)
        p.set_bf_mat(W / np.linalg.norm(W, axis=0, keepdims=True))
        T = 3100
    xs = []
    for k in range(NCALL):
        x = z["sig_in"][[k % 3, (k + 1) % 3], :T] * (0.5 + rng.rand()) + 0.2 * rng.randn(2, T, 7)
        xs.append(p.to_device(x))
    torch.cuda.synchronize()
    return p, xs


def _run(p, xs, stream, torch, start=None):
    outs = []
    with torch.cuda.stream(stream):
        if start is not None:
            start.wait()
        for x in xs:
            outs.append(p.snn_pipeline(x, want_spikes=True, want_power=True))
    stream.synchronize()
    return [(o["spikes"].cpu().numpy(), o["power"].cpu().numpy(), o["argmax"].cpu().numpy()) for o in outs]


def test_two_host_threads_equal_the_serial_run(cfg2, torch):
    from haghighatshoarmuir2024_amd import _lib

    lib = _lib.load()
    work = [_make(cfg2, i, torch) for i in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    serial = [_run(p, xs, s, torch) for (p, xs), s in zip(work, streams)]
    # the serial results are the oracle's (first call of each thread's list)
    z = golden("trials_cfg2.npz")
    x00 = work[0][1][0].cpu().numpy()
    ref = O.snn_chain(x00[0], cfg2["kernel"], cfg2["b"], cfg2["a"], cfg2["robust_width"], True, cfg2["nir"], cfg2["bf_mat"], want=("spikes", "power"))
    np.testing.assert_array_equal(serial[0][0][0][0], ref["spikes"])
    np.testing.assert_allclose(serial[0][0][1][0], ref["power"], rtol=1e-12)

    results, errors, last_err = [None, None], [], [None, None]
    start = threading.Barrier(2)
    a_failed, b_read = threading.Event(), threading.Event()

    def worker(i):
        try:
            p, xs = work[i]
            results[i] = _run(p, xs, streams[i], torch, start)
            if i == 0:
                # provoke a HIP error in THIS thread only, harmlessly: a device ordinal that does not exist reaches hipGetDeviceProperties
                # (hipErrorInvalidDevice) and comes back as MICLOC_ERR_HIP
                h = ctypes.c_void_p()
                st = lib.micloc_stream_create_cu_range(99, 0, 4, ctypes.byref(h))
                assert not h.value
                last_err[0] = (st, lib.micloc_last_hip_error())
                a_failed.set()
                b_read.wait(60)
                # ... and a later successful call of this thread leaves the record alone (it is "the last error", not a status)
                p.snn_pipeline(xs[0], want_power=True)
                torch.cuda.synchronize()
                last_err[0] += (lib.micloc_last_hip_error(),)
            else:
                a_failed.wait(120)
                last_err[1] = lib.micloc_last_hip_error()
                b_read.set()
        except BaseException as e:  # noqa: BLE001 (reported by the main thread)
            errors.append((i, repr(e)))
            a_failed.set()
            b_read.set()

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    for i in range(2):
        assert len(results[i]) == NCALL
        for k in range(NCALL):
            for got, want in zip(results[i][k], serial[i][k]):
                np.testing.assert_array_equal(got, want, err_msg=f"thread {i} call {k}")  # bit for bit, power included
    st, code, code_later = last_err[0]
    assert st == _lib.MICLOC_ERR_HIP and code != 0 and code_later == code
    assert last_err[1] == 0, "thread 2 saw thread 1's HIP error"
