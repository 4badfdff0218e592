"""Monte-Carlo DoA sweep (restatement of paper_plots/target_snn_localization.py:435-467 as a batched,
shardable harness).  The script's loop carries no state between trials except the RNG stream, so trials
shard contiguously across ranks (one process per GPU); the only exchange is one small all-gather of
per-trial results at the end (RCCL over xGMI when the process group is "nccl").

parity mode      every rank replays the reference's global NumPy stream (rand(1) then randn(T, M) per trial,
                 legacy MT19937) and keeps its own shard, so results are identical to the single-process
                 reference for any world size.
throughput mode  DoAs from a seeded host generator, clean array signals synthesised on the device
                 (csrc/synth.hip, bit-exact with np.interp), noise from the Philox-4x32-10 + Box-Muller kernel
                 (csrc/rng.hip), numbered by global trial so that the draw does not depend on the sharding.

Sweeps: noisy_target_sweep (target_snn_localization.py:435-467), speech_target_sweep (:213-245), xylo_target_sweep
(target_xylo_localization.py:540-608; integer-LIF stage parity-unpinned).
"""
import numpy as np

from .snn_beamformer import synthesize_array_signal


def shard_range(total, rank, world_size):
    """Contiguous shard [lo, hi) of `total` trials for `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def doa_error(doa_est, doa_true):
    """arcsin|sin(est - true)| (target_snn_localization.py:466; pi-periodic by construction)."""
    return np.arcsin(np.abs(np.sin(doa_est - doa_true)))


class ShardStore:
    """Per-shard result persistence of a sweep (`out_dir=`): every finished device batch of a rank is written as ONE small `.npy`
    (a structured array {trial i8, doa f8, index i8, pmax f8}, 32 bytes per trial, written to a temporary name and renamed: a file
    either exists completely or not at all), under a directory keyed by everything the results depend on -- sweep, seed, mode, the
    SNR of every trial, the DoA grid, the hash of bf_mat and of the test signal -- so a rerun with the same arguments finds its own
    results and any change of them starts a fresh directory.  A rerun loads what exists and computes only the trials that are
    missing; in parity mode the reference's MT19937 stream is still replayed for the skipped trials, so the results are identical to an
    uninterrupted run -- also when the world size changed in between (coverage is per trial, not per shard).
    The reference keeps its sweeps' results the same way, at the end of the script (ref:paper_plots/snn_localization_benchmark.py:588-592,
    ref:paper_plots/target_snn_localization.py:525); a 16 384-trial sweep over 8 ranks should not restart from zero."""

    REC = np.dtype([("trial", "<i8"), ("doa", "<f8"), ("index", "<i8"), ("pmax", "<f8")])

    def __init__(self, out_dir, sweep, total, **key):
        import hashlib
        import json
        import os

        def h(v):
            if isinstance(v, np.ndarray):
                a = np.ascontiguousarray(v)
                return {"sha256": hashlib.sha256(a.tobytes()).hexdigest(), "shape": list(a.shape), "dtype": str(a.dtype)}
            if isinstance(v, (np.integer, np.floating)):
                return v.item()
            return v

        self.meta = {"sweep": sweep, "total": int(total), "format": 1, **{k: h(v) for k, v in sorted(key.items())}}
        blob = json.dumps(self.meta, sort_keys=True).encode()
        self.key = hashlib.sha256(blob).hexdigest()[:16]
        self.dir = os.path.join(str(out_dir), f"{sweep}-{self.key}")
        os.makedirs(self.dir, exist_ok=True)
        meta_path = os.path.join(self.dir, "meta.json")
        if not os.path.exists(meta_path):
            tmp = f"{meta_path}.tmp-{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(blob)
            os.replace(tmp, meta_path)  # (several ranks may race: they all write the same bytes)
        self.total = int(total)
        self.have = np.zeros(self.total, dtype=bool)
        self.rec = np.zeros(self.total, dtype=self.REC)
        self.files_loaded = 0
        self.trials_loaded = 0
        self.files_written = 0
        for name in sorted(os.listdir(self.dir)):
            if not (name.startswith("trials_") and name.endswith(".npy")):
                continue
            try:
                a = np.load(os.path.join(self.dir, name))
            except (OSError, ValueError):
                continue  # (cannot happen with the rename protocol; a damaged file is simply recomputed)
            if a.dtype != self.REC or a.ndim != 1 or len(a) == 0 or a["trial"].min() < 0 or a["trial"].max() >= self.total:
                continue
            self.rec[a["trial"]] = a
            self.have[a["trial"]] = True
            self.files_loaded += 1
        self.trials_loaded = int(self.have.sum())

    def covered(self, lo, hi):
        return bool(self.have[lo:hi].all())

    def put(self, trials, doa, index, pmax):
        """Persist one finished batch (any set of trial numbers) and mark it done."""
        import os

        trials = np.asarray(trials, dtype=np.int64)
        if len(trials) == 0:
            return
        a = np.zeros(len(trials), dtype=self.REC)
        a["trial"], a["doa"], a["index"], a["pmax"] = trials, doa, index, pmax
        # (the name carries a checksum of the trial numbers: two different sets with the same bounds and size -- resumed runs under
        #  different world sizes -- never overwrite each other's records)
        import zlib

        name = f"trials_{int(trials.min()):08d}_{int(trials.max()) + 1:08d}_{len(trials)}_{zlib.crc32(np.ascontiguousarray(trials).tobytes()):08x}.npy"
        tmp = os.path.join(self.dir, f".tmp-{os.getpid()}-{name}")
        with open(tmp, "wb") as f:
            np.save(f, a)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, os.path.join(self.dir, name))
        self.rec[trials] = a
        self.have[trials] = True
        self.files_written += 1

    def stats(self):
        return {"dir": self.dir, "files_loaded": self.files_loaded, "trials_loaded": self.trials_loaded, "files_written": self.files_written}


def gather_shards(local, total, rank, world_size, group=None, bounds=None, stats=None):
    """The sweep's one exchange step: all-gather the per-trial result arrays of contiguous shards; returns the
    full-length arrays on every rank.  `local` is a dict name -> 1-d numpy array (this rank's shard; any mix of dtypes).
    ONE collective whatever the number of arrays: every rank packs its arrays into a struct-of-arrays byte record (each array
    padded to the widest shard and to 8 bytes), one `all_gather_into_tensor` moves the records (RCCL over xGMI when the group is
    "nccl": one host -> device copy, one collective, one device -> host copy), every rank unpacks.  `bounds(r) -> (lo, hi)`
    overrides the default `shard_range(total, r, world_size)` partition.  `stats` (a dict, optional) receives `exchange_ms` (wall
    time of pack + collective + unpack), `bytes_per_rank` and `collectives`."""
    import time

    if bounds is None:
        bounds = lambda r: shard_range(total, r, world_size)  # noqa: E731
    if world_size == 1:
        if stats is not None:
            stats.update(exchange_ms=0.0, bytes_per_rank=0, collectives=0)
        return {k: np.asarray(v) for k, v in local.items()}
    import torch
    import torch.distributed as dist

    t0 = time.perf_counter()
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    spans = [bounds(r) for r in range(world_size)]
    width = max(hi - lo for lo, hi in spans)  # padded shard length (items)
    keys = list(local)
    arrs = {k: np.ascontiguousarray(local[k]) for k in keys}
    n_local = spans[rank][1] - spans[rank][0]
    for k in keys:
        if arrs[k].ndim != 1 or len(arrs[k]) != n_local:
            raise ValueError(f"gather_shards: '{k}' has shape {arrs[k].shape}, this rank's shard holds {n_local} items")
    # record layout: the arrays one behind the other, each `width` items of its dtype, 8-byte aligned
    offs, off = {}, 0
    for k in keys:
        offs[k] = off
        off += (width * arrs[k].dtype.itemsize + 7) & ~7
    rec = np.zeros(max(off, 8), dtype=np.uint8)
    for k in keys:
        v = arrs[k]
        rec[offs[k] : offs[k] + v.nbytes] = v.view(np.uint8)
    buf = torch.from_numpy(rec).to(dev)
    full = torch.empty(len(rec) * world_size, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(full, buf, group=group)
    full = full.cpu().numpy().reshape(world_size, len(rec))
    out = {}
    for k in keys:
        dt = arrs[k].dtype
        parts = [full[r, offs[k] : offs[k] + (hi - lo) * dt.itemsize].view(dt) for r, (lo, hi) in enumerate(spans)]
        out[k] = np.concatenate(parts)
    if stats is not None:
        stats.update(exchange_ms=(time.perf_counter() - t0) * 1e3, bytes_per_rank=int(len(rec)), collectives=1)
    return out


def sharded_design(design_fn, doa_list, rank=0, world_size=1, group=None):
    """`design_from_template` sharded over the DoA grid (the G columns of bf_mat are independent units):
    every rank designs the columns of its contiguous DoA shard with `design_fn(doa_sublist) -> [2M, n_local]` and one
    all-gather assembles the full [2M, G] matrix on every rank (RCCL when the group is "nccl")."""
    doa_list = np.asarray(doa_list, dtype=np.float64)
    G = len(doa_list)
    lo, hi = shard_range(G, rank, world_size)
    local = np.asarray(design_fn(doa_list[lo:hi]), dtype=np.float64)
    if local.ndim != 2 or local.shape[1] != hi - lo:
        raise ValueError(f"design_fn returned shape {local.shape} for {hi - lo} DoAs")
    rows = local.shape[0]
    if world_size == 1:
        return local
    # one gather of `rows` doubles per DoA: flatten column-major so that a shard is a contiguous run of items
    flat = gather_shards({"cols": np.ascontiguousarray(local.T).ravel()}, G * rows, rank, world_size, group=group,
                         bounds=lambda r: tuple(rows * v for v in shard_range(G, r, world_size)))
    return flat["cols"].reshape(G, rows).T.copy()


def device_localizer(beamf, bf_mat, max_batch=1100):
    """Default localizer: the HIP pipeline (power + arg-max, no T x G temporary)."""

    def run(sig_batch, time_vec):
        am, pm = [], []
        for s in range(0, len(sig_batch), max_batch):
            # (the non-spiking complex Beamformer has no neuron kernel, hence no time axis to pass: ref:paper_plots/target_localization.py)
            kw = dict(time_vec=time_vec) if hasattr(beamf, "tau_vec") else {}
            out = beamf.localize_batch(bf_mat, sig_batch[s : s + max_batch], **kw)
            a = out["argmax"].cpu().numpy().astype(np.int64)
            p = out["power"].cpu().numpy()
            am.append(a)
            pm.append(p[np.arange(len(a)), a])
        return np.concatenate(am), np.concatenate(pm)

    return run


def _throughput_pipelined(beamf, bf_mat, time_test, sig_test, doa_all, snr_db_trial, lo, hi, batch_trials, seed, streams, scan_lane_cus,
                          ranges=None, on_done=None):
    """Trials [lo, hi) of a throughput-mode sweep with several batches in flight (runtime.StreamPipeline): batch k is synthesised
    (micloc_synth_awgn_f64: delayed template + Philox noise numbered by global trial) and localised on stream k % streams, its
    arg-max / power rows return through page-locked memory, nothing on the host waits before the end.  Long recordings (the encoder
    is time-chunked) run their serial checkpoint scans on the pipeline's scan lane.  The same bits as one batch at a time.
    `ranges`: the batches [(s0, s1), ...] to compute (default: all of [lo, hi) in steps of batch_trials -- a resumed sweep skips the
    finished ones); `on_done(s0, s1, argmax, pmax)` is called for every batch as soon as its results are known to be on the host (an
    event behind its copies, polled when the next batch is enqueued -- no extra synchronisation) and for the rest at the end.
    Returns (argmax [hi - lo] int64, pmax [hi - lo]); rows of batches that were not in `ranges` are zero."""
    import torch

    from . import runtime, synthesis
    from .snn_beamformer import neuron_impulse_response

    fs = beamf.fs
    geometry = beamf.geometry
    M = len(geometry)
    time_in, sig_in = synthesis._resample(time_test, sig_test, fs)
    tpl = runtime.Template(time_in, sig_in, fs, device=beamf.device)
    dev, T = tpl.device, tpl.T
    if ranges is None:
        ranges = [(s0, min(hi, s0 + batch_trials)) for s0 in range(lo, hi, batch_trials)]
    nplans = max(1, min(int(streams), len(ranges)))
    nir = neuron_impulse_response(time_in, beamf.tau_vec)
    plans = [beamf.plan()] + [beamf.new_plan() for _ in range(nplans - 1)]
    for pl in plans:
        pl.set_neuron_kernel(nir)
        pl.set_bf_mat(np.asarray(bf_mat, dtype=np.float64))
    G = plans[0].G
    Bmax = max(s1 - s0 for s0, s1 in ranges)
    chunked = plans[0].encoder_chunks(Bmax, T) > 1
    try:
        pipe = runtime.StreamPipeline(plans, scan_lane=scan_lane_cus if (chunked and nplans > 1) else 0)
    except (ValueError, RuntimeError):  # (MiclocError is a RuntimeError)
        # no compute-unit masks here (a partitioned device, fewer compute units per XCD than the lane wants, a driver that refuses
        # them): ordinary streams -- slower on long recordings, the same results
        pipe = runtime.StreamPipeline(plans, scan_lane=0)
    # per plan: input batch and noise workspace (a batch in flight owns them until its stream has moved on); per batch: result rows
    xs = [torch.empty((Bmax, T, M), dtype=torch.float64, device=dev) for _ in plans]
    wss = [runtime.awgn_workspace(Bmax, T, M, dev) for _ in plans]
    outs = [None] * nplans
    snr_dev = torch.from_numpy(np.ascontiguousarray(snr_db_trial[lo:hi], dtype=np.float64)).to(dev)
    am_host = torch.zeros((hi - lo,), dtype=torch.int32).pin_memory()
    pw_host = torch.zeros((hi - lo, G), dtype=torch.float64).pin_memory()
    cur = {}

    def before(i):
        s0, s1 = cur["range"]
        # delays from NumPy's cos like the host path (bit-exact with np.interp on them), one global minimum per trial (:257)
        delays = geometry.delays(doa_all[s0:s1], normalized=False)
        delays = delays - delays.min(axis=1, keepdims=True)
        d_delays = torch.from_numpy(np.ascontiguousarray(delays[:, None, :])).pin_memory().to(dev, non_blocking=True)
        runtime.synth_awgn(tpl, "apply_to_template", snr_dev[s0 - lo : s1 - lo], seed=seed, first_trial=s0, ws=wss[i], delays=d_delays,
                           out=xs[i][: s1 - s0])

    pending = []  # (event behind the batch's device -> host copies, s0, s1)

    def after(i, out):
        s0, s1 = cur["range"]
        am_host[s0 - lo : s1 - lo].copy_(out["argmax"], non_blocking=True)
        pw_host[s0 - lo : s1 - lo].copy_(out["power"], non_blocking=True)
        if on_done is not None:
            ev = torch.cuda.Event()
            ev.record()  # (on the batch's stream: StreamPipeline runs `after` under it)
            pending.append((ev, s0, s1))

    def report(all_=False):
        while pending and (all_ or pending[0][0].query()):
            _, s0, s1 = pending.pop(0)
            a = am_host[s0 - lo : s1 - lo].numpy().astype(np.int64)
            on_done(s0, s1, a, pw_host[s0 - lo : s1 - lo].numpy()[np.arange(s1 - s0), a])

    for k, (s0, s1) in enumerate(ranges):
        cur["range"] = (s0, s1)
        i = k % nplans
        if outs[i] is not None and outs[i]["argmax"].shape[0] != s1 - s0:
            outs[i] = None  # (a last, shorter batch: its own result tensors)
        pipe.snn_pipeline(lambda j: xs[j][: cur["range"][1] - cur["range"][0]], before=before, after=after, index=i, out=outs, want_power=True)
        report()
    pipe.synchronize()
    report(all_=True)
    am = am_host.numpy().astype(np.int64)
    return am, pw_host.numpy()[np.arange(hi - lo), am]


def _template_sweep(beamf, bf_mat, doa_list, time_test, sig_test, snr_db_trial, num_sim, seed, mode, rank, world_size, group,
                    localizer, batch_trials, streams=4, scan_lane_cus=4, out_dir=None, sweep_name="template"):
    """The Monte-Carlo loop shared by the noisy-target and the speech sweep (target_snn_localization.py:447-467 / :224-245):
    per trial `doa = rand(1)[0] * 2 pi`, apply_to_template at `snr_db_trial[trial]`, power, arg-max, pi-periodic error.
    Trials are processed in batches of `batch_trials` (host memory: a speech trial is 18.6 MB).  out_dir: per-batch result files and
    resume (ShardStore): finished trials are loaded, only the missing ones are computed, the result is the uninterrupted run's."""
    total = len(snr_db_trial)
    lo, hi = shard_range(total, rank, world_size)
    store = None
    if out_dir is not None:
        store = ShardStore(out_dir, sweep_name, total, seed=int(seed), mode=mode, snr_db_trial=np.asarray(snr_db_trial, dtype=np.float64),
                           doa_list=np.asarray(doa_list, dtype=np.float64), bf_mat=np.asarray(bf_mat), time_test=np.asarray(time_test, dtype=np.float64),
                           sig_test=np.asarray(sig_test, dtype=np.float64), fs=float(beamf.fs), num_mic=len(beamf.geometry),
                           r_vec=np.asarray(beamf.geometry.r_vec, dtype=np.float64), theta_vec=np.asarray(beamf.geometry.theta_vec, dtype=np.float64))
    done = store.have.copy() if store is not None else np.zeros(total, dtype=bool)
    pipelined = mode == "throughput" and localizer is None and streams > 0 and hi > lo
    localizer = localizer or device_localizer(beamf, bf_mat, max_batch=batch_trials)
    M = len(beamf.geometry)
    doa_all = np.zeros(total)
    argmax = np.zeros(hi - lo, dtype=np.int64)
    pmax = np.zeros(hi - lo)
    if store is not None:  # what earlier runs finished of this rank's shard
        argmax[done[lo:hi]] = store.rec["index"][lo:hi][done[lo:hi]]
        pmax[done[lo:hi]] = store.rec["pmax"][lo:hi][done[lo:hi]]

    def finished(trials, a, p):
        trials = np.asarray(trials, dtype=np.int64)
        argmax[trials - lo] = np.asarray(a, dtype=np.int64)
        pmax[trials - lo] = np.asarray(p, dtype=np.float64)
        if store is not None:
            store.put(trials, doa_all[trials], a, p)

    def flush(sig_batch, time_in, trials):
        a, p = localizer(sig_batch, time_in)
        finished(trials, a, p)

    if mode == "parity":
        # the reference's global MT19937 stream, replayed on every rank; a rank keeps the trials of its shard (a resumed sweep: the
        # ones of its shard that no earlier run finished -- the stream is drawn for every trial all the same)
        np.random.seed(seed)
        T = len(np.arange(time_test.min(), time_test.max(), step=1 / beamf.fs))
        sigs, ids, time_in = [], [], None
        for trial in range(total):
            doa = np.random.rand(1)[0] * 2 * np.pi
            doa_all[trial] = doa
            if lo <= trial < hi and not done[trial]:
                time_in, sig = synthesize_array_signal(beamf.geometry, beamf.fs, time_test, sig_test, doa)
                sig += np.sqrt(np.mean(sig**2)) / np.sqrt(10 ** (snr_db_trial[trial] / 10)) * np.random.randn(*sig.shape)
                sigs.append(sig)
                ids.append(trial)
                if len(sigs) == batch_trials:
                    flush(np.stack(sigs), time_in, ids)
                    sigs, ids = [], []
            else:
                np.random.randn(T, M)  # keep the stream aligned: the reference draws T x M normals for every trial
        if sigs:
            flush(np.stack(sigs), time_in, ids)
    elif mode == "throughput":
        from . import synthesis

        rng = np.random.RandomState(seed)
        doa_all[:] = rng.rand(total) * 2 * np.pi
        ranges = [(s0, min(hi, s0 + batch_trials)) for s0 in range(lo, hi, batch_trials)]
        ranges = [(s0, s1) for s0, s1 in ranges if not done[s0:s1].all()]  # (a batch with any trial missing is recomputed whole)
        if pipelined and ranges:
            # the default localizer: several batches in flight, no host synchronisation between them (same bits)
            _throughput_pipelined(beamf, bf_mat, time_test, sig_test, doa_all, snr_db_trial, lo, hi, batch_trials, seed, streams, scan_lane_cus,
                                  ranges=ranges, on_done=lambda s0, s1, a, p: finished(np.arange(s0, s1), a, p))
        for s0, s1 in (() if pipelined else ranges):
            # noise-free array signals synthesised on the device (bit-exact with the host np.interp path), noise from the
            # Philox kernel, numbered by GLOBAL trial: the same draw for any sharding
            time_in, x = beamf.synthesize_batch((time_test, sig_test), doa_all[s0:s1])
            synthesis.add_noise_(x, snr_db_trial[s0:s1], seed=seed, first_trial=s0)
            flush(x, time_in, np.arange(s0, s1))
    else:
        raise ValueError("mode must be 'parity' or 'throughput'")

    # the one exchange step: {argmax i64, pmax f64} per trial in ONE all-gather (the DoAs come from the shared stream: every rank has them)
    exchange = {}
    full = gather_shards({"argmax": argmax, "pmax": pmax}, total, rank, world_size, group, stats=exchange)
    err = doa_error(np.asarray(doa_list)[full["argmax"]], doa_all)
    shape = (total // num_sim, num_sim)
    res = dict(doa=doa_all.reshape(shape), argmax=full["argmax"].reshape(shape), pmax=full["pmax"].reshape(shape), err=err.reshape(shape),
               mae_deg=np.mean(err.reshape(shape), axis=1) * 180 / np.pi, exchange=exchange)
    if store is not None:
        res["persistence"] = store.stats()
    return res


def noisy_target_sweep(beamf, bf_mat, doa_list, snr_db_vec=None, num_sim=100, seed=0, mode="parity", rank=0, world_size=1,
                       group=None, freq_design=2000.0, test_duration=100e-3, snr_gain_due_to_bandwidth=None, localizer=None,
                       batch_trials=1100, streams=4, out_dir=None):
    """paper_plots/target_snn_localization.py:435-467.  Returns dict(doa, argmax, err, pmax: [num_snr, num_sim];
    mae_deg [num_snr]) on every rank.  out_dir: every finished batch is written there and a rerun with the same arguments resumes
    (ShardStore; `persistence` in the result says what was loaded and written)."""
    fs = beamf.fs
    snr_db_vec = np.asarray(np.linspace(-10, 20, 11) if snr_db_vec is None else snr_db_vec, dtype=np.float64)
    if snr_gain_due_to_bandwidth is None:
        snr_gain_due_to_bandwidth = (fs / 2) / 1000.0  # (fs/2)/(f_max - f_min) with the paper's [1, 2] kHz band
    time_test = np.arange(0, test_duration, step=1 / fs)
    sig_test = np.sin(2 * np.pi * freq_design * time_test)
    snr_trial = np.repeat(snr_db_vec - 10 * np.log10(snr_gain_due_to_bandwidth), num_sim)  # :449
    res = _template_sweep(beamf, bf_mat, doa_list, time_test, sig_test, snr_trial, num_sim, seed, mode, rank, world_size, group,
                          localizer, batch_trials, streams, out_dir=out_dir, sweep_name="noisy")
    res["snr_db_vec"] = snr_db_vec
    return res


def speech_source(fs, flac_path=None, pcm16=None, rate=None):
    """The speech test signal of target_snn_localization.py:148-154: the LibriSpeech utterance (FLAC file decoded by
    haghighatshoarmuir2024_amd.flac, or already-decoded int16 PCM), resampled to `fs` with np.interp on a linspace grid.
    Returns (time_fs, sig_fs)."""
    if pcm16 is None:
        from . import flac

        pcm, rate, _ = flac.decode(open(flac_path, "rb").read())
        sig = pcm[:, 0].astype(np.float64) / 32768.0  # soundfile.read returns float64 in [-1, 1)
    else:
        sig = np.asarray(pcm16).astype(np.float64) / 32768.0
    rate = int(rate)
    time_test = np.arange(len(sig)) / rate
    time_fs = np.linspace(time_test[0], time_test[-1], int(len(sig) / rate * fs))
    return time_fs, np.interp(time_fs, time_test, sig)


def speech_target_sweep(beamf, bf_mat, doa_list, source, snr_db_vec=None, num_sim=20, seed=0, mode="parity", rank=0, world_size=1,
                        group=None, localizer=None, batch_trials=None, streams=4, out_dir=None):
    """The speech accuracy sweep of paper_plots/target_snn_localization.py:213-245: `source` = (time_fs, sig_test) from
    `speech_source`, 11 SNRs x 20 trials, NO bandwidth correction of the SNR (`snr_db_target = snr_db`, :227).
    batch_trials: trials per device batch (default: 25 in parity mode -- a trial is 18.6 MB on the host --, 125 in throughput mode,
    where the batches are synthesised on the device and `streams` of them are in flight; streams=0: one batch at a time).
    out_dir: per-batch result files and resume, as in noisy_target_sweep."""
    if batch_trials is None:
        batch_trials = 125 if mode == "throughput" else 25
    snr_db_vec = np.asarray(np.linspace(-10, 20, 11) if snr_db_vec is None else snr_db_vec, dtype=np.float64)
    time_fs, sig_test = source
    res = _template_sweep(beamf, bf_mat, doa_list, np.asarray(time_fs, dtype=np.float64), np.asarray(sig_test, dtype=np.float64),
                          np.repeat(snr_db_vec, num_sim), num_sim, seed, mode, rank, world_size, group, localizer, batch_trials, streams,
                          out_dir=out_dir, sweep_name="speech")
    res["snr_db_vec"] = snr_db_vec
    return res


def xylo_target_sweep(demo, snr_db_vec=None, num_sim=100, seed=0, mode="parity", rank=0, world_size=1, group=None,
                      test_duration=1000e-3, snr_gain_due_to_bandwidth=None, batch_trials=None, device_delays=None, peak=None, out_dir=None):
    """The Xylo accuracy sweep of paper_plots/target_xylo_localization.py:540-608 (and its `_unipolar` twin): chirp test
    signal over the design band (:549-560), per trial `signal_from_template` -> AWGN -> `spike_encoding` -> `xylo_process`
    -> spike rate -> `find_peak_location(win_size)` with win_size = 2 * ((num_grid // 32) // 2) + 1 (:600-603) -> error.

    PARITY UNPINNED for the integer-LIF stage (rockpool / XyloSim absent: xylo_snn_localization.py module docstring);
    everything around it follows the reference's arithmetic.  `demo` is a xylo_snn_localization.Demo with one band.
    batch_trials: trials per device batch -- default 50 in parity mode (host arrays of 2.7 MB per trial), 1100 in throughput mode: the
    integer LIF is one serial chain per trial, a launch takes as long for 50 trials as for 1100."""
    from . import synthesis
    from .utils import find_peak_location
    from .xylo_snn_localization import signal_from_template

    fs = demo.fs
    snr_db_vec = np.asarray(np.linspace(-10, 20, 11) if snr_db_vec is None else snr_db_vec, dtype=np.float64)
    f_min, f_max = [float(v) for v in demo.freq_bands[0]]
    if snr_gain_due_to_bandwidth is None:
        snr_gain_due_to_bandwidth = (fs / 2) / (f_max - f_min)
    time_test = np.arange(0, test_duration, step=1 / fs)
    period = time_test[-1]
    freq_inst = f_min + (f_max - f_min) * (time_test % period) / period
    sig_test = np.sin(2 * np.pi * np.cumsum(freq_inst) * 1 / fs)
    geometry = demo.beamfs[0].geometry
    doa_list = demo.doa_list
    num_grid = len(doa_list)
    win_size = 2 * ((num_grid // 32) // 2) + 1
    total = len(snr_db_vec) * num_sim
    snr_trial = np.repeat(snr_db_vec - 10 * np.log10(snr_gain_due_to_bandwidth), num_sim)
    lo, hi = shard_range(total, rank, world_size)
    if batch_trials is None:
        batch_trials = 1100 if mode == "throughput" else 50
    if device_delays is None:
        device_delays = mode == "throughput"
    doa_all = np.zeros(total)
    index = np.zeros(hi - lo, dtype=np.int64)

    if peak is None:
        peak = "device" if mode == "throughput" else "host"
    store = None
    if out_dir is not None:
        store = ShardStore(out_dir, "xylo", total, seed=int(seed), mode=mode, snr_db_trial=snr_trial, doa_list=np.asarray(doa_list, dtype=np.float64),
                           bf_mat=np.asarray(demo.bf_mats[0]), fs=float(fs), num_mic=len(geometry), bipolar=bool(demo.bipolar_spikes),
                           time_test=time_test, sig_test=sig_test, peak=peak, device_delays=bool(device_delays), win_size=int(win_size),
                           r_vec=np.asarray(geometry.r_vec, dtype=np.float64), theta_vec=np.asarray(geometry.theta_vec, dtype=np.float64))
    done = store.have.copy() if store is not None else np.zeros(total, dtype=bool)
    if store is not None:
        index[done[lo:hi]] = store.rec["index"][lo:hi][done[lo:hi]]

    def flush(x, trials):
        trials = np.asarray(trials, dtype=np.int64)
        if peak == "device":  # find_peak_location on the device (exact integer window sums): only indices come back
            idx = demo.peak_batch(x, win_size).cpu().numpy().astype(np.int64)
            demo.network().check()  # (the copy above synchronised: a broken ticket-queue launch raises here)
        else:
            rate = demo.rate_batch(x).cpu().numpy()  # [B, G]: mean(spikes_out) * fs per DoA
            demo.network().check()
            idx = []
            for p in rate:
                mx = p.max()
                p = p / mx if mx > 0 else p  # :595 (an all-silent output divides 0 by 0 in the reference)
                idx.append(int(find_peak_location(sig_in=p, win_size=win_size)))
            idx = np.asarray(idx, dtype=np.int64)
        index[trials - lo] = idx
        if store is not None:
            store.put(trials, doa_all[trials], idx, np.zeros(len(trials)))

    if mode == "parity":
        np.random.seed(seed)
        T, M = len(time_test), len(geometry)
        sigs, ids = [], []
        for trial in range(total):
            doa = np.random.rand(1)[0] * 2 * np.pi
            doa_all[trial] = doa
            if lo <= trial < hi and not done[trial]:
                sig = signal_from_template(geometry, (time_test, sig_test, doa))
                noise_sigma = np.sqrt(np.mean(sig**2) / 10 ** (snr_trial[trial] / 10))
                sigs.append(sig + noise_sigma * np.random.randn(*sig.shape))
                ids.append(trial)
                if len(sigs) == batch_trials:
                    flush(np.stack(sigs), ids)
                    sigs, ids = [], []
            else:
                np.random.randn(T, M)
        if sigs:
            flush(np.stack(sigs), ids)
    elif mode == "throughput":
        rng = np.random.RandomState(seed)
        doa_all[:] = rng.rand(total) * 2 * np.pi
        for s0 in range(lo, hi, batch_trials):
            s1 = min(hi, s0 + batch_trials)
            if done[s0:s1].all():
                continue
            x = synthesis.signal_from_template_batch(geometry, (time_test, sig_test), doa_all[s0:s1], device=demo.device, device_delays=device_delays)
            synthesis.add_noise_(x, snr_trial[s0:s1], seed=seed, first_trial=s0)
            flush(x, np.arange(s0, s1))
    else:
        raise ValueError("mode must be 'parity' or 'throughput'")

    full = gather_shards({"index": index}, total, rank, world_size, group)
    err = doa_error(np.asarray(doa_list)[full["index"]], doa_all)
    shape = (len(snr_db_vec), num_sim)
    res = dict(doa=doa_all.reshape(shape), index=full["index"].reshape(shape), err=err.reshape(shape),
               mae_deg=np.mean(err.reshape(shape), axis=1) * 180 / np.pi, snr_db_vec=snr_db_vec, win_size=win_size, parity="unpinned (integer LIF)")
    if store is not None:
        res["persistence"] = store.stats()
    return res


def main(argv=None):
    """`python -m haghighatshoarmuir2024_amd.sweep [--sweep noisy|speech|xylo]`: the accuracy sweeps of the paper scripts
    (paper_plots/target_snn_localization.py:309-520 noisy target, :97-300 speech target; target_xylo_localization.py:540-608),
    design + 11 SNRs x num_sim trials, printing what the scripts print (SNR vector and mean absolute errors in degrees)."""
    import argparse
    import os

    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--sweep", choices=["noisy", "speech", "xylo"], default="noisy")
    ap.add_argument("--num-sim", type=int, default=None, help="trials per SNR (scripts: 100 noisy / xylo, 20 speech)")
    ap.add_argument("--grid", type=int, default=64 * 7 + 1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--mode", choices=["parity", "throughput"], default="parity")
    ap.add_argument("--flac", default=None, help="speech sweep: the LibriSpeech utterance (84-121123-0020.flac of the reference's paper_plots/)")
    ap.add_argument("--pcm-npz", default=None, help="speech sweep: an .npz with `pcm16` and `rate` instead of the FLAC file")
    ap.add_argument("--svd", choices=["host", "device"], default="host", help="design_from_template decompositions")
    args = ap.parse_args(argv)

    from .array_geometry import CenterCircularArray
    from .snn_beamformer import SNNBeamformer

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
        dist.init_process_group("nccl")
    fs, freq_design = 48_000, 2000.0
    freq_range = [0.5 * freq_design, freq_design]
    tau = 1.0 / (2 * np.pi * freq_design)
    geometry = CenterCircularArray(radius=4.5e-2, num_mic=7)
    doa_list = np.linspace(-np.pi, np.pi, args.grid)
    if args.sweep == "xylo":
        from .xylo_snn_localization import Demo

        demo = Demo(geometry=geometry, freq_bands=[freq_range], doa_list=doa_list, recording_duration=0.25, bipolar_spikes=True, fs=fs)
        res = xylo_target_sweep(demo, num_sim=args.num_sim or 100, seed=args.seed, mode=args.mode, rank=rank, world_size=world)
    else:
        beamf = SNNBeamformer(geometry, kernel_duration=10.0e-3, tau_vec=np.asarray([tau, tau]), freq_range=freq_range, fs=fs, bipolar_spikes=True)
        time_temp = np.arange(0, 1.0, step=1 / fs)
        period = time_temp[-1]
        freq_inst = freq_range[0] + (freq_range[1] - freq_range[0]) * (time_temp % period) / period
        sig_temp = np.sin(2 * np.pi * np.cumsum(freq_inst) / fs)
        # the design is sharded over the DoA grid as well (one all-gather of the columns)
        bf_mat = sharded_design(lambda doas: beamf.design_from_template((time_temp, sig_temp), doas, svd=args.svd), doa_list, rank, world)
        if args.sweep == "speech":
            if args.pcm_npz:
                z = np.load(args.pcm_npz)
                src = speech_source(fs, pcm16=z["pcm16"], rate=int(z["rate"]))
            elif args.flac:
                src = speech_source(fs, flac_path=args.flac)
            else:
                ap.error("--sweep speech needs --flac or --pcm-npz")
            res = speech_target_sweep(beamf, bf_mat, doa_list, src, num_sim=args.num_sim or 20, seed=args.seed, mode=args.mode, rank=rank, world_size=world)
        else:
            res = noisy_target_sweep(beamf, bf_mat, doa_list, num_sim=args.num_sim or 100, seed=args.seed, mode=args.mode, rank=rank, world_size=world)
    if rank == 0:
        print(f"SNR: {res['snr_db_vec']}")
        print(f"Mean aboslute errors: {res['mae_deg']}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
