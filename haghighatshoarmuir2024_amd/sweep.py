"""Monte-Carlo DoA sweep (restatement of paper_plots/target_snn_localization.py:435-467 as a batched,
shardable harness).  The script's loop carries no state between trials except the RNG stream, so trials
shard contiguously across ranks (one process per GPU); the only exchange is one small all-gather of
per-trial results at the end (RCCL over xGMI when the process group is "nccl").

parity mode      every rank replays the reference's global NumPy stream (rand(1) then randn(T, M) per trial,
                 legacy MT19937) and keeps its own shard, so results are identical to the single-process
                 reference for any world size.
throughput mode  DoAs from a seeded host generator, clean array signals synthesised on the device
                 (micloc_synth_delay_f64, bit-exact with np.interp), noise drawn on the device (torch Philox generator
                 seeded per rank).
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


def gather_shards(local, total, rank, world_size, group=None, bounds=None):
    """All-gather equal-dtype 1-d arrays from contiguous shards; returns the full-length arrays on every rank.
    `local` is a dict name -> 1-d numpy array (this rank's shard).  Uses torch.distributed when world_size > 1.
    `bounds(r) -> (lo, hi)` overrides the default `shard_range(total, r, world_size)` partition."""
    if bounds is None:
        bounds = lambda r: shard_range(total, r, world_size)  # noqa: E731
    if world_size == 1:
        return {k: np.asarray(v) for k, v in local.items()}
    import torch
    import torch.distributed as dist

    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    width = max(bounds(r)[1] - bounds(r)[0] for r in range(world_size))  # padded shard length
    out = {}
    for k, v in local.items():
        v = np.asarray(v)
        buf = torch.zeros(width, dtype=torch.from_numpy(v[:0].copy()).dtype, device=dev)
        buf[: len(v)] = torch.from_numpy(np.ascontiguousarray(v)).to(dev)
        full = torch.empty(width * world_size, dtype=buf.dtype, device=dev)
        dist.all_gather_into_tensor(full, buf, group=group)
        full = full.cpu().numpy().reshape(world_size, width)
        parts = []
        for r in range(world_size):
            lo, hi = bounds(r)
            parts.append(full[r, : hi - lo])
        out[k] = np.concatenate(parts)
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
            out = beamf.localize_batch(bf_mat, sig_batch[s : s + max_batch], time_vec=time_vec)
            a = out["argmax"].cpu().numpy().astype(np.int64)
            p = out["power"].cpu().numpy()
            am.append(a)
            pm.append(p[np.arange(len(a)), a])
        return np.concatenate(am), np.concatenate(pm)

    return run


def noisy_target_sweep(beamf, bf_mat, doa_list, snr_db_vec=None, num_sim=100, seed=0, mode="parity", rank=0, world_size=1,
                       group=None, freq_design=2000.0, test_duration=100e-3, snr_gain_due_to_bandwidth=None, localizer=None):
    """Returns dict(doa, argmax, err, pmax: [num_snr, num_sim]; mae_deg [num_snr]) on every rank."""
    fs = beamf.fs
    if snr_db_vec is None:
        snr_db_vec = np.linspace(-10, 20, 11)
    snr_db_vec = np.asarray(snr_db_vec, dtype=np.float64)
    if snr_gain_due_to_bandwidth is None:
        snr_gain_due_to_bandwidth = (fs / 2) / 1000.0  # (fs/2)/(f_max - f_min) with the paper's [1, 2] kHz band
    time_test = np.arange(0, test_duration, step=1 / fs)
    sig_test = np.sin(2 * np.pi * freq_design * time_test)
    total = len(snr_db_vec) * num_sim
    lo, hi = shard_range(total, rank, world_size)
    localizer = localizer or device_localizer(beamf, bf_mat)

    doa_all = np.zeros(total)
    sigs = []
    time_in = None
    if mode == "parity":
        np.random.seed(seed)
        for trial in range(total):
            snr_db = snr_db_vec[trial // num_sim] - 10 * np.log10(snr_gain_due_to_bandwidth)
            doa = np.random.rand(1)[0] * 2 * np.pi
            doa_all[trial] = doa
            if lo <= trial < hi:
                time_in, sig = synthesize_array_signal(beamf.geometry, fs, time_test, sig_test, doa)
                sig += np.sqrt(np.mean(sig**2)) / np.sqrt(10 ** (snr_db / 10)) * np.random.randn(*sig.shape)
                sigs.append(sig)
            else:
                # keep the global stream aligned: the reference draws T x M normals for every trial
                np.random.randn(len(time_test) - 1, len(beamf.geometry))
        sig_batch = np.stack(sigs) if sigs else np.zeros((0, len(time_test) - 1, len(beamf.geometry)))
    elif mode == "throughput":
        import torch

        rng = np.random.RandomState(seed)
        doa_all[:] = rng.rand(total) * 2 * np.pi
        # noise-free array signals synthesised on the device (bit-exact with the host np.interp path)
        time_in, clean = beamf.synthesize_batch((time_test, sig_test), doa_all[lo:hi])
        gen = torch.Generator(device=clean.device)
        gen.manual_seed(seed * 1_000_003 + rank)
        snr_db = torch.from_numpy(snr_db_vec[np.arange(lo, hi) // num_sim] - 10 * np.log10(snr_gain_due_to_bandwidth)).to(clean.device)
        sigma = torch.sqrt(torch.mean(clean**2, dim=(1, 2))) / torch.sqrt(10 ** (snr_db / 10))
        sig_batch = clean + sigma[:, None, None] * torch.randn(clean.shape, generator=gen, device=clean.device, dtype=torch.float64)
    else:
        raise ValueError("mode must be 'parity' or 'throughput'")

    if hi > lo:
        argmax, pmax = localizer(sig_batch, time_in)
    else:
        argmax, pmax = np.zeros(0, dtype=np.int64), np.zeros(0)
    full = gather_shards({"argmax": np.asarray(argmax, dtype=np.int64), "pmax": np.asarray(pmax, dtype=np.float64)}, total, rank, world_size, group)
    doa_list = np.asarray(doa_list)
    err = doa_error(doa_list[full["argmax"]], doa_all)
    shape = (len(snr_db_vec), num_sim)
    return dict(doa=doa_all.reshape(shape), argmax=full["argmax"].reshape(shape), pmax=full["pmax"].reshape(shape), err=err.reshape(shape),
                mae_deg=np.mean(err.reshape(shape), axis=1) * 180 / np.pi, snr_db_vec=snr_db_vec)


def main(argv=None):
    """`python -m haghighatshoarmuir2024_amd.sweep`: the noisy-target accuracy sweep of
    paper_plots/target_snn_localization.py:309-520 (design + 11 SNRs x num_sim trials), printing what the script prints."""
    import argparse
    import os

    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("--num-sim", type=int, default=100)
    ap.add_argument("--grid", type=int, default=64 * 7 + 1)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--mode", choices=["parity", "throughput"], default="parity")
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
    beamf = SNNBeamformer(CenterCircularArray(radius=4.5e-2, num_mic=7), kernel_duration=10.0e-3, tau_vec=np.asarray([tau, tau]),
                          freq_range=freq_range, fs=fs, bipolar_spikes=True)
    time_temp = np.arange(0, 1.0, step=1 / fs)
    period = time_temp[-1]
    freq_inst = freq_range[0] + (freq_range[1] - freq_range[0]) * (time_temp % period) / period
    sig_temp = np.sin(2 * np.pi * np.cumsum(freq_inst) / fs)
    doa_list = np.linspace(-np.pi, np.pi, args.grid)
    # the design is sharded over the DoA grid as well (one all-gather of the columns)
    bf_mat = sharded_design(lambda doas: beamf.design_from_template((time_temp, sig_temp), doas), doa_list, rank, world)
    res = noisy_target_sweep(beamf, bf_mat, doa_list, num_sim=args.num_sim, seed=args.seed, mode=args.mode, rank=rank, world_size=world)
    if rank == 0:
        print(f"SNR: {res['snr_db_vec']}")
        print(f"Mean aboslute errors: {res['mae_deg']}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
