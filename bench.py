#!/usr/bin/env python3
"""Headline benchmark: audio frames/s through the full STHT -> RZCC -> SNN-beamform pipeline.

Workload (BASELINE.json configs[1], target_snn_localization.py): 7-mic centre-circular array, 48 kHz,
0.1 s noisy 2 kHz test tone (T = 4799 frames), 11 SNRs x 100 Monte-Carlo trials = 1100 trials per step,
DoA grid 360 (BASELINE's nominal grid; `--grid 449` selects the script-exact one), bf_mat designed from the
1 s 1->2 kHz chirp with design_from_template on the device before timing.  One "step" = one pass of the
hot path over the 1100-trial batch (inputs resident in HBM): STHT, band-pass, RZCC, LIF, beamforming,
power, arg-max, DoA error / MAE.  Weak scaling: every rank processes its own 1100-trial batch.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--grid G] [--no-cpu-baseline]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the field definitions).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# HBM bytes per launch of the dominant kernel for the DEFAULT workload (1100 trials, T=4799, G=360), from separate
# rocprofv3 --pmc passes (profiles/r1/pmc_summary.csv): 2 x FETCH_SIZE (gfx950 counts wide reads at half,
# MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both reported in KiB: 2 * 36409 + 60088 KiB.
BEAMFORM_TRAFFIC_BYTES_DEFAULT = (2 * 36409 + 60088) * 1024
# stage key in `stages_ms` -> device symbol that dominates it (what rocprofv3 lists)
KERNEL_SYMBOL = {"beamform_kernel": "beamform_ws_kernel", "stht_kernel": "stht_kernel", "bandpass_rzcc_kernel": "bandpass_rzcc_fast_kernel"}
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X datasheet fp64 matrix = fp64 vector = 1/2 of the 157.3 TF fp32 rate in MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", type=int, default=360)
    ap.add_argument("--trials", type=int, default=1100, help="trials per rank per step (11 SNRs x 100)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1100, help="trials timed on the CPU oracle (the whole batch: about 15 s of CPU work)")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams consecutive steps are pipelined over (1 = serial)")
    ap.add_argument("--traffic-bytes", type=float, default=None, help="HBM bytes per dominant-kernel launch from a separate rocprofv3 --pmc pass")
    return ap.parse_args()


def build_workload(args, rank, device):
    import torch

    from haghighatshoarmuir2024_amd.array_geometry import CenterCircularArray
    from haghighatshoarmuir2024_amd.snn_beamformer import SNNBeamformer, neuron_impulse_response

    fs, num_mic, freq_design = 48_000, 7, 2000.0
    freq_range = [0.5 * freq_design, freq_design]
    tau = 1.0 / (2 * np.pi * freq_design)
    geometry = CenterCircularArray(radius=4.5e-2, num_mic=num_mic)
    beamf = SNNBeamformer(geometry=geometry, kernel_duration=10.0e-3, tau_vec=np.asarray([tau, tau]), freq_range=freq_range, fs=fs,
                          bipolar_spikes=True, device=device)
    # chirp template and DoA grid (target_snn_localization.py:345-371)
    time_temp = np.arange(0, 1.0, step=1 / fs)
    period = time_temp[-1]
    freq_inst = freq_range[0] + (freq_range[1] - freq_range[0]) * (time_temp % period) / period
    sig_temp = np.sin(2 * np.pi * np.cumsum(freq_inst) / fs)
    doa_list = np.linspace(-np.pi, np.pi, args.grid)
    bf_mat = beamf.design_from_template((time_temp, sig_temp), doa_list)

    # test signals (target_snn_localization.py:435-455): synthetic, generated here, noise drawn on the device
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * freq_design * time_test)
    B = args.trials
    rng = np.random.RandomState(1000 + rank)
    doa = rng.rand(B) * 2 * np.pi
    snr_db_vec = np.linspace(-10, 20, 11)
    snr_db = snr_db_vec[(np.arange(B) * len(snr_db_vec)) // B] - 10 * np.log10((fs / 2) / (freq_range[1] - freq_range[0]))
    time_in, clean = beamf.synthesize_batch((time_test, sig_test), doa)  # device synthesis, bit-exact with np.interp
    gen = torch.Generator(device=device)
    gen.manual_seed(1234 + rank)
    sigma = torch.sqrt(torch.mean(clean**2, dim=(1, 2))) / torch.sqrt(10 ** (torch.from_numpy(snr_db).to(device) / 10))
    x = (clean + sigma[:, None, None] * torch.randn(clean.shape, generator=gen, device=device, dtype=torch.float64)).contiguous()
    del clean

    plan = beamf.plan()
    nir = neuron_impulse_response(time_in, beamf.tau_vec)
    plan.set_neuron_kernel(nir)
    plan.set_bf_mat(bf_mat)
    return dict(beamf=beamf, plan=plan, x=x, doa=torch.from_numpy(doa).to(device), doa_list=torch.from_numpy(doa_list).to(device),
                bf_mat=bf_mat, nir=nir, snr_groups=len(snr_db_vec), fs=fs)


def make_step(wl, nstreams):
    """One step = one pass of the hot path over the batch.  Consecutive steps are independent batches, so they are
    dispatched round-robin over `nstreams` HIP streams (one plan/workspace each): the latency-bound RZCC kernel of
    one step overlaps the throughput-bound STHT / beamforming kernels of its neighbours."""
    import torch

    from haghighatshoarmuir2024_amd.runtime import StreamPipeline

    x, doa, doa_list, S = wl["x"], wl["doa"], wl["doa_list"], wl["snr_groups"]
    plans = [wl["plan"]]
    for _ in range(nstreams - 1):
        p = wl["beamf"].new_plan()
        p.set_neuron_kernel(wl["nir"])
        p.set_bf_mat(wl["bf_mat"])
        plans.append(p)
    pipe = StreamPipeline(plans)

    def body(plan, cov=False):
        if cov == "f32":
            out = plan.snn_pipeline_f32bf(x)
        else:
            out = plan.snn_pipeline_cov(x, want_power=True) if cov else plan.snn_pipeline(x, want_power=True)
        est = doa_list[out["argmax"].long()]
        err = torch.arcsin(torch.abs(torch.sin(est - doa)))
        mae = err.reshape(S, -1).mean(dim=1)
        return out, mae

    # one HIP graph per stream (pipeline kernels + the DoA-error / MAE ops), replayed round-robin
    replay_direct = pipe.capture(lambda plan: body(plan, False))
    replay_cov = pipe.capture(lambda plan: body(plan, True)) if x.shape[2] * 2 <= 64 else None

    replay_f32 = pipe.capture(lambda plan: body(plan, "f32")) if x.shape[2] * 2 <= 64 else None

    def step(cov=False):
        if cov == "f32":
            return replay_f32()
        return replay_cov() if cov else replay_direct()

    return step, pipe


def stage_times(wl, iters):
    """Average duration of each stage of the pipeline (HIP events on the launch stream), in ms."""
    import torch

    plan, x = wl["plan"], wl["x"]
    B, T, M = x.shape
    h = plan.stht(x)
    _, spikes = plan.bandpass_rzcc(h, T, want_pre=False, want_spikes=True)
    res = {}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(iters)]
        for i in range(iters):
            e0[i].record()
            fn()
            e1[i].record()
        torch.cuda.synchronize()
        return float(np.mean([a.elapsed_time(b) for a, b in zip(e0, e1)]))

    res["stht_kernel"] = timed(lambda: plan.stht(x))
    res["bandpass_rzcc_kernel"] = timed(lambda: plan.bandpass_rzcc(h, T, want_pre=False, want_spikes=True))
    res["beamform_kernel"] = timed(lambda: plan.lif_beamform(spikes, want_power=True))
    return res


def cpu_baseline(wl, n):
    """The oracle (CPU restatement, scalar C, one thread) on the first n trials of the same batch."""
    from oracle import oracle as O

    beamf = wl["beamf"]
    x = wl["x"][:n].cpu().numpy()
    b, a = beamf.bandpass_filter
    O.lib()
    t0 = time.perf_counter()
    pw, am = O.snn_chain_batch(x, beamf.kernel, b, a, beamf.spk_encoder.robust_width, True, wl["nir"], wl["bf_mat"])
    dt = time.perf_counter() - t0
    return dict(value=n * x.shape[1] / dt, unit="frames/s", cores=1, kind="port",
                sample=f"{n} of the {wl['x'].shape[0]} trials of the same batch (T={x.shape[1]}, M={x.shape[2]}, G={wl['bf_mat'].shape[1]}), "
                       f"{dt:.1f} s, oracle/micloc_oracle.c single thread"), am


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and world > 1:
        args.gpus = world
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # MICLOC_FORCE_DIST=1 runs the collective code path with a 1-rank RCCL group (to exercise it on a 1-GPU box)
    use_dist = world > 1 or os.environ.get("MICLOC_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    wl = build_workload(args, rank, device)
    step, pipe = make_step(wl, max(1, args.streams))
    B, T, M = wl["x"].shape
    G = wl["bf_mat"].shape[1]

    def barrier():
        pipe.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, mae = step()
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # the sweep's one exchange step: gather the per-rank MAE curves (RCCL)
        gathered = [torch.empty_like(mae) for _ in range(world)]
        dist.all_gather(gathered, mae)
        mae = torch.stack(gathered).mean(dim=0)
    dt = float(tmax.item())
    frames = world * B * T * args.steps
    value = frames / dt
    argmax_direct = out["argmax"].clone()  # graph outputs are static buffers: keep a copy for the comparisons below

    # separately reported algorithmic variant: covariance-form power, same K steps, same inputs
    cov_variant = None
    if M * 2 <= 64:
        for _ in range(args.warmup):
            step(cov=True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out_c, mae_c = step(cov=True)
        barrier()
        dtc = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if use_dist:
            dist.all_reduce(dtc, op=dist.ReduceOp.MAX)
        dtc = float(dtc.item())
        cov_variant = {"value": frames / dtc, "unit": "frames/s", "ms_per_step": dtc / args.steps * 1e3,
                       "argmax_equal_to_direct": bool(torch.equal(out_c["argmax"], argmax_direct)),
                       "note": "power = w^T (V^T V / T) w instead of mean_t (V w)^2: algebraically identical, 2C^2 instead of 2CG flops per frame; not the headline"}

    # second separately reported variant: fp32-MFMA beamforming tail (fp64 up to the spikes)
    f32_variant = None
    if M * 2 <= 64:
        for _ in range(args.warmup):
            step(cov="f32")
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out_f, mae_f = step(cov="f32")
        barrier()
        dtf = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if use_dist:
            dist.all_reduce(dtf, op=dist.ReduceOp.MAX)
        dtf = float(dtf.item())
        relerr = float((out_f["power"] / out["power"] - 1).abs().max().item())
        f32_variant = {"value": frames / dtf, "unit": "frames/s", "ms_per_step": dtf / args.steps * 1e3,
                       "argmax_equal_to_f64": int((out_f["argmax"] == argmax_direct).sum().item()), "trials": int(B),
                       "max_rel_power_err_vs_f64": relerr,
                       "note": "LIF + beamforming + power on v_mfma_f32_16x16x4_f32 (157 TF peak); STHT / band-pass / RZCC stay fp64; not the headline"}

    result = None
    if rank == 0:
        st = stage_times(wl, max(5, min(args.steps, 20)))
        n_nir = len(wl["nir"])
        C = 2 * M
        # algorithmic work per frame (SURVEY 8d / DESIGN.md): dense-tap STHT as the reference computes it
        flops = {
            "stht_kernel": 2 * len(wl["beamf"].kernel) * M,
            "bandpass_rzcc_kernel": 17 * C,
            "beamform_kernel": 2 * n_nir * C + 2 * C * G + 2 * G,
        }
        dom = max(st, key=st.get)
        frames_launch = B * T
        if dom == "bandpass_rzcc_kernel":
            # latency-bound sequential stage: price it against HBM with its algorithmic bytes (8 B in per channel sample + 1 B spike out)
            achieved = frames_launch * (8 * C + C) / (st[dom] * 1e-3) / 1e9
            roof = dict(kernel=dom, bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                        traffic=args.traffic_bytes)
        else:
            achieved = frames_launch * flops[dom] / (st[dom] * 1e-3) / 1e12
            traffic = args.traffic_bytes
            if traffic is None and dom == "beamform_kernel" and (B, T, M, G) == (1100, 4799, 7, 360):
                traffic = float(BEAMFORM_TRAFFIC_BYTES_DEFAULT)
            roof = dict(kernel=KERNEL_SYMBOL.get(dom, dom), bound="mfma", achieved=achieved, peak=FP64_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                        frac=achieved / FP64_MFMA_PEAK_TFLOPS, traffic=traffic)
        roof["avg_launch_ms"] = st[dom]
        roof["stages_ms"] = st
        result = {
            "metric": "audio samples/sec through STHT+RZCC+SNN beamform, 7-mic 48kHz 360-DoA; DoA MAE vs ref",
            "value": value,
            "unit": "frames/s (one frame = one 7-mic audio sample instant)",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"target_snn_localization noisy sweep: 7-mic centre-circular, 48 kHz, T={T}, {B} trials/GPU/step (11 SNR x {B // 11}), "
                                   f"{G}-DoA grid, bipolar RZCC, bf_mat designed on device from the 1 s chirp",
                       "trials_per_gpu": B, "frames_per_trial": T, "num_mic": M, "num_doa": G, "mic_samples_per_s": value * M,
                       "parallelism": f"trial-sharded x{world}", "hip_streams": max(1, args.streams), "hip_graphs": True},
            "mae_deg_per_snr": [float(v) for v in (mae * 180 / np.pi).cpu().numpy()],
            "roofline": roof,
            # the north star also asks for the fraction of the HBM roofline: algorithmic bytes of the fused sweep
            # (one fp64 frame in, int8 spikes out and back in = 8M + 2M bytes per frame) over the whole job
            "hbm_fraction": {"bytes_per_frame": 10 * M, "achieved_GBs": value / world * 10 * M / 1e9, "peak_GBs": HBM_PEAK_GBS,
                             "frac_per_gpu": value / world * 10 * M / 1e9 / HBM_PEAK_GBS,
                             "note": "compute-bound path (about 300 flop/B): small by construction, the binding roof is in `roofline`"},
            "variants": {"covariance_power": cov_variant, "f32_mfma_beamform": f32_variant},
        }
        if not args.no_cpu_baseline and world == 1:
            cb, am_cpu = cpu_baseline(wl, min(args.cpu_sample, B))
            am_gpu = argmax_direct[: len(am_cpu)].cpu().numpy()
            cb["argmax_equal_to_gpu"] = bool(np.array_equal(am_cpu, am_gpu))
            result["cpu_baseline"] = cb
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio on stdout; flush it first so the JSON line is the last line
        import ctypes

        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
