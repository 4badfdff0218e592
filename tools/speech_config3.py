#!/usr/bin/env python3
"""BASELINE config 3 shape on one GPU: the per-GPU share (125 trials) of the 1000-trial speech sweep, T = 332 157 frames
(6.9 s at 48 kHz), 7 mics, 360 DoAs.  Synthetic speech-like source (band-limited noise with a syllabic envelope; the
real FLAC source is exercised for parity in tests/test_speech_config.py).  With so few, so long streams the band-pass /
RZCC kernel is pure latency (T sequential steps on 28 workgroups, about 84 ns each) and dominates the step; splitting
the trials into sub-batches on several HIP streams was measured and is slower (60 ms against 36 ms), so the share runs
as one batch."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=125)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    import torch

    from haghighatshoarmuir2024_amd.array_geometry import CenterCircularArray
    from haghighatshoarmuir2024_amd.snn_beamformer import SNNBeamformer, neuron_impulse_response

    fs, M, G, T0 = 48_000, 7, 360, 332_158
    tau = 1 / (2 * np.pi * 2000.0)
    beamf = SNNBeamformer(CenterCircularArray(radius=4.5e-2, num_mic=M), 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=fs)
    rng = np.random.RandomState(3)
    W = rng.randn(2 * M, G)
    W /= np.linalg.norm(W, axis=0, keepdims=True)
    t = np.arange(T0) / fs
    src = rng.randn(T0)
    src = np.convolve(src, np.hanning(25) / 12, mode="same") * (0.6 + 0.4 * np.sin(2 * np.pi * 4 * t)) ** 2
    doa = rng.rand(args.trials) * 2 * np.pi
    time_in, clean = beamf.synthesize_batch((t, src), doa)
    gen = torch.Generator(device=clean.device)
    gen.manual_seed(11)
    x = (clean + 0.3 * torch.randn(clean.shape, generator=gen, device=clean.device, dtype=torch.float64)).contiguous()
    del clean
    T = x.shape[1]
    nir = neuron_impulse_response(time_in[:4800], beamf.tau_vec)
    plan = beamf.new_plan()
    plan.set_neuron_kernel(nir)
    plan.set_bf_mat(W)
    plan.snn_pipeline(x, want_power=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.snn_pipeline(x, want_power=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(f"config 3 (per-GPU share): {args.trials} trials x {T} frames x {M} mics, G={G}: {dt * 1e3:.1f} ms/step = "
          f"{args.trials * T / dt:.3e} frames/s ({T / fs / (dt / args.trials):.0f} x real time per trial stream)")


if __name__ == "__main__":
    main()
