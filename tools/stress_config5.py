#!/usr/bin/env python3
"""BASELINE config 5 (stress shape): 64-mic random planar array, 96 kHz (960-tap STHT, w = 24, 71-tap neuron kernel),
1440-DoA grid, T = 9599 frames.  Times the fused pipeline on one GPU for --trials trials (the full sweep is 16 384
trials over 8 GPUs = 2048 per GPU); parity of this shape against the CPU oracle is
tests/test_hip_fullsize.py::test_config5_shape_vs_oracle.  bf_mat: random unit-norm columns
(designing 1440 DoAs x 1 s x 64 mics is a separate, one-off cost and does not change the hot path's work)."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch

    from haghighatshoarmuir2024_amd.array_geometry import Random2DArray
    from haghighatshoarmuir2024_amd.snn_beamformer import SNNBeamformer, neuron_impulse_response

    fs, M, G = 96_000, 64, 1440
    np.random.seed(1)
    geometry = Random2DArray(radius=0.2, num_mic=M)
    tau = 1 / (2 * np.pi * 2000.0)
    beamf = SNNBeamformer(geometry, 10e-3, [1000.0, 2000.0], np.asarray([tau, tau]), bipolar_spikes=True, fs=fs)
    rng = np.random.RandomState(5)
    W = rng.randn(2 * M, G)
    W /= np.linalg.norm(W, axis=0, keepdims=True)
    time_test = np.arange(0, 100e-3, step=1 / fs)
    sig_test = np.sin(2 * np.pi * 2000 * time_test)
    doa = rng.rand(args.trials) * 2 * np.pi
    time_in, clean = beamf.synthesize_batch((time_test, sig_test), doa)
    gen = torch.Generator(device=clean.device)
    gen.manual_seed(7)
    x = (clean + 0.5 * torch.randn(clean.shape, generator=gen, device=clean.device, dtype=torch.float64)).contiguous()
    del clean
    B, T, _ = x.shape
    plan = beamf.plan()
    nir = neuron_impulse_response(time_in, beamf.tau_vec)
    plan.set_neuron_kernel(nir)
    plan.set_bf_mat(W)
    out = plan.snn_pipeline(x, want_spikes=True, want_power=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = plan.snn_pipeline(x, want_power=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    flops = 2 * len(nir) * 2 * M + 2 * 2 * M * G + 2 * G
    print(f"config 5: {B} trials x {T} frames x {M} mics, G={G}: {dt * 1e3:.1f} ms/step, {B * T / dt:.3e} frames/s, "
          f"beamform+LIF algorithmic {B * T * flops / dt / 1e12:.1f} TFLOP/s (whole pipeline time)")


if __name__ == "__main__":
    main()
