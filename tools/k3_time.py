"""Time the LIF + beamforming + power stage alone (config 2 shape, synthetic ternary spikes).

python tools/k3_time.py [G] [B] — prints the average launch time over 20 launches (HIP events on the launch stream).
Used for kernel ablations; the spike content does not change the MFMA work.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from haghighatshoarmuir2024_amd import runtime  # noqa: E402


def main():
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 360
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1100
    T, M = 4799, 7
    rng = np.random.default_rng(0)
    kernel = rng.standard_normal(480)
    kernel[::2] = 0.0
    b = np.array([1.0, 0, 0, 0, 0, 0, 0, 0, 0])
    a = np.array([1.0, 0, 0, 0, 0, 0, 0, 0, 0])
    plan = runtime.Plan(M, kernel, b, a, 8, True)
    plan.set_neuron_kernel(rng.standard_normal(35))
    plan.set_bf_mat(rng.standard_normal((2 * M, G)))
    spikes = torch.from_numpy((rng.random((B, T, 2 * M)) < 0.1).astype(np.int8) * rng.choice([-1, 1], size=(B, T, 2 * M)).astype(np.int8)).cuda()
    for _ in range(3):
        plan.lif_beamform(spikes)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        plan.lif_beamform(spikes)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flop = (2 * 35 * 14 + 2 * 14 * G + 2 * G) * B * T
    print(f"G={G} B={B}: {ms:.4f} ms/launch (incl. power_argmax), {flop / ms / 1e9:.1f} TFLOP/s algorithmic, env={ {k: v for k, v in os.environ.items() if k.startswith('MICLOC_')} }")


if __name__ == "__main__":
    main()
