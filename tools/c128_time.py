"""Time the complex Beamformer's contraction stage alone (config-2 shape: 1100 x 4799 frames, 7 mics).

python tools/c128_time.py [G] [B] [want_y] -- average launch time over 20 launches (HIP events on the launch stream),
algorithmic rate on 8 M G + 4 G flop per frame (micloc/beamformer.py:290 + the power) and, with y stored, the HBM write rate.
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
    want_y = len(sys.argv) > 3 and sys.argv[3] not in ("0", "")
    T, M = 4799, 7
    rng = np.random.default_rng(0)
    kernel = rng.standard_normal(480)
    kernel[::2] = 0.0
    b = np.array([1.0, 0, 0, 0, 0])
    a = np.array([1.0, 0, 0, 0, 0])
    plan = runtime.Plan(M, kernel, b, a, 1, False)
    plan.set_bf_mat(rng.standard_normal((M, G)) + 1j * rng.standard_normal((M, G)))
    Ts = plan.padded_T(T)
    pre = torch.randn((B, 2 * M, Ts), dtype=torch.float64, device="cuda")
    out = plan.beamform_c128(pre, T, want_y=want_y, want_power=True)
    for _ in range(3):
        plan.beamform_c128(pre, T, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        plan.beamform_c128(pre, T, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flop = (8 * M * G + 4 * G) * B * T
    extra = f", y store {B * T * G * 16 / ms / 1e9:.2f} TB/s" if want_y else ""
    print(f"c128 G={G} B={B} want_y={want_y}: {ms:.4f} ms/launch (incl. power_argmax), {flop / ms / 1e9:.1f} TFLOP/s algorithmic "
          f"= {flop / ms / 1e9 / 78.6:.3f} of 78.6, {B * T / ms / 1e6:.3f} G frames/s{extra}")


if __name__ == "__main__":
    main()
