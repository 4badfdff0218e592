"""Time the LIF + beamforming + power stage alone (config 2 shape, synthetic ternary spikes).

python tools/k3_time.py [G] [B] [want_y] — prints the average launch time over 20 launches (HIP events on the launch
stream).  Used for kernel ablations; the spike content does not change the MFMA work.
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
    b = np.array([1.0, 0, 0, 0, 0, 0, 0, 0, 0])
    a = np.array([1.0, 0, 0, 0, 0, 0, 0, 0, 0])
    plan = runtime.Plan(M, kernel, b, a, 8, True)
    plan.set_neuron_kernel(rng.standard_normal(35))
    plan.set_bf_mat(rng.standard_normal((2 * M, G)))
    spikes = torch.from_numpy((rng.random((B, T, 2 * M)) < 0.1).astype(np.int8) * rng.choice([-1, 1], size=(B, T, 2 * M)).astype(np.int8)).cuda()
    y = torch.empty((B, T, G), dtype=torch.float64, device="cuda") if want_y else None

    def run():
        if not want_y:
            return plan.lif_beamform(spikes)
        # reuse one output buffer: time the kernel, not the allocator
        import ctypes
        from haghighatshoarmuir2024_amd import _lib
        ws, nbytes = plan.workspace(B, T)
        power = torch.empty((B, G), dtype=torch.float64, device="cuda")
        argmax = torch.empty((B,), dtype=torch.int32, device="cuda")
        _lib.check(plan.lib.micloc_lif_beamform_f64(plan.handle, runtime._ptr(spikes), B, T, runtime._ptr(y), runtime._ptr(power), runtime._ptr(argmax),
                                                    runtime._ptr(ws), nbytes, runtime._stream(plan.device)), "lif_beamform")

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flop = (2 * 35 * 14 + 2 * 14 * G + 2 * G) * B * T
    extra = f", y store {B * T * G * 8 / ms / 1e9:.2f} TB/s" if want_y else ""
    print(f"G={G} B={B} want_y={want_y}: {ms:.4f} ms/launch (incl. power_argmax), {flop / ms / 1e9:.1f} TFLOP/s algorithmic, {B * T / ms / 1e6:.3f} G frames/s{extra}")


if __name__ == "__main__":
    main()
