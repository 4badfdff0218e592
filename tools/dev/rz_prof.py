"""Per-wave work / barrier-wait cycles of the encoder kernel (needs tools/_variants/libmicloc_hip_prof.so, see
make_rz_prof_variant.py).  usage: python tools/dev/rz_prof.py noisy|xylo [chunk] [B]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from haghighatshoarmuir2024_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "_variants", "libmicloc_hip_prof%s.so" % os.environ.get("RZ_VARIANT", ""))
import torch
from haghighatshoarmuir2024_amd import runtime, synthesis
from haghighatshoarmuir2024_amd.array_geometry import CenterCircularArray
from haghighatshoarmuir2024_amd.xylo_snn_localization import Demo
from haghighatshoarmuir2024_amd.snn_beamformer import SNNBeamformer

cfg = sys.argv[1] if len(sys.argv) > 1 else "xylo"
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else -1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1100
dev = torch.device("cuda", 0)
fs, M, G = 48000, 7, 360
geometry = CenterCircularArray(radius=4.5e-2, num_mic=M)
rng = np.random.RandomState(0)
doa = rng.rand(B) * 2 * np.pi
snr_vec = np.linspace(-10, 20, 11)
snr_db = snr_vec[(np.arange(B) * 11) // B] - 10 * np.log10(24)
if cfg == "xylo":
    demo = Demo(geometry=geometry, freq_bands=[[1000.0, 2000.0]], doa_list=np.linspace(-np.pi, np.pi, G), recording_duration=0.25, bipolar_spikes=True, fs=fs, device=dev)
    t = np.arange(0, 1.0, 1 / fs)
    s = np.sin(2 * np.pi * np.cumsum(1000 + 1000 * (t % t[-1]) / t[-1]) / fs)
    x = synthesis.signal_from_template_batch(geometry, (t, s), doa, device=dev, device_delays=True)
    enc = demo.beamfs[0].spk_encoder
    bb, aa = demo.filterbank.ba_list[0]
    plan = runtime.Plan(M, demo.beamfs[0].kernel, bb, aa, enc.robust_width, enc.bipolar, device=dev)
else:
    tau = 1 / (2 * np.pi * 2000.0)
    beamf = SNNBeamformer(geometry=geometry, kernel_duration=10e-3, tau_vec=np.asarray([tau, tau]), freq_range=[1000.0, 2000.0], fs=fs, bipolar_spikes=True, device=dev)
    t = np.arange(0, 100e-3, 1 / fs)
    _, x = beamf.synthesize_batch((t, np.sin(2 * np.pi * 2000 * t)), doa)
    plan = beamf.plan()
synthesis.add_noise_(x, snr_db, seed=4321, first_trial=0)
T = x.shape[1]
hq = plan.stht(x)
plan.set_encoder_chunk(chunk)
lib = _lib.load()
lib.micloc_debug_rz_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * 24)()
plan.bandpass_rzcc(hq, T, want_pre=False, want_spikes=True)
torch.cuda.synchronize()
lib.micloc_debug_rz_prof(buf, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
plan.bandpass_rzcc(hq, T, want_pre=False, want_spikes=True)
e1.record()
torch.cuda.synchronize()
lib.micloc_debug_rz_prof(buf, 1)
v = np.array(list(buf), dtype=np.float64).reshape(3, 8)
P = plan.encoder_chunks(B, T)
ntile = (T + 15) // 16
print(f"{cfg} B={B} T={T} chunk={chunk} P={P}: stage {e0.elapsed_time(e1):.3f} ms (incl. zero fill, scan, fallback)")
names = ["loader0", "loader1", "filter", "detect", "select+", "select-", "writer", "resolver"]
for w in range(8):
    if v[2, w] == 0:
        continue
    n = v[2, w]
    # clock64 = s_memtime: 100 MHz constant clock on gfx9? report raw ticks per tile and the work share
    print(f"  wave {w} {names[w]:8s}: waves {int(n):6d}  work {v[0, w] / n / (ntile / P):9.1f} ticks/tile  wait {v[1, w] / n / (ntile / P):9.1f} ticks/tile  work share {v[0, w] / (v[0, w] + v[1, w]):.2f}")
