"""A recording delivered tile by tile: SNNBeamformer.apply_to_signal's chain with exact state hand-off across tiles.

The reference has two ways of consuming audio: `apply_to_signal` on a whole recording (micloc/snn_beamformer.py:283-370) and
the live demo, which restarts the chain on every 0.25 s frame (micloc/localization_demo_snn.py:125-193; here:
localization_demo_snn.Demo.process_frame).  This module adds the third, which the reference lacks: the recording is ONE
stream but arrives in tiles, and the result is bit-identical to the one-shot call whatever the tiling --

  STHT        the quadrature FIR sees the last L - 1 frames of the previous tile (kept here), the in-phase channel is the input
              delayed by L / 2 frames; its first L / 2 frames are np.roll's wrap-around, i.e. the LAST L / 2 frames of the
              recording (snn_beamformer.py:325-326), which a caller that knows them passes as `wrap_tail` (offline tiling of
              a long file); a live source does not, and gets zeros there like any causal implementation would;
  band-pass   DF2T state carried in the device-side stream state;
  RZCC        running sum, detector state, open clusters (candidate ring) and selection cursors carried as well
              (csrc/rzcc.hip "streaming", micloc_stream_encode_f64); spikes of a cluster are emitted when it closes;
  LIF / beamforming / power   run over the finished int8 raster (14 B per frame stay resident; the fp64 intermediates,
              112 B per frame, exist for one tile at a time).
"""
import ctypes

import numpy as np

from . import _lib, runtime
from .snn_beamformer import neuron_impulse_response


class StreamingLocalizer:
    def __init__(self, beamf, bf_mat, batch, total_frames, wrap_tail=None):
        """beamf: SNNBeamformer; bf_mat [2M, G]; `batch` recordings of `total_frames` frames each are streamed in lock step.
        wrap_tail [batch, L // 2, M]: the last L // 2 frames of every recording (np.roll's wrap-around), or None (zeros)."""
        torch = runtime._torch()
        self.beamf = beamf
        self.plan = beamf.new_plan()
        self.device = self.plan.device
        self.B, self.T, self.M = int(batch), int(total_frames), len(beamf.geometry)
        self.L = len(beamf.kernel)
        self.halo = -(-(self.L - 1) // 8) * 8
        self.plan.set_neuron_kernel(neuron_impulse_response(np.arange(self.T) / beamf.fs, beamf.tau_vec))
        self.plan.set_bf_mat(np.asarray(bf_mat, dtype=np.float64))
        self.lib = _lib.load()
        self.nstate = self.lib.micloc_stream_state_bytes(self.plan.handle, self.B)
        self.state = torch.empty(int(self.nstate), dtype=torch.uint8, device=self.device)
        self.spikes = torch.empty((self.B, self.T, 2 * self.M), dtype=torch.int8, device=self.device)
        self.hist = torch.zeros((self.B, self.halo, self.M), dtype=torch.float64, device=self.device)  # zero history (lfilter)
        self.wrap = None
        if wrap_tail is not None:
            wrap_tail = self.plan.to_device(np.asarray(wrap_tail, dtype=np.float64) if isinstance(wrap_tail, np.ndarray) else wrap_tail)
            if tuple(wrap_tail.shape) != (self.B, self.L // 2, self.M):
                raise ValueError(f"wrap_tail must be [batch, {self.L // 2}, num_mic]")
            self.wrap = wrap_tail
        self.t = 0
        self.done = False

    def push(self, x_tile):
        """x_tile [batch, n, M] (numpy or device tensor); n must be a multiple of 16 except for the last tile."""
        torch = runtime._torch()
        if self.done:
            raise _lib.MiclocError("the stream has ended")
        x = self.plan.to_device(x_tile)
        B, n, M = x.shape
        if B != self.B or M != self.M:
            raise ValueError(f"number of channels in the input siganl {M} should be the same as the number of microphones {self.M}!")
        final = self.t + n == self.T
        if self.t + n > self.T or (not final and n % 16 != 0):
            raise ValueError("tiles must be multiples of 16 frames (except the last) and add up to total_frames")
        ext = torch.cat([self.hist, x], dim=1).contiguous()  # [B, halo + n, M]
        Text = ext.shape[1]
        Ts = self.plan.padded_T(Text)
        # one spare row: the encoder's loader may read up to `halo` elements past a row it was handed at an offset
        h = torch.empty((B * 2 * M + 1, Ts), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.micloc_stht_f64(self.plan.handle, runtime._ptr(ext), B, Text, runtime._ptr(h), Ts, runtime._stream(self.device)), "stht")
        hv = h[: B * 2 * M].view(B, 2 * M, Ts)
        if self.t < self.L // 2:
            # np.roll's wrap-around: in-phase[t] = x[T - L/2 + t] for t < L/2 (zeros if the caller could not know them)
            k = min(self.L // 2 - self.t, n)
            src = self.wrap[:, self.t : self.t + k, :].transpose(1, 2) if self.wrap is not None else 0.0
            hv[:, :M, self.halo : self.halo + k] = src
        h_tile = ctypes.c_void_p(h.data_ptr() + 8 * self.halo)
        _lib.check(self.lib.micloc_stream_encode_f64(self.plan.handle, h_tile, B, n, Ts, self.t, int(self.t == 0), int(final), runtime._ptr(self.spikes),
                                                     self.T, runtime._ptr(self.state), self.nstate, runtime._stream(self.device)), "stream_encode")
        self.hist = ext[:, Text - self.halo :, :].contiguous()
        self.t += n
        self.done = final

    def finish(self, want_spikes=False):
        """-> dict(power [B, G], argmax [B] int32, spikes [B, T, 2M] int8 or None) as device tensors."""
        torch = runtime._torch()
        if not self.done:
            raise _lib.MiclocError(f"the stream is incomplete: {self.t} of {self.T} frames pushed")
        lost = ctypes.c_int(0)
        _lib.check(self.lib.micloc_stream_overflow(runtime._ptr(self.state), ctypes.byref(lost), runtime._stream(self.device)), "stream_overflow")
        if lost.value:
            raise _lib.MiclocError(f"{lost.value} stream(s) overflowed the candidate ring (out-of-band input): use the one-shot call, "
                                   "which redoes such streams exactly")
        G = self.plan.G
        power = torch.empty((self.B, G), dtype=torch.float64, device=self.device)
        argmax = torch.empty((self.B,), dtype=torch.int32, device=self.device)
        nbytes = self.lib.micloc_lif_beamform_workspace_bytes(self.plan.handle, self.B, self.T)  # partial sums only
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.micloc_lif_beamform_f64(self.plan.handle, runtime._ptr(self.spikes), self.B, self.T, None, runtime._ptr(power),
                                                    runtime._ptr(argmax), runtime._ptr(ws), nbytes, runtime._stream(self.device)), "lif_beamform")
        return dict(power=power, argmax=argmax, spikes=self.spikes if want_spikes else None)
