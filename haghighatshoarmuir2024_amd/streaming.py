"""A recording delivered tile by tile: SNNBeamformer.apply_to_signal's chain with exact state hand-off across tiles, localised
INCREMENTALLY -- a running power spectrum / DoA after every tile, O(tile) memory, no host synchronisation in push().

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
              (csrc/rzcc.hip "streaming"); spikes of a cluster are emitted when it closes, into a sliding WINDOW of the int8
              raster (micloc_stream_encode_window_f64);
  LIF / beamforming / power   after every tile the device decides which frames can no longer receive a spike, filters and
              beamforms the 256-frame chunks that became final and adds their sum of y^2 to a persistent [B, G] accumulator in
              the order of the one-shot call's time reduction (micloc_stream_localize_f64): power and arg-max after the last
              tile equal the one-shot call bit for bit; in between they are the running estimate the live loop wants.

Memory: the window (default: tile + 4096 frames, 14 B per frame and trial, twice), one tile of fp64 intermediates, 2 x G doubles per
trial.  push() allocates nothing and never synchronises (the ready range lives on the device); finish() / status() do.
"""
import ctypes

import numpy as np

from . import _lib, runtime
from .snn_beamformer import neuron_impulse_response


class StreamingLocalizer:
    def __init__(self, beamf, bf_mat, batch, total_frames=None, wrap_tail=None, max_tile=12_000, lag_frames=4096, keep_raster=False):
        """beamf: SNNBeamformer; bf_mat [2M, G]; `batch` recordings are streamed in lock step.
        total_frames  length of the recordings if known (the last tile is then recognised by itself, and the neuron kernel is
                      normalised over exactly that many samples like apply_to_signal does, snn_beamformer.py:342-361); None: a
                      live source -- pass final=True with the last tile; the kernel is normalised over 1 s (the sum has converged
                      to the last bit long before).
        wrap_tail     [batch, L // 2, M]: the last L // 2 frames of every recording (np.roll's wrap-around), or None (zeros).
        max_tile      longest tile push() will be given; lag_frames: how far spikes may trail the input (an open cluster holds
                      its frames back) before status() reports a lag failure.
        keep_raster   (tests) also assemble the full spike raster [batch, total_frames, 2M]; needs total_frames."""
        torch = runtime._torch()
        self.beamf = beamf
        self.plan = beamf.new_plan()
        self.device = self.plan.device
        self.B, self.M = int(batch), len(beamf.geometry)
        self.C = 2 * self.M
        self.T = None if total_frames is None else int(total_frames)
        self.L = len(beamf.kernel)
        self.halo = -(-(self.L - 1) // 8) * 8
        nir_frames = self.T if self.T is not None else int(beamf.fs)
        self.plan.set_neuron_kernel(neuron_impulse_response(np.arange(nir_frames) / beamf.fs, beamf.tau_vec))
        self.plan.set_bf_mat(np.asarray(bf_mat, dtype=np.float64))
        self.lib = _lib.load()
        self.G = self.plan.G
        self.CH = self.lib.micloc_stream_chunk_frames(self.plan.handle)
        if self.CH <= 0:
            _lib.check(self.CH, "stream_chunk_frames")
        self.max_tile = -(-int(max_tile) // 16) * 16
        # window: the tile being encoded + the frames that may still be waiting for their spikes + one chunk of LIF history
        self.cap = -(-(self.max_tile + int(lag_frames) + 2 * self.CH) // self.CH) * self.CH
        dev = self.device
        self.nstate = self.lib.micloc_stream_state_bytes(self.plan.handle, self.B)
        self.state = torch.empty(int(self.nstate), dtype=torch.uint8, device=dev)
        self.nloc = self.lib.micloc_stream_localize_state_bytes(self.plan.handle, self.B)
        self.loc = torch.empty(int(self.nloc), dtype=torch.uint8, device=dev)
        self.nws = self.lib.micloc_stream_localize_workspace_bytes(self.plan.handle, self.B, self.cap)
        self.ws = torch.empty(int(self.nws), dtype=torch.uint8, device=dev)
        self.win = [torch.empty((self.B, self.cap, self.C), dtype=torch.int8, device=dev) for _ in range(2)]
        self.cur = 0
        self.base = 0
        # tile workspace, allocated once: [history | tile] frames and their planar STHT output (+ one spare row, see push)
        self.hist = torch.zeros((self.B, self.halo, self.M), dtype=torch.float64, device=dev)  # zero history (lfilter's zero state)
        self.ext = torch.empty(self.B * (self.halo + self.max_tile) * self.M, dtype=torch.float64, device=dev)
        self.h = torch.empty((self.B * self.C + 1) * self.plan.padded_T(self.halo + self.max_tile), dtype=torch.float64, device=dev)
        self.power = torch.zeros((self.B, self.G), dtype=torch.float64, device=dev)
        self.argmax = torch.zeros((self.B,), dtype=torch.int32, device=dev)
        self.wrap = None
        if wrap_tail is not None:
            wrap_tail = self.plan.to_device(np.asarray(wrap_tail, dtype=np.float64) if isinstance(wrap_tail, np.ndarray) else wrap_tail)
            if tuple(wrap_tail.shape) != (self.B, self.L // 2, self.M):
                raise ValueError(f"wrap_tail must be [batch, {self.L // 2}, num_mic]")
            self.wrap = wrap_tail
        self.raster = None
        if keep_raster:
            if self.T is None:
                raise ValueError("keep_raster needs total_frames")
            self.raster = torch.zeros((self.B, self.T, self.C), dtype=torch.int8, device=dev)
        self.t = 0
        self.done = False

    # ---- one tile -------------------------------------------------------------------------------------------------------
    def push(self, x_tile, final=None):
        """x_tile [batch, n, M] (numpy or device tensor); n a multiple of 16 except for the last tile, n <= max_tile.
        final: this is the last tile (default: inferred from total_frames).  Returns the running (power, argmax) device
        tensors (overwritten by the next push; over the frames beamformed so far)."""
        if self.done:
            raise _lib.MiclocError("the stream has ended")
        x = self.plan.to_device(x_tile)
        B, n, M = x.shape
        if B != self.B or M != self.M:
            raise ValueError(f"number of channels in the input siganl {M} should be the same as the number of microphones {self.M}!")
        if final is None:
            final = self.T is not None and self.t + n == self.T
        if n < 1 or n > self.max_tile or (not final and n % 16 != 0) or (self.T is not None and (self.t + n > self.T or (final and self.t + n != self.T))):
            raise ValueError("tiles must be multiples of 16 frames (except the last), at most max_tile long, and add up to total_frames")
        st = runtime._stream(self.device)
        lib, plan = self.lib, self.plan
        # the window must cover [.., t + n): slide it forward (whole chunks) if it does not
        if self.t + n > self.base + self.cap:
            new_base = -(-(self.t + n - self.cap) // self.CH) * self.CH
            if self.raster is not None:
                self._save_window(self.t)
            src, dst = self.win[self.cur], self.win[1 - self.cur]
            _lib.check(lib.micloc_stream_window_shift(plan.handle, runtime._ptr(self.loc), runtime._ptr(src), runtime._ptr(dst), B, self.cap, self.base,
                                                      new_base, st), "stream_window_shift")
            self.cur = 1 - self.cur
            self.base = new_base
        ext = self.ext[: B * (self.halo + n) * M].view(B, self.halo + n, M)  # contiguous [history | tile] of this tile length
        ext[:, : self.halo, :].copy_(self.hist)
        ext[:, self.halo :, :].copy_(x)
        Text = self.halo + n
        Ts = plan.padded_T(Text)
        h = self.h[: (B * self.C + 1) * Ts]  # one spare row: the encoder's loader may read up to `halo` elements past its last row
        _lib.check(lib.micloc_stht_f64(plan.handle, runtime._ptr(ext), B, Text, runtime._ptr(h), Ts, st), "stht")
        hv = h[: B * self.C * Ts].view(B, self.C, Ts)
        if self.t < self.L // 2:
            # np.roll's wrap-around: in-phase[t] = x[T - L/2 + t] for t < L/2 (zeros if the caller could not know them)
            k = min(self.L // 2 - self.t, n)
            hv[:, :M, self.halo : self.halo + k] = self.wrap[:, self.t : self.t + k, :].transpose(1, 2) if self.wrap is not None else 0.0
        h_tile = ctypes.c_void_p(h.data_ptr() + 8 * self.halo)
        win = self.win[self.cur]
        first = int(self.t == 0)
        _lib.check(lib.micloc_stream_encode_window_f64(plan.handle, h_tile, B, n, Ts, self.t, first, int(final), runtime._ptr(win), self.cap, self.base,
                                                       runtime._ptr(self.state), self.nstate, st), "stream_encode_window")
        _lib.check(lib.micloc_stream_localize_f64(plan.handle, runtime._ptr(self.state), runtime._ptr(self.loc), self.nloc, runtime._ptr(win), B, self.cap,
                                                  self.base, self.t + n, first, int(final), runtime._ptr(self.power), runtime._ptr(self.argmax),
                                                  runtime._ptr(self.ws), self.nws, st), "stream_localize")
        # history for the next tile's quadrature FIR: the last `halo` frames of [history | tile]
        self.hist.copy_(ext[:, Text - self.halo :, :])
        self.t += n
        self.done = bool(final)
        return self.power, self.argmax

    def _save_window(self, t_end):
        """(keep_raster) copy the window's frames [base, t_end) into the full raster: later copies carry more final data."""
        n = min(t_end, self.base + self.cap) - self.base
        if n > 0:
            self.raster[:, self.base : self.base + n, :].copy_(self.win[self.cur][:, :n, :])

    # ---- results ----------------------------------------------------------------------------------------------------------
    def status(self):
        """dict(chunks, frames, lag_failures, overflow): synchronises the stream."""
        st4 = (ctypes.c_int * 4)()
        _lib.check(self.lib.micloc_stream_localize_status(runtime._ptr(self.loc), st4, runtime._stream(self.device)), "stream_localize_status")
        lost = ctypes.c_int(0)
        _lib.check(self.lib.micloc_stream_overflow(runtime._ptr(self.state), ctypes.byref(lost), runtime._stream(self.device)), "stream_overflow")
        return dict(chunks=int(st4[0]), frames=int(st4[1]), lag_failures=int(st4[2]), overflow=int(lost.value))

    def finish(self, want_spikes=False):
        """-> dict(power [B, G], argmax [B] int32, spikes [B, T, 2M] int8 (keep_raster only) or None) as device tensors."""
        if not self.done:
            raise _lib.MiclocError(f"the stream is incomplete: {self.t} frames pushed and no final tile")
        s = self.status()
        if s["overflow"]:
            raise _lib.MiclocError(f"{s['overflow']} stream(s) overflowed the candidate ring or the raster window (out-of-band input): use the "
                                   "one-shot call, which redoes such streams exactly")
        if s["lag_failures"] or s["frames"] != self.t:
            raise _lib.MiclocError(f"the raster window ({self.cap} frames) slid past frames whose spikes were not final yet "
                                   f"({s['frames']} of {self.t} frames beamformed): raise lag_frames")
        spikes = None
        if want_spikes:
            if self.raster is None:
                raise ValueError("spikes are only assembled with keep_raster=True")
            self._save_window(self.t)
            spikes = self.raster
        return dict(power=self.power, argmax=self.argmax, spikes=spikes)
