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
              raster (micloc_stream_encode_tile_f64);
  LIF / beamforming / power   after every tile the device decides which frames can no longer receive a spike, filters and
              beamforms the 256-frame chunks that became final and adds their sum of y^2 to a persistent [B, G] accumulator in
              the order of the one-shot call's time reduction (micloc_stream_localize_tile_f64): power and arg-max after the last
              tile equal the one-shot call bit for bit; in between they are the running estimate the live loop wants.

The stream's CLOCK (frames pushed so far, base of the raster window) lives on the device as well: no launch of a tile carries an
absolute time, so a tile of a given length is one replayable hipGraph -- push_replay(), the reference's live loop
(micloc/localization_demo_snn.py:125-193) as one graph launch per 0.25 s frame.

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
        self.win = torch.empty((self.B, self.cap, self.C), dtype=torch.int8, device=dev)
        self.win_tmp = torch.empty_like(self.win)  # the slide's staging copy
        self.base = 0  # host mirror of the device clock's window base (the schedule depends on the tile sizes only)
        # tile workspace, allocated once: [history | tile] frames and their planar STHT output (+ one spare row, see _tile)
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
        self._seen, self._graphs = set(), {}
        # the stream's clock (frames pushed, window base) lives on the device: zeroed here with the states and the window
        _lib.check(self.lib.micloc_stream_reset(self.plan.handle, self.B, runtime._ptr(self.state), self.nstate, runtime._ptr(self.loc), self.nloc,
                                                runtime._ptr(self.win), self.cap, runtime._stream(dev)), "stream_reset")

    # ---- one tile -------------------------------------------------------------------------------------------------------
    def _check_tile(self, B, n, M, final):
        if self.done:
            raise _lib.MiclocError("the stream has ended")
        if B != self.B or M != self.M:
            raise ValueError(f"number of channels in the input siganl {M} should be the same as the number of microphones {self.M}!")
        if final is None:
            final = self.T is not None and self.t + n == self.T
        if n < 1 or n > self.max_tile or (not final and n % 16 != 0) or (self.T is not None and (self.t + n > self.T or (final and self.t + n != self.T))):
            raise ValueError("tiles must be multiples of 16 frames (except the last), at most max_tile long, and add up to total_frames")
        if self.t + n > 0x7FFFFFFF:
            raise ValueError("the stream's clock is a 32-bit frame counter")
        return bool(final)

    def _tile(self, x, n, final):
        """The launches of one tile (nothing else: no allocation, no synchronisation, no absolute time by value -- the clock is a
        device word), so that a tile of a given length is ONE replayable hipGraph (push_replay)."""
        lib, plan, B, M = self.lib, self.plan, self.B, self.M
        st = runtime._stream(self.device)
        # clock: t_end = t + n; the window slides forward (whole chunks) if it does not cover [.., t + n)
        _lib.check(lib.micloc_stream_begin_tile(plan.handle, runtime._ptr(self.loc), runtime._ptr(self.win), runtime._ptr(self.win_tmp), B, n, self.cap, st),
                   "stream_begin_tile")
        ext = self.ext[: B * (self.halo + n) * M].view(B, self.halo + n, M)  # contiguous [history | tile] of this tile length
        ext[:, : self.halo, :].copy_(self.hist)
        ext[:, self.halo :, :].copy_(x)
        Text = self.halo + n
        Ts = plan.padded_T(Text)
        h = self.h[: (B * self.C + 1) * Ts]  # one spare row: the encoder's loader may read up to `halo` elements past its last row
        _lib.check(lib.micloc_stht_f64(plan.handle, runtime._ptr(ext), B, Text, runtime._ptr(h), Ts, st), "stht")
        # np.roll's wrap-around: in-phase[t] = x[T - L/2 + t] for t < L/2 (zeros if the caller could not know them); a no-op later
        _lib.check(lib.micloc_stream_wrap_rows_f64(plan.handle, runtime._ptr(self.loc), runtime._ptr(h), B, Ts, self.halo, n, runtime._ptr(self.wrap), st),
                   "stream_wrap_rows")
        h_tile = ctypes.c_void_p(h.data_ptr() + 8 * self.halo)
        _lib.check(lib.micloc_stream_encode_tile_f64(plan.handle, h_tile, B, n, Ts, int(final), runtime._ptr(self.win), self.cap, runtime._ptr(self.state),
                                                     self.nstate, runtime._ptr(self.loc), st), "stream_encode_tile")
        _lib.check(lib.micloc_stream_localize_tile_f64(plan.handle, runtime._ptr(self.state), runtime._ptr(self.loc), self.nloc, runtime._ptr(self.win), B,
                                                       self.cap, int(final), runtime._ptr(self.power), runtime._ptr(self.argmax), runtime._ptr(self.ws),
                                                       self.nws, st), "stream_localize_tile")
        # history for the next tile's quadrature FIR: the last `halo` frames of [history | tile]
        self.hist.copy_(ext[:, Text - self.halo :, :])

    def _advance(self, n, final):
        """Host mirror of the device clock (rz_stream_clock_begin_kernel's schedule) and the end-of-stream bookkeeping."""
        if self.t + n > self.base + self.cap:
            self.base = -(-(self.t + n - self.cap) // self.CH) * self.CH
        self.t += n
        self.done = bool(final)

    def _before_slide(self, n):
        if self.raster is not None and self.t + n > self.base + self.cap:
            self._save_window(self.t)  # (keep_raster) the rows about to leave the window

    def push(self, x_tile, final=None):
        """x_tile [batch, n, M] (numpy or device tensor); n a multiple of 16 except for the last tile, n <= max_tile.
        final: this is the last tile (default: inferred from total_frames).  Returns the running (power, argmax) device
        tensors (overwritten by the next push; over the frames beamformed so far)."""
        x = self.plan.to_device(x_tile)
        B, n, M = x.shape
        final = self._check_tile(B, n, M, final)
        self._before_slide(n)
        self._tile(x, n, final)
        self._seen.add(n)
        self._advance(n, final)
        return self.power, self.argmax

    def push_replay(self, x_tile):
        """push() for the steady state of a live source (micloc/localization_demo_snn.py:125-193: one 0.25 s frame after the other):
        a non-final tile whose length has been pushed before is ONE hipGraph launch -- captured on its second occurrence, replayed from
        then on; the tile is copied into the graph's input buffer first.  Same results as push()."""
        torch = runtime._torch()
        x = self.plan.to_device(x_tile)
        B, n, M = x.shape
        if n not in self._seen or (self.T is not None and self.t + n == self.T):
            return self.push(x)  # first tile of this length (lazy kernel set-up must not happen inside a capture) / the final tile
        self._check_tile(B, n, M, False)
        g = self._graphs.get(n)
        if g is None:
            x_in = torch.empty((B, n, M), dtype=torch.float64, device=self.device)
            graph = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream(device=self.device)
            s.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.graph(graph, stream=s, capture_error_mode="thread_local"):
                self._tile(x_in, n, False)
            torch.cuda.current_stream(self.device).wait_stream(s)
            g = self._graphs[n] = (graph, x_in)
        self._before_slide(n)
        g[1].copy_(x)
        g[0].replay()
        self._advance(n, False)
        return self.power, self.argmax

    def _save_window(self, t_end):
        """(keep_raster) copy the window's frames [base, t_end) into the full raster: later copies carry more final data."""
        n = min(t_end, self.base + self.cap) - self.base
        if n > 0:
            self.raster[:, self.base : self.base + n, :].copy_(self.win[:, :n, :])

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
