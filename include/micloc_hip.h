/*
 * micloc_hip.h -- C-ABI of libmicloc_hip.so, the MI355X (gfx950) implementation of the micloc
 * hot path (STHT -> band-pass -> RZCC spike encoding -> alpha-kernel "LIF" filter -> beamforming ->
 * power / arg-max) of synsense/HaghighatshoarMuir2024.
 *
 * The reference has no FFI: its boundary is the Python class surface in micloc/.  Each entry point
 * below therefore names the reference *method lines* it replaces (paths relative to the reference
 * repository root); INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / HIP types in the signatures (`stream` is a
 *     hipStream_t passed as void*; NULL = the default stream).
 *   - Pointers documented "host" are read synchronously during the call; everything else is a
 *     caller-owned DEVICE buffer on the plan's device.  No allocation happens in the stage calls:
 *     scratch comes from the caller (`ws`, sized by micloc_workspace_bytes).
 *   - All stage calls are asynchronous on `stream`, re-entrant per plan+stream, and safe to capture
 *     in a hipGraph.  Return value: MICLOC_OK (0) or a negative micloc_status.
 *   - A plan's calls launch on the plan's device; the entry points without a plan (synthesis, random
 *     numbers, DoA error, design vectors, peak location, Xylo LIF, stand-alone operators) launch on the
 *     device that OWNS their output / workspace buffer -- in both cases whatever the caller's current
 *     device is (the previous current device is restored before the call returns).
 *   - All arithmetic is IEEE binary64 unless a name says otherwise (float32 before the
 *     encoder flips spikes).
 *   - PRECONDITION: recordings are FINITE (no Inf / NaN samples).  The bit-exactness contract (time chunking,
 *     streaming and the one-shot call give the same spikes) rests on it: the checkpoint scan of a chunked encoder
 *     skips the products with a band-pass numerator coefficient that is exactly zero (fma(0, x, z) == z for finite
 *     x, up to the sign of a zero that no comparison, sum or spike sees), which the one-pass encoder multiplies --
 *     0 * Inf would be NaN there.  The reference's own chain is NaN-poisoned from the first non-finite sample on
 *     (lfilter -> cumsum -> find_peaks), so there is no result to reproduce; signed zeros and subnormals are fine.
 *
 * Device layouts
 *   x        [B][T][M]      row-major: trial b is the reference's `sig_in_vec` (T x num_mic).
 *   planar   [B][C][Ts]     C = 2M channels ordered re_0..re_{M-1}, im_0..im_{M-1}
 *                           (np.hstack([real, imag]), snn_beamformer.py:335); Ts = micloc_padded_T(T).
 *   spikes   [B][T][C]      int8 in {-1,0,+1}: trial b is the reference's spikes_vec (T x 2M).
 *   y        [B][T][G]      trial b is the reference's return value of apply_to_signal (T x num_grid);
 *                           complex variant: [B][T][G][2] (re, im) == numpy complex128.
 *   power    [B][G]         mean_t |y|^2  (target_snn_localization.py:462);  argmax [B] int32.
 */
#ifndef MICLOC_HIP_H
#define MICLOC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MICLOC_ABI_VERSION 1
#define MICLOC_MAX_IIR 9 /* len(b) = len(a) <= 9, i.e. band-pass order <= 4 */

typedef enum {
    MICLOC_OK = 0,
    MICLOC_ERR_INVALID = -1,     /* bad argument (NULL, non-positive size, unsupported length) */
    MICLOC_ERR_SHAPE = -2,       /* channel count / matrix shape mismatch (the reference's ValueError) */
    MICLOC_ERR_WORKSPACE = -3,   /* ws too small or misaligned */
    MICLOC_ERR_NOT_SET = -4,     /* neuron kernel / bf_mat not set on the plan */
    MICLOC_ERR_HIP = -5,         /* a HIP runtime call failed; see micloc_last_hip_error */
    MICLOC_ERR_NO_DEVICE = -6    /* device ordinal out of range, or the device is not a gfx950 (checked in plan_create) */
} micloc_status;

typedef struct micloc_plan micloc_plan;

/* Parameters fixed at construction time of SNNBeamformer / Beamformer (host pointers, copied). */
typedef struct {
    int device;                 /* HIP device ordinal */
    int num_mic;                /* M */
    int stht_len;               /* L = int(fs * kernel_duration)            snn_beamformer.py:50 */
    const double *stht_kernel;  /* host [L] fftshift(imag(hilbert(delta)))  snn_beamformer.py:51-53 */
    int iir_len;                /* n = len(b) = len(a) (zero-padded), 1..MICLOC_MAX_IIR */
    const double *iir_b;        /* host [n]  scipy.signal.butter(...)[0]    snn_beamformer.py:68-72 */
    const double *iir_a;        /* host [n]  a[0] != 0 (normalised inside) */
    int robust_width;           /* find_peaks(distance=...)                 snn_beamformer.py:75-80 */
    int bipolar;                /* 0: +1 spikes only, 1: +1 / -1            spike_encoder.py:131-135 */
} micloc_config;

/* ---- plan life cycle -------------------------------------------------------------------------- */
int micloc_plan_create(const micloc_config *cfg, micloc_plan **out);
void micloc_plan_destroy(micloc_plan *plan);

/* neuron impulse response (host [n]); replaces the kernel built in snn_beamformer.py:342-361 */
int micloc_plan_set_neuron_kernel(micloc_plan *plan, const double *nir, int n);
/* real beamforming matrix (host, row-major [C][G], C must equal 2M); SNNBeamformer bf_mat */
int micloc_plan_set_bf_mat(micloc_plan *plan, const double *W, int C, int G);
/* complex beamforming matrix (host, row-major [M][G] re and im); Beamformer bf_mat, applied as
 * sig @ bf_mat.conj() (beamformer.py:290) */
int micloc_plan_set_bf_mat_c128(micloc_plan *plan, const double *Wre, const double *Wim, int M, int G);

/* Counter that changes whenever a hipGraph captured from this plan's stage calls has become stale: a device table was
 * RE-ALLOCATED (set_neuron_kernel / set_bf_mat with a table larger than any before: the graph holds the raw pointer) or
 * replaced by a table of ANOTHER SHAPE (other channel / DoA count, real vs complex, other neuron-kernel length: the graph
 * holds those dimensions by value).  Re-capture (or refuse to replay) when the value differs from the one read at capture
 * time.  A table of the same shape is overwritten in place after a device synchronisation and leaves the counter -- and
 * captured graphs -- valid. */
int micloc_plan_generation(const micloc_plan *plan);

/* Time chunking of the band-pass / RZCC stage.  The stage is serial in time per (trial, channel) stream; with few, long
 * streams it is cut into chunks of `chunk_frames` owned frames that run side by side, each restarted from the exact
 * state a serial scan stored at its boundary -- results are bit-identical for every setting.  0 (default): automatic
 * (chunks only when the launch would not fill the chip); < 0: never; > 0: this many frames per chunk.  Changes the
 * workspace size: query micloc_workspace_bytes again.  micloc_plan_encoder_chunks reports the chunks per stream a
 * (B, T) launch will use. */
int micloc_plan_set_encoder_chunk(micloc_plan *plan, int chunk_frames);
int micloc_plan_encoder_chunks(const micloc_plan *plan, int B, int T);

/* A HIP stream whose kernels run only on compute units [cu_lo, cu_hi) of EVERY XCD of `device` (MI355X: 8 XCDs of 32 compute
 * units; hipExtStreamCreateWithCUMask with the mask layout of gfx950).  For callers that keep several batches in flight: the serial
 * scan of a chunked encoder (MICLOC_STAGE_ENCODE_SCAN below) is a latency chain of few workgroups that runs at a fraction of its
 * pace when its SIMD also issues another kernel's matrix instructions; on a stream of its own with e.g. [0, 4), beside work streams
 * restricted to [4, 32), consecutive batches' scans run one after the other at full pace while the other stages overlap them
 * (runtime.StreamPipeline(scan_lane=...)).  Scheduling only: results never depend on where a kernel runs.  The reference has no
 * counterpart (one NumPy process per trial, target_snn_localization.py:447-467). */
int micloc_stream_create_cu_range(int device, int cu_lo, int cu_hi, void **stream);
int micloc_stream_destroy(void *stream);

/* padded time stride of planar buffers (multiple of 8 samples) */
int micloc_padded_T(int T);
/* bytes of scratch needed by any stage / pipeline call with this (B, T) */
size_t micloc_workspace_bytes(const micloc_plan *plan, int B, int T);

/* ---- stages (device pointers) ----------------------------------------------------------------- */
/* STHT: roll by L/2 + 1j * FIR.  Replaces snn_beamformer.py:325-327 / :158-160,
 * beamformer.py:281-283, xylo_snn_localization.py:329-331.   x [B][T][M] -> h planar [B][2M][Ts].
 * The FIR is the chain acc = fma(ker[k], x[t - k], acc) over the non-zero taps in ascending k, from +0.  When every second tap
 * is zero (every even-length Hilbert kernel) it runs as a Toeplitz product on the fp64 matrix cores -- the same chain, the same
 * bits -- otherwise (or with MICLOC_STHT_VALU=1 in the environment, for A/B timing) on the vector ALU. */
int micloc_stht_f64(const micloc_plan *plan, const double *x, int B, int T, double *h, int Ts, void *stream);

/* Band-pass (DF2T, lfilter(b,a,.)) + RZCC encoder.  Replaces snn_beamformer.py:330-338 and
 * spike_encoder.py:115-137.  h planar in; `pre` (planar, post-filter signal) and `spikes` may each be
 * NULL.  spikes must not alias anything; it is fully overwritten. */
int micloc_bandpass_rzcc_f64(const micloc_plan *plan, const double *h, int B, int T, int Ts, double *pre,
                             int8_t *spikes, void *ws, size_t ws_bytes, void *stream);

/* alpha-kernel FIR ("LIF") + real beamforming + power/argmax.  Replaces snn_beamformer.py:364-368 and
 * target_snn_localization.py:462-464.  y, power, argmax may each be NULL. */
int micloc_lif_beamform_f64(const micloc_plan *plan, const int8_t *spikes, int B, int T, double *y,
                            double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream);

/* complex beamforming of the band-passed analytic signal: pre (planar) @ conj(W).
 * Replaces beamformer.py:290.  y is [B][T][G][2]. */
int micloc_beamform_c128_f64(const micloc_plan *plan, const double *pre, int B, int T, int Ts, double *y,
                             double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream);

/* ---- fused pipelines -------------------------------------------------------------------------- */
/* SNNBeamformer.apply_to_signal (snn_beamformer.py:283-370) + power/argmax for B trials.
 * spikes / y / power / argmax may each be NULL (at least one must be given). */
int micloc_snn_pipeline_f64(const micloc_plan *plan, const double *x, int B, int T, int8_t *spikes, double *y,
                            double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream);
/* The same pipeline, launched in parts: `stages` selects which of its three kernel groups this call enqueues.  The
 * intermediate results live in the workspace (and in `spikes` when given), so the parts of one batch must use the same
 * workspace, in order; a caller may enqueue them on different streams with its own event dependencies -- e.g. the STHT
 * of batch i+1 next to the latency-bound band-pass/RZCC of batch i. */
#define MICLOC_STAGE_STHT 1     /* snn_beamformer.py:325-327 (quadrature FIR)                    */
#define MICLOC_STAGE_ENCODE 2   /* snn_beamformer.py:330-338 (band-pass, re/im stack, RZCC)      */
#define MICLOC_STAGE_BEAMFORM 4 /* snn_beamformer.py:342-368 (LIF, beamforming) + power/arg-max  */
#define MICLOC_STAGE_ALL 7
/* MICLOC_STAGE_ENCODE in two parts.  Long recordings are encoded in time chunks that restart from checkpoints a serial scan of
 * every stream stores first (micloc_plan_set_encoder_chunk): few workgroups, one dependent chain each -- latency, not throughput,
 * and a chain that shares its SIMD with another kernel's waves runs at a fraction of its pace.  A caller with several batches in
 * flight can give the scans a stream (and compute units) of their own: _SCAN enqueues only the scan (nothing when the launch is not
 * chunked), _REST everything else of the stage; same workspace, scan first.  MICLOC_STAGE_ENCODE == both. */
#define MICLOC_STAGE_ENCODE_SCAN 8
#define MICLOC_STAGE_ENCODE_REST 16
int micloc_snn_pipeline_stages_f64(const micloc_plan *plan, const double *x, int B, int T, int8_t *spikes, double *y,
                                   double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream, int stages);
/* Beamformer.apply_to_signal (beamformer.py:260-292) + power/argmax for B trials. */
int micloc_beamformer_pipeline_f64(const micloc_plan *plan, const double *x, int B, int T, double *y,
                                   double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream);

/* ---- stand-alone operators (no plan) ---------------------------------------------------------- */
/* ZeroCrossingSpikeEncoder.evolve (spike_encoder.py:115-137) on row-major sig [B][T][C]. */
size_t micloc_rzcc_workspace_bytes(int B, int T, int C);
int micloc_rzcc_encode_f64(const double *sig, int B, int T, int C, int robust_width, int bipolar, int8_t *spikes,
                           void *ws, size_t ws_bytes, void *stream);
/* the same with an explicit time-chunking choice (see micloc_plan_set_encoder_chunk) */
size_t micloc_rzcc_workspace_bytes_ex(int B, int T, int C, int robust_width, int chunk_frames);
int micloc_rzcc_encode_ex_f64(const double *sig, int B, int T, int C, int robust_width, int bipolar, int chunk_frames,
                              int8_t *spikes, void *ws, size_t ws_bytes, void *stream);
/* scipy.signal.lfilter(b, a, x, axis=0) on row-major x [B][T][C] (Filterbank.evolve,
 * filterbank.py:25-46); b, a are host [n], zero-padded to the same length n <= MICLOC_MAX_IIR. */
size_t micloc_lfilter_workspace_bytes(int B, int T, int C);
int micloc_lfilter_f64(const double *b, const double *a, int n, const double *x, int B, int T, int C, double *y,
                       void *ws, size_t ws_bytes, void *stream);

/* ---- fp32-MFMA variant of the beamforming tail ------------------------------------------------------ */
/* Same contract as micloc_lif_beamform_f64 (power / arg-max only, real bf_mat, up to 64 channels) with the LIF filter
 * and the contraction on v_mfma_f32_16x16x4_f32: the spikes are exact in fp32, the power agrees with the fp64 path to
 * ~1e-6 relative (BASELINE's north star allows 1e-5 float32 for this stage).  A VARIANT, reported separately. */
int micloc_lif_beamform_f32(const micloc_plan *plan, const int8_t *spikes, int B, int T, double *power, int32_t *argmax,
                            void *ws, size_t ws_bytes, void *stream);

/* ---- covariance form ------------------------------------------------------------------ */
/* cov[b] = V^T V / (T - t_start) over frames t >= t_start of the membrane signal V = lfilter(nir,[1],spikes)
 * (the matrix design_from_template needs, snn_beamformer.py:176-191, with t_start = T // 4), and
 * power[b][g] = w_g^T (V^T V / T') w_g  ==  mean_t (V @ bf_mat)^2  without forming T x G.  An algebraically
 * identical VARIANT of micloc_lif_beamform_f64's power (2 C^2 instead of 2 C G flops per frame); it agrees to
 * ~1e-15 relative but is reported separately.  cov [B][C][C], power [B][G], argmax [B] may each be NULL.
 * Supports C <= 128 channels (lif_cov_kernel up to 64, lif_cov_wide_kernel beyond). */
int micloc_lif_covariance_f64(const micloc_plan *plan, const int8_t *spikes, int B, int T, int t_start, double *cov,
                              double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream);
/* Gram matrix of a planar signal: gram[b] = sum_{t >= t_start} x[b][:, t] x[b][:, t]^T (divided by T - t_start when `normalise`),
 * x planar [B][C][Ts] (C <= 128), gram [B][C][C].  Beamformer.design_from_template (beamformer.py:142-150) takes the complex
 * covariance conj(h)^T h / T' of the STHT output over the stable part of each delayed template; with the planar rows
 * [re_0..re_{M-1}, im_0..im_{M-1}] it is the fold (R_rr + R_ii) + j (R_ri - R_ir) of this real 2M x 2M matrix.  fp64 MFMA,
 * deterministic (fixed-order chunk sums).  ws from micloc_planar_gram_workspace_bytes. */
size_t micloc_planar_gram_workspace_bytes(int B, int T, int C, int t_start);
int micloc_planar_gram_f64(const double *planar, int B, int C, int T, int Ts, int t_start, int normalise, double *gram, void *ws,
                           size_t ws_bytes, void *stream);
/* whole chain (STHT, band-pass, RZCC, LIF) with the covariance-form tail */
int micloc_snn_pipeline_cov_f64(const micloc_plan *plan, const double *x, int B, int T, int t_start, int8_t *spikes,
                                double *cov, double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream);

/* ---- streaming: a recording delivered tile by tile, localised while it arrives ----------------------------------- */
/* apply_to_signal (micloc/snn_beamformer.py:283-370) treats a recording as ONE stream; the reference's live loop restarts the chain
 * on every 0.25 s frame instead (micloc/localization_demo_snn.py:125-193).  Here the recording is one stream that arrives in tiles:
 *   - the band-pass / RZCC stage is a resumable machine: a tile of T_tile frames per call, the complete state of every stream (DF2T
 *     registers, running sum, detector state, the candidate ring with its open clusters and selection cursors) kept in `enc_state`
 *     (micloc_stream_state_bytes, 256-B aligned), so that the tiles together perform exactly the operations of one call on the whole
 *     recording: the spikes are bit-identical for any tiling (spike_encoder.py:115-137); tile lengths are multiples of 16 except the
 *     last, final_tile closes the clusters still open.  A stream whose candidate ring overflows (> 63 pending candidates: out-of-band
 *     input) cannot be redone from its start; micloc_stream_overflow returns how many were lost (synchronises) -- a non-zero count
 *     is an error;
 *   - the spikes go into a sliding WINDOW of the int8 raster ([B][window_frames][2M] over the absolute frames [base, base +
 *     window_frames), multiples of micloc_stream_chunk_frames), and after every tile the DEVICE decides which frames can no longer
 *     receive a spike (the earliest position an open cluster or an incomplete candidate of any stream can still emit), runs LIF +
 *     beamforming on the chunks that became final and adds their sums of y^2 to an accumulator in `loc_state` in the order of the
 *     one-shot time reduction: power / argmax (may be NULL) are the running estimate over the frames beamformed so far and, after
 *     the last tile, the one-shot result bit for bit;
 *   - the CLOCK of the stream (frames pushed, base of the window) lives in the control words of `loc_state`, on the device: no call
 *     takes an absolute time, every launch of a tile of n frames has the same arguments, so a tile is ONE capturable hipGraph.
 * One tile of n frames =
 *   micloc_stream_begin_tile        clock: t_end = t + n; the window slides forward by whole chunks when it must (through
 *                                   scratch_window; a slide past rows the next chunk still needs is counted as a lag failure)
 *   micloc_stht_f64                 on [history | tile] (the last L - 1 frames of the previous tile in front), by the caller
 *   micloc_stream_wrap_rows_f64     np.roll's wrap-around rows of the in-phase channels while t < L/2 (wrap_tail [B][L/2][M] or NULL:
 *                                   zeros -- a live source cannot know them; the tile's frame i sits at column first_col + i of
 *                                   h [B][2M][Ts])
 *   micloc_stream_encode_tile_f64   h = the tile's first column (row stride row_stride), spikes into the window
 *   micloc_stream_localize_tile_f64 horizon, LIF + beamforming of the final chunks, accumulation; ends with the clock's tick
 * micloc_stream_reset zero-fills the clock, both states and the window (once, before the first tile).  The last tile of a recording
 * is issued with final_tile = 1 (both calls).  micloc_stream_localize_status: {chunks beamformed, frames beamformed, lag failures,
 * open block fill}; synchronises the stream.  ws: micloc_stream_localize_workspace_bytes(B, window_frames). */
size_t micloc_lif_beamform_workspace_bytes(const micloc_plan *plan, int B, int T); /* ws of micloc_lif_beamform_f64 / micloc_beamform_c128_f64 alone (bf_mat set) */
size_t micloc_stream_state_bytes(const micloc_plan *plan, int B);
size_t micloc_stream_localize_state_bytes(const micloc_plan *plan, int B);
size_t micloc_stream_localize_workspace_bytes(const micloc_plan *plan, int B, int window_frames);
int micloc_stream_chunk_frames(const micloc_plan *plan);
int micloc_stream_reset(const micloc_plan *plan, int B, void *enc_state, size_t enc_bytes, void *loc_state, size_t loc_bytes, int8_t *window,
                        int window_frames, void *stream);
int micloc_stream_begin_tile(const micloc_plan *plan, void *loc_state, int8_t *window, int8_t *scratch_window, int B, int n, int window_frames,
                             void *stream);
int micloc_stream_wrap_rows_f64(const micloc_plan *plan, const void *loc_state, double *h, int B, int Ts, int first_col, int n,
                                const double *wrap_tail, void *stream);
int micloc_stream_encode_tile_f64(const micloc_plan *plan, const double *h, int B, int T_tile, int row_stride, int final_tile, int8_t *window,
                                  int window_frames, void *enc_state, size_t enc_bytes, const void *loc_state, void *stream);
int micloc_stream_localize_tile_f64(const micloc_plan *plan, const void *enc_state, void *loc_state, size_t loc_bytes, const int8_t *window, int B,
                                    int window_frames, int final_tile, double *power, int32_t *argmax, void *ws, size_t ws_bytes, void *stream);
int micloc_stream_overflow(const void *state, int *count, void *stream);
int micloc_stream_localize_status(const void *loc_state, int *status4, void *stream);

/* ---- beamforming vectors from membrane covariances (design_from_template's decomposition step) ------ */
/* Replaces the per-DoA np.linalg.svd calls of SNNBeamformer.design_from_template (snn_beamformer.py:183-203) and
 * _find_dc_removed_sing_vec (:372-422) by one batched kernel: column g0 + i of bf_mat [C][G] from cov[i] [C][C]
 * (micloc_lif_covariance_f64's output).  bipolar = 0: the DC-removed conditional singular vector (secular equation solved by
 * the reference's bisection to rel_prec; the result does not depend on the signs of the singular vectors).  bipolar = 1: the
 * leading left singular vector of the folded complex covariance, stacked [Re; Im]; its unit phase -- arbitrary by
 * definition -- is fixed by making the FIRST component real and negative (what LAPACK's zgesdd leaves on these matrices, to
 * ~2e-4 of the component's modulus; a convention that is continuous along the DoA grid), so columns agree with the reference's
 * up to that residual phase and |W^H W| agrees.  C <= 32: cyclic two-sided Jacobi in LDS, one wave per DoA; 32 < C <= 128
 * (C even): one-sided (Hestenes) Jacobi on the matrix / the real embedding of the complex fold, one workgroup per DoA.
 * MICLOC_ERR_SHAPE beyond that.  Device buffers. */
int micloc_design_vectors_f64(const double *cov, int n_doa, int C, int bipolar, double rel_prec, double *bf_mat, int G, int g0,
                              void *stream);

/* ---- array-signal synthesis ------------------------------------------------------------------- */
/* Noise-free part of SNNBeamformer.apply_to_template for a constant DoA per trial (snn_beamformer.py:246-267):
 * x[b][t][m] = np.interp(max(time[t] - delays[b][m], time[0]), time, sig), bit-exact with NumPy.  All pointers are
 * DEVICE buffers: time [T] (the resampled grid, np.arange(t.min(), t.max(), 1/fs)), sig [T], slopes [T-1] =
 * diff(sig)/diff(time), delays [B][M] (already shifted so that their minimum is 0), x [B][T][M]. */
int micloc_synth_delay_f64(const double *time, const double *sig, const double *slopes, int T, const double *delays,
                           int B, int M, double fs, double *x, void *stream);

/* General form: every noise-free signal generator of the reference (all pointers are DEVICE buffers).
 *   x[b][t][m] = sum_k gain[b][k][t] * np.interp(arg, time, sig),   arg = max(time[t] - (d - shift[b]), time[0])   (mode 0)
 *                                                                   arg = time[t] + d                             (mode 1)
 *   d = delay of microphone m for target k of trial b at time t (geometry.delays(doa, normalized=False))
 * mode MICLOC_SYNTH_APPLY_TO_TEMPLATE   = SNNBeamformer / Beamformer.apply_to_template (snn_beamformer.py:239-267,
 *      beamformer.py:220-245; `shift` = delays.min() of :257), also with a moving DoA (`moving` = 1);
 * mode MICLOC_SYNTH_SIGNAL_FROM_TEMPLATE = signal_from_template (xylo_snn_localization.py:44-71: no shift, no clamp, np.interp
 *      saturates) and, with K > 1 and `gain`, signal_multiple_targets (paper_plots/multiple_targets_snn.py:87-160).
 * Delay source: `delays` [B][K][Td][M] (Td = T if moving else 1) computed by the caller (NumPy cos: bit-exact parity with
 * the reference), or, with delays == NULL, computed on the device from `doa` [B][K][Td] and the geometry r_vec / theta_vec
 * [M], `speed` (-r cos(theta_m - doa) / speed; device cos, within an ulp of NumPy's: throughput runs, nothing of size
 * B x T x M crosses PCIe).  `shift` [B] and `gain` [B][K][T] may be NULL.  time / sig [T], slopes [T-1], x [B][T][M]. */
#define MICLOC_SYNTH_APPLY_TO_TEMPLATE 0
#define MICLOC_SYNTH_SIGNAL_FROM_TEMPLATE 1
typedef struct micloc_synth_args {
    const double *time, *sig, *slopes;
    int T, B, K, M;
    const double *delays;
    const double *doa;
    int moving;
    const double *r_vec, *theta_vec;
    double speed;
    const double *shift;
    const double *gain;
    int mode;
    double fs;
    double *x;
} micloc_synth_args;
int micloc_synth_targets_f64(const micloc_synth_args *args, void *stream);
/* shift[b] = min over targets, time steps and microphones of -r cos(theta_m - doa) / speed  (snn_beamformer.py:257's
 * `delays.min()`), from doa [B][K][moving_T]; device buffers. */
int micloc_delay_min_f64(const double *doa, int B, int K, int moving_T, const double *r_vec, const double *theta_vec, int M,
                         double speed, double *shift, void *stream);

/* ---- counter-based random numbers (throughput-mode sweeps) --------------------------------------- */
/* Philox-4x32-10 keyed by `seed`; counter = (pair index, epoch, trial, substream); the uniforms use the reserved trial word
 * 0xFFFFFFFF, so for one seed no two draws of any (generator, trial, substream, epoch) share a block.  The reference draws from NumPy's sequential
 * global MT19937 stream (target_snn_localization.py:452, snn_beamformer.py:273); parity runs replay that on the host,
 * throughput runs use these: out[i] = lo + (hi - lo) * u_i, u in [0, 1) with 53 bits (np.random.rand's range), and
 * x[b] += sigma_b * N(0, 1) (Box-Muller, fp64) with sigma_b = sqrt(mean(x[b]^2)) / sqrt(10^(snr_db[b] / 10))
 * (snn_beamformer.py:270-275) computed on the device from snr_db [B], or taken from `sigma` [B] when it is not NULL.
 * Trial b of the call uses counter word `first_trial + b`, so a sweep sharded over ranks draws the same noise for a
 * given global trial whatever the world size (first_trial + B <= 0xFFFFFFFF).  `epoch` (device uint32, may be NULL: 0) is read when
 * the kernel runs and is a counter word of its own: a HIP graph that contains the generators draws fresh numbers on every replay once micloc_counter_add_u32 (also in
 * the graph) advances it.  Device buffers; ws from micloc_awgn_workspace_bytes (256-B aligned). */
int micloc_uniform_f64(double *out, size_t n, uint64_t seed, uint32_t substream, const uint32_t *epoch, double lo, double hi,
                       void *stream);
size_t micloc_awgn_workspace_bytes(int B, int T, int M);
int micloc_awgn_f64(double *x, int B, int T, int M, const double *snr_db, const double *sigma, uint64_t seed, uint32_t substream,
                    const uint32_t *epoch, uint32_t first_trial, void *ws, size_t ws_bytes, void *stream);
/* micloc_synth_targets_f64 followed by micloc_awgn_f64 (sigma from snr_db) WITHOUT storing the noise-free signal: pass 1 recomputes
 * the synthesis and reduces its squares (the unfused order: sigma is bit-identical), pass 2 recomputes it, adds the normals of
 * micloc_awgn_f64 (same counters) and stores x once -- one trip through HBM instead of four; same bits as the two calls.
 * ws from micloc_synth_awgn_workspace_bytes(B, T, M, K). */
size_t micloc_synth_awgn_workspace_bytes(int B, int T, int M, int K);
int micloc_synth_awgn_f64(const micloc_synth_args *args, const double *snr_db, uint64_t seed, uint32_t substream, const uint32_t *epoch,
                          uint32_t first_trial, void *ws, size_t ws_bytes, void *stream);
int micloc_counter_add_u32(uint32_t *counter, uint32_t inc, void *stream); /* *counter += inc (device word) */

/* ---- Monte-Carlo results ---------------------------------------------------------------------- */
/* err[b] = arcsin|sin(doa_list[argmax[b]] - doa_true[b])| (target_snn_localization.py:464-466; pi-periodic) and
 * mae[s] = mean of err over the `B / groups` consecutive trials of SNR group s (:520).  All pointers are DEVICE buffers
 * (doa_list [G], doa_true [B], err [B], mae [groups]); err or mae may be NULL; B must be a multiple of groups. */
int micloc_doa_error_f64(const int32_t *argmax, const double *doa_list, int G, const double *doa_true, int B, int groups,
                         double *err, double *mae, void *stream);

/* ---- Xylo-A2 hidden-layer integer LIF (BASELINE config 4) -------------------------------------- */
/* Replaces XyloSim.evolve as called by Demo.xylo_process (xylo_snn_localization.py:358-377) with the network built at
 * :173-290: bit-shift decay, int8 weights, saturating int16 state, subtractive reset, one shared recurrent weight.
 * PARITY UNPINNED: the reference's arithmetic is rockpool/xylosim (third party, unpinned, absent); this follows the
 * published update rule (oracle/micloc_oracle.c oracle_xylo_lif).  spikes_in [B][T][Cin] uint8 (device),
 * W_in [Cin][N] / dash_syn [N] / dash_mem [N] / thr [N] host arrays; spikes_out [B][T][N] uint8 and rate [B][N]
 * int32 (device) may each be NULL.  Synchronises the stream once (coefficient upload). */
size_t micloc_xylo_workspace_bytes(int Cin, int N);
int micloc_xylo_lif_i16(const uint8_t *spikes_in, int B, int T, int Cin, const int8_t *W_in, int N, int w_rec,
                        const uint8_t *dash_syn, const uint8_t *dash_mem, const int16_t *thr, int max_spikes,
                        uint8_t *spikes_out, int32_t *rate, void *ws, size_t ws_bytes, void *stream);

/* The same network in two phases: micloc_xylo_upload places the packed weights / constants in `ws` (synchronises once);
 * micloc_xylo_lif_resident_i16 then runs any number of batches without touching host memory or synchronising, so it can
 * be captured into a HIP graph.  ternary_channels > 0: `spikes_in` is the encoder's int8 raster [B][T][ternary_channels]
 * in {-1, 0, +1} and Cin = 2 * ternary_channels: input channel c carries the +1 events of raster channel c, channel
 * ternary_channels + c its -1 events -- Demo.spike_encoding's split (xylo_snn_localization.py:350-354) folded into the
 * kernel's staging loop.  ternary_channels == 0: spikes_in is uint8 [B][T][Cin] as above.
 * With a ternary raster, w_rec == 0 and a 16-byte aligned `spikes_in` the launch takes the sweep kernel (two neurons per
 * lane in packed 16-bit arithmetic, input currents on the int8 matrix cores); anything else the general one.  Same results. */
int micloc_xylo_upload(int Cin, const int8_t *W_in, int N, const uint8_t *dash_syn, const uint8_t *dash_mem, const int16_t *thr,
                       void *ws, size_t ws_bytes, void *stream);
int micloc_xylo_lif_resident_i16(const void *spikes_in, int ternary_channels, int B, int T, int Cin, int N, int w_rec,
                                 int max_spikes, uint8_t *spikes_out, int32_t *rate, void *ws, size_t ws_bytes, void *stream);
/* The sweep's form of the same call (paper_plots/target_xylo_localization.py:590-594 needs only rec["Spikes"].mean(0): the spike
 * COUNTS): ternary raster [B][T][ternary_channels] in, rate [B][N] out, w_rec == 0.  The launch is a set of persistent workgroups
 * that take (trial, time chunk) tickets from a device counter and hand a trial's integer state on through `scratch`
 * (micloc_xylo_sweep_scratch_bytes(B) bytes, 256-byte aligned, contents irrelevant on entry; one buffer per concurrent call) --
 * every SIMD carries the same number of chains from start to end.  Same counts as micloc_xylo_lif_resident_i16; shapes the queue does
 * not serve (N > 512, Cin > 64, unaligned raster) take that call's kernels.  workers_per_cu: persistent workgroups per CU, 0 = 4 (three
 * waves each: three per SIMD); fewer leave room for the kernels of other streams.  No host access, no synchronisation (graph-capturable). */
size_t micloc_xylo_sweep_scratch_bytes(int B);
/* status2 = {tickets handed out (>= chunks x trials once the launch has drained), workers that gave up waiting for a predecessor
 * (must be 0: the wait is bounded at ~half a minute only to keep a broken launch from hanging the device)} of the last
 * micloc_xylo_lif_sweep_i16 call on `scratch`; synchronises the stream. */
int micloc_xylo_sweep_status(const void *scratch, int *status2, void *stream);
int micloc_xylo_lif_sweep_i16(const int8_t *raster, int ternary_channels, int B, int T, int Cin, int N, int max_spikes, int32_t *rate,
                              void *ws, size_t ws_bytes, void *scratch, size_t scratch_bytes, int workers_per_cu, void *stream);
/* Demo.spike_encoding's channel bookkeeping (micloc/xylo_snn_localization.py:339-354: the per-band rasters side by side,
 * astype(int64), then [spikes > 0 | spikes < 0]) for ONE band of `bands`: raster int8 [B][T][C] in {-1, 0, +1} -> its channel block
 * of out [B][T][bands * C * (2 if BIPOLAR else 1)].  TERNARY: the value itself at channel band*C + c (the concatenated raster the
 * sweep kernel reads); UNIPOLAR: (s > 0) there; BIPOLAR: (s > 0) there and (s < 0) at bands*C + band*C + c. */
#define MICLOC_PACK_TERNARY 0
#define MICLOC_PACK_UNIPOLAR 1
#define MICLOC_PACK_BIPOLAR 2
int micloc_pack_events_u8(const int8_t *raster, int B, int T, int C, int band, int bands, int mode, uint8_t *out, void *stream);
/* Demo.extract_rate (micloc/xylo_snn_localization.py:379-398) from the spike COUNTS of the hidden layer: rate[b][g] = mean over bands of
 * (counts[b][f * G + g] / T) * fs.  counts int32 [B][bands * G], rate double [B][G] (device). */
int micloc_rate_from_counts_f64(const int32_t *counts, int B, int G, int bands, int T, double fs, double *rate, void *stream);
/* index[b] = find_peak_location(power_b, win_size) (micloc/utils.py:84-121; paper_plots/target_xylo_localization.py:594-604)
 * with power_b[g] = sum over `bands` of rate[b][f * G + g] (the per-DoA spike counts; the reference's positive scale
 * factors T / fs and 1 / max do not move the arg-max): box-car of win_size samples, FULL non-circular convolution, first
 * maximum, minus win_size // 2, modulo G.  Exact integer window sums.  rate [B][bands * G] int32, index [B] (device). */
int micloc_peak_location_i32(const int32_t *rate, int B, int G, int bands, int win_size, int32_t *index, void *stream);

/* Moving-target tracking: Envelope.evolve (micloc/utils.py:36-81) on the beamformer output and the per-time-step arg-max over the DoA
 * grid, `doa_index = np.argmax(sig_bf_env, axis=1)` (paper_plots/target_snn_localization.py:599-622), without the T x G array ever
 * leaving the device.  y [B][T][G] (apply_to_signal's result, real);  env [B][T][G]: env[t] = the envelope state after sample t,
 * state_0 = |y_0|, state_t = (1 - 1/w) state_{t-1} + 1/w |y_t| [|y_t| >= state_{t-1}] with w = the rise window when the bracket holds,
 * the fall window otherwise -- NumPy's order of operations, nothing fused: bit-identical to the reference class.  The caller passes
 * a_rise = 1 - 1/w_rise, i_rise = 1/w_rise, a_fall = 1 - 1/w_fall as NumPy computes them (w = int(fs * time) >= 1).  env is always
 * written (a caller that only wants the indices passes scratch); index [B][T] int32 (may be NULL) = first maximum of every row. */
int micloc_envelope_track_f64(const double *y, int B, int T, int G, double a_rise, double i_rise, double a_fall, double *env, int32_t *index,
                              void *stream);
/* The same for the other arrays the reference's scripts hand to Envelope.evolve: the complex Beamformer's output
 * (paper_plots/target_localization.py:597-600; y [B][T][G][2] = numpy complex128, |z| = hypot(re, im) -- the device library's, < 1 ulp) and
 * integer spike rasters (paper_plots/target_xylo_localization.py:757-768: `sig_bf = spikes_out`; exact).  env stays double [B][T][G]. */
#define MICLOC_ENV_F64 0
#define MICLOC_ENV_C128 1
#define MICLOC_ENV_U8 2
#define MICLOC_ENV_I32 3
#define MICLOC_ENV_I64 4
int micloc_envelope_track_any(const void *y, int kind, int B, int T, int G, double a_rise, double i_rise, double a_fall, double *env, int32_t *index,
                              void *stream);

/* ---- misc ------------------------------------------------------------------------------------- */
int micloc_abi_version(void);
const char *micloc_status_string(int status);
int micloc_last_hip_error(void); /* last hipError_t seen by this thread (0 if none) */

#ifdef __cplusplus
}
#endif
#endif /* MICLOC_HIP_H */
