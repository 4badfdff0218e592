// Internal declarations shared by the HIP translation units of libmicloc_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/micloc_hip.h"

namespace micloc {

// ---- measurement variants -------------------------------------------------------------------------------------------
// The shipped library is built with both `false`: what runs never depends on the environment.  tools/dev/make_variant.py builds
// tools/_variants/libmicloc_hip_<name>.so from a copy of this header with ONE of them flipped (A/B runs on one box and
// tests/test_hip_parity.py::test_stht_vector_form_still_exact); results are identical in every variant.
constexpr bool VARIANT_WS_FOUR_KSTEPS = false;    // ws_k4: beamform_ws_kernel multiplies four k-steps whatever the channel count
constexpr bool VARIANT_STHT_VECTOR_FORM = false;  // stht_valu: stride-2 STHT kernels on the vector ALU instead of the matrix cores
constexpr bool VARIANT_STHT_WIDE_TWO_TILES = false;  // stht_wide2: the 480-tap walking STHT (96 kHz) with two time tiles per wave and ONE workgroup per CU (98 KB) instead of one tile / two workgroups
constexpr bool VARIANT_STHT_ONE_TILE = false;     // stht_one_tile: the matrix-core STHT with one time tile per workgroup (round 3's form) instead of the walk

// ---- XCD-aware workgroup order (speed only, never correctness) -------------------------------------------------------
// Workgroup L of a launch runs on XCD L % 8, and every XCD has its own L2.  Kernels whose NEIGHBOURING work items read the same
// cache lines (stream groups that end in the middle of a trial: a frame row [t][0..M) is shorter than a line; time tiles that
// share a halo) walk the items so that XCD x gets the contiguous range [x n/8, (x + 1) n/8) in ascending order; the n % 8
// items at the end keep their place.  A bijection of [0, n).
__device__ __forceinline__ int xcd_walk(int L, int n)
{
    const int per = n >> 3;
    return L < 8 * per ? (L & 7) * per + (L >> 3) : L;
}

// ---- STHT ---------------------------------------------------------------------------------------
constexpr int STHT_R = 8;                // consecutive output samples per lane (register window; groups of STHT_R delays)
constexpr int STHT_TILE = 64 * STHT_R;   // outputs per wave-task
constexpr int STHT_MAX_MB = 8;           // mics per block (one wave each)

struct SthtTaps {
    // compact tap table on device: taps[j] = ker[klo + j * kstep], zero padded to a multiple of 8/kstep
    const double *taps;
    int ngroups;  // number of groups of (STHT_R / kstep) taps
    int klo;      // delay of the first tap
    int kstep;    // 1 or 2
    int halo;     // Hh: samples staged before the tile; Hh == klo (mod STHT_R), Hh >= largest delay
    int shift;    // L / 2 (np.roll amount)
};

hipError_t launch_stht(const SthtTaps &tp, const double *x, double *h, int B, int T, int M, int Ts,
                       hipStream_t stream, bool write_re = true);
size_t stht_lds_bytes(const SthtTaps &tp, int M);

// ---- band-pass + RZCC ---------------------------------------------------------------------------
struct IirCoef {
    double b[MICLOC_MAX_IIR];
    double a[MICLOC_MAX_IIR];
    int n;
};

// Scratch of the encoder: flagged-unit list, chunk checkpoints, fallback candidate lists (rzcc.hip "Time chunking").
// chunk_frames: 0 = automatic, < 0 = never chunk, > 0 = owned frames per chunk.
size_t rzcc_scratch_bytes(int nlanes, int T, int robust_width, int chunk_frames);
int rzcc_chunks(int nlanes, int T, int robust_width, int chunk_frames);  // chunks per stream the launcher will use
// phases: the serial scan of a chunked launch (RZ_PHASE_SCAN: checkpoints into the scratch; nothing when the launch is not chunked)
// and everything else (RZ_PHASE_ENCODE) may be enqueued by separate calls, scan first, with the same arguments
constexpr int RZ_PHASE_SCAN = 1, RZ_PHASE_ENCODE = 2, RZ_PHASE_ALL = 3;
hipError_t launch_bandpass_rzcc(const IirCoef &coef, const double *h, int nlanes, int C, int T, int Ts,
                                int robust_width, int bipolar, double *pre, int8_t *spikes, void *scratch,
                                hipStream_t stream, const double *xin = nullptr, int M = 0, int shift = 0,
                                int chunk_frames = 0, int phases = RZ_PHASE_ALL);
// streaming: one tile per launch, exact state hand-off (rzcc.hip "streaming")
size_t rzcc_stream_state_bytes(int nlanes);
hipError_t launch_stream_encode(const IirCoef &coef, const double *h, int nlanes, int C, int T, int Ts, int robust_width,
                                int bipolar, int8_t *spikes, int Ttot, long long t_base, int first_tile, int final_tile,
                                void *state, hipStream_t stream, int pos_lo = 0, const int *clk = nullptr);
// streaming localisation (rzcc.hip "which frames of the spike raster are final"); ctl: 64 device ints
hipError_t launch_stream_horizon(const void *enc_state, int nlanes, int bipolar, int t_end, int final_, int chunk_frames, int base_chunk,
                                 int nwin, int *ctl, hipStream_t stream, int clocked = 0);
// the device clock of a stream: control words of loc_state (ints).  [0] chunks done, [1] chunks of the open block, [4..7] + [10] chunk
// range / window length of the beamforming launch, [8] frames beamformed, [12..15] status -- and the clock:
constexpr int STREAM_CLK_T = 16;      // frames pushed so far (absolute time of the next tile's first frame)
constexpr int STREAM_CLK_BASE = 17;   // absolute frame of the raster window's row 0
constexpr int STREAM_CLK_SHIFT = 18;  // frames the window slides in front of the current tile
constexpr int STREAM_CLK_TEND = 19;   // t + n of the current tile
hipError_t launch_stream_begin_tile(int *ctl, int8_t *win, int8_t *tmp, int B, size_t row_bytes, int C, int n, int cap, int chunk_frames,
                                    hipStream_t stream);
hipError_t launch_stream_tick(int *ctl, hipStream_t stream);
hipError_t launch_stht_wrap_rows(double *h, const double *wrap, int B, int M, int Ts, int col0, int n, int half, const int *ctl,
                                 hipStream_t stream);
hipError_t launch_stream_commit(int *ctl, int block_chunks, hipStream_t stream);
hipError_t launch_stream_accumulate(const double *partial, int B, int nwin, int Gp, int G, const int *range, const int *ctl,
                                    double *acc, const int *frames_ptr, double *power, int32_t *argmax, hipStream_t stream);
constexpr int STREAM_BLOCK_CHUNKS = 32;  // == PA_BLOCK of the one-shot time reduction (beamform.hip)
hipError_t launch_zero_fill(void *ptr, size_t bytes, hipStream_t stream);
// row-major [B][T][C] <-> planar [B][C][Ts]
hipError_t launch_pack_planar(const double *src, double *dst, int B, int T, int C, int Ts, hipStream_t stream);
hipError_t launch_unpack_planar(const double *src, double *dst, int B, int T, int C, int Ts, hipStream_t stream);

// ---- LIF + beamforming --------------------------------------------------------------------------
constexpr int BF_WAVES = 8;                         // waves per block
constexpr int BF_NT = 4;                            // 16-frame tiles per wave
constexpr int BF_CHUNK = BF_WAVES * BF_NT * 16;     // frames per block (512)

struct BeamformW {
    const double *Wp;  // device, zero padded [Kp][Gp] row-major; Kp = 16 * CT, Gp = 16 * GT
    int CT, GT;
    int C, G;          // logical sizes (G counts real columns: 2 * G_complex for the complex variant)
    int complex_pairs; // 0: real bf_mat; 1: columns [0,G/2) are Re, [Gp/2 ...) see api
    // device {lo, hi, -, trial stride in frames} or nullptr: only the chunks lo <= chunk < hi of every trial are computed (the
    // others leave their row of `partial` untouched), trial b starts at spikes + b * stride * C -- the streaming path beamforms the chunks whose spikes have become final, the range is decided on
    // the device and the launch stays sync-free
    const int *chunk_range = nullptr;
};

struct NeuronTab {
    const double *tab;  // device: zero padded Toeplitz lookup, see beamform.hip
    int n;              // taps
    int NK;             // k-steps of 4 past samples: 4 * NK >= n + 15
};

// partial sums: [B][nchunks][Gp] doubles
size_t beamform_partial_bytes(int B, int T, int Gp);
// *nchunks: number of partial-sum rows per trial the chosen kernel wrote (input of launch_power_argmax)
hipError_t launch_lif_beamform(const BeamformW &W, const NeuronTab &nt, const int8_t *spikes, int B, int T,
                               double *y, double *partial, hipStream_t stream, int *nchunks);
// *nchunks: partial-sum rows per trial the chosen kernel wrote
hipError_t launch_planar_beamform(const BeamformW &W, const double *pre, int B, int T, int Ts, double *y,
                                  int y_complex, double *partial, hipStream_t stream, int *nchunks);
hipError_t launch_power_argmax(const double *partial, int B, int T, int nchunks, int Gp, int G, int complex_pairs,
                               int Ghalf_pad, double *power, int32_t *argmax, hipStream_t stream);
int beamform_nchunks(int T);
int beamform_nchunks_ct(int T, int CT);  // chunking of the kernel family that serves CT channel tiles
int lif_beamform_chunk_frames(const BeamformW &W, const NeuronTab &nt);  // frames per chunk of the kernel launch_lif_beamform picks (power only)

// fp32-MFMA variant of LIF + beamforming + power (up to 64 channels, bf_mat must fit in LDS)
hipError_t launch_lif_beamform_f32(const BeamformW &W, const NeuronTab &nt, const int8_t *spikes, int B, int T,
                                   double *partial, hipStream_t stream, int *nchunks);

// ---- covariance-form power / membrane covariance ------------------------------------------------------------
size_t cov_partial_bytes(int B, int T, int CT);
hipError_t launch_lif_cov(const NeuronTab &nt, const int8_t *spikes, int B, int T, int C, int CT, int t_start,
                          double *partial, hipStream_t stream);
hipError_t launch_cov_power(const double *partial, int B, int T, int CT, int C, int Tn, const double *Wp, int Gp, int G,
                            double *cov_out, double *power, int32_t *argmax, hipStream_t stream);

// Gram matrix of a planar signal [B][C][Ts] over frames t >= t_start (complex covariance of Beamformer.design_from_template)
size_t planar_gram_partial_bytes(int B, int T, int C, int t_start);
hipError_t launch_planar_gram(const double *x, int B, int C, int T, int Ts, int t_start, int normalise, double *gram, double *partial,
                              hipStream_t stream);

// ---- Xylo integer LIF (parity unpinned) ------------------------------------------------------------------
size_t xylo_ws_bytes(int Cin, int N);
hipError_t launch_xylo(const uint8_t *spikes_in, int B, int T, int Cin, const int8_t *W_in_host, int N, int w_rec,
                       const uint8_t *dash_syn_host, const uint8_t *dash_mem_host, const int16_t *thr_host,
                       int max_spikes, uint8_t *spikes_out, int32_t *rate, void *ws, hipStream_t stream);

hipError_t xylo_upload(int Cin, const int8_t *W_in_host, int N, const uint8_t *dash_syn_host, const uint8_t *dash_mem_host,
                       const int16_t *thr_host, void *ws, hipStream_t stream);
hipError_t launch_xylo_resident(const void *spikes_in, int ternary_C, int B, int T, int Cin, int N, int w_rec, int max_spikes,
                                uint8_t *spikes_out, int32_t *rate, void *ws, hipStream_t stream);

// counts-only sweep form (ternary raster, no recurrence): persistent workgroups on a (trial, time chunk) ticket queue
size_t xylo_sweep_scratch_bytes(int B);
hipError_t launch_xylo_sweep(const int8_t *raster, int ternary_C, int B, int T, int Cin, int N, int max_spikes, int32_t *rate, void *ws,
                             void *scratch, int workers_per_cu, hipStream_t stream);

// ---- array-signal synthesis ---------------------------------------------------------------------------------
hipError_t launch_synth(const double *xp, const double *fp, const double *slopes, int T, const double *delays, int B,
                        int M, double inv_step, double *out, hipStream_t stream);
struct SynthArgs {
    const double *time, *sig, *slopes;  // template on the fs grid
    int T, B, K, M;
    const double *delays;               // [B][K][Td][M] or nullptr
    const double *doa;                  // [B][K][Td] (delays == nullptr)
    int moving;                         // Td = moving ? T : 1
    const double *r_vec, *theta_vec;    // [M] geometry (delays == nullptr)
    double speed;
    const double *shift;                // [B] or nullptr
    const double *gain;                 // [B][K][T] or nullptr
    int mode;                           // 0: t - (d - shift), clamped at t0;  1: t + d
    double inv_step;
    double *x;                          // [B][T][M]
};
hipError_t launch_synth_targets(const SynthArgs &a, hipStream_t stream);
hipError_t launch_delay_min(const double *doa, int B, int K, int Td, const double *r_vec, const double *theta_vec, int M,
                            double speed, double *shift, hipStream_t stream);

// ---- counter-based random numbers (Philox-4x32-10) ---------------------------------------------------------------
hipError_t launch_uniform(double *out, size_t n, uint64_t seed, uint32_t substream, const uint32_t *epoch, double lo, double hi,
                          hipStream_t stream);
hipError_t launch_counter_add(uint32_t *counter, uint32_t inc, hipStream_t stream);
size_t awgn_ws_bytes(int B, size_t n);
hipError_t launch_awgn(double *x, int B, size_t n, int M, const double *snr_db, const double *sigma, uint64_t seed, uint32_t substream,
                       const uint32_t *epoch, uint32_t trial0, void *ws, hipStream_t stream);

// fused: x = synth(args) + sigma N(0, 1) without ever storing the noise-free signal (same bits as synth_targets + awgn)
size_t synth_awgn_ws_bytes(int B, size_t n, int K, int M);
hipError_t launch_synth_awgn(const SynthArgs &a, const double *snr_db, uint64_t seed, uint32_t substream, const uint32_t *epoch,
                             uint32_t trial0, void *ws, hipStream_t stream);

// ---- sweep results ---------------------------------------------------------------------------------------------
hipError_t launch_doa_error(const int32_t *argmax, const double *doa_list, int G, const double *doa_true, int B, int groups,
                            double *err, double *mae, hipStream_t stream);

hipError_t launch_design_vec(const double *cov, int n_doa, int C, int bipolar, double rel_prec, double *bf, int G, int g0,
                             hipStream_t stream);
hipError_t launch_pack_events(const int8_t *raster, size_t rows, int C, uint8_t *out, int stride, int pos_off, int neg_off, int mode,
                              hipStream_t stream);
hipError_t launch_rate_from_counts(const int32_t *counts, int B, int G, int F, int T, double fs, double *rate, hipStream_t stream);
hipError_t launch_envelope_track(const void *y, int kind, int B, int T, int G, double a_rise, double i_rise, double a_fall, double *env,
                                 int32_t *index, hipStream_t stream);
hipError_t launch_peak_location(const int32_t *rate, int B, int G, int F, int win, int32_t *index, hipStream_t stream);

}  // namespace micloc
