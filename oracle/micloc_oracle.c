/*
 * micloc_oracle.c -- CPU restatement of the micloc hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the *checker* for the HIP path.  Nothing under haghighatshoarmuir2024_amd/ may
 * import, link or call it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * It restates, stage by stage, the algorithm of the reference (citations are relative to
 * /root/reference/):
 *
 *   oracle_stht            micloc/snn_beamformer.py:325-327  (np.roll + 1j*lfilter(kernel,[1],x))
 *                          micloc/beamformer.py:281-283, micloc/xylo_snn_localization.py:329-331
 *   oracle_iir_df2t        micloc/snn_beamformer.py:330-331  (scipy.signal.lfilter(b,a,.) = DF2T)
 *   oracle_rzcc            micloc/spike_encoder.py:115-137   (cumsum + find_peaks(distance=w))
 *   oracle_local_maxima    scipy.signal._peak_finding_utils._local_maxima_1d   (scipy 1.15.3)
 *   oracle_select_by_distance  scipy.signal._peak_finding_utils._select_by_peak_distance
 *   oracle_lif_fir         micloc/snn_beamformer.py:364      (lfilter(neuron_impulse_response,[1],spikes))
 *   oracle_beamform        micloc/snn_beamformer.py:368      (vmem @ bf_mat)
 *   oracle_power_argmax    paper_plots/target_snn_localization.py:462-464
 *   oracle_snn_chain       micloc/snn_beamformer.py:283-370  (apply_to_signal, whole chain)
 *   oracle_beamformer_chain micloc/beamformer.py:260-292     (Beamformer.apply_to_signal)
 *
 * Parity pin: tests/test_oracle_golden.py checks every function here against golden vectors that
 * tests/golden/make_golden.py produced by importing the reference itself in the build container
 * (spikes bit-exact, pre-encoder signal |err| <= 1e-11, power rel err <= 1e-10, same argmax).
 *
 * Arithmetic contract (shared with the HIP kernels so that HIP == oracle *bit for bit* up to and
 * including the spikes and the membrane signal):
 *   - everything is IEEE binary64;
 *   - every multiply-add is a single-rounding fma() and -ffp-contract=off is used for the rest, so
 *     the result does not depend on the host compiler's contraction choices;
 *   - FIR sums run over taps k ascending, starting from +0.0, skipping taps that are exactly 0.0
 *     (fma(0,x,acc) == acc for finite x, so skipping is exact);
 *   - DF2T recurrences are evaluated as  y = fma(b0,x,z0);  z_i = fma(-a_{i+1}, y, fma(b_{i+1}, x, z_{i+1})).
 *   - cumsum is the strictly sequential  c_t = c_{t-1} + r_t.
 * The reference (NumPy/SciPy) uses other summation orders; it agrees with this contract to ~1e-12
 * before the encoder and exactly on the spikes for every golden vector.
 */
#include "micloc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * STHT: in-phase = circular roll by L/2, quadrature = causal FIR with the Hilbert kernel.
 * x, re, im: [T][M] row-major (the reference's `sig_in_vec`, T x num_mic).
 * ---------------------------------------------------------------------------------------------- */
void oracle_stht(const double *x, int T, int M, const double *ker, int L, double *re, double *im)
{
    const int shift = L / 2; /* np.roll(x, kernel_length // 2, axis=0): out[(t+shift) % T] = in[t] */
    for (int t = 0; t < T; ++t) {
        int src = (t - (shift % T) + T) % T;
        for (int m = 0; m < M; ++m) re[(size_t)t * M + m] = x[(size_t)src * M + m];
    }
    /* lfilter(b,[1],x) FIR branch == np.convolve(b, x)[:T]: y[t] = sum_k b[k] x[t-k], zero history. */
    for (int m = 0; m < M; ++m) {
        for (int t = 0; t < T; ++t) {
            double acc = 0.0;
            int kmax = (t < L - 1) ? t : L - 1;
            for (int k = 0; k <= kmax; ++k) {
                double c = ker[k];
                if (c != 0.0) acc = fma(c, x[(size_t)(t - k) * M + m], acc);
            }
            im[(size_t)t * M + m] = acc;
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Direct-form-II-transposed IIR over one strided channel; n = len(b) = len(a) (pad with zeros),
 * a[0] must be 1 (scipy normalises by a[0] first; Butterworth designs already have a[0] == 1).
 * ---------------------------------------------------------------------------------------------- */
void oracle_iir_df2t(const double *b, const double *a, int n, const double *x, int T, int stride,
                     double *y, int ystride)
{
    double z[ORACLE_MAX_IIR];
    for (int i = 0; i < ORACLE_MAX_IIR; ++i) z[i] = 0.0;
    if (n == 1) {
        for (int t = 0; t < T; ++t) y[(size_t)t * ystride] = fma(b[0], x[(size_t)t * stride], 0.0);
        return;
    }
    for (int t = 0; t < T; ++t) {
        double xv = x[(size_t)t * stride];
        double yv = fma(b[0], xv, z[0]);
        for (int i = 0; i < n - 2; ++i) z[i] = fma(-a[i + 1], yv, fma(b[i + 1], xv, z[i + 1]));
        z[n - 2] = fma(-a[n - 1], yv, b[n - 1] * xv);
        y[(size_t)t * ystride] = yv;
    }
}

/* ------------------------------------------------------------------------------------------------
 * scipy _local_maxima_1d: strict rise, optional plateau, strict fall; plateau -> midpoint
 * (left+right)//2; first and last sample are never peaks.  Returns number of peaks.
 * ---------------------------------------------------------------------------------------------- */
int oracle_local_maxima(const double *x, int n, int *mid)
{
    int m = 0;
    int i = 1;
    const int imax = n - 1;
    while (i < imax) {
        if (x[i - 1] < x[i]) {
            int ia = i + 1;
            while (ia < imax && x[ia] == x[i]) ++ia;
            if (x[ia] < x[i]) {
                mid[m++] = (i + ia - 1) / 2;
                i = ia;
            }
        }
        ++i;
    }
    return m;
}

typedef struct {
    double pri;
    int pos;
} pri_item;

static int pri_cmp(const void *pa, const void *pb)
{
    const pri_item *a = (const pri_item *)pa, *b = (const pri_item *)pb;
    if (a->pri < b->pri) return -1;
    if (a->pri > b->pri) return 1;
    return (a->pos > b->pos) - (a->pos < b->pos); /* stable: equal priority keeps index order */
}

/* ------------------------------------------------------------------------------------------------
 * scipy _select_by_peak_distance: visit peaks by descending priority (equal priority: the later
 * peak first), every still-kept visited peak removes all neighbours closer than `distance`
 * (strict <).  keep[] is filled with 0/1.
 * ---------------------------------------------------------------------------------------------- */
void oracle_select_by_distance(const int *peaks, const double *priority, int n, int distance,
                               unsigned char *keep)
{
    pri_item *order = (pri_item *)malloc(sizeof(pri_item) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        order[i].pri = priority[i];
        order[i].pos = i;
        keep[i] = 1;
    }
    qsort(order, (size_t)n, sizeof(pri_item), pri_cmp);
    for (int i = n - 1; i >= 0; --i) {
        int j = order[i].pos;
        if (!keep[j]) continue;
        int k = j - 1;
        while (k >= 0 && peaks[j] - peaks[k] < distance) keep[k--] = 0;
        k = j + 1;
        while (k < n && peaks[k] - peaks[j] < distance) keep[k++] = 0;
    }
    free(order);
}

/* find_peaks(x, distance=w) restricted to what the encoder uses; returns #kept, fills out[]. */
static int find_peaks_distance(const double *x, int n, int w, int *out, int *tmp_pk, double *tmp_pr,
                               unsigned char *tmp_keep)
{
    int m = oracle_local_maxima(x, n, tmp_pk);
    for (int i = 0; i < m; ++i) tmp_pr[i] = x[tmp_pk[i]];
    /* scipy: distance >= 1 required; distance==1 keeps everything (no two peaks are < 1 apart). */
    oracle_select_by_distance(tmp_pk, tmp_pr, m, w, tmp_keep);
    int k = 0;
    for (int i = 0; i < m; ++i)
        if (tmp_keep[i]) out[k++] = tmp_pk[i];
    return k;
}

/* ------------------------------------------------------------------------------------------------
 * RZCC encoder.  r: [T][C] row-major (stride C), spikes: [T][C] int8 in {-1,0,+1}.
 * Per channel: c = cumsum(r); peaks of c -> +1; if bipolar, peaks of -c -> -1 (written second).
 * ---------------------------------------------------------------------------------------------- */
void oracle_rzcc(const double *r, int T, int C, int robust_width, int bipolar, signed char *spikes)
{
    double *c = (double *)malloc(sizeof(double) * (size_t)(T > 0 ? T : 1));
    double *pr = (double *)malloc(sizeof(double) * (size_t)(T > 0 ? T : 1));
    int *pk = (int *)malloc(sizeof(int) * (size_t)(T > 0 ? T : 1));
    int *sel = (int *)malloc(sizeof(int) * (size_t)(T > 0 ? T : 1));
    unsigned char *keep = (unsigned char *)malloc((size_t)(T > 0 ? T : 1));
    memset(spikes, 0, (size_t)T * (size_t)C);
    for (int ch = 0; ch < C; ++ch) {
        double acc = 0.0;
        for (int t = 0; t < T; ++t) {
            acc = acc + r[(size_t)t * C + ch];
            c[t] = acc;
        }
        int n = find_peaks_distance(c, T, robust_width, sel, pk, pr, keep);
        for (int i = 0; i < n; ++i) spikes[(size_t)sel[i] * C + ch] = 1;
        if (bipolar) {
            for (int t = 0; t < T; ++t) c[t] = -c[t];
            n = find_peaks_distance(c, T, robust_width, sel, pk, pr, keep);
            for (int i = 0; i < n; ++i) spikes[(size_t)sel[i] * C + ch] = -1;
        }
    }
    free(c);
    free(pr);
    free(pk);
    free(sel);
    free(keep);
}

/* vmem[t][c] = sum_{k<n} nir[k] * spikes[t-k][c]  (zero history); the past samples are visited in
 * chronological order (k descending), fma chain from 0 -- the order of the Toeplitz MFMA on the GPU. */
void oracle_lif_fir(const signed char *spikes, int T, int C, const double *nir, int n, double *vmem)
{
    for (int t = 0; t < T; ++t) {
        int kmax = (t < n - 1) ? t : n - 1;
        for (int ch = 0; ch < C; ++ch) {
            double acc = 0.0;
            for (int k = kmax; k >= 0; --k)
                acc = fma((double)spikes[(size_t)(t - k) * C + ch], nir[k], acc);
            vmem[(size_t)t * C + ch] = acc;
        }
    }
}

/* y[t][g] = sum_c v[t][c] * W[c][g], ascending c, fma chain from 0 (what a K-ordered f64 MFMA does). */
void oracle_beamform(const double *v, int T, int C, const double *W, int G, double *y)
{
    for (int t = 0; t < T; ++t) {
        const double *vt = v + (size_t)t * C;
        double *yt = y + (size_t)t * G;
        for (int g = 0; g < G; ++g) yt[g] = 0.0;
        for (int ch = 0; ch < C; ++ch) {
            const double a = vt[ch];
            const double *w = W + (size_t)ch * G;
            for (int g = 0; g < G; ++g) yt[g] = fma(a, w[g], yt[g]);
        }
    }
}

/* power[g] = mean_t y[t][g]^2 ; returns argmax (first maximum, like np.argmax). */
int oracle_power_argmax(const double *y, int T, int G, double *power)
{
    for (int g = 0; g < G; ++g) power[g] = 0.0;
    for (int t = 0; t < T; ++t) {
        const double *yt = y + (size_t)t * G;
        for (int g = 0; g < G; ++g) power[g] = fma(yt[g], yt[g], power[g]);
    }
    int best = 0;
    for (int g = 0; g < G; ++g) {
        power[g] = power[g] / (double)T;
        if (power[g] > power[best]) best = g;
    }
    return best;
}

/* ------------------------------------------------------------------------------------------------
 * Whole SNNBeamformer.apply_to_signal chain for one trial, streaming the T x G product through a
 * row buffer so the oracle can run speech-length signals without a 1 GB temporary.
 * Any of pre_enc [T][2M], spikes [T][2M], vmem [T][2M], y [T][G], power [G] may be NULL.
 * Returns argmax of power.
 * ---------------------------------------------------------------------------------------------- */
int oracle_snn_chain(const double *x, int T, int M, const double *ker, int L, const double *b,
                     const double *a, int nba, int robust_width, int bipolar, const double *nir,
                     int n_nir, const double *W, int G, double *pre_enc, signed char *spikes,
                     double *vmem, double *y, double *power)
{
    const int C = 2 * M;
    double *re = (double *)malloc(sizeof(double) * (size_t)T * M);
    double *im = (double *)malloc(sizeof(double) * (size_t)T * M);
    double *r = pre_enc ? pre_enc : (double *)malloc(sizeof(double) * (size_t)T * C);
    signed char *s = spikes ? spikes : (signed char *)malloc((size_t)T * C);
    double *v = vmem ? vmem : (double *)malloc(sizeof(double) * (size_t)T * C);
    double *pw = power ? power : (double *)malloc(sizeof(double) * (size_t)G);

    oracle_stht(x, T, M, ker, L, re, im);
    /* hstack([real, imag]): channel order re_0..re_{M-1}, im_0..im_{M-1} (snn_beamformer.py:335) */
    for (int m = 0; m < M; ++m) {
        oracle_iir_df2t(b, a, nba, re + m, T, M, r + m, C);
        oracle_iir_df2t(b, a, nba, im + m, T, M, r + M + m, C);
    }
    oracle_rzcc(r, T, C, robust_width, bipolar, s);
    oracle_lif_fir(s, T, C, nir, n_nir, v);

    int best;
    if (y) {
        oracle_beamform(v, T, C, W, G, y);
        best = oracle_power_argmax(y, T, G, pw);
    } else {
        double *row = (double *)malloc(sizeof(double) * (size_t)G);
        for (int g = 0; g < G; ++g) pw[g] = 0.0;
        for (int t = 0; t < T; ++t) {
            oracle_beamform(v + (size_t)t * C, 1, C, W, G, row);
            for (int g = 0; g < G; ++g) pw[g] = fma(row[g], row[g], pw[g]);
        }
        best = 0;
        for (int g = 0; g < G; ++g) {
            pw[g] = pw[g] / (double)T;
            if (pw[g] > pw[best]) best = g;
        }
        free(row);
    }
    free(re);
    free(im);
    if (!pre_enc) free(r);
    if (!spikes) free(s);
    if (!vmem) free(v);
    if (!power) free(pw);
    return best;
}

/* ------------------------------------------------------------------------------------------------
 * Beamformer.apply_to_signal (non-spiking): STHT + band-pass, then  h @ conj(W).
 * Wre/Wim: [M][G].  yre/yim: [T][G] (may be NULL), power[g] = mean_t |y|^2 (may be NULL).
 * Real-arithmetic expansion of (hr + j hi)(wr - j wi):  re = hr wr + hi wi ; im = hi wr - hr wi,
 * summed as one 2M-long fma chain over the stacked channels [hr_0..hr_{M-1}, hi_0..hi_{M-1}].
 * ---------------------------------------------------------------------------------------------- */
int oracle_beamformer_chain(const double *x, int T, int M, const double *ker, int L,
                            const double *b, const double *a, int nba, const double *Wre,
                            const double *Wim, int G, double *pre, double *yre, double *yim,
                            double *power)
{
    const int C = 2 * M;
    double *re = (double *)malloc(sizeof(double) * (size_t)T * M);
    double *im = (double *)malloc(sizeof(double) * (size_t)T * M);
    double *r = pre ? pre : (double *)malloc(sizeof(double) * (size_t)T * C);
    double *pw = power ? power : (double *)malloc(sizeof(double) * (size_t)G);
    /* stacked real weight matrices: A = [Wre; Wim] (real part), Bm = [-Wim; Wre] (imag part) */
    double *A = (double *)malloc(sizeof(double) * (size_t)C * G);
    double *Bm = (double *)malloc(sizeof(double) * (size_t)C * G);
    for (int m = 0; m < M; ++m)
        for (int g = 0; g < G; ++g) {
            A[(size_t)m * G + g] = Wre[(size_t)m * G + g];
            A[(size_t)(M + m) * G + g] = Wim[(size_t)m * G + g];
            Bm[(size_t)m * G + g] = -Wim[(size_t)m * G + g];
            Bm[(size_t)(M + m) * G + g] = Wre[(size_t)m * G + g];
        }
    oracle_stht(x, T, M, ker, L, re, im);
    for (int m = 0; m < M; ++m) {
        oracle_iir_df2t(b, a, nba, re + m, T, M, r + m, C);
        oracle_iir_df2t(b, a, nba, im + m, T, M, r + M + m, C);
    }
    double *rowr = (double *)malloc(sizeof(double) * (size_t)G);
    double *rowi = (double *)malloc(sizeof(double) * (size_t)G);
    for (int g = 0; g < G; ++g) pw[g] = 0.0;
    for (int t = 0; t < T; ++t) {
        oracle_beamform(r + (size_t)t * C, 1, C, A, G, rowr);
        oracle_beamform(r + (size_t)t * C, 1, C, Bm, G, rowi);
        for (int g = 0; g < G; ++g) {
            pw[g] = fma(rowr[g], rowr[g], pw[g]);
            pw[g] = fma(rowi[g], rowi[g], pw[g]);
        }
        if (yre) memcpy(yre + (size_t)t * G, rowr, sizeof(double) * (size_t)G);
        if (yim) memcpy(yim + (size_t)t * G, rowi, sizeof(double) * (size_t)G);
    }
    int best = 0;
    for (int g = 0; g < G; ++g) {
        pw[g] = pw[g] / (double)T;
        if (pw[g] > pw[best]) best = g;
    }
    free(rowr);
    free(rowi);
    free(A);
    free(Bm);
    free(re);
    free(im);
    if (!pre) free(r);
    if (!power) free(pw);
    return best;
}

/* Batched driver used by bench.py's cpu_baseline leg: B trials of [T][M], power [B][G], argmax [B]. */
void oracle_snn_chain_batch(const double *x, int B, int T, int M, const double *ker, int L,
                            const double *b, const double *a, int nba, int robust_width,
                            int bipolar, const double *nir, int n_nir, const double *W, int G,
                            double *power, int *argmax)
{
    for (int i = 0; i < B; ++i) {
        argmax[i] = oracle_snn_chain(x + (size_t)i * T * M, T, M, ker, L, b, a, nba, robust_width,
                                     bipolar, nir, n_nir, W, G, NULL, NULL, NULL, NULL,
                                     power + (size_t)i * G);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Xylo-A2 (SYNS61201) hidden-layer integer LIF, restated from the PUBLIC description of the chip /
 * XyloSim (bit-shift decay, 8-bit weights, 16-bit state, subtractive reset).  PARITY UNPINNED: the
 * arithmetic of the reference for this stage lives in rockpool/xylosim (third party, not vendored, not
 * pinned in setup.py:19, not installed here) and the reference holds no test vectors for it, so this
 * function follows the published update rule, not a checked-against-XyloSim one.
 * Call sites in the reference: micloc/xylo_snn_localization.py:269-290 (XyloSim.from_config),
 * :358-377 (xylo_process: reset_state, evolve, rec["Spikes"]).
 *
 * Per time step t and hidden neuron g (all state int16, saturating):
 *   isyn <- decay(isyn, dash_syn[g]);  vmem <- decay(vmem, dash_mem[g])
 *        decay(v, d) = v - dv,  dv = v >> d (arithmetic), and if dv == 0 and v != 0: dv = sign(v)
 *   isyn <- sat16(isyn + sum_c W_in[c][g] * s_in[t][c] + w_rec * (sum_g' s_out[t-1][g']))
 *   vmem <- sat16(vmem + isyn)
 *   n = 0; while vmem >= thr[g] and n < max_spikes: vmem -= thr[g]; ++n      (subtractive reset)
 *   s_out[t][g] = n
 * spikes_in: [T][Cin] uint8 (event counts), W_in: [Cin][N] int8, w_rec: one shared recurrent weight
 * (Demo uses an all-equal w_rec = -0.1/N, xylo_snn_localization.py:231-232).
 * spikes_out ([T][N] uint8) and rate ([N] int32, sum over t) may each be NULL.
 * ---------------------------------------------------------------------------------------------- */
static int sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }

static int xylo_decay(int v, int dash)
{
    int dv = v >> dash; /* arithmetic shift of a (sign-extended) int16 value */
    if (dv == 0 && v != 0) dv = v > 0 ? 1 : -1;
    return v - dv;
}

void oracle_xylo_lif(const unsigned char *spikes_in, int T, int Cin, const signed char *W_in, int N, int w_rec,
                     const unsigned char *dash_syn, const unsigned char *dash_mem, const short *thr, int max_spikes,
                     unsigned char *spikes_out, int *rate)
{
    int *isyn = (int *)calloc((size_t)N, sizeof(int));
    int *vmem = (int *)calloc((size_t)N, sizeof(int));
    int prev_total = 0;
    if (rate)
        for (int g = 0; g < N; ++g) rate[g] = 0;
    for (int t = 0; t < T; ++t) {
        const unsigned char *s = spikes_in + (size_t)t * Cin;
        int total = 0;
        for (int g = 0; g < N; ++g) {
            int in = 0;
            for (int c = 0; c < Cin; ++c) in += (int)W_in[(size_t)c * N + g] * (int)s[c];
            int i = xylo_decay(isyn[g], dash_syn[g]);
            int v = xylo_decay(vmem[g], dash_mem[g]);
            i = sat16(i + in + w_rec * prev_total);
            v = sat16(v + i);
            int n = 0;
            while (v >= thr[g] && n < max_spikes) {
                v -= thr[g];
                ++n;
            }
            isyn[g] = i;
            vmem[g] = v;
            total += n;
            if (spikes_out) spikes_out[(size_t)t * N + g] = (unsigned char)n;
            if (rate) rate[g] += n;
        }
        prev_total = total;
    }
    free(isyn);
    free(vmem);
}


/* ------------------------------------------------------------------------------------------------
 * Counter-based random numbers of the throughput-mode sweep (csrc/rng.hip): Philox-4x32-10
 * (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the Random123
 * reference implementation, v1.14 philox.h: multipliers 0xD2511F53 / 0xCD9E8D57, Weyl keys 0x9E3779B9 /
 * 0xBB67AE85), pinned by the known-answer vectors of its kat_vectors file (tests/test_rng_cpu.py).
 * The reference itself draws from NumPy's sequential MT19937 (paper_plots/target_snn_localization.py:452,
 * micloc/snn_beamformer.py:270-275); this stream replaces it only where bit parity of the noise is not
 * asked for.  Same counter layout as the device: key = seed, counter = (pair index, epoch, trial, substream); the uniforms
 * carry the reserved trial word 0xFFFFFFFF (disjoint from every trial's normals for any substream and epoch).
 * ---------------------------------------------------------------------------------------------- */
void oracle_philox4x32_10(const unsigned int ctr[4], const unsigned int key[2], unsigned int out[4])
{
    unsigned int c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0;
        const unsigned long long p1 = 0xCD9E8D57ull * c2;
        const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c1 ^ k0;
        const unsigned int n1 = (unsigned int)p1;
        const unsigned int n2 = (unsigned int)(p0 >> 32) ^ c3 ^ k1;
        const unsigned int n3 = (unsigned int)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u53_co(unsigned int lo, unsigned int hi) { return (double)((((unsigned long long)hi << 32) | lo) >> 11) * 0x1.0p-53; }
static double u53_oc(unsigned int lo, unsigned int hi) { return (double)(((((unsigned long long)hi << 32) | lo) >> 11) + 1) * 0x1.0p-53; }

/* out[i] = lo + (hi - lo) * u_i, u in [0, 1): two per Philox call */
void oracle_uniform(double *out, long long n, unsigned long long seed, unsigned int substream, unsigned int epoch, double lo, double hi)
{
    const unsigned int key[2] = {(unsigned int)seed, (unsigned int)(seed >> 32)};
    const double span = hi - lo;
    for (long long pair = 0; 2 * pair < n; ++pair) {
        const unsigned int ctr[4] = {(unsigned int)pair, epoch, 0xFFFFFFFFu, substream};
        unsigned int r[4];
        oracle_philox4x32_10(ctr, key, r);
        out[2 * pair] = lo + span * u53_co(r[0], r[1]);
        if (2 * pair + 1 < n) out[2 * pair + 1] = lo + span * u53_co(r[2], r[3]);
    }
}

/* z[e] for the flat [T][M] block of trial `trial`: Box-Muller, pair i -> elements 2i (cos), 2i+1 (sin) */
void oracle_normals(double *z, long long n, unsigned long long seed, unsigned int substream, unsigned int epoch, unsigned int trial)
{
    const unsigned int key[2] = {(unsigned int)seed, (unsigned int)(seed >> 32)};
    for (long long pair = 0; 2 * pair < n; ++pair) {
        const unsigned int ctr[4] = {(unsigned int)pair, epoch, trial, substream};
        unsigned int r[4];
        oracle_philox4x32_10(ctr, key, r);
        const double u1 = u53_oc(r[0], r[1]);
        const double u2 = u53_co(r[2], r[3]);
        const double rad = sqrt(-2.0 * log(u1));
        const double ang = 6.283185307179586476925286766559 * u2;
        z[2 * pair] = rad * cos(ang);
        if (2 * pair + 1 < n) z[2 * pair + 1] = rad * sin(ang);
    }
}
