/*
 * micloc_oracle.h -- CPU restatement of the micloc hot path.  TEST INFRASTRUCTURE ONLY:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
 * See micloc_oracle.c for the reference citations and the arithmetic contract.
 */
#ifndef MICLOC_ORACLE_H
#define MICLOC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_MAX_IIR 16

void oracle_stht(const double *x, int T, int M, const double *ker, int L, double *re, double *im);
void oracle_iir_df2t(const double *b, const double *a, int n, const double *x, int T, int stride,
                     double *y, int ystride);
int oracle_local_maxima(const double *x, int n, int *mid);
void oracle_select_by_distance(const int *peaks, const double *priority, int n, int distance,
                               unsigned char *keep);
void oracle_rzcc(const double *r, int T, int C, int robust_width, int bipolar, signed char *spikes);
void oracle_lif_fir(const signed char *spikes, int T, int C, const double *nir, int n, double *vmem);
void oracle_beamform(const double *v, int T, int C, const double *W, int G, double *y);
int oracle_power_argmax(const double *y, int T, int G, double *power);
int oracle_snn_chain(const double *x, int T, int M, const double *ker, int L, const double *b,
                     const double *a, int nba, int robust_width, int bipolar, const double *nir,
                     int n_nir, const double *W, int G, double *pre_enc, signed char *spikes,
                     double *vmem, double *y, double *power);
int oracle_beamformer_chain(const double *x, int T, int M, const double *ker, int L,
                            const double *b, const double *a, int nba, const double *Wre,
                            const double *Wim, int G, double *pre, double *yre, double *yim,
                            double *power);
void oracle_snn_chain_batch(const double *x, int B, int T, int M, const double *ker, int L,
                            const double *b, const double *a, int nba, int robust_width,
                            int bipolar, const double *nir, int n_nir, const double *W, int G,
                            double *power, int *argmax);

void oracle_xylo_lif(const unsigned char *spikes_in, int T, int Cin, const signed char *W_in, int N, int w_rec,
                     const unsigned char *dash_syn, const unsigned char *dash_mem, const short *thr, int max_spikes,
                     unsigned char *spikes_out, int *rate);

void oracle_philox4x32_10(const unsigned int ctr[4], const unsigned int key[2], unsigned int out[4]);
void oracle_uniform(double *out, long long n, unsigned long long seed, unsigned int substream, unsigned int epoch, double lo, double hi);
void oracle_normals(double *z, long long n, unsigned long long seed, unsigned int substream, unsigned int epoch, unsigned int trial);

#ifdef __cplusplus
}
#endif
#endif
