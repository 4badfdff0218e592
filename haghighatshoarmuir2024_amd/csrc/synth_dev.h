// Device functions shared by the synthesis kernels (synth.hip) and the fused synthesis + noise kernels (rng.hip).
#pragma once
#include "micloc_internal.h"

namespace micloc {

// np.interp's bracket for x >= xp[0]: the largest j with xp[j] <= x (T - 1 when x >= xp[T-1]); the guess from the (nominally)
// uniform grid is corrected against the stored grid
__device__ __forceinline__ int interp_bracket(const double *__restrict__ xp, int T, double x, double x0, double inv_step)
{
    int j = (int)((x - x0) * inv_step);
    j = j < 0 ? 0 : (j > T - 1 ? T - 1 : j);
    while (j > 0 && xp[j] > x) --j;
    while (j < T - 1 && xp[j + 1] <= x) ++j;
    return j;
}

__device__ __forceinline__ double interp_at(const double *__restrict__ xp, const double *__restrict__ fp, const double *__restrict__ slopes, int T,
                                            double x, int j)
{
    if (j == T - 1) return fp[j];  // x >= xp[T-1]: right = fp[T-1]
    const double xj = xp[j];
    return (xj == x) ? fp[j] : slopes[j] * (x - xj) + fp[j];
}

__device__ __forceinline__ double interp_one(const double *__restrict__ xp, const double *__restrict__ fp,
                                             const double *__restrict__ slopes, int T, double x, double x0, double inv_step)
{
    if (x < x0) return fp[0];  // np.interp: left = fp[0]
    return interp_at(xp, fp, slopes, T, x, interp_bracket(xp, T, x, x0, inv_step));
}

__device__ __forceinline__ double mic_delay(const SynthArgs &a, int b, int k, int t, int m)
{
    const int Td = a.moving ? a.T : 1;
    const int td = a.moving ? t : 0;
    const size_t bk = (size_t)b * a.K + k;
    double d;
    if (a.delays) {
        d = a.delays[(bk * Td + td) * a.M + m];
    } else {
        // ArrayGeometry.delays (array_geometry.py:52): -r_vec * cos(theta_vec - theta) / speed, in that order
        d = -a.r_vec[m] * cos(a.theta_vec[m] - a.doa[bk * Td + td]) / a.speed;
    }
    return d;
}


// One output sample x[b][t][m] of the general generator (K targets, moving DoAs, per-sample gains, either delay source, both
// conventions): exactly the arithmetic of synth_targets_kernel.  `dl` (may be null): the K x M delays of a constant-DoA trial,
// computed once per workgroup.
__device__ __forceinline__ double synth_sample(const SynthArgs &a, const double *__restrict__ dl, int b, int t, int m, double x0, double shift)
{
    const double tt = a.time[t];
    double acc = 0.0;
    for (int k = 0; k < a.K; ++k) {
        double d = dl ? dl[k * a.M + m] : mic_delay(a, b, k, t, m);
        double x;
        if (a.mode == 0) {
            if (a.shift) d = d - shift;  // delays - delays.min()  (snn_beamformer.py:257)
            x = tt - d;                  // :259
            x = x < x0 ? x0 : x;         // :260
        } else {
            x = tt + d;  // xylo_snn_localization.py:64, multiple_targets_snn.py:147
        }
        double r = interp_one(a.time, a.sig, a.slopes, a.T, x, x0, a.inv_step);
        if (a.gain) r = a.gain[((size_t)b * a.K + k) * a.T + t] * r;  // multiple_targets_snn.py:155
        acc = (a.K == 1) ? r : acc + r;                               // :157 (sig_in = 0; sig_in += sig_target)
    }
    return acc;
}

}  // namespace micloc
