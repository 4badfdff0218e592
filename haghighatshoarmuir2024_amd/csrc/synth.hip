// Array-signal synthesis on the device: the noise-free part of SNNBeamformer.apply_to_template
// (reference micloc/snn_beamformer.py:246-267) for a constant DoA per trial:
//     x[b][t][m] = np.interp(max(time[t] - delay[b][m], time[0]), time, sig)
// Bit-exact with NumPy's arr_interp: same bracket j (xp[j] <= x < xp[j+1], found from a uniform-grid guess and
// corrected against the actual grid values), slope (fp[j+1]-fp[j])/(xp[j+1]-xp[j]) taken from a host-computed table
// (NumPy pre-computes the same table), result slope*(x - xp[j]) + fp[j] as an UNFUSED multiply-add (NumPy's C is
// built without FMA contraction; this file is compiled with -ffp-contract=off).  The per-trial delays
// (-r cos(theta_m - doa)/c minus the minimum) come from the host so that cos() is NumPy's.
// This replaces the reference's T-calls-per-trial Python loop over geometry.delays (60 % of its per-trial time).
#include "micloc_internal.h"

namespace micloc {

__global__ __launch_bounds__(256) void synth_kernel(const double *__restrict__ xp, const double *__restrict__ fp,
                                                     const double *__restrict__ slopes, int T,
                                                     const double *__restrict__ delays, int M, double inv_step,
                                                     double *__restrict__ out)
{
    const int b = blockIdx.y;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // flat (t, m)
    if (idx >= (size_t)T * M) return;
    const int t = (int)(idx / M);
    const int m = (int)(idx - (size_t)t * M);
    const double x0 = xp[0];
    double x = xp[t] - delays[(size_t)b * M + m];
    x = x < x0 ? x0 : x;
    // bracket: guess from the (nominally) uniform grid, then correct against the stored grid
    int j = (int)((x - x0) * inv_step);
    j = j < 0 ? 0 : (j > T - 1 ? T - 1 : j);
    while (j > 0 && xp[j] > x) --j;
    while (j < T - 1 && xp[j + 1] <= x) ++j;
    double r;
    if (j == T - 1) {
        r = fp[j];
    } else {
        const double xj = xp[j];
        r = (xj == x) ? fp[j] : slopes[j] * (x - xj) + fp[j];
    }
    out[(size_t)b * T * M + idx] = r;
}

hipError_t launch_synth(const double *xp, const double *fp, const double *slopes, int T, const double *delays, int B,
                        int M, double inv_step, double *out, hipStream_t stream)
{
    const size_t n = (size_t)T * M;
    dim3 grid((unsigned)((n + 255) / 256), B), block(256);
    hipLaunchKernelGGL(synth_kernel, grid, block, 0, stream, xp, fp, slopes, T, delays, M, inv_step, out);
    return hipGetLastError();
}

}  // namespace micloc
