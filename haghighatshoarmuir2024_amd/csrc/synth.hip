// Array-signal synthesis on the device: the noise-free part of every signal generator of the reference.
//
//   SNNBeamformer.apply_to_template     micloc/snn_beamformer.py:239-267     x = interp(max(t - (d - d.min()), t0), t, s)
//   Beamformer.apply_to_template        micloc/beamformer.py:220-245         (same arithmetic)
//   signal_from_template (Xylo sweep)   micloc/xylo_snn_localization.py:44-71   x = interp(t + d, t, s)  (no shift, no clamp)
//   signal_multiple_targets             paper_plots/multiple_targets_snn.py:87-160   x = sum_k power_k[t] * interp(t + d_k, t, s)
//
// with d = geometry.delays(doa, normalized=False) = -r_m cos(theta_m - doa) / c per time step (the DoA may move).
// The interpolation is bit-exact with NumPy's arr_interp: same bracket j (xp[j] <= x < xp[j+1], found from a
// uniform-grid guess and corrected against the actual grid values), slope (fp[j+1]-fp[j])/(xp[j+1]-xp[j]) taken from
// a host-computed table (NumPy pre-computes the same table), result slope*(x - xp[j]) + fp[j] as an UNFUSED
// multiply-add (NumPy's C is built without FMA contraction; this file is compiled with -ffp-contract=off), left /
// right saturation at fp[0] / fp[T-1].
//
// Two sources for the delays:
//   * a table from the host ([B][K][Td][M], NumPy's cos): bit-exact with the reference -- the parity path;
//   * computed here from the DoA series and the geometry (device cos, within an ulp of NumPy's): nothing of size
//     B x T x M crosses PCIe -- the throughput path.  This replaces the reference's T-calls-per-trial Python loop over
//     geometry.delays (60 % of its per-trial time, snn_beamformer.py:254-256).
#include "micloc_internal.h"
#include "synth_dev.h"

namespace micloc {

// constant-DoA, one target, host delays: the original fast path (one delay load per output)
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
    out[(size_t)b * T * M + idx] = interp_one(xp, fp, slopes, T, x, x0, inv_step);
}

hipError_t launch_synth(const double *xp, const double *fp, const double *slopes, int T, const double *delays, int B,
                        int M, double inv_step, double *out, hipStream_t stream)
{
    const size_t n = (size_t)T * M;
    dim3 grid((unsigned)((n + 255) / 256), B), block(256);
    hipLaunchKernelGGL(synth_kernel, grid, block, 0, stream, xp, fp, slopes, T, delays, M, inv_step, out);
    return hipGetLastError();
}

// general form: K targets, moving DoAs, per-sample gains, either delay source, both conventions
constexpr int SY_DMAX = 512;  // delays of a constant-DoA trial kept in LDS (K x M)

__global__ __launch_bounds__(256) void synth_targets_kernel(SynthArgs a)
{
    // constant DoAs: the K x M delays of the trial do not depend on time -- computed (one cos each) once per workgroup
    // instead of once per output sample
    __shared__ double dl[SY_DMAX];
    const int b = blockIdx.y;
    const bool cached = !a.moving && a.K * a.M <= SY_DMAX;
    if (cached) {
        for (int e = threadIdx.x; e < a.K * a.M; e += 256) dl[e] = mic_delay(a, b, e / a.M, 0, e % a.M);
        __syncthreads();
    }
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;  // flat (t, m)
    if (idx >= (size_t)a.T * a.M) return;
    const int t = (int)(idx / a.M);
    const int m = (int)(idx - (size_t)t * a.M);
    const double shift = a.shift ? a.shift[b] : 0.0;
    a.x[(size_t)b * a.T * a.M + idx] = synth_sample(a, cached ? dl : nullptr, b, t, m, a.time[0], shift);
}

hipError_t launch_synth_targets(const SynthArgs &a, hipStream_t stream)
{
    const size_t n = (size_t)a.T * a.M;
    dim3 grid((unsigned)((n + 255) / 256), a.B), block(256);
    hipLaunchKernelGGL(synth_targets_kernel, grid, block, 0, stream, a);
    return hipGetLastError();
}

// shift[b] = min over targets, time steps and microphones of the un-normalised delays (apply_to_template's
// `delays.min()`, snn_beamformer.py:257), from the DoA series.  One workgroup per trial, fixed-order reduction.
__global__ __launch_bounds__(256) void delay_min_kernel(const double *__restrict__ doa, int K, int Td, const double *__restrict__ r_vec,
                                                         const double *__restrict__ theta_vec, int M, double speed,
                                                         double *__restrict__ shift)
{
    __shared__ double red[256];
    const int b = blockIdx.x;
    const size_t n = (size_t)K * Td * M;
    double mn = __builtin_inf();
    for (size_t i = threadIdx.x; i < n; i += 256) {
        const size_t kt = i / M;
        const int m = (int)(i - kt * M);
        const double d = -r_vec[m] * cos(theta_vec[m] - doa[(size_t)b * K * Td + kt]) / speed;
        mn = d < mn ? d : mn;
    }
    red[threadIdx.x] = mn;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] = red[threadIdx.x + h] < red[threadIdx.x] ? red[threadIdx.x + h] : red[threadIdx.x];
        __syncthreads();
    }
    if (threadIdx.x == 0) shift[b] = red[0];
}

hipError_t launch_delay_min(const double *doa, int B, int K, int Td, const double *r_vec, const double *theta_vec, int M,
                            double speed, double *shift, hipStream_t stream)
{
    hipLaunchKernelGGL(delay_min_kernel, dim3(B), dim3(256), 0, stream, doa, K, Td, r_vec, theta_vec, M, speed, shift);
    return hipGetLastError();
}

}  // namespace micloc
