// Per-trial results of the Monte-Carlo loop on the device.
// Reference: paper_plots/target_snn_localization.py:464-467 (DoA estimate from the arg-max, pi-periodic error) and :520
// (mean absolute error per SNR).  Keeps the timed sweep free of framework element-wise launches: one tiny kernel after
// power_argmax_kernel instead of index / sin / abs / asin / mean.
#include "micloc_internal.h"

namespace micloc {

// One workgroup per SNR group (contiguous runs of `per_group` trials).  The per-group mean is a fixed-order two-level
// sum (strided per-thread partial sums, then a binary tree in LDS): deterministic, independent of the launch.
__global__ __launch_bounds__(256) void doa_error_kernel(const int32_t *__restrict__ argmax, const double *__restrict__ doa_list,
                                                         int G, const double *__restrict__ doa_true, int per_group,
                                                         double *__restrict__ err, double *__restrict__ mae)
{
    __shared__ double red[256];
    const int s = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < per_group; i += 256) {
        const int b = s * per_group + i;
        int a = argmax[b];
        a = a < 0 ? 0 : (a >= G ? G - 1 : a);
        const double e = asin(fabs(sin(doa_list[a] - doa_true[b])));
        if (err) err[b] = e;
        acc += e;
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0 && mae) mae[s] = red[0] / (double)per_group;
}

hipError_t launch_doa_error(const int32_t *argmax, const double *doa_list, int G, const double *doa_true, int B, int groups,
                            double *err, double *mae, hipStream_t stream)
{
    hipLaunchKernelGGL(doa_error_kernel, dim3(groups), dim3(256), 0, stream, argmax, doa_list, G, doa_true, B / groups, err, mae);
    return hipGetLastError();
}

}  // namespace micloc
