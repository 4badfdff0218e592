// v_mfma_f64_16x16x4_f64 throughput against the number of independent accumulator chains per wave and of waves per SIMD.
// hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_f64_chains tools/mfma_f64_chains.hip && /tmp/mfma_f64_chains
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(1024) void k(double *out, int iters, double a0, double b0)
{
    double4_t acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH>
static void run(int waves_per_simd)
{
    const int iters = 4096 / CH;
    const int threads = 64 * 4 * waves_per_simd;  // one workgroup per CU, waves spread over its 4 SIMDs
    double *out;
    hipMalloc(&out, sizeof(double) * 256 * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0, 2.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * CH * waves_per_simd;
    printf("chains/wave %d  waves/SIMD %d: %.3f ms -> %.1f cycles per MFMA per SIMD (2.4 GHz)\n", CH, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / mfma_per_simd);
    hipFree(out);
}

int main()
{
    for (int w : {1, 2, 4}) {
        run<1>(w);
        run<2>(w);
        run<4>(w);
        run<8>(w);
    }
    return 0;
}
