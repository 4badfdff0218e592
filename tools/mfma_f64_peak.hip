// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 rate on gfx950 (evidence for the roofline `peak`).
// hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ void mfma_loop(double *out, int iters, double a0, double b0)
{
    double4_t acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
void run(int waves_per_simd, int iters)
{
    const int blocks = 256, threads = 256 * waves_per_simd;  // one block per CU, 4 SIMDs x waves_per_simd waves
    double *out;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001, 0.9999);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_loop<CHAINS>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0001, 0.9999);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma = (double)blocks * (threads / 64) * iters * CHAINS;
    const double tflops = n_mfma * 2048.0 / (ms * 1e-3) / 1e12;
    printf("chains=%d waves/SIMD=%d: %.3f ms, %.1f TFLOP/s fp64 MFMA (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", CHAINS, waves_per_simd, ms,
           tflops, (ms * 1e-3 * 2.4e9) / (iters * CHAINS * (double)waves_per_simd));
    hipFree(out);
}

int main()
{
    for (int w = 1; w <= 4; ++w) {
        run<1>(w, 20000);
        run<4>(w, 5000);
    }
    return 0;
}
