// Microbenchmark: what one wave (and two waves per SIMD) can issue in fp64 VALU on gfx950.
// The band-pass / RZCC stage is bound by single-wave issue + dependent latency, so these numbers set its floor:
//   indep   : 8 independent v_fma_f64 chains            -> cycles per instruction (issue)
//   dep     : 1 chain                                     -> cycles per dependent instruction (latency)
//   half    : 8 independent chains, lanes 32..63 masked   -> does a half-empty wave issue faster?
//   iir     : the DF2T order-4 step (9 fma + 1 mul + 1 add, 2 on the recurrence)
// each with 1 and 2 waves per SIMD (block 256 / 512, one block per CU).
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o valu_f64_issue tools/valu_f64_issue.hip && ./valu_f64_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int ITER = 4096;

template <int MODE>
__global__ __launch_bounds__(512) void k(double *out, long long *cyc, double a, double b)
{
    double acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    const bool on = MODE != 2 || (threadIdx.x & 63) < 32;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (MODE == 0 || MODE == 2) {
        if (on) {
            for (int it = 0; it < ITER; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_fma(acc[i], a, b);
            }
        }
    } else if (MODE == 1) {
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[0] = __builtin_fma(acc[0], a, b);
        }
    } else {
        // DF2T order 4: z = acc[0..3], running sum acc[4]; input derived from the iteration (no memory)
        double x = acc[5];
        for (int it = 0; it < ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const double y = __builtin_fma(b, x, acc[0]);
                acc[0] = __builtin_fma(-a, y, __builtin_fma(b, x, acc[1]));
                acc[1] = __builtin_fma(-a, y, __builtin_fma(b, x, acc[2]));
                acc[2] = __builtin_fma(-a, y, __builtin_fma(b, x, acc[3]));
                acc[3] = __builtin_fma(-a, y, b * x);
                acc[4] = acc[4] + y;
                x = -x;
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int block, double per_iter_instr)
{
    double *out;
    long long *cyc;
    const int grid = 256;
    hipMalloc(&out, sizeof(double) * grid * 512);
    hipMalloc(&cyc, sizeof(long long) * grid);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), 0, 0, out, cyc, 0.999, 1e-3);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(block), 0, 0, out, cyc, 0.999, 1e-3);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), cyc, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += v;
    mean /= grid;
    // s_memtime runs at a fixed 100 MHz on this part; report wall time per instruction as well
    printf("%-8s block %3d (%d wave/SIMD): %8.0f ticks, %.3f ms -> %.2f ns per wave-instruction (%.1f cycles at 2.4 GHz)\n", name, block,
           block / 256, mean, ms, ms * 1e6 / (ITER * per_iter_instr), ms * 1e6 / (ITER * per_iter_instr) * 2.4);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int block : {256, 512}) {
        run<0>("indep", block, 8);
        run<1>("dep", block, 8);
        run<2>("half", block, 8);
        run<3>("iir", block, 8 * 11);
    }
    return 0;
}
