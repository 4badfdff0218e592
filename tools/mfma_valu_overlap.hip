// Microbenchmark: what can run in the shadow of v_mfma_f64_16x16x4_f64 on a gfx950 SIMD?
// hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o /tmp/mfma_ovl && /tmp/mfma_ovl
//
// (A) intra-wave: every wave runs 4 independent MFMA chains with NF filler instructions pinned behind each MFMA
//     (sched_group_barrier); fillers: fp64 FMA, fp32 FMA, int32 mad, LDS read.
// (B) inter-wave: waves 0..3 of a workgroup (one per SIMD) run pure MFMA chains, waves 4..7 run pure filler loops.
// Reported: cycles per MFMA per SIMD (64 = the matrix pipe is never waiting).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));

enum { F_NONE = 0, F_F64 = 1, F_F32 = 2, F_I32 = 3, F_LDS = 4 };

template <int KIND, int NF>
__global__ __launch_bounds__(256, 2) void intra(double *out, int iters, double a0, double b0)
{
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = i;
    __syncthreads();
    double4_t acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    double fd[4] = {a, b, a + b, a - b};
    float ff[6] = {(float)a, (float)b, 1.5f, 2.5f, 1.0000001f, 1e-9f};
    int fi[5] = {(int)threadIdx.x, 3, 5, 7, (int)threadIdx.x * 77};
    const unsigned ldsaddr = (unsigned)(size_t)lds + 8 * (threadIdx.x & 63);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (KIND == F_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(fd[(c + f) & 3]) : "v"(a), "v"(b));
                if (KIND == F_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ff[(c + f) & 3]) : "v"(ff[4]), "v"(ff[5]));
                if (KIND == F_I32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(fi[(c + f) & 3]) : "v"(fi[4]));
                if (KIND == F_LDS) asm volatile("ds_read_b64 %0, %1" : "=v"(fd[(c + f) & 3]) : "v"(ldsaddr));
            }
        }
        if (KIND == F_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    double s = 0;
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3] + fd[c] + ff[c] + fi[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// waves 0..3: MFMA only; waves 4..7: filler only, `fill_iters` iterations of 16 fillers (0 = idle companion)
template <int KIND>
__global__ __launch_bounds__(512) void inter(double *out, int iters, int fill_iters, double a0, double b0)
{
    __shared__ double lds[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const int wv = threadIdx.x >> 6;
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    double s = 0;
    if (wv < 4) {
        double4_t acc[4];
        for (int c = 0; c < 4; ++c) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
        for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    } else {
        double fd[8] = {a, b, a + b, a - b, a * 2, b * 2, a * 3, b * 3};
        float ff[10] = {(float)a, (float)b, 1.5f, 2.5f, 3.5f, 4.5f, 5.5f, 6.5f, 1.0000001f, 1e-9f};
        int fi[9] = {(int)threadIdx.x, 3, 5, 7, 9, 11, 13, 15, (int)threadIdx.x * 77};
        const unsigned ldsaddr = (unsigned)(size_t)lds + 8 * (threadIdx.x & 63);
        for (int i = 0; i < fill_iters; ++i) {
#pragma unroll
            for (int f = 0; f < 16; ++f) {
                if (KIND == F_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(fd[f & 7]) : "v"(a), "v"(b));
                if (KIND == F_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ff[f & 7]) : "v"(ff[8]), "v"(ff[9]));
                if (KIND == F_I32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(fi[f & 7]) : "v"(fi[8]));
                if (KIND == F_LDS) asm volatile("ds_read_b64 %0, %1" : "=v"(fd[f & 7]) : "v"(ldsaddr));
            }
            if (KIND == F_LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        for (int c = 0; c < 8; ++c) s += fd[c] + ff[c] + fi[c];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double *g_out;
static const char *kname[] = {"none", "v_fma_f64", "v_fma_f32", "v_mad_i32", "ds_read_b64"};

template <int KIND, int NF>
void run_intra(int iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((intra<KIND, NF>), dim3(256), dim3(256), 0, 0, g_out, iters, 1.0001, 0.9999);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("intra-wave  filler=%-11s x%d per MFMA: %.3f ms, %.1f cycles/MFMA/SIMD @2.4GHz\n", kname[KIND], NF, ms,
           ms * 1e-3 * 2.4e9 / (iters * 4.0));
}

template <int KIND>
void run_inter(int iters, int fill_iters)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((inter<KIND>), dim3(256), dim3(512), 0, 0, g_out, iters, fill_iters, 1.0001, 0.9999);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
    }
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("inter-wave  companion=%-11s %6d x16 fillers: %.3f ms, %.1f cycles/MFMA/SIMD @2.4GHz, %.1f cycles/filler\n",
           kname[KIND], fill_iters, ms, ms * 1e-3 * 2.4e9 / (iters * 4.0),
           fill_iters ? ms * 1e-3 * 2.4e9 / (fill_iters * 16.0) : 0.0);
}

int main()
{
    (void)hipMalloc(&g_out, sizeof(double) * 256 * 512);
    const int it = 5000;
    run_intra<F_NONE, 0>(it);
    run_intra<F_F64, 1>(it);
    run_intra<F_F64, 2>(it);
    run_intra<F_F64, 4>(it);
    run_intra<F_F32, 1>(it);
    run_intra<F_F32, 4>(it);
    run_intra<F_F32, 8>(it);
    run_intra<F_I32, 1>(it);
    run_intra<F_I32, 4>(it);
    run_intra<F_I32, 8>(it);
    run_intra<F_I32, 12>(it);
    run_intra<F_LDS, 1>(it);
    run_intra<F_LDS, 2>(it);
    run_intra<F_LDS, 4>(it);
    // companions sized to last about as long as the MFMA waves (20000 MFMAs x 64 cycles = 1.28 M cycles)
    run_inter<F_NONE>(it, 0);
    run_inter<F_F64>(it, 0);
    run_inter<F_F64>(it, 5000);
    run_inter<F_F64>(it, 10000);
    run_inter<F_F64>(it, 20000);
    run_inter<F_F32>(it, 10000);
    run_inter<F_F32>(it, 20000);
    run_inter<F_I32>(it, 10000);
    run_inter<F_I32>(it, 20000);
    run_inter<F_LDS>(it, 5000);
    run_inter<F_LDS>(it, 10000);
    (void)hipFree(g_out);
    return 0;
}
