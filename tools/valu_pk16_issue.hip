// Microbenchmark: issue rate of the packed 16-bit integer ops the Xylo LIF kernel is made of, against 32-bit integer ops, on
// gfx950: 8 independent chains per lane, 1 / 2 / 4 waves per SIMD (block 256 / 512 / 1024, one block per CU).
//   hipcc -w -O3 --offload-arch=gfx950 -o /tmp/valu_pk16_issue tools/valu_pk16_issue.hip && /tmp/valu_pk16_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef short short2_t __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

// MODE 0: v_pk_add_i16 clamp   1: v_pk_max_i16   2: v_pk_ashrrev_i16   3: v_add_u32   4: v_max_i32   5: the packed LIF step (18 ops)
template <int MODE>
__global__ __launch_bounds__(1024) void k(int *out, long long *cyc, int a, int b)
{
    int acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x * 3 + i;
    auto s2 = [](int w) { return __builtin_bit_cast(short2_t, w); };
    auto i1 = [](short2_t v) { return __builtin_bit_cast(int, v); };
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) acc[i] = i1(__builtin_elementwise_add_sat(s2(acc[i]), s2(a)));
            if (MODE == 1) acc[i] = i1(__builtin_elementwise_max(s2(acc[i]), s2(a)));
            if (MODE == 2) acc[i] = i1(s2(acc[i]) >> s2(b));
            if (MODE == 3) acc[i] = acc[i] + a;
            if (MODE == 4) acc[i] = acc[i] > a ? acc[i] : a;
        }
        if (MODE == 5) {
            // two independent neuron pairs (acc[0..1] = isyn, vmem; acc[2..3] likewise), the kernel's step
#pragma unroll
            for (int p = 0; p < 4; p += 2) {
                const short2_t one = {1, 1}, fifteen = {15, 15};
                short2_t is = s2(acc[p]), vm = s2(acc[p + 1]);
                short2_t i2 = is - __builtin_elementwise_max(is >> s2(b), __builtin_elementwise_min(is, one));
                short2_t v2 = vm - __builtin_elementwise_max(vm >> s2(b), __builtin_elementwise_min(vm, one));
                i2 = __builtin_elementwise_add_sat(i2, s2(a));
                v2 = __builtin_elementwise_add_sat(v2, i2);
                const short2_t d = __builtin_elementwise_sub_sat(v2, s2(0x00400040));
                const short2_t m = d >> fifteen;
                acc[p] = i1(i2);
                acc[p + 1] = (i1(m) & i1(v2)) | (~i1(m) & i1(d));
                acc[4] += i1(m);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    int s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char *name, int ninstr)
{
    int *out;
    long long *cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(int));
    hipMalloc(&cyc, 256 * sizeof(long long));
    for (int block : {256, 512, 1024}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(block), 0, 0, out, cyc, 0x00030005, 0x00020002);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(block), 0, 0, out, cyc, 0x00030005, 0x00020002);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)ITER * ninstr;  // wave-instructions per wave
        const double waves_per_simd = block / 256.0;
        printf("%-22s %d wave/SIMD: %.3f ms -> %.2f cycles per wave-instruction per SIMD (2.4 GHz)\n", name, (int)waves_per_simd, ms,
               ms * 1e-3 * 2.4e9 / (n * waves_per_simd));
    }
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    run<0>("v_pk_add_i16 clamp", 8);
    run<1>("v_pk_max_i16", 8);
    run<2>("v_pk_ashrrev_i16", 8);
    run<3>("v_add_u32", 8);
    run<4>("v_max_i32", 8);
    run<5>("packed LIF step x2", 2 * 15);
    return 0;
}
