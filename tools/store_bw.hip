// Microbenchmark: HBM write bandwidth of gfx950 for the store patterns a [rows][G] fp64 result matrix can be written
// with (the API-faithful `y` output of the LIF + beamforming stage, rows = B x T frames, G DoA columns).
//   flat        every wave-level store is 512 contiguous bytes, grid-stride              (what torch.fill_ does)
//   flat2       16 bytes per lane (1 KB per store)
//   tile        the MFMA accumulator layout stored directly: one store = 4 rows x 128 B; a wave owns 64 rows and walks
//               the 16-column tiles (gt outer, time tile inner)
//   tile_t      same stores, time tile outer / gt inner (a wave finishes 16 rows before it moves on)
//   row         after a 4 x 4 exchange between the 16-lane rows of a wave: one store = 1 row x 512 B
//   ws tile/row/flat   the 8 waves of a workgroup write the same 16-row block at about the same time (see ws_kernel)
// each with normal and non-temporal stores, for G = 360 (64-byte aligned rows) and G = 449 (8-byte aligned rows).
//   hipcc -O3 --offload-arch=gfx950 -o store_bw tools/store_bw.hip && ./store_bw
#include <hip/hip_runtime.h>

#include <cstdio>

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            printf("%s: %s\n", #x, hipGetErrorString(e_));                              \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

template <bool NT>
__device__ __forceinline__ void st(double *p, double v)
{
    if (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
}

template <bool NT>
__global__ __launch_bounds__(256) void flat_kernel(double *y, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) st<NT>(y + i, (double)i);
}

template <bool NT>
__global__ __launch_bounds__(256) void flat2_kernel(double *y, size_t n)
{
    typedef double d2 __attribute__((ext_vector_type(2)));
    const size_t stride = (size_t)gridDim.x * 256;
    d2 *y2 = reinterpret_cast<d2 *>(y);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n / 2; i += stride) {
        d2 v = {(double)i, 1.0};
        if (NT)
            __builtin_nontemporal_store(v, y2 + i);
        else
            y2[i] = v;
    }
}

// workgroup = 8 waves x 64 rows = 512 rows; MODE 0: gt outer, 1: time tile outer, 2: row-contiguous 512 B stores
template <bool NT, int MODE>
__global__ __launch_bounds__(512) void tile_kernel(double *y, size_t rows, int G)
{
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, lc = l & 15, q = l >> 4;
    const size_t r0 = (size_t)blockIdx.x * 512 + wv * 64;
    const int GT = (G + 15) / 16;
    if (MODE == 0) {
        for (int gt = 0; gt < GT; ++gt) {
            const int g = 16 * gt + lc;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t row = r0 + 16 * tt + q + 4 * r;
                    if (row < rows && g < G) st<NT>(y + row * G + g, (double)g);
                }
        }
    } else if (MODE == 1) {
        for (int tt = 0; tt < 4; ++tt)
            for (int gt = 0; gt < GT; ++gt) {
                const int g = 16 * gt + lc;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t row = r0 + 16 * tt + q + 4 * r;
                    if (row < rows && g < G) st<NT>(y + row * G + g, (double)g);
                }
            }
    } else {
        // four 16-column tiles at a time; after the exchange lane l holds column 64*gq + l of one row
        const int GQ = (G + 63) / 64;
        for (int gq = 0; gq < GQ; ++gq) {
            const int g = 64 * gq + l;
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    const size_t row = r0 + 16 * tt + rr;
                    if (row < rows && g < G) st<NT>(y + row * G + g, (double)g);
                }
        }
    }
}

// the bf_mat-stationary ownership: the 8 waves of a workgroup walk the time tiles of a 256-row chunk together, so every
// 16-row block (contiguous in memory) is written by the whole workgroup at about the same time.
//   MODE 0: wave owns tiles gt = wv + 8j, store = 4 rows x 128 B     MODE 1: wave owns columns [64 wv, 64 wv + 64), store
//   = 1 row x 512 B     MODE 2: all 512 threads write the block front to back (as after staging it in LDS)
template <bool NT, int MODE>
__global__ __launch_bounds__(512) void ws_kernel(double *y, size_t rows, int G)
{
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63, lc = l & 15, q = l >> 4;
    const size_t r0 = (size_t)blockIdx.x * 256;
    const int GT = (G + 15) / 16;
    for (int t = 0; t < 16; ++t) {
        const size_t rb = r0 + 16 * t;
        if (MODE == 0) {
            for (int gt = wv; gt < GT; gt += 8) {
                const int g = 16 * gt + lc;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t row = rb + q + 4 * r;
                    if (row < rows && g < G) st<NT>(y + row * G + g, (double)g);
                }
            }
        } else if (MODE == 1) {
            const int g = 64 * wv + l;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const size_t row = rb + rr;
                if (row < rows && g < G) st<NT>(y + row * G + g, (double)g);
            }
        } else {
            const size_t lo = rb * G, hi = (rb + 16 < rows ? rb + 16 : rows) * G;
            for (size_t e = lo + threadIdx.x; e < hi; e += 512) st<NT>(y + e, (double)e);
        }
    }
}

int main()
{
    const size_t rows = (size_t)1100 * 4799;
    double *y;
    CHECK(hipMalloc(&y, rows * 449 * sizeof(double)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int reps = 5;
    auto time = [&](const char *name, int G, auto launch) {
        launch();
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        for (int i = 0; i < reps; ++i) launch();
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        printf("%-12s G=%d  %.3f ms  %.2f TB/s\n", name, G, ms, rows * G * 8.0 / ms / 1e9);
        return 0;
    };
    for (int G : {360, 449}) {
        const size_t n = rows * G;
        const unsigned wg = (unsigned)((rows + 511) / 512);
        time("flat", G, [&] { hipLaunchKernelGGL(flat_kernel<false>, dim3(256 * 16), dim3(256), 0, 0, y, n); });
        time("flat nt", G, [&] { hipLaunchKernelGGL(flat_kernel<true>, dim3(256 * 16), dim3(256), 0, 0, y, n); });
        time("flat2", G, [&] { hipLaunchKernelGGL(flat2_kernel<false>, dim3(256 * 16), dim3(256), 0, 0, y, n); });
        time("flat2 nt", G, [&] { hipLaunchKernelGGL(flat2_kernel<true>, dim3(256 * 16), dim3(256), 0, 0, y, n); });
        time("tile", G, [&] { hipLaunchKernelGGL((tile_kernel<false, 0>), dim3(wg), dim3(512), 0, 0, y, rows, G); });
        time("tile nt", G, [&] { hipLaunchKernelGGL((tile_kernel<true, 0>), dim3(wg), dim3(512), 0, 0, y, rows, G); });
        time("tile_t", G, [&] { hipLaunchKernelGGL((tile_kernel<false, 1>), dim3(wg), dim3(512), 0, 0, y, rows, G); });
        time("tile_t nt", G, [&] { hipLaunchKernelGGL((tile_kernel<true, 1>), dim3(wg), dim3(512), 0, 0, y, rows, G); });
        time("row", G, [&] { hipLaunchKernelGGL((tile_kernel<false, 2>), dim3(wg), dim3(512), 0, 0, y, rows, G); });
        const unsigned wg2 = (unsigned)((rows + 255) / 256);
        time("ws tile", G, [&] { hipLaunchKernelGGL((ws_kernel<false, 0>), dim3(wg2), dim3(512), 0, 0, y, rows, G); });
        time("ws tile nt", G, [&] { hipLaunchKernelGGL((ws_kernel<true, 0>), dim3(wg2), dim3(512), 0, 0, y, rows, G); });
        time("ws row", G, [&] { hipLaunchKernelGGL((ws_kernel<false, 1>), dim3(wg2), dim3(512), 0, 0, y, rows, G); });
        time("ws row nt", G, [&] { hipLaunchKernelGGL((ws_kernel<true, 1>), dim3(wg2), dim3(512), 0, 0, y, rows, G); });
        time("ws flat", G, [&] { hipLaunchKernelGGL((ws_kernel<false, 2>), dim3(wg2), dim3(512), 0, 0, y, rows, G); });
        time("ws flat nt", G, [&] { hipLaunchKernelGGL((ws_kernel<true, 2>), dim3(wg2), dim3(512), 0, 0, y, rows, G); });
        time("row nt", G, [&] { hipLaunchKernelGGL((tile_kernel<true, 2>), dim3(wg), dim3(512), 0, 0, y, rows, G); });
    }
    CHECK(hipMemset(y, 0, 64));
    (void)hipFree(y);
    return 0;
}
