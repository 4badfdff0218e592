// Where do the waves of small workgroups land?  Launches G workgroups of W waves (with some LDS, all resident at once, kept alive
// for a while) and prints how many waves each SIMD of each CU received (HW_ID: simd, cu, sh, se, and XCC_ID).
// hipcc -O3 --offload-arch=gfx950 -o /tmp/wave_placement tools/wave_placement.hip && /tmp/wave_placement 1100 3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

__global__ void probe(unsigned *out, int spin)
{
    extern __shared__ int lds[];
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    lds[threadIdx.x] = spin;
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);  // stay resident while the others are placed
    if ((threadIdx.x & 63) == 0) {
        out[2 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6))] = hw;
        out[2 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) + 1] = xcc;
    }
}

int main(int argc, char **argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 1100, W = argc > 2 ? atoi(argv[2]) : 3;
    unsigned *d;
    hipMalloc(&d, sizeof(unsigned) * 2 * G * W);
    hipLaunchKernelGGL(probe, dim3(G), dim3(64 * W), 16384, 0, d, 200);
    std::vector<unsigned> h(2 * G * W);
    hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * G * W, hipMemcpyDeviceToHost);
    // HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
    std::map<unsigned, std::vector<int>> per_cu;  // key: (xcc, se, sh, cu) -> waves per simd
    for (int i = 0; i < G * W; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        auto &v = per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu];
        if (v.empty()) v.assign(4, 0);
        v[simd]++;
    }
    long tot[4] = {0, 0, 0, 0};
    int mx = 0;
    std::map<int, int> hist;
    for (auto &kv : per_cu) {
        for (int s = 0; s < 4; ++s) {
            tot[s] += kv.second[s];
            mx = kv.second[s] > mx ? kv.second[s] : mx;
            hist[kv.second[s]]++;
        }
    }
    printf("%d workgroups x %d waves on %zu CUs: waves per SIMD id 0..3 = %ld %ld %ld %ld, max on one SIMD %d, histogram (waves on a SIMD: SIMDs):",
           G, W, per_cu.size(), tot[0], tot[1], tot[2], tot[3], mx);
    for (auto &kv : hist) printf(" %d:%d", kv.first, kv.second);
    printf("\n");
    return 0;
}
