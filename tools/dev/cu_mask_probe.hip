// Which CUs does a hipExtStreamCreateWithCUMask bit select on this GPU?  Launches G workgroups of 3 waves on a stream masked to a set
// of bits and prints per XCC how many distinct CUs and workgroups were used.
// hipcc -O3 --offload-arch=gfx950 -o /tmp/cu_mask_probe tools/dev/cu_mask_probe.hip && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

__global__ void probe(unsigned *out, int spin)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
}

static void run(const char *name, const std::vector<int> &bits, int G)
{
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<uint32_t> mask((ncu + 31) / 32, 0);
    for (int b : bits) mask[b / 32] |= 1u << (b % 32);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
        printf("%s: stream creation failed\n", name);
        return;
    }
    unsigned *d;
    hipMalloc(&d, sizeof(unsigned) * 2 * G);
    hipLaunchKernelGGL(probe, dim3(G), dim3(192), 0, s, d, 2000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * G);
    hipMemcpy(h.data(), d, sizeof(unsigned) * 2 * G, hipMemcpyDeviceToHost);
    std::map<unsigned, std::map<unsigned, int>> per;  // xcc -> (se, sh, cu) -> workgroups
    for (int i = 0; i < G; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        per[xcc][(hw >> 8) & 0xff]++;
    }
    printf("%-28s %3zu bits, %d workgroups:", name, bits.size(), G);
    for (auto &kv : per) {
        int mx = 0;
        for (auto &c : kv.second) mx = c.second > mx ? c.second : mx;
        printf("  xcc%u: %zu CUs (max %d wg/CU)", kv.first, kv.second.size(), mx);
    }
    printf("\n");
    hipFree(d);
    hipStreamDestroy(s);
}

int main()
{
    int ncu = 0;
    hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
    printf("%d CUs\n", ncu);
    auto range = [](int a, int b, int step = 1) {
        std::vector<int> v;
        for (int i = a; i < b; i += step) v.push_back(i);
        return v;
    };
    run("bits [0, 32)", range(0, 32), 28);
    run("bits [0, 32), 256 wgs", range(0, 32), 256);
    run("bits [224, 256)", range(224, 256), 28);
    run("bits [32, 64)", range(32, 64), 28);
    run("bits [0, 8)", range(0, 8), 28);
    run("bit 0", range(0, 1), 8);
    run("bit 1", range(1, 2), 8);
    run("bit 8", range(8, 9), 8);
    run("every 8th bit", range(0, ncu, 8), 28);
    run("every 8th bit, 256 wgs", range(0, ncu, 8), 256);
    run("bits [32, 256)", range(32, ncu), 28);
    run("bits [32, 256), 2048 wgs", range(32, ncu), 2048);
    run("all bits", range(0, ncu), 28);
    return 0;
}
