// Which CUs does a stream made by hipExtStreamCreateWithCUMask see?  A kernel of many small
// workgroups notes (XCC, SE, CU) of the CU each ran on; the host prints, for a few masks, how many
// distinct CUs per XCC were used.  Build: hipcc --offload-arch=gfx950 -O2 cu_mask.hip -o cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
__global__ void where(uint32_t* out, int spin) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xF) << 16) | (hw & 0xFFFF);
}
int main() {
    const int blocks = 4096;
    uint32_t* d;
    hipMalloc(&d, blocks * 4);
    std::vector<uint32_t> h(blocks);
    struct Case { int first, count; };
    const Case cases[] = {{0, 256}, {0, 32}, {0, 48}, {0, 64}, {32, 224}, {48, 208}, {64, 192}, {0, 8}, {8, 8}, {0, 1}, {1, 1}, {8, 1}, {32, 1}};
    for (const Case& c : cases) {
        uint32_t mask[32] = {};
        for (int k = c.first; k < c.first + c.count; ++k) mask[k >> 5] |= 1u << (k & 31);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 32, mask) != hipSuccess) { printf("mask failed\n"); continue; }
        hipLaunchKernelGGL(where, dim3(blocks), dim3(64), 0, s, d, 20000);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), d, blocks * 4, hipMemcpyDeviceToHost);
        std::map<int, std::set<int>> per_xcc;
        for (uint32_t v : h) {
            const int xcc = v >> 16, hw = v & 0xFFFF;
            const int cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
            per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
        }
        int total = 0;
        printf("bits [%d, %d):", c.first, c.first + c.count);
        for (auto& kv : per_xcc) { printf(" xcc%d:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
        printf("  = %d CUs\n", total);
        hipStreamDestroy(s);
    }
    return 0;
}
