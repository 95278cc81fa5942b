// Micro-benchmark: what does an LDS-DMA request (global_load_lds_dwordx4: 64 lanes x 16 B = 1 KiB,
// destination M0 + lane * 16) cost the matrix pipe when it is issued between the
// v_mfma_f32_16x16x4_f32 of a 512-thread block (2 waves per SIMD, one block per CU: the shape of
// dbh_forward_kernel, which moves ~527 KB of weights per window this way)?  And does the
// instruction's immediate offset move BOTH addresses (then one M0 serves four pieces)?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/dma_issue.hip -o tools/microbench/_build/dma_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

#define M(i) "v_mfma_f32_16x16x4_f32 %" #i ", %12, %13, %" #i "\n"
#define PURE M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) \
             M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11)
// %14 = lane * 16 (vector offset), %15 = global base (scalar pair), %16..%19 = LDS bases (scalar)
#define D(m, off) "s_mov_b32 m0, %" #m "\ns_nop 0\nglobal_load_lds_dwordx4 %14, %15 offset:" #off "\n"
#define DSAME(off) "global_load_lds_dwordx4 %14, %15 offset:" #off "\n"
#define SETM0(m) "s_mov_b32 m0, %" #m "\ns_nop 0\n"
// plain loads to registers for comparison
#define G(d, off) "global_load_dwordx4 %" #d ", %14, %15 offset:" #off "\n"

#define PAT_D3 \
    M(0) M(1) M(2) M(3) D(16, 0) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) D(17, 0) \
    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) D(18, 0) M(8) M(9) M(10) M(11)
#define PAT_D6 \
    M(0) M(1) D(16, 0) M(2) M(3) M(4) M(5) D(17, 0) M(6) M(7) M(8) M(9) D(18, 0) M(10) M(11) \
    M(0) M(1) D(19, 0) M(2) M(3) M(4) M(5) D(16, 0) M(6) M(7) M(8) M(9) D(17, 0) M(10) M(11)
#define PAT_D3_SAME_M0 \
    SETM0(16) M(0) M(1) M(2) M(3) DSAME(0) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) DSAME(1024) \
    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) DSAME(2048) M(8) M(9) M(10) M(11)
#define PAT_D6_SAME_M0 \
    SETM0(16) M(0) M(1) DSAME(0) M(2) M(3) M(4) M(5) DSAME(1024) M(6) M(7) M(8) M(9) DSAME(2048) M(10) M(11) \
    M(0) M(1) DSAME(3072) M(2) M(3) M(4) M(5) DSAME(0) M(6) M(7) M(8) M(9) DSAME(1024) M(10) M(11)
#define PAT_D3_BUNCHED D(16, 0) D(17, 0) D(18, 0) PURE
#define PAT_D3_BUNCHED_SAME SETM0(16) DSAME(0) DSAME(1024) DSAME(2048) PURE
#define PAT_D3_BACK PURE D(16, 0) D(17, 0) D(18, 0)
#define PAT_G3 \
    M(0) M(1) M(2) M(3) G(20, 0) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) G(21, 1024) \
    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) G(22, 2048) M(8) M(9) M(10) M(11)

#define KERNEL(NAME, PATTERN)                                                                      \
    __global__ __launch_bounds__(512) void NAME(float* out, const float* src, long long* cyc,     \
                                                int steps) {                                      \
        __shared__ f4 sh[8192];   /* 128 KiB: one block per CU */                                 \
        for (int i = threadIdx.x; i < 8192; i += blockDim.x) sh[i] = f4{1, 2, 3, 4};              \
        f4 acc[12], ld[3];                                                                        \
        for (int i = 0; i < 12; ++i) acc[i] = f4{0, 0, 0, 0};                                     \
        for (int i = 0; i < 3; ++i) ld[i] = f4{0, 0, 0, 0};                                       \
        float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;                                 \
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                        \
        const unsigned lane_bytes = (threadIdx.x & 63) * 16;                                      \
        const float* g = src + (size_t)wave * 4096;       /* 16 KiB per wave, L2 resident */      \
        const unsigned l0 = (unsigned)(size_t)(__attribute__((address_space(3))) f4*)sh +         \
                            wave * 16384;                                                         \
        const unsigned l1 = l0 + 4096, l2 = l0 + 8192, l3 = l0 + 12288;                           \
        __syncthreads();                                                                          \
        const long long t0 = __builtin_readcyclecounter();                                        \
        for (int s = 0; s < steps; ++s) {                                                         \
            asm volatile(PATTERN                                                                  \
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]),  \
                           "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]),  \
                           "+v"(acc[10]), "+v"(acc[11]), "+v"(a), "+v"(b)                         \
                         : "v"(lane_bytes), "s"(g), "s"(l0), "s"(l1), "s"(l2), "s"(l3),           \
                           "v"(ld[0]), "v"(ld[1]), "v"(ld[2])                                     \
                         : "memory");                                                             \
        }                                                                                         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
        const long long t1 = __builtin_readcyclecounter();                                        \
        __syncthreads();                                                                          \
        f4 s4 = sh[threadIdx.x] + ld[0] + ld[1] + ld[2];                                          \
        for (int i = 0; i < 12; ++i) s4 += acc[i];                                                \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s4.x + s4.y + s4.z + s4.w;                   \
        if ((threadIdx.x & 63) == 0) {                                                            \
            cyc[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2] = t0;                                   \
            cyc[(blockIdx.x * 8 + (threadIdx.x >> 6)) * 2 + 1] = t1;                               \
        }                                                                                         \
    }

KERNEL(k_pure, PURE)
KERNEL(k_d3, PAT_D3)
KERNEL(k_d6, PAT_D6)
KERNEL(k_d3_same, PAT_D3_SAME_M0)
KERNEL(k_d6_same, PAT_D6_SAME_M0)
KERNEL(k_d3_bunched, PAT_D3_BUNCHED)
KERNEL(k_d3_bunched_same, PAT_D3_BUNCHED_SAME)
KERNEL(k_d3_back, PAT_D3_BACK)
KERNEL(k_g3, PAT_G3)

// Where do the bytes land?  One wave copies 4 KiB with ONE M0 and offsets 0 / 1024 / 2048 / 3072.
__global__ __launch_bounds__(64) void k_where(const float* src, float* dst) {
    __shared__ f4 sh[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) sh[i] = f4{-1, -1, -1, -1};
    __syncthreads();
    const unsigned lane_bytes = threadIdx.x * 16;
    const unsigned l0 = (unsigned)(size_t)(__attribute__((address_space(3))) f4*)sh;
    asm volatile("s_mov_b32 m0, %0\ns_nop 0\n"
                 "global_load_lds_dwordx4 %1, %2 offset:0\n"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n"
                 "global_load_lds_dwordx4 %1, %2 offset:3072\n"
                 "s_waitcnt vmcnt(0)\n" ::"s"(l0), "v"(lane_bytes), "s"(src) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 64) reinterpret_cast<f4*>(dst)[i] = sh[i];
}

typedef void (*kern_t)(float*, const float*, long long*, int);

static double run(const char* name, kern_t k, double base, int dmas) {
    float *out, *src;
    long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&src, 8 * 4096 * 4);
    (void)hipMemset(src, 0, 8 * 4096 * 4);
    (void)hipMalloc(&cyc, 256 * 8 * 16);
    const int steps = 200;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, src, cyc, steps);
    (void)hipDeviceSynchronize();
    static long long h[256 * 16];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double span = 0;
    for (int b = 0; b < 256; ++b) {
        long long lo = h[b * 16], hi = h[b * 16 + 1];
        for (int w = 1; w < 8; ++w) {
            lo = std::min(lo, h[(b * 8 + w) * 2]);
            hi = std::max(hi, h[(b * 8 + w) * 2 + 1]);
        }
        span += double(hi - lo);
    }
    span /= 256;
    const double per_step = span / steps;        // cycles per step of 48 MFMAs per SIMD
    printf("%-58s %8.1f cycles/step", name, per_step);
    if (base > 0 && dmas > 0)
        printf("   = +%6.1f per request and wave (%d per wave and step)", (per_step - base) / dmas, dmas);
    printf("\n");
    (void)hipFree(out);
    (void)hipFree(src);
    (void)hipFree(cyc);
    return per_step;
}

int main() {
    const double base = run("24 MFMA per wave (48 per SIMD)", k_pure, 0, 0);
    run("+ 3 LDS-DMA pieces spread, M0 set for each", k_d3, base, 3);
    run("+ 6 LDS-DMA pieces spread, M0 set for each", k_d6, base, 6);
    run("+ 3 pieces spread, ONE M0, immediate offsets", k_d3_same, base, 3);
    run("+ 6 pieces spread, ONE M0, immediate offsets", k_d6_same, base, 6);
    run("+ 3 pieces bunched in front, M0 set for each", k_d3_bunched, base, 3);
    run("+ 3 pieces bunched in front, ONE M0", k_d3_bunched_same, base, 3);
    run("+ 3 pieces bunched behind, M0 set for each", k_d3_back, base, 3);
    run("+ 3 global_load_dwordx4 to registers, spread", k_g3, base, 3);
    // placement check
    std::vector<float> hsrc(1024 * 4), hdst(1024 * 4);
    for (size_t i = 0; i < hsrc.size(); ++i) hsrc[i] = (float)i;
    float *src, *dst;
    (void)hipMalloc(&src, hsrc.size() * 4);
    (void)hipMalloc(&dst, hdst.size() * 4);
    (void)hipMemcpy(src, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_where, dim3(1), dim3(64), 0, 0, src, dst);
    (void)hipMemcpy(hdst.data(), dst, hdst.size() * 4, hipMemcpyDeviceToHost);
    size_t same = 0;
    for (size_t i = 0; i < hsrc.size(); ++i) same += hdst[i] == hsrc[i];
    printf("one M0, offsets 0/1024/2048/3072: %zu of %zu floats where a plain copy puts them "
           "(LDS[256]=%g LDS[512]=%g LDS[768]=%g)\n", same, hsrc.size(), hdst[256], hdst[512], hdst[768]);
    return 0;
}
