// Micro-benchmark: what does it cost to issue VALU / LDS instructions between the
// v_mfma_f32_16x16x4_f32 of a 512-thread block (2 waves per SIMD, one block per CU, the shape of
// dbh_forward_kernel)?  Every pattern is one hand-written asm block of 24 MFMAs (12 accumulators,
// each used twice) so the compiler cannot move anything; the figure printed is the block's span
// (first wave's start to last wave's end) per MFMA per SIMD - 32.0 is the matrix pipe's limit.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_issue.hip -o tools/microbench/_build/mfma_issue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// operands: %0-%11 acc, %12 a, %13 b, %14-%17 scalar VALU chains, %18-%21 packed VALU chains
//           (also the ds_read_b64 destinations), %22 k1, %23 k2, %24 pk1, %25 pk2,
//           %26-%29 ds_read_b128 destinations, %30 lds address, %31 global address (per lane)
#define M(i) "v_mfma_f32_16x16x4_f32 %" #i ", %12, %13, %" #i "\n"
#define V(j) "v_fma_f32 %" #j ", %" #j ", %22, %23\n"
#define P(j) "v_pk_fma_f32 %" #j ", %" #j ", %24, %25\n"
#define L128(d, off) "ds_read_b128 %" #d ", %30 offset:" #off "\n"
#define L64(d, off) "ds_read_b64 %" #d ", %30 offset:" #off "\n"
#define SN "s_nop 0\n"
#define G2(d, off) "global_load_dwordx2 %" #d ", %31, off offset:" #off "\n"
#define G4(d, off) "global_load_dwordx4 %" #d ", %31, off offset:" #off "\n"

#define PAT_PURE \
    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) \
    M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11)
#define X1(i, v) M(i) V(v)
#define PAT_V1 \
    X1(0,14) X1(1,15) X1(2,16) X1(3,17) X1(4,14) X1(5,15) X1(6,16) X1(7,17) X1(8,14) X1(9,15) X1(10,16) X1(11,17) \
    X1(0,14) X1(1,15) X1(2,16) X1(3,17) X1(4,14) X1(5,15) X1(6,16) X1(7,17) X1(8,14) X1(9,15) X1(10,16) X1(11,17)
#define X2(i, v, w) M(i) V(v) V(w)
#define PAT_V2 \
    X2(0,14,15) X2(1,16,17) X2(2,14,15) X2(3,16,17) X2(4,14,15) X2(5,16,17) X2(6,14,15) X2(7,16,17) X2(8,14,15) X2(9,16,17) X2(10,14,15) X2(11,16,17) \
    X2(0,14,15) X2(1,16,17) X2(2,14,15) X2(3,16,17) X2(4,14,15) X2(5,16,17) X2(6,14,15) X2(7,16,17) X2(8,14,15) X2(9,16,17) X2(10,14,15) X2(11,16,17)
#define X4(i) M(i) V(14) V(15) V(16) V(17)
#define PAT_V4 \
    X4(0) X4(1) X4(2) X4(3) X4(4) X4(5) X4(6) X4(7) X4(8) X4(9) X4(10) X4(11) \
    X4(0) X4(1) X4(2) X4(3) X4(4) X4(5) X4(6) X4(7) X4(8) X4(9) X4(10) X4(11)
#define XP(i, v) M(i) P(v)
#define PAT_P1 \
    XP(0,18) XP(1,19) XP(2,20) XP(3,21) XP(4,18) XP(5,19) XP(6,20) XP(7,21) XP(8,18) XP(9,19) XP(10,20) XP(11,21) \
    XP(0,18) XP(1,19) XP(2,20) XP(3,21) XP(4,18) XP(5,19) XP(6,20) XP(7,21) XP(8,18) XP(9,19) XP(10,20) XP(11,21)
// 24 VALU first, then the 24 MFMAs back to back
#define PAT_V1_BUNCHED \
    V(14) V(15) V(16) V(17) V(14) V(15) V(16) V(17) V(14) V(15) V(16) V(17) \
    V(14) V(15) V(16) V(17) V(14) V(15) V(16) V(17) V(14) V(15) V(16) V(17) PAT_PURE
// 12 packed VALU first (the same 24 flops per lane as PAT_V1_BUNCHED), then the 24 MFMAs
#define PAT_P_BUNCHED \
    P(18) P(19) P(20) P(21) P(18) P(19) P(20) P(21) P(18) P(19) P(20) P(21) PAT_PURE
#define PAT_P24_BUNCHED \
    P(18) P(19) P(20) P(21) P(18) P(19) P(20) P(21) P(18) P(19) P(20) P(21) \
    P(18) P(19) P(20) P(21) P(18) P(19) P(20) P(21) P(18) P(19) P(20) P(21) PAT_PURE
// 12 LDS reads, one after every second MFMA
#define PAT_L128 \
    M(0) M(1) L128(26,0) M(2) M(3) L128(27,1024) M(4) M(5) L128(28,2048) M(6) M(7) L128(29,3072) \
    M(8) M(9) L128(26,4096) M(10) M(11) L128(27,5120) M(0) M(1) L128(28,6144) M(2) M(3) L128(29,7168) \
    M(4) M(5) L128(26,8192) M(6) M(7) L128(27,9216) M(8) M(9) L128(28,10240) M(10) M(11) L128(29,11264)
#define PAT_L64 \
    M(0) M(1) L64(18,0) M(2) M(3) L64(19,1024) M(4) M(5) L64(20,2048) M(6) M(7) L64(21,3072) \
    M(8) M(9) L64(18,4096) M(10) M(11) L64(19,5120) M(0) M(1) L64(20,6144) M(2) M(3) L64(21,7168) \
    M(4) M(5) L64(18,8192) M(6) M(7) L64(19,9216) M(8) M(9) L64(20,10240) M(10) M(11) L64(21,11264)
// 12 LDS reads bunched at the top of the step
#define PAT_L128_BUNCHED \
    L128(26,0) L128(27,1024) L128(28,2048) L128(29,3072) L128(26,4096) L128(27,5120) \
    L128(28,6144) L128(29,7168) L128(26,8192) L128(27,9216) L128(28,10240) L128(29,11264) PAT_PURE
// the F(4,3) phase-0 mix: 24 MFMA, 22 VALU, 6 b128 + 5 b64
#define PAT_MIX \
    M(0) V(14) M(1) V(15) L128(26,0) M(2) V(16) M(3) V(17) L64(18,1024) M(4) V(14) M(5) V(15) L128(27,2048) \
    M(6) V(16) M(7) V(17) L64(19,3072) M(8) V(14) M(9) V(15) L128(28,4096) M(10) V(16) M(11) V(17) L64(20,5120) \
    M(0) V(14) M(1) V(15) L128(29,6144) M(2) V(16) M(3) V(17) L64(21,7168) M(4) V(14) M(5) V(15) L128(26,8192) \
    M(6) V(16) M(7) V(17) L64(18,9216) M(8) V(14) M(9) V(15) L128(27,10240) M(10) M(11)
// the same with packed transforms: 11 v_pk_fma
#define PAT_MIX_PK \
    M(0) M(1) L128(26,0) M(2) P(18) M(3) L64(19,1024) M(4) M(5) P(20) L128(27,2048) \
    M(6) M(7) P(21) L64(18,3072) M(8) M(9) P(19) L128(28,4096) M(10) M(11) P(20) L64(21,5120) \
    M(0) M(1) P(18) L128(29,6144) M(2) M(3) P(19) L64(20,7168) M(4) M(5) P(21) L128(26,8192) \
    M(6) M(7) P(18) L64(19,9216) M(8) M(9) P(20) L128(27,10240) M(10) P(21) M(11)
// global loads to registers (L2-resident data): 5 x 8 bytes or 3 x 16 bytes per lane per step
#define PAT_G2 \
    M(0) M(1) M(2) M(3) G2(18,0) M(4) M(5) M(6) M(7) G2(19,512) M(8) M(9) M(10) M(11) G2(20,1024) \
    M(0) M(1) M(2) M(3) G2(21,1536) M(4) M(5) M(6) M(7) G2(18,2048) M(8) M(9) M(10) M(11)
#define PAT_G4 \
    M(0) M(1) M(2) M(3) M(4) M(5) G4(26,0) M(6) M(7) M(8) M(9) M(10) M(11) G4(27,1024) \
    M(0) M(1) M(2) M(3) M(4) M(5) G4(28,2048) M(6) M(7) M(8) M(9) M(10) M(11)
#define PAT_G2_BUNCHED \
    G2(18,0) G2(19,512) G2(20,1024) G2(21,1536) G2(18,2048) PAT_PURE
// the F(4,3) phase-0 step as the kernel has it: 4 b64 + 6 b128, 14 v_fma in one bunch, 24 MFMAs
#define V14 V(14) V(15) V(16) V(17) V(14) V(15) V(16) V(17) V(14) V(15) V(16) V(17) V(14) V(15)
#define PAT_STEP_FRONT \
    L64(18,0) L64(19,1024) L64(20,2048) L64(21,3072) L128(26,4096) L128(27,5120) L128(28,6144) \
    L128(29,7168) L128(26,8192) L128(27,9216) V14 PAT_PURE
// the same work software-pipelined inside the wave: the VALU bunch in the middle of the MFMA
// block, the LDS reads one after every second MFMA
#define PAT_STEP_MID \
    M(0) M(1) L64(18,0) M(2) M(3) L64(19,1024) M(4) M(5) L64(20,2048) M(6) M(7) L64(21,3072) \
    M(8) M(9) L128(26,4096) M(10) M(11) V14 M(0) M(1) L128(27,5120) M(2) M(3) L128(28,6144) \
    M(4) M(5) L128(29,7168) M(6) M(7) L128(26,8192) M(8) M(9) L128(27,9216) M(10) M(11)
// two bunches of 7
#define V7 V(14) V(15) V(16) V(17) V(14) V(15) V(16)
#define PAT_STEP_MID2 \
    M(0) M(1) L64(18,0) M(2) M(3) L64(19,1024) M(4) M(5) L64(20,2048) M(6) M(7) V7 L64(21,3072) \
    M(8) M(9) L128(26,4096) M(10) M(11) M(0) M(1) L128(27,5120) M(2) M(3) L128(28,6144) \
    M(4) M(5) L128(29,7168) M(6) M(7) V7 L128(26,8192) M(8) M(9) L128(27,9216) M(10) M(11)
// dependent chains: 24 MFMAs over 1, 2, 3, 4, 6 accumulators (reuse distance = that many issues)
#define PAT_DEP1 M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0)
#define PAT_DEP2 M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1) M(0) M(1)
#define PAT_DEP3 M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2) M(0) M(1) M(2)
#define PAT_DEP4 M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3) M(0) M(1) M(2) M(3)
#define PAT_DEP6 M(0) M(1) M(2) M(3) M(4) M(5) M(0) M(1) M(2) M(3) M(4) M(5) M(0) M(1) M(2) M(3) M(4) M(5) M(0) M(1) M(2) M(3) M(4) M(5)
// s_nop between MFMAs (pure issue-slot cost)
#define XS(i) M(i) SN
#define PAT_SNOP \
    XS(0) XS(1) XS(2) XS(3) XS(4) XS(5) XS(6) XS(7) XS(8) XS(9) XS(10) XS(11) \
    XS(0) XS(1) XS(2) XS(3) XS(4) XS(5) XS(6) XS(7) XS(8) XS(9) XS(10) XS(11)

#define KERNEL(NAME, PATTERN) KERNEL_T(NAME, PATTERN, 512)
#define KERNEL_T(NAME, PATTERN, THREADS)                                                           \
    __global__ __launch_bounds__(THREADS) void NAME(float* out, long long* cyc, int steps) {      \
        __shared__ f4 sh[6144];   /* 96 KiB: one block per CU */                                  \
        for (int i = threadIdx.x; i < 1024; i += blockDim.x) sh[i] = f4{1, 2, 3, 4};              \
        f4 acc[12], ld[4];                                                                        \
        for (int i = 0; i < 12; ++i) acc[i] = f4{0, 0, 0, 0};                                     \
        for (int i = 0; i < 4; ++i) ld[i] = f4{0, 0, 0, 0};                                       \
        float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f, v[4], k1 = 1.0001f, k2 = 0.5f;  \
        f2 pv[4], pk1 = f2{1.0001f, 1.0001f}, pk2 = f2{0.5f, 0.5f};                               \
        for (int i = 0; i < 4; ++i) { v[i] = threadIdx.x + i; pv[i] = f2{v[i], v[i] + 1}; }       \
        const unsigned addr = (unsigned)(size_t)sh + (threadIdx.x & 63) * 16;                     \
        const float* gp = out + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 1024;              \
        __syncthreads();                                                                          \
        const long long t0 = __builtin_readcyclecounter();                                        \
        for (int s = 0; s < steps; ++s) {                                                         \
            asm volatile(PATTERN "s_waitcnt lgkmcnt(0)\n"                                        \
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]),  \
                           "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]), "+v"(acc[8]), "+v"(acc[9]),  \
                           "+v"(acc[10]), "+v"(acc[11]), "+v"(a), "+v"(b), "+v"(v[0]),            \
                           "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(pv[0]), "+v"(pv[1]),          \
                           "+v"(pv[2]), "+v"(pv[3]), "+v"(k1), "+v"(k2), "+v"(pk1), "+v"(pk2),    \
                           "+v"(ld[0]), "+v"(ld[1]), "+v"(ld[2]), "+v"(ld[3])                     \
                         : "v"(addr), "v"(gp)                                                     \
                         : "memory");                                                             \
        }                                                                                         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
        const long long t1 = __builtin_readcyclecounter();                                        \
        __syncthreads();                                                                          \
        f4 s4 = ld[0] + ld[1] + ld[2] + ld[3];                                                    \
        for (int i = 0; i < 12; ++i) s4 += acc[i];                                                \
        float sv = 0;                                                                             \
        for (int i = 0; i < 4; ++i) sv += v[i] + pv[i].x + pv[i].y;                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s4.x + s4.y + s4.z + s4.w + sv;              \
        if ((threadIdx.x & 63) == 0) {                                                            \
            cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0;                                  \
            cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1;                              \
        }                                                                                         \
    }

KERNEL(k_pure, PAT_PURE)
KERNEL(k_v1, PAT_V1)
KERNEL(k_v2, PAT_V2)
KERNEL(k_v4, PAT_V4)
KERNEL(k_p1, PAT_P1)
KERNEL(k_v1_bunched, PAT_V1_BUNCHED)
KERNEL(k_p_bunched, PAT_P_BUNCHED)
KERNEL(k_p24_bunched, PAT_P24_BUNCHED)
KERNEL(k_l128, PAT_L128)
KERNEL(k_l64, PAT_L64)
KERNEL(k_l128_bunched, PAT_L128_BUNCHED)
KERNEL(k_mix, PAT_MIX)
KERNEL(k_mix_pk, PAT_MIX_PK)
KERNEL(k_snop, PAT_SNOP)
KERNEL_T(k_dep1, PAT_DEP1, 256)
KERNEL_T(k_dep2, PAT_DEP2, 256)
KERNEL_T(k_dep3, PAT_DEP3, 256)
KERNEL_T(k_dep4, PAT_DEP4, 256)
KERNEL_T(k_dep6, PAT_DEP6, 256)
KERNEL(k_dep2_2w, PAT_DEP2)
KERNEL(k_dep3_2w, PAT_DEP3)
KERNEL_T(k_pure_1wave, PAT_PURE, 256)
KERNEL_T(k_pure_3wave, PAT_PURE, 768)
KERNEL_T(k_pure_4wave, PAT_PURE, 1024)
KERNEL_T(k_front_1wave, PAT_STEP_FRONT, 256)
KERNEL_T(k_front_4wave, PAT_STEP_FRONT, 1024)
KERNEL_T(k_mid_1wave, PAT_STEP_MID, 256)
KERNEL_T(k_mid2_1wave, PAT_STEP_MID2, 256)
KERNEL_T(k_v1_1wave, PAT_V1, 256)
KERNEL_T(k_l128_1wave, PAT_L128, 256)
KERNEL(k_step_front, PAT_STEP_FRONT)
KERNEL(k_step_mid, PAT_STEP_MID)
KERNEL(k_step_mid2, PAT_STEP_MID2)
KERNEL(k_g2, PAT_G2)
KERNEL(k_g4, PAT_G4)
KERNEL(k_g2b, PAT_G2_BUNCHED)

typedef void (*kern_t)(float*, long long*, int);

static void run(const char* name, kern_t k, int threads = 512) {
    float* out;
    long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMalloc(&cyc, 256 * 16 * 16);
    const int steps = 200;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, steps);
    (void)hipDeviceSynchronize();
    static long long h[256 * 32];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double span = 0, w0 = 0;
    for (int b = 0; b < 256; ++b) {
        long long lo = h[b * 32], hi = h[b * 32 + 1];
        for (int w = 1; w < threads / 64; ++w) {
            lo = std::min(lo, h[(b * 16 + w) * 2]);
            hi = std::max(hi, h[(b * 16 + w) * 2 + 1]);
        }
        span += double(hi - lo);
        w0 += double(h[b * 32 + 1] - h[b * 32]);
    }
    span /= 256;
    w0 /= 256;
    printf("%-60s %6.2f cycles/MFMA/SIMD   (wave 0: %6.2f per own MFMA)\n", name,
           span / (steps * 24.0 * (threads / 256.0)), w0 / (steps * 24.0));
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    run("24 MFMA", k_pure);
    run("24 MFMA, s_nop after each", k_snop);
    run("24 MFMA, 1 v_fma after each", k_v1);
    run("24 MFMA, 2 v_fma after each", k_v2);
    run("24 MFMA, 4 v_fma after each", k_v4);
    run("24 MFMA, 1 v_pk_fma after each", k_p1);
    run("24 v_fma bunched, then 24 MFMA", k_v1_bunched);
    run("12 v_pk_fma bunched, then 24 MFMA", k_p_bunched);
    run("24 v_pk_fma bunched, then 24 MFMA", k_p24_bunched);
    run("24 MFMA, ds_read_b128 after every 2nd", k_l128);
    run("24 MFMA, ds_read_b64 after every 2nd", k_l64);
    run("12 ds_read_b128 bunched, then 24 MFMA", k_l128_bunched);
    run("F(4,3) mix: 24 MFMA + 22 v_fma + 6 b128 + 5 b64", k_mix);
    run("F(4,3) mix packed: 24 MFMA + 11 v_pk_fma + 6 b128 + 5 b64", k_mix_pk);
    run("24 MFMA, 5 global_load_dwordx2 spread", k_g2);
    run("24 MFMA, 3 global_load_dwordx4 spread", k_g4);
    run("5 global_load_dwordx2 bunched, then 24 MFMA", k_g2b);
    run("24 MFMA, 1 wave per SIMD", k_pure_1wave, 256);
    run("24 MFMA, 3 waves per SIMD", k_pure_3wave, 768);
    run("24 MFMA, 4 waves per SIMD", k_pure_4wave, 1024);
    run("1 wave/SIMD, accumulator reused every MFMA", k_dep1, 256);
    run("1 wave/SIMD, accumulator reused every 2nd MFMA", k_dep2, 256);
    run("1 wave/SIMD, accumulator reused every 3rd MFMA", k_dep3, 256);
    run("1 wave/SIMD, accumulator reused every 4th MFMA", k_dep4, 256);
    run("1 wave/SIMD, accumulator reused every 6th MFMA", k_dep6, 256);
    run("2 waves/SIMD, accumulator reused every 2nd MFMA", k_dep2_2w);
    run("2 waves/SIMD, accumulator reused every 3rd MFMA", k_dep3_2w);
    run("step (front), 1 wave per SIMD", k_front_1wave, 256);
    run("step (front), 4 waves per SIMD", k_front_4wave, 1024);
    run("step (mid), 1 wave per SIMD", k_mid_1wave, 256);
    run("step (2 bunches), 1 wave per SIMD", k_mid2_1wave, 256);
    run("24 MFMA + 1 v_fma after each, 1 wave per SIMD", k_v1_1wave, 256);
    run("24 MFMA + ds_read_b128 after every 2nd, 1 wave per SIMD", k_l128_1wave, 256);
    run("step: 10 LDS, 14 v_fma, then 24 MFMA", k_step_front);
    run("step: 12 MFMA, 14 v_fma, 12 MFMA, LDS spread", k_step_mid);
    run("step: 2 bunches of 7 v_fma inside 24 MFMA, LDS spread", k_step_mid2);
    return 0;
}
