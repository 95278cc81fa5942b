// Round-2 verdict, item 7 (bounded experiment; nothing of it ships): "fp32 by split bf16" as a
// stage-B tile loop.  One unit = what a wave does for one N tile of an F(4,3) layer: six GEMMs of
// [16 quads x 48 channels] . [48 x 16]:
//   fp32     72 x v_mfma_f32_16x16x4_f32 (the kernel's loop: three ds_read_b128 of B per step);
//   bf16x3  108 x v_mfma_f32_16x16x16_bf16: each operand cut into three bf16 pieces (hi + mid + lo),
//           six of the nine partial products kept (hh, hm, mh, hl, lh, mm), fp32 accumulation -
//           as exact as fp32 (tools/split_bf16_error.py).  A pieces in registers (the transformed
//           inputs U, cut once per layer), B pieces from LDS (cut on the host: 6 bytes per weight
//           instead of 4);
//   +split  the same with the cutting of U in the loop: 11 vector instructions per pair of values
//           (cvt_pk, 2 x widen, 2 x subtract, twice; cvt_pk), a third of a layer's 396 per unit;
//   bf16x2   54 MFMAs: two pieces, three products (hh, hm, mh) - 2e-5 of the output scale for
//           F(4,3), at the edge of the stage tolerance.
// Prints cycles per unit and per fp32-equivalent MFMA (cycles / 72; the fp32 pipe's limit is 32).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/split_bf16.hip -o tools/microbench/_build/split_bf16
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef short s4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f4 mfma_f32(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f4 mfma_bf16(s4 a, s4 b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
template <int OFF>
__device__ __forceinline__ f4 ds_read_f4(unsigned addr) {
    f4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ s4 ds_read_s4(unsigned addr) {
    s4 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// MODE 0 fp32; 1 bf16 x 3 pieces; 2 the same + the cutting of U; 3 bf16 x 2 pieces
template <int MODE>
__global__ __launch_bounds__(512) void k_unit(float* out, long long* cyc, int units) {
    __shared__ __attribute__((aligned(16))) float lds[24 * 1024];      // 96 KB: one workgroup per CU
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 24 * 1024; i += 512) lds[i] = 1.0f / (float)(1 + (i & 255));
    __syncthreads();
    const unsigned b_addr = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)lds +
                            (unsigned)lane * 16u;
    f4 acc[6];
    for (int x = 0; x < 6; ++x) acc[x] = f4{0.f, 0.f, 0.f, 0.f};
    // the A side: 72 fp32 values (fp32), or their bf16 pieces, four per register pair
    f2 u[6][6];
    s4 a_piece[3][6][3];          // [piece][matrix][K16 step]
    for (int x = 0; x < 6; ++x)
        for (int sp = 0; sp < 6; ++sp) u[x][sp] = f2{(float)(lane + x), (float)(sp + 1) * 0.5f};
    for (int p = 0; p < 3; ++p)
        for (int x = 0; x < 6; ++x)
            for (int s = 0; s < 3; ++s)
                a_piece[p][x][s] = s4{(short)(0x3f80 + lane + p), (short)(0x3f80 + x),
                                      (short)(0x3f80 + s), (short)0x3f80};
    const long long t0 = __builtin_readcyclecounter();
    for (int unit = 0; unit < units; ++unit) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int sp = 0; sp < 6; ++sp) {
                f4 b[3];
                b[0] = ds_read_f4<0>(b_addr + (unsigned)(sp * 3 + 0) * 1024u);
                b[1] = ds_read_f4<0>(b_addr + (unsigned)(sp * 3 + 1) * 1024u);
                b[2] = ds_read_f4<0>(b_addr + (unsigned)(sp * 3 + 2) * 1024u);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    acc[2 * p] = mfma_f32(u[2 * p][sp].x, b[p][0], acc[2 * p]);
                    acc[2 * p + 1] = mfma_f32(u[2 * p + 1][sp].x, b[p][2], acc[2 * p + 1]);
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    acc[2 * p] = mfma_f32(u[2 * p][sp].y, b[p][1], acc[2 * p]);
                    acc[2 * p + 1] = mfma_f32(u[2 * p + 1][sp].y, b[p][3], acc[2 * p + 1]);
                }
            }
        } else {
            constexpr int PIECES = MODE == 3 ? 2 : 3;
#pragma unroll
            for (int s = 0; s < 3; ++s) {           // K = 16 channels per step
                if constexpr (MODE == 2) {
                    // cut two of the step's six (matrix) value quartets: 2 x 2 pairs x 11 = 44
                    // vector instructions per step, 132 per unit = a third of a layer's 396
#pragma unroll
                    for (int x = 0; x < 2; ++x)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            float lo = u[x + 2 * (unit & 1)][2 * s + h].x, hi = u[x][2 * s + h].y;
                            const unsigned p0 = cvt_pk_bf16(lo, hi);
                            lo -= __builtin_bit_cast(float, p0 << 16);
                            hi -= __builtin_bit_cast(float, p0 & 0xffff0000u);
                            const unsigned p1 = cvt_pk_bf16(lo, hi);
                            lo -= __builtin_bit_cast(float, p1 << 16);
                            hi -= __builtin_bit_cast(float, p1 & 0xffff0000u);
                            const unsigned p2 = cvt_pk_bf16(lo, hi);
                            a_piece[0][x][s][2 * h] = (short)p0;
                            a_piece[1][x][s][2 * h] = (short)p1;
                            a_piece[2][x][s][2 * h] = (short)p2;
                        }
                }
                // the step's B pieces for all six matrices, then product by product ACROSS the
                // matrices: consecutive MFMAs write different accumulators (one accumulator six
                // times in a row is a dependent chain: 18 cycles per MFMA instead of 8)
                s4 b[6][PIECES];
#pragma unroll
                for (int x = 0; x < 6; ++x)
#pragma unroll
                    for (int p = 0; p < PIECES; ++p)
                        b[x][p] = ds_read_s4<0>(b_addr / 2u + (unsigned)(((s * 6 + x) * 3 + p) * 512));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int x = 0; x < 6; ++x) acc[x] = mfma_bf16(a_piece[0][x][s], b[x][0], acc[x]);
#pragma unroll
                for (int x = 0; x < 6; ++x) acc[x] = mfma_bf16(a_piece[0][x][s], b[x][1], acc[x]);
#pragma unroll
                for (int x = 0; x < 6; ++x) acc[x] = mfma_bf16(a_piece[1][x][s], b[x][0], acc[x]);
                if constexpr (PIECES == 3) {
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] = mfma_bf16(a_piece[0][x][s], b[x][2], acc[x]);
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] = mfma_bf16(a_piece[2][x][s], b[x][0], acc[x]);
#pragma unroll
                    for (int x = 0; x < 6; ++x) acc[x] = mfma_bf16(a_piece[1][x][s], b[x][1], acc[x]);
                }
            }
        }
#pragma unroll
        for (int x = 0; x < 6; ++x) asm volatile("" : "+v"(acc[x]));
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int x = 0; x < 6; ++x) s += acc[x][0] + acc[x][3];
    out[blockIdx.x * 512 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + (tid >> 6)] = t1 - t0;
}

template <int MODE>
double run(const char* name, int mfmas_per_unit) {
    const int blocks = 256, units = 2000;
    float* out;
    long long* cyc;
    hipMalloc(&out, blocks * 512 * sizeof(float));
    hipMalloc(&cyc, blocks * 8 * sizeof(long long));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_unit<MODE>, dim3(blocks), dim3(512), 0, 0, out, cyc, units);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(blocks * 8);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    // two waves per SIMD share the pipe: a SIMD does 2 units in the time a wave takes for one
    const double per_unit = (double)h[h.size() / 2] / units / 2.0;
    std::printf("%-34s %4d MFMAs per unit: %8.1f cycles per unit and SIMD = %6.2f per MFMA, "
                "%6.2f per fp32-equivalent MFMA (/72)\n",
                name, mfmas_per_unit, per_unit, per_unit / mfmas_per_unit, per_unit / 72.0);
    hipFree(out);
    hipFree(cyc);
    return per_unit;
}

int main() {
    const double fp32 = run<0>("fp32 (v_mfma_f32_16x16x4_f32)", 72);
    const double b3 = run<1>("bf16 x 3 pieces, 6 products", 108);
    const double b3s = run<2>("bf16 x 3 pieces + cutting U", 108);
    const double b2 = run<3>("bf16 x 2 pieces, 3 products", 54);
    std::printf("speed-up over fp32: x3 %.2f, x3 with the cutting %.2f, x2 %.2f\n", fp32 / b3,
                fp32 / b3s, fp32 / b2);
    return 0;
}
