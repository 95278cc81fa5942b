// Micro-benchmark of the stage-B tile loop of dbh_forward.hip (w43_tile): per step three
// ds_read_b128 of B fragments (double-buffered, hand-counted waits) and twelve
// v_mfma_f32_16x16x4_f32 whose A operands are 72 register-resident values.  Variants isolate what
// costs what: B as six ds_read_b64, no LDS at all, one wave per SIMD, reuse distance of the
// accumulators.  Prints cycles per MFMA per SIMD (32.0 = the matrix pipe's limit).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/tile_loop.hip -o tools/microbench/_build/tile_loop
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
template <int OFF>
__device__ __forceinline__ f4 ds_read_f4(unsigned addr) {
    f4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ f2 ds_read_f2(unsigned addr) {
    f2 v;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF) : "memory");
    return v;
}
struct U72 { f2 u[6][6]; };

// MODE 0: 3 x b128 per step; 1: 6 x b64 per step; 2: no LDS (B fixed in registers)
template <int MODE, int SP>
__device__ __forceinline__ void load_b(f4 (&b)[3], unsigned addr) {
    if constexpr (MODE == 0) {
        b[0] = ds_read_f4<((SP * 3 + 0) * 256) * 4>(addr);
        b[1] = ds_read_f4<((SP * 3 + 1) * 256) * 4>(addr);
        b[2] = ds_read_f4<((SP * 3 + 2) * 256) * 4>(addr);
    } else if constexpr (MODE == 1) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            f2 lo, hi;
            if (p == 0) { lo = ds_read_f2<((SP * 6 + 0) * 128) * 4>(addr); hi = ds_read_f2<((SP * 6 + 1) * 128) * 4>(addr); }
            if (p == 1) { lo = ds_read_f2<((SP * 6 + 2) * 128) * 4>(addr); hi = ds_read_f2<((SP * 6 + 3) * 128) * 4>(addr); }
            if (p == 2) { lo = ds_read_f2<((SP * 6 + 4) * 128) * 4>(addr); hi = ds_read_f2<((SP * 6 + 5) * 128) * 4>(addr); }
            b[p] = f4{lo.x, lo.y, hi.x, hi.y};
        }
    }
}
template <int MODE, int SP>
__device__ __forceinline__ void step(const U72& U, unsigned addr, f4 (&buf)[2][3], f4 (&acc)[6]) {
    if constexpr (MODE != 2) {
        if constexpr (SP + 1 < 6) {
            load_b<MODE, SP + 1>(buf[(SP + 1) & 1], addr);
            if (MODE == 0) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    f4(&b)[3] = buf[MODE == 2 ? 0 : (SP & 1)];
#pragma unroll
    for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(b[p]));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        acc[2 * p] = mfma4(U.u[2 * p][SP].x, b[p][0], acc[2 * p]);
        acc[2 * p + 1] = mfma4(U.u[2 * p + 1][SP].x, b[p][2], acc[2 * p + 1]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        acc[2 * p] = mfma4(U.u[2 * p][SP].y, b[p][1], acc[2 * p]);
        acc[2 * p + 1] = mfma4(U.u[2 * p + 1][SP].y, b[p][3], acc[2 * p + 1]);
    }
#pragma unroll
    for (int x = 0; x < 6; ++x) asm volatile("" : "+v"(acc[x]));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SP + 1 < 6) step<MODE, SP + 1>(U, addr, buf, acc);
}

template <int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_tile(float* out, long long* cyc, int tiles) {
    __shared__ f4 sh[6144];   // 96 KiB: one block per CU
    for (int i = threadIdx.x; i < 6144; i += blockDim.x) sh[i] = f4{1, 2, 3, 4} * (1.f / (1 + i));
    U72 U;
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int s = 0; s < 6; ++s) U.u[x][s] = f2{threadIdx.x * 0.001f + x, s * 0.01f};
    f4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f4{0, 0, 0, 0};
    const unsigned addr = (unsigned)(size_t)sh + (threadIdx.x & 63) * 16;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < tiles; ++t) {
        f4 buf[2][3];
        if (MODE == 2) { buf[0][0] = f4{1, 2, 3, 4}; buf[0][1] = f4{2, 3, 4, 5}; buf[0][2] = f4{3, 4, 5, 6}; }
        else load_b<MODE, 0>(buf[0], addr);
        step<MODE, 0>(U, addr, buf, acc);
    }
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    f4 s4 = acc[0] + acc[1] + acc[2] + acc[3] + acc[4] + acc[5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s4.x + s4.y + s4.z + s4.w;
    if ((threadIdx.x & 63) == 0) {
        cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0;
        cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1;
        // HW_ID (hardware register 4): SIMD_ID = bits 5:4, CU_ID = bits 11:8
        if (blockIdx.x == 0) cyc[256 * 32 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}


// ---------------------------------------------------------------------------------------------
// The ladder: the same loop with the ingredients of dbh_forward.hip's stage-B tiles 1 and 2 added
// one at a time (FLAGS: 1 = s_setprio by progress through a pair of tiles, 2 = the F(4,3) output
// transform + ReLU of the tile before (10 packed + 8 scalar VALU per half) inside steps 1 and 3,
// 4 = its 8 ds_write_b32 per half, 8 = transform WITHOUT the eight v_max), to see which of them
// takes the loop from 32.6 cycles per MFMA to the 38 the kernel's timeline shows.
// ---------------------------------------------------------------------------------------------
template <int FLAGS, int STEP0, int SP>
__device__ __forceinline__ void lstep(const U72& U, unsigned addr, f4 (&buf)[2][3], f4 (&acc)[6],
                                      const f4 (&prev)[6], float* out_lane, f2& sink) {
    if constexpr (SP + 1 < 6) {
        load_b<0, SP + 1>(buf[(SP + 1) & 1], addr);
        asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    f4(&b)[3] = buf[SP & 1];
#pragma unroll
    for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(b[p]));
    if constexpr (FLAGS & 1) __builtin_amdgcn_s_setprio(3 - (4 * (STEP0 + SP)) / 12);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        acc[2 * p] = mfma4(U.u[2 * p][SP].x, b[p][0], acc[2 * p]);
        acc[2 * p + 1] = mfma4(U.u[2 * p + 1][SP].x, b[p][2], acc[2 * p + 1]);
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        acc[2 * p] = mfma4(U.u[2 * p][SP].y, b[p][1], acc[2 * p]);
        acc[2 * p + 1] = mfma4(U.u[2 * p + 1][SP].y, b[p][3], acc[2 * p + 1]);
    }
#pragma unroll
    for (int x = 0; x < 6; ++x) asm volatile("" : "+v"(acc[x]));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr ((FLAGS & (2 | 4 | 8)) != 0 && (SP == 1 || SP == 3)) {
        constexpr int h = SP == 1 ? 0 : 1;
        // 16 / 32: the epilogue block at the highest / lowest priority, then back to the step's
        if constexpr (FLAGS & 16) __builtin_amdgcn_s_setprio(3);
        if constexpr (FLAGS & 32) __builtin_amdgcn_s_setprio(0);
        f2 y0 = f2{prev[0][2 * h], prev[0][2 * h + 1]}, y1 = y0, y2 = y0, y3 = y0;
        if constexpr (FLAGS & (2 | 8)) {
            const f2 k2 = f2{2.f, 2.f}, k4 = f2{4.f, 4.f}, k8 = f2{8.f, 8.f};
            const f2 a0 = f2{prev[0][2 * h], prev[0][2 * h + 1]}, a1 = f2{prev[1][2 * h], prev[1][2 * h + 1]};
            const f2 a2 = f2{prev[2][2 * h], prev[2][2 * h + 1]}, a3 = f2{prev[3][2 * h], prev[3][2 * h + 1]};
            const f2 a4 = f2{prev[4][2 * h], prev[4][2 * h + 1]}, a5 = f2{prev[5][2 * h], prev[5][2 * h + 1]};
            const f2 s12 = a1 + a2, d12 = a1 - a2, s34 = a3 + a4, d34 = a3 - a4;
            y0 = a0 + s12 + s34;
            y1 = __builtin_elementwise_fma(k2, d34, d12);
            y2 = __builtin_elementwise_fma(k4, s34, s12);
            y3 = __builtin_elementwise_fma(k8, d34, d12) + a5;
            if constexpr (FLAGS & 2) {
                y0 = f2{fmaxf(y0.x, 0.f), fmaxf(y0.y, 0.f)};
                y1 = f2{fmaxf(y1.x, 0.f), fmaxf(y1.y, 0.f)};
                y2 = f2{fmaxf(y2.x, 0.f), fmaxf(y2.y, 0.f)};
                y3 = f2{fmaxf(y3.x, 0.f), fmaxf(y3.y, 0.f)};
            }
        }
        if constexpr (FLAGS & 4) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float* dst = out_lane + (8 * h + e) * 4 * 50;
                dst[0] = y0[e];
                dst[50] = y1[e];
                dst[100] = y2[e];
                dst[150] = y3[e];
            }
        } else {
            sink = sink + y0 + y1 + y2 + y3;
        }
        if constexpr (FLAGS & (16 | 32)) __builtin_amdgcn_s_setprio(3 - (4 * (STEP0 + SP)) / 12);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SP + 1 < 6) lstep<FLAGS, STEP0, SP + 1>(U, addr, buf, acc, prev, out_lane, sink);
}

template <int FLAGS>
__global__ __launch_bounds__(512) void k_ladder(float* out, long long* cyc, int tiles) {
    __shared__ f4 sh[6144 + 1700];   // 96 KiB of "weights" + 27 KB of "activations"
    for (int i = threadIdx.x; i < 6144 + 1700; i += blockDim.x) sh[i] = f4{1, 2, 3, 4} * (1.f / (1 + i));
    U72 U;
#pragma unroll
    for (int x = 0; x < 6; ++x)
#pragma unroll
        for (int s = 0; s < 6; ++s) U.u[x][s] = f2{threadIdx.x * 0.001f + x, s * 0.01f};
    f4 acc[2][6];
    for (int i = 0; i < 6; ++i) acc[0][i] = acc[1][i] = f4{0, 0, 0, 0};
    const unsigned addr = (unsigned)(size_t)sh + (threadIdx.x & 63) * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* out_lane = reinterpret_cast<float*>(sh + 6144) + (wave * 16 + 2 * (lane >> 4)) * 4 * 50 % 6000 + (lane & 15);
    f2 sink = f2{0.f, 0.f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int t = 0; t < tiles; t += 2) {
        f4 buf[2][3];
        load_b<0, 0>(buf[0], addr);
        lstep<FLAGS, 0, 0>(U, addr, buf, acc[1], acc[0], out_lane, sink);
        load_b<0, 0>(buf[0], addr);
        lstep<FLAGS, 6, 0>(U, addr, buf, acc[0], acc[1], out_lane, sink);
    }
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    f4 s4 = acc[0][0] + acc[0][1] + acc[0][2] + acc[0][3] + acc[0][4] + acc[0][5] + acc[1][0] + acc[1][5];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s4.x + s4.y + s4.z + s4.w + sink.x + sink.y;
    if ((threadIdx.x & 63) == 0) {
        cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0;
        cyc[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1;
        if (blockIdx.x == 0) cyc[256 * 32 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    }
}

typedef void (*kern_t)(float*, long long*, int);
static void run(const char* name, kern_t k, int threads) {
    float* out;
    long long* cyc;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMalloc(&cyc, 256 * 16 * 16 + 16 * 8);
    const int tiles = 100;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, tiles);
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
    printf("%-64s %6.2f cycles/MFMA/SIMD   (wave 0: %6.2f per own MFMA)\n", name,
           span / (tiles * 72.0 * (threads / 256.0)), w0 / (tiles * 72.0));
    // block 0, wave by wave: which SIMD it ran on, when it started and ended (relative to the
    // block's first start): how the two waves of a SIMD share the matrix pipe
    long long ids[16];
    (void)hipMemcpy(ids, cyc + 256 * 32, sizeof(ids), hipMemcpyDeviceToHost);
    long long first = h[0];
    for (int w = 1; w < threads / 64; ++w) first = std::min(first, h[w * 2]);
    for (int w = 0; w < threads / 64; ++w)
        printf("    wave %d  simd %lld  cu %lld  start %8lld  end %8lld  (%.2f per own MFMA)\n", w,
               (ids[w] >> 4) & 3, (ids[w] >> 8) & 15, h[w * 2] - first, h[w * 2 + 1] - first,
               double(h[w * 2 + 1] - h[w * 2]) / (tiles * 72.0));
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    run("tile loop: 3 b128 + 12 MFMA per step, 2 waves/SIMD", k_tile<0, 512>, 512);
    run("tile loop: 3 b128 + 12 MFMA per step, 1 wave/SIMD", k_tile<0, 256>, 256);
    run("tile loop: 6 b64 + 12 MFMA per step, 2 waves/SIMD", k_tile<1, 512>, 512);
    run("tile loop: 6 b64 + 12 MFMA per step, 1 wave/SIMD", k_tile<1, 256>, 256);
    run("tile loop: no LDS, 12 MFMA per step, 2 waves/SIMD", k_tile<2, 512>, 512);
    run("tile loop: no LDS, 12 MFMA per step, 1 wave/SIMD", k_tile<2, 256>, 256);
    run("ladder 0: plain", k_ladder<0>, 512);
    run("ladder 1: + setprio by progress", k_ladder<1>, 512);
    run("ladder 3: + setprio + transform + ReLU (VALU only)", k_ladder<3>, 512);
    run("ladder 9: + setprio + transform, no ReLU (VALU only)", k_ladder<9>, 512);
    run("ladder 5: + setprio + stores only", k_ladder<5>, 512);
    run("ladder 7: + setprio + transform + ReLU + stores", k_ladder<7>, 512);
    run("ladder 6: transform + ReLU + stores, no setprio", k_ladder<6>, 512);
    run("ladder 23: 7 with the epilogue block at priority 3", k_ladder<23>, 512);
    run("ladder 39: 7 with the epilogue block at priority 0", k_ladder<39>, 512);
    return 0;
}
