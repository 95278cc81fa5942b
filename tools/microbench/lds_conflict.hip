// Micro-benchmark: LDS cycles per ds_read_b64 / ds_read_b128 for the MFMA-fragment access pattern
// lane -> base + (lane & 15) * n_stride + (lane >> 4) * q_stride, as a function of the activation
// row stride.  One 512-thread block per CU, every wave issues 64 reads back to back.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_conflict.hip -o tools/microbench/_build/lds_conflict
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int WIDTH>
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int n_stride, int q_stride) {
    extern __shared__ float sh[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) sh[i] = i;
    const int lane = threadIdx.x & 63;
    const unsigned addr = (unsigned)(size_t)sh + (lane & 15) * n_stride + (lane >> 4) * q_stride;
    f4 acc = f4{0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 16; ++it) {
        f4 a = f4{0, 0, 0, 0}, b = a, c = a, d = a;
        if (WIDTH == 8) {
            f2 x, y, z, w;
            asm volatile("ds_read_b64 %0, %4\nds_read_b64 %1, %4 offset:8\n"
                         "ds_read_b64 %2, %4 offset:16\nds_read_b64 %3, %4 offset:24\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(x), "=&v"(y), "=&v"(z), "=&v"(w) : "v"(addr) : "memory");
            a.x = x.x; b.x = y.x; c.x = z.x; d.x = w.x;
        } else {
            asm volatile("ds_read_b128 %0, %4\nds_read_b128 %1, %4 offset:16\n"
                         "ds_read_b128 %2, %4 offset:32\nds_read_b128 %3, %4 offset:48\n"
                         "s_waitcnt lgkmcnt(0)\n"
                         : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(addr) : "memory");
        }
        acc += a + b + c + d;
    }
    const long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int WIDTH>
static double run(int n_stride, int q_stride) {
    static float* out = nullptr;
    static long long* cyc = nullptr;
    if (!out) {
        (void)hipMalloc(&out, 256 * 512 * 4);
        (void)hipMalloc(&cyc, 256 * 8);
    }
    for (int r = 0; r < 2; ++r)
        hipLaunchKernelGGL(k<WIDTH>, dim3(256), dim3(512), 65536, 0, out, cyc, n_stride, q_stride);
    (void)hipDeviceSynchronize();
    long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0;
    for (int i = 0; i < 256; ++i) avg += double(h[i]);
    // 8 waves x 64 reads per block share the CU's LDS
    return avg / 256 / (8 * 64);
}

int main() {
    printf("contiguous: b64 %.2f  b128 %.2f cycles per wave-read (LDS pipe, per CU)\n",
           run<8>(8, 128), run<16>(16, 256));
    printf("%4s %22s %22s %22s\n", "S", "direct (1 row) b64", "F(2,3) (2 rows) b64", "F(4,3) (4 rows) b64");
    for (int S = 48; S <= 68; S += (S >= 48 && S < 54) ? 1 : 2)
        printf("%4d %22.2f %22.2f %22.2f\n", S, run<8>(S * 4, 8), run<8>(2 * S * 4, 8), run<8>(4 * S * 4, 8));
    printf("weights: b128 lane*16 %.2f, b64 lane*8 %.2f\n", run<16>(16, 256), run<8>(8, 128));
    return 0;
}
