// How many hardware threads does this box really give a process?  N threads each spin on private
// arithmetic (no memory traffic, no system calls) for a fixed time; the aggregate rate against N
// tells a CPU quota (cgroup cpu.max) or an affinity mask from the 256 "online" CPUs the OS lists.
// A second pass does the same with a private 64 KB memcpy loop (cache-resident).
// g++ -O2 -pthread tools/microbench/cpu_scaling.cpp -o cpu_scaling && ./cpu_scaling
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double run(int threads, int kind) {
    std::atomic<bool> go(false), stop(false);
    std::vector<unsigned long long> counts((size_t)threads, 0);
    std::vector<std::thread> team;
    for (int t = 0; t < threads; ++t)
        team.emplace_back([&, t] {
            std::vector<char> a(65536, 1), b(65536, 2);
            unsigned long long n = 0, x = 88172645463325252ull + (unsigned)t;
            while (!go.load()) {}
            while (!stop.load(std::memory_order_relaxed)) {
                if (kind == 0) {
                    for (int i = 0; i < 4096; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; }
                    n += 4096;
                } else {
                    std::memcpy(a.data(), b.data(), a.size());
                    a[x & 65535] ^= 1;
                    n += 1;
                }
            }
            counts[(size_t)t] = n + (x & 1);
        });
    const auto t0 = std::chrono::steady_clock::now();
    go = true;
    std::this_thread::sleep_for(std::chrono::milliseconds(400));
    stop = true;
    for (auto& th : team) th.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long total = 0;
    for (auto c : counts) total += c;
    return (double)total / dt;
}

int main() {
    std::printf("hardware_concurrency %u\n", std::thread::hardware_concurrency());
    for (int kind = 0; kind < 2; ++kind) {
        const double one = run(1, kind);
        std::printf("%s: 1 thread = %.3g/s; speed-up at", kind == 0 ? "xorshift" : "memcpy 64K", one);
        for (int n : {2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256})
            std::printf("  %d: %.1f", n, run(n, kind) / one);
        std::printf("\n");
    }
    return 0;
}
