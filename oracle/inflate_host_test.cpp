// TEST INFRASTRUCTURE (oracle/): the GPU inflate's phase-1 decoder (deepbinner_amd/csrc/
// dbh_inflate_core.h, the very header the kernels are compiled from) run on the CPU, one lane at a
// time, with a sequential stand-in for phase 2 - so that tests/test_inflate.py can hold it to
// zlib (Python's zlib module = the library libhdf5 inflates Signal chunks with) on the build box,
// without a GPU.  Not part of the product; nothing in deepbinner_amd/ calls it.
//   cases file: u32 n; per case: u32 comp_bytes, u32 out_cap, bytes
//   result file: per case: i32 status, i32 ended, i32 adler_ok, u32 n_tokens, u32 out_bytes, bytes
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../deepbinner_amd/csrc/dbh_inflate_core.h"

struct HostMem {
    uint16_t lit_[dbi::kLitEntries], dist_[dbi::kDistEntries], work_[dbi::kMaxSyms];
    uint8_t lens_[dbi::kMaxLens];
    uint16_t lit(int e) const { return lit_[check(e, dbi::kLitEntries)]; }
    uint16_t dist(int e) const { return dist_[check(e, dbi::kDistEntries)]; }
    void set_lit(int e, uint16_t v) { lit_[check(e, dbi::kLitEntries)] = v; }
    void set_dist(int e, uint16_t v) { dist_[check(e, dbi::kDistEntries)] = v; }
    int len(int i) const { return lens_[check(i, dbi::kMaxLens)]; }
    void set_len(int i, int v) { lens_[check(i, dbi::kMaxLens)] = (uint8_t)v; }
    int work(int i) const { return work_[check(i, dbi::kMaxSyms)]; }
    void set_work(int i, int v) { work_[check(i, dbi::kMaxSyms)] = (uint16_t)v; }
    static int check(int i, int n) {
        if (i < 0 || i >= n) {
            std::fprintf(stderr, "table index %d outside [0, %d)\n", i, n);
            std::abort();
        }
        return i;
    }
};

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    FILE* in = std::fopen(argv[1], "rb");
    FILE* out = std::fopen(argv[2], "wb");
    if (!in || !out) return 2;
    uint32_t n_cases = 0;
    if (std::fread(&n_cases, 4, 1, in) != 1) return 2;
    for (uint32_t c = 0; c < n_cases; ++c) {
        uint32_t comp_bytes = 0, out_cap = 0;
        if (std::fread(&comp_bytes, 4, 1, in) != 1 || std::fread(&out_cap, 4, 1, in) != 1) return 2;
        std::vector<uint8_t> comp((size_t)comp_bytes + 64, 0);       // padded like the device buffer
        if (comp_bytes && std::fread(comp.data(), 1, comp_bytes, in) != comp_bytes) return 2;
        HostMem mem;
        std::memset(&mem, 0, sizeof(mem));
        dbi::Lane L;
        dbi::lane_start(L, comp.data(), (int64_t)comp_bytes, (int64_t)out_cap);
        std::vector<uint32_t> tokens;
        long guard = 0;
        while (L.state != dbi::kDone) {
            if (++guard > 100000000L) {
                std::fprintf(stderr, "case %u does not terminate\n", c);
                return 3;
            }
            if (L.state == dbi::kNeedBlock) {
                dbi::lane_block(L, mem);
                continue;
            }
            uint32_t token = 0;
            // (kDecode: the hot path the kernel runs; the rest through the general step)
            if (L.state == dbi::kDecode ? dbi::lane_decode(L, mem, &token)
                                        : dbi::lane_step(L, mem, &token)) {
                tokens.push_back(token);
            }
        }
        // phase 2, sequentially
        std::vector<uint8_t> bytes;
        int status = L.status;
        if (status == dbi::kOk) {
            for (uint32_t t : tokens) {
                if (t & dbi::kMatchFlag) {
                    const size_t len = t & 0x1FF, dist = ((t >> 9) & 0x7FFF) + 1;
                    if (dist > bytes.size()) {
                        status = dbi::kBadDistance;
                        break;
                    }
                    for (size_t k = 0; k < len; ++k) bytes.push_back(bytes[bytes.size() - dist]);
                } else {
                    bytes.push_back((uint8_t)t);
                }
            }
        }
        int adler_ok = -1;
        if (status == dbi::kOk && L.ended) {
            uint32_t s1 = 1, s2 = 0;
            for (uint8_t b : bytes) {
                s1 = (s1 + b) % 65521u;
                s2 = (s2 + s1) % 65521u;
            }
            adler_ok = ((s2 << 16) | s1) == L.adler ? 1 : 0;
            if (!adler_ok) status = dbi::kBadChecksum;
        }
        if (status != dbi::kOk) bytes.clear();
        const int32_t head[3] = {status, L.ended, adler_ok};
        const uint32_t sizes[2] = {(uint32_t)tokens.size(), (uint32_t)bytes.size()};
        std::fwrite(head, 4, 3, out);
        std::fwrite(sizes, 4, 2, out);
        if (!bytes.empty()) std::fwrite(bytes.data(), 1, bytes.size(), out);
    }
    std::fclose(in);
    std::fclose(out);
    return 0;
}
