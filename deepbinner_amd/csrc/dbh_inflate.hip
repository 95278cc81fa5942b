// dbh_inflate.hip - inflating the Signal chunks of fast5 files ON THE GPU (C ABI: the
// "compressed input" section of include/deepbinner_hip.h).
//
// The reference inflates on the host: h5py -> libhdf5 -> zlib, one chunk after the other
// (deepbinner/load_fast5s.py:33-43).  For this package's native loader that is ~100 of the ~118 us
// a 55 KB read costs a CPU core, and the box hands a process 16 cores: one host stops at ~130 k
// reads/s while ONE GPU classifies 214 k (profiles/r03_*).  A container of 4,000 reads is 4,000
// independent zlib streams - latency-bound, branchy, byte-granular work that a CPU runs 16 at a
// time and a GPU runs thousands at a time.
//
// Two kernels (RFC 1950 / 1951; bit-exact with zlib, same accept / reject decisions; the decoder
// core is dbh_inflate_core.h + dbh_inflate_wave.h, which the CPU test harness compiles too):
//   1. Huffman codes -> tokens (literal | match {length, distance}): no output bytes, no window -
//      nothing here depends on memory the kernel wrote itself.  Two forms, same tokens:
//      inflate_tokens_wave_kernel (what runs) - ONE WAVEFRONT PER STREAM: the 64 lanes decode 64
//      consecutive pieces of the same Huffman block at once, each from a guessed first bit, and
//      re-decode until every piece begins where the one before it ended (prefix codes
//      re-synchronise: ~2.7 rounds; dbh_inflate_wave.h); 4,000 streams are 4,000 waves, a dozen
//      per CU, and a launch lasts 2.4 ms instead of 12.9 (84 -> 14 ms both kernels when the
//      reads' lengths are log-normal: a long read is no longer one lane's work);
//      inflate_tokens_kernel (DEEPBINNER_INFLATE_KERNEL=lane) - ONE LANE PER STREAM, 64 streams
//      per wavefront: rounds 3 and 4's kernel, the same work in 30 % less CU time but 63 waves
//      for as long as the longest stream lasts.  All lanes of a wave step together; streams
//      deflated with the same settings reach their block boundaries (every 16,383 symbols with
//      zlib's defaults) on the same step, so the code builds line up; a lane that is done takes
//      its next stream off a counter there.
//   2. tokens -> bytes, ONE WAVE PER STREAM, up to 64 tokens per step: a wave-wide prefix sum of
//      the token lengths gives every token its output position.  Two forms, same bytes:
//      inflate_resolve_pre_kernel (what runs; described in front of it) - an 8 KiB ring in LDS
//      and the stream's own flushed output behind it, twenty streams per CU; the short matches
//      whose source is complete before their step (four fifths of all) read it at the step
//      boundary and are stored with the literals, the others go in rounds by the exact rule:
//      0.84 ms per 4,000 x 54 KB streams;
//      inflate_resolve_kernel (DEEPBINNER_INFLATE_RESOLVE=rounds; rounds 3-5) - the whole 32 KiB
//      window as a ring in LDS, five streams per CU; literals are stored at once, every match
//      copies from the ring as soon as everything it reads has been written (the lanes go in
//      rounds, the first waiting match deciding who may go; a step in which a far-reaching match
//      could be overtaken by a write that wraps around the ring goes in token order instead:
//      dbh_inflate_core.h, ring_hazard): 3.04 ms.
//      The ring is written out in coalesced 256-byte pieces, with the Adler-32 sums on the way.
// What dbh_inflate_dev launches is both of them AT ONCE, a pair of waves per stream
// (inflate_pair_kernel: wave 0 is kernel 1, wave 1 is kernel 2 resolving the tokens as they come;
// described in front of it) - a stream then lasts as long as the slower of its halves, not their
// sum; DEEPBINNER_INFLATE_PAIR=0: the two launches.
// HBM traffic per read (55 KB of samples, ~22 k tokens): 35 KB compressed in, 88 KB of tokens out
// and in again, 55 KB of samples out - latency and instruction issue, not bandwidth, are what
// both kernels are bound by.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/deepbinner_hip.h"
#include "dbh_inflate_core.h"
#include "dbh_inflate_wave.h"

namespace dbh_inflate_detail {

using dbi::Lane;

// streams per workgroup of kernel 1 = the active lanes of its one wavefront.  A token costs a wave
// the same time whether 64 or 16 of its lanes decode; fewer lanes per wave buy LDS per lane - room
// for the first-level decode tables of dbh_inflate_core.h (-DDBI_LANES=16 -DDBI_LIT_BITS=11: an
// experiment of round 5, 15 % faster on four times the waves; what ships is 64 lanes, no tables).
#ifndef DBI_LANES
#define DBI_LANES 64
#endif
constexpr int kLanes = DBI_LANES;
static_assert(kLanes == 16 || kLanes == 32 || kLanes == 64, "");
// LDS of kernel 1, every array interleaved by lane ([entry][lane]): consecutive lanes hit
// consecutive addresses whatever entry each of them wants.
constexpr int kRingOff = 0;                                              // u32 [34][lanes]
constexpr int kLitPairOff = kRingOff + dbi::kRingStore * kLanes * 4;     // u32 [16][lanes]
constexpr int kDistPairOff = kLitPairOff + 16 * kLanes * 4;              // u32 [16][lanes]
constexpr int kLitSymOff = kDistPairOff + 16 * kLanes * 4;               // u16 [288][lanes]
constexpr int kCntOff = kLitSymOff + dbi::kLitSyms * kLanes * 2;         // u16 [16][lanes]
constexpr int kLitTabOff = kCntOff + 16 * kLanes * 2;                    // u16 [2^kLitBits][lanes]
constexpr int kDistTabOff = kLitTabOff + (dbi::kTables ? (1 << dbi::kLitBits) * kLanes * 2 : 0);
constexpr int kDistSymOff = kDistTabOff + (dbi::kTables ? (1 << dbi::kDistBits) * kLanes * 2 : 0);  // u8 [32][lanes]
constexpr int kLensOff = kDistSymOff + dbi::kDistSyms * kLanes;          // u8 [320][lanes]
constexpr int kLdsBytes1 = kLensOff + dbi::kMaxLens * kLanes;
static_assert(kLdsBytes1 <= 160 * 1024, "kernel 1's LDS no longer fits a CU");

struct LdsMem {
    static constexpr int kClTableBits = 0;             // (no table for the code-length code: 64 per wave)
    __device__ __forceinline__ uint32_t cl_tab(int) const { return 0u; }
    __device__ __forceinline__ void set_cl_tab(int, uint32_t) {}
    uint32_t* ring_;
    uint32_t* lit_pair_;
    uint32_t* dist_pair_;
    uint16_t* lit_sym_;
    uint16_t* cnt_;
    uint16_t* lit_tab_;
    uint16_t* dist_tab_;
    uint8_t* dist_sym_;
    uint8_t* lens_;
    // asynchronous reads for the decode front (dbh_inflate_core.h: issue_*): requested here, waited
    // for by the caller's lds_landed<>()
    static __device__ __forceinline__ unsigned addr_of(const void* p) {
        return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
    }
    __device__ __forceinline__ void ring3_issue(int r, uint32_t& a, uint32_t& b, uint32_t& c) const {
        uint64_t ab;
        asm volatile("ds_read2_b32 %0, %2 offset1:%3\n\tds_read_b32 %1, %2 offset:%4"
                     : "=&v"(ab), "=&v"(c)
                     : "v"(addr_of(ring_ + r * kLanes)), "n"(kLanes), "n"(kLanes * 8)
                     : "memory");
        a = (uint32_t)ab;
        b = (uint32_t)(ab >> 32);
    }
    __device__ __forceinline__ void lit_pair_issue(int l, uint32_t& v) const {
        asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr_of(lit_pair_ + l * kLanes)) : "memory");
    }
    __device__ __forceinline__ void dist_pair_issue(int l, uint32_t& v) const {
        asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr_of(dist_pair_ + l * kLanes)) : "memory");
    }
    __device__ __forceinline__ void lit_sym_issue(int i, uint32_t& v) const {
        asm volatile("ds_read_u16 %0, %1" : "=v"(v) : "v"(addr_of(lit_sym_ + i * kLanes)) : "memory");
    }
    __device__ __forceinline__ void dist_sym_issue(int i, uint32_t& v) const {
        asm volatile("ds_read_u8 %0, %1" : "=v"(v) : "v"(addr_of(dist_sym_ + i * kLanes)) : "memory");
    }
    __device__ __forceinline__ uint32_t lit_tab(int i) const { return lit_tab_[i * kLanes]; }
    __device__ __forceinline__ void set_lit_tab(int i, uint32_t v) { lit_tab_[i * kLanes] = (uint16_t)v; }
    __device__ __forceinline__ uint32_t dist_tab(int i) const { return dist_tab_[i * kLanes]; }
    __device__ __forceinline__ void set_dist_tab(int i, uint32_t v) { dist_tab_[i * kLanes] = (uint16_t)v; }
    __device__ __forceinline__ uint32_t ring(int r) const { return ring_[r * kLanes]; }
    __device__ __forceinline__ void set_ring(int r, uint32_t v) { ring_[r * kLanes] = v; }
    __device__ __forceinline__ int len(int i) const { return lens_[i * kLanes]; }
    __device__ __forceinline__ void set_len(int i, int v) { lens_[i * kLanes] = (uint8_t)v; }
    __device__ __forceinline__ int cnt(int l) const { return cnt_[l * kLanes]; }
    __device__ __forceinline__ void set_cnt(int l, int v) { cnt_[l * kLanes] = (uint16_t)v; }
    __device__ __forceinline__ uint32_t lit_pair(int l) const { return lit_pair_[l * kLanes]; }
    __device__ __forceinline__ void set_lit_pair(int l, uint32_t v) { lit_pair_[l * kLanes] = v; }
    __device__ __forceinline__ uint32_t dist_pair(int l) const { return dist_pair_[l * kLanes]; }
    __device__ __forceinline__ void set_dist_pair(int l, uint32_t v) { dist_pair_[l * kLanes] = v; }
    __device__ __forceinline__ uint32_t lit_sym(int i) const { return lit_sym_[i * kLanes]; }
    __device__ __forceinline__ void set_lit_sym(int i, uint32_t v) { lit_sym_[i * kLanes] = (uint16_t)v; }
    __device__ __forceinline__ uint32_t dist_sym(int i) const { return dist_sym_[i * kLanes]; }
    __device__ __forceinline__ void set_dist_sym(int i, uint32_t v) { dist_sym_[i * kLanes] = (uint8_t)v; }
};

// what kernel 1 leaves for kernel 2 (and for the caller) per stream
struct StreamInfo {
    int32_t status;
    int32_t ended;
    uint32_t adler;
    int32_t n_tokens;
    int64_t produced;
};

// STREAMS PER LANE (round 5; DBI_PER_LANE, default 1).  A token is one dependent chain of ~190
// vector instructions with five dependent LDS reads in it, and a container's 4,000 streams are 63
// waves for the GPU's 1,024 SIMDs: a wave decodes alone, nothing fills its latencies.  The obvious
// answer - every lane carries TWO independent streams, each with its own decoder state and its
// own 1.2 KB of LDS, the two "fronts" of a round (look at the next token: dbh_inflate_core.h) side
// by side - was built three ways and measured (profiles/r05_inflate/README.md; 4,000 streams of
// 54 KB, both kernels): one slot 15.7 ms; two slots, front after front 28.3 ms; as one software
// pipeline with hand-counted LDS waits (each request in flight behind the other slot's
// arithmetic) 27.7 ms; in lockstep, the two chains alternating instruction by instruction
// 28.5 ms.  Twice the tokens per wave cost twice the time whatever the arrangement: a wave alone
// on its SIMD is bound by instruction ISSUE (~6.3 cycles per instruction, dependent or not), not
// by the latencies between its instructions.  What is left is the number of instructions per
// token, i.e. decode tables instead of the canonical compare chains (a third of all
// instructions) - 2 to 4 KB per lane, half or a quarter of the lanes per wave.  The kernel keeps
// the general form (the slots of a lane are a compile-time array) with one slot.
#ifndef DBI_PER_LANE
#define DBI_PER_LANE 1
#endif
#ifndef DBI_DEFAULT_WAVE
#define DBI_DEFAULT_WAVE 1
#endif
constexpr int kPerLane = DBI_PER_LANE;
static_assert(kPerLane * kLdsBytes1 <= 160 * 1024 && (kPerLane == 1 || kPerLane == 2),
              "kernel 1's LDS no longer fits a CU");

__global__ __launch_bounds__(kLanes) void inflate_tokens_kernel(
    const uint8_t* __restrict__ comp, int64_t comp_total,
    const dbh_inflate_stream* __restrict__ streams, int n_streams, uint32_t* __restrict__ tokens,
    StreamInfo* __restrict__ info, int* __restrict__ next_stream) {
    __shared__ __attribute__((aligned(16))) uint8_t lds_all[kPerLane * kLdsBytes1];
    const int lane = threadIdx.x;
    LdsMem mem[kPerLane];
    Lane L[kPerLane];
    uint32_t* tok[kPerLane];
    int n_tok[kPerLane], cur[kPerLane];
#pragma unroll
    for (int s = 0; s < kPerLane; ++s) {
        uint8_t* lds = lds_all + s * kLdsBytes1;
        mem[s].ring_ = reinterpret_cast<uint32_t*>(lds + kRingOff) + lane;
        mem[s].lit_pair_ = reinterpret_cast<uint32_t*>(lds + kLitPairOff) + lane;
        mem[s].dist_pair_ = reinterpret_cast<uint32_t*>(lds + kDistPairOff) + lane;
        mem[s].lit_sym_ = reinterpret_cast<uint16_t*>(lds + kLitSymOff) + lane;
        mem[s].cnt_ = reinterpret_cast<uint16_t*>(lds + kCntOff) + lane;
        mem[s].lit_tab_ = reinterpret_cast<uint16_t*>(lds + kLitTabOff) + lane;
        mem[s].dist_tab_ = reinterpret_cast<uint16_t*>(lds + kDistTabOff) + lane;
        mem[s].dist_sym_ = lds + kDistSymOff + lane;
        mem[s].lens_ = lds + kLensOff + lane;
        L[s].state = dbi::kDone;
        L[s].status = dbi::kOk;
        L[s].ended = 0;
        L[s].adler = 0;
        L[s].out_pos = 0;
        L[s].out_cap = 0;
        L[s].final_block = 0;
        L[s].stored_left = 0;
#pragma unroll
        for (int l = 0; l < 15; ++l) L[s].lim_lit[l] = L[s].lim_dist[l] = 0;
        L[s].br.in = comp;
        L[s].br.limit_bits = 0;
        L[s].br.bp = L[s].br.wr = 0;
        L[s].br.pending = 0;
        L[s].br.fetch_cap = 0;
        tok[s] = tokens;
        n_tok[s] = 0;
        cur[s] = -1;              // the stream this slot of the lane is decoding
    }
    bool more = true;             // streams may be left to fetch
    for (;;) {
        // A slot that has finished its stream takes the next one off the counter - HERE, where
        // no lane is inside a block: zlib ends a block after a fixed number of symbols, so
        // streams that start together reach their block headers together (and the hot loop
        // below ends when the last lane has left its block); a stream taken up in between
        // would make every lane of the wave wait for its headers, each time, alone.
#pragma unroll
        for (int s = 0; s < kPerLane; ++s) {
            while (L[s].state == dbi::kDone && more) {
                if (cur[s] >= 0) {
                    StreamInfo r;
                    r.status = L[s].status;
                    r.ended = L[s].ended;
                    r.adler = L[s].adler;
                    r.n_tokens = n_tok[s];
                    r.produced = L[s].out_pos;
                    info[cur[s]] = r;
                }
                cur[s] = atomicAdd(next_stream, 1);
                if (cur[s] >= n_streams) {
                    cur[s] = -1;
                    more = false;
                    break;
                }
                const dbh_inflate_stream st = streams[cur[s]];
                n_tok[s] = 0;
                if (st.mode == DBH_INFLATE_ZLIB) {
                    // (the caller's buffer is readable for 64 bytes beyond comp_total)
                    dbi::lane_start(L[s], mem[s], comp + st.comp_offset, st.comp_bytes, st.out_bytes,
                                    comp_total + 64 - st.comp_offset);
                    tok[s] = tokens + st.out_offset;          // one token slot per byte of output
                } else {
                    L[s].status = dbi::kOk;                  // nothing to decode: kernel 2 copies it
                    L[s].ended = 0;
                    L[s].adler = 0;
                    L[s].out_pos = 0;
                }
            }
            // (a slot left with a finished stream when the counter ran out: its record)
            if (L[s].state == dbi::kDone && !more && cur[s] >= 0) {
                StreamInfo r;
                r.status = L[s].status;
                r.ended = L[s].ended;
                r.adler = L[s].adler;
                r.n_tokens = n_tok[s];
                r.produced = L[s].out_pos;
                info[cur[s]] = r;
                cur[s] = -1;
            }
        }
        if (!__any(L[0].state != dbi::kDone || L[kPerLane - 1].state != dbi::kDone)) break;
        // the rare states: a block header (with its two code builds), a stored block's bytes
#pragma unroll
        for (int s = 0; s < kPerLane; ++s) {
            if (L[s].state == dbi::kNeedBlock) {
                dbi::lane_block(L[s], mem[s]);
            } else if (L[s].state == dbi::kStored) {
                uint32_t token;
                if (dbi::lane_stored(L[s], mem[s], &token)) tok[s][n_tok[s]++] = token;
            }
        }
        // the hot loop: every slot that is inside a Huffman block decodes four tokens per round
        // (slots that have left their block wait for the others)
        while (__any(L[0].state == dbi::kDecode || L[kPerLane - 1].state == dbi::kDecode)) {
            uint32_t t[kPerLane][4];
            bool p[kPerLane][4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                dbi::Decoded dec[kPerLane];
                if constexpr (kPerLane == 1 && dbi::kTables) {
                    // through the first-level tables; a code longer than their index (no entry)
                    // sends the whole wave the canonical way for this token
                    const bool fast = dbi::lane_decode_fast(L[0], mem[0], dec[0]);
#ifndef DBI_ABL_NO_FALLBACK
                    if (__any(!fast && L[0].state == dbi::kDecode))
#else
                    if (false)
#endif
                        dec[0] = dbi::lane_decode_front(L[0], mem[0]);
                } else {
                    dbi::lane_decode_fronts<kPerLane, LdsMem>(L, mem, dec);
                }
#pragma unroll
                for (int s = 0; s < kPerLane; ++s) {
                    t[s][k] = 0;
                    p[s][k] = dbi::lane_decode_commit(L[s], dec[s], &t[s][k]);
                }
            }
#pragma unroll
            for (int s = 0; s < kPerLane; ++s) {
                // (a store per token and lane would be 64 partial cache lines per step: four
                // tokens go out as one 16-byte store - all four real in all but a handful of rounds)
                if (p[s][0] && p[s][1] && p[s][2] && p[s][3]) {
                    const uint32_t four[4] = {t[s][0], t[s][1], t[s][2], t[s][3]};
                    __builtin_memcpy(tok[s] + n_tok[s], four, 16);
                    n_tok[s] += 4;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (p[s][k]) tok[s][n_tok[s]++] = t[s][k];
                }
                L[s].br.checkpoint(mem[s]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel 1, second form: ONE WAVE PER STREAM (dbh_inflate_wave.h - the rounds, why they end, what
// they cost).  One wavefront per workgroup, one stream per workgroup, in the caller's order (the
// longest first, if the caller has sorted them).  LDS: the staged chunk (4.4 KB) and ONE set of
// canonical code tables (1.3 KB) - a dozen and more waves per CU, which is what hides the LDS
// round trips of a token's dependent chain here.
// ---------------------------------------------------------------------------------------------
struct WaveLds {
    uint32_t stage[dbi::kStageDwords];
    uint32_t ring[dbi::kRingStore], lit_pair[16], dist_pair[16];
    uint16_t lit_sym[dbi::kLitSyms], cnt[16];
    uint16_t wave_lit_tab[dbi::kWaveLitEntries], wave_dist_tab[dbi::kWaveDistEntries];
    uint8_t dist_sym[dbi::kDistSyms], lens[dbi::kMaxLens];
};
// (the code-length code's table - 128 bytes, lane 0's, needed only while a block header is read -
// lies in the chunk's stage: no chunk is staged then.  128 bytes more would make a CU's LDS hold
// 15 of these instead of 16.)
static_assert(sizeof(((WaveLds*)nullptr)->stage) >= 128, "");
struct WaveMem {
    WaveLds* m;
#ifndef DBI_WAVE_CL_TABLE
#define DBI_WAVE_CL_TABLE 7
#endif
    static constexpr int kClTableBits = DBI_WAVE_CL_TABLE;      // (7, or 0 = without the table)
    __device__ __forceinline__ uint32_t cl_tab(int i) const {
        return reinterpret_cast<const uint8_t*>(m->stage)[i];
    }
    __device__ __forceinline__ void set_cl_tab(int i, uint32_t v) {
        reinterpret_cast<uint8_t*>(m->stage)[i] = (uint8_t)v;
    }
    __device__ __forceinline__ uint32_t stage(int i) const { return m->stage[i]; }
    __device__ __forceinline__ uint32_t lit_tab(int) const { return 0u; }      // (no decode tables)
    __device__ __forceinline__ void set_lit_tab(int, uint32_t) {}
    __device__ __forceinline__ uint32_t dist_tab(int) const { return 0u; }
    __device__ __forceinline__ void set_dist_tab(int, uint32_t) {}
    __device__ __forceinline__ uint32_t wave_lit_tab(int i) const { return m->wave_lit_tab[i]; }
    __device__ __forceinline__ uint32_t wave_dist_tab(int i) const { return m->wave_dist_tab[i]; }
    __device__ __forceinline__ uint32_t ring(int r) const { return m->ring[r]; }
    __device__ __forceinline__ void set_ring(int r, uint32_t v) { m->ring[r] = v; }
    __device__ __forceinline__ int len(int i) const { return m->lens[i]; }
    __device__ __forceinline__ void set_len(int i, int v) { m->lens[i] = (uint8_t)v; }
    __device__ __forceinline__ int cnt(int l) const { return m->cnt[l]; }
    __device__ __forceinline__ void set_cnt(int l, int v) { m->cnt[l] = (uint16_t)v; }
    __device__ __forceinline__ uint32_t lit_pair(int l) const { return m->lit_pair[l]; }
    __device__ __forceinline__ void set_lit_pair(int l, uint32_t v) { m->lit_pair[l] = v; }
    __device__ __forceinline__ uint32_t dist_pair(int l) const { return m->dist_pair[l]; }
    __device__ __forceinline__ void set_dist_pair(int l, uint32_t v) { m->dist_pair[l] = v; }
    __device__ __forceinline__ uint32_t lit_sym(int i) const { return m->lit_sym[i]; }
    __device__ __forceinline__ void set_lit_sym(int i, uint32_t v) { m->lit_sym[i] = (uint16_t)v; }
    __device__ __forceinline__ uint32_t dist_sym(int i) const { return m->dist_sym[i]; }
    __device__ __forceinline__ void set_dist_sym(int i, uint32_t v) { m->dist_sym[i] = (uint8_t)v; }
};

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t uni(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ int wave_scan_i32(int v);
// a value of the lane before (lane 0: its own)
__device__ __forceinline__ uint32_t from_lane_before(uint32_t v) { return (uint32_t)__shfl_up((int)v, 1); }

// a compiler barrier that also drains the wave's LDS queue: what other lanes wrote is there (a
// wave's LDS operations execute in order; this is all the synchronisation ONE wave needs)
__device__ __forceinline__ void lds_settle() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct NoProgress {
    __device__ __forceinline__ void tokens(int) const {}
};

// One stream through kernel 1's one-wavefront-per-stream form: the tokens to `tokens` + the
// stream's out_offset, its record returned (lane 0's copy is the stream's state).  `lds` is this
// wave's own, the wave synchronises with nobody (inflate_tokens_wave_kernel: one wave per
// workgroup; inflate_pair_kernel: beside the wave that resolves the same stream's tokens, which
// `progress.tokens(n)` tells how many are there - behind every chunk and every stored run).
template <class Progress>
__device__ __forceinline__ StreamInfo tokens_wave_stream(
    WaveLds& lds, const uint8_t* __restrict__ comp, int64_t comp_total, const dbh_inflate_stream& st,
    uint32_t* tokens, int lane, const Progress& progress) {
    using namespace dbi;
    if (st.mode != DBH_INFLATE_ZLIB)                     // nothing to decode: kernel 2 copies it
        return StreamInfo{kOk, 0, 0u, 0, 0};
    WaveMem mem{&lds};
    // Lane 0's copy of L is the stream's state; the serial code (header, block headers with their
    // code builds, stored bytes) runs on lane 0 alone, and what the whole wave needs of it is
    // handed round afterwards.
    Lane L;
    uint32_t* const tok = tokens + st.out_offset;        // one token slot per byte of output
    int n_tok = 0;
    // (the caller's buffer is readable for 64 bytes beyond comp_total)
    lane_start(L, mem, comp + st.comp_offset, st.comp_bytes, st.out_bytes,
               comp_total + 64 - st.comp_offset);        // (every lane the same: uniform so far)
    WaveBlock B;
    while (uni(L.state) != kDone) {
        const int state = uni(L.state);
        if (state == kNeedBlock) {
            if (lane == 0) lane_block(L, mem);
#pragma unroll
            for (int l = 0; l < 15; ++l) {
                B.lim_lit[l] = uni(L.lim_lit[l]);
                B.lim_dist[l] = uni(L.lim_dist[l]);
            }
            lds_settle();                                 // (the tables lane 0 wrote: for all lanes)
            if (kWaveTables && uni(L.state) == kDecode) {
                // the first-level tables of this block's two codes: every lane its share of the
                // indices, decoded the canonical way (dbh_inflate_wave.h)
                for (int k = lane; k < kWaveLitEntries; k += kWaveLanes)
                    lds.wave_lit_tab[k] = (uint16_t)wave_lit_entry((uint32_t)k, B.lim_lit, mem);
                for (int k = lane; k < kWaveDistEntries; k += kWaveLanes)
                    lds.wave_dist_tab[k] = (uint16_t)wave_dist_entry((uint32_t)k, B.lim_dist, mem);
                lds_settle();    
            }
            continue;
        }
        if (state == kStored) {
            // the bytes of a stored block: literal tokens, 64 at a time
            L.stored_left = uni(L.stored_left);
            L.out_pos = uni(L.out_pos);
            L.br.bp = uni(L.br.bp);
            L.final_block = uni(L.final_block);
            const int n = stored_run(L);
            const uint8_t* src = comp + st.comp_offset + (L.br.bp >> 3);
            for (int k = lane; k < n; k += kWaveLanes) tok[n_tok + k] = src[k];
            n_tok += n;
            progress.tokens(n_tok);
            if (stored_advance(L, n) && lane == 0) L.br.seek(mem, L.br.bp);
            continue;
        }
        // ---- inside a Huffman block: one chunk of 64 homes ----
        const uint32_t bp = uni(L.br.bp);
        int out_pos = uni(L.out_pos);
        const int out_cap = uni(L.out_cap);
        const uint32_t first_dword = bp >> 5, rel0 = bp & 31u;
        const uint32_t fetch_cap = uni(L.br.fetch_cap);
        const uint8_t* in = comp + st.comp_offset;
        lds_settle();                                     // (nobody still reads the chunk before)
        for (int piece = lane; piece < kStageDwords / 4; piece += kWaveLanes) {
            U4 v;
            __builtin_memcpy(&v, in + stage_piece_at(first_dword, piece, fetch_cap), 16);
            *reinterpret_cast<U4*>(&lds.stage[4 * piece]) = v;
        }
        lds_settle();    
        B.limit_rel = uni(L.br.limit_bits) - first_dword * 32u;
        uint32_t x = sub_start(rel0, lane);
        const uint32_t stop = sub_start(rel0, lane + 1);
        // (the walks keep their tokens at the end of the stream's token region, if there is room)
        const KeepTokens kept{keep_room(n_tok, st.out_bytes) ? tok + (st.out_bytes - kKeepSlots) : nullptr, lane};
        SubResult r = sub_decode(B, mem, x, stop, kept);
        int last;
        for (;;) {
            const uint32_t prev_end = from_lane_before(r.end);
            const int prev_flag = (int)from_lane_before((uint32_t)r.flag);
            const uint32_t want = lane == 0 ? rel0 : prev_flag != kSubNone ? sub_start(rel0, lane) : prev_end;
            const bool moved = want != x;
            const unsigned long long m_moved = __ballot(moved), m_flag = __ballot(r.flag != kSubNone);
            const int first_moved = m_moved ? __ffsll((long long)m_moved) - 1 : kWaveLanes;
            const int first_flag = m_flag ? __ffsll((long long)m_flag) - 1 : kWaveLanes;
            if (first_moved > first_flag || first_moved == kWaveLanes) {
                last = uni(first_flag < kWaveLanes ? first_flag : kWaveLanes - 1);
                break;
            }
            if (moved) {
                x = want;
                r = sub_decode(B, mem, x, stop, kept);
            }
        }
        // where every lane's tokens and bytes go; the lane in which the wanted number of bytes is
        // exceeded ends the chunk, if that comes first
        const int cnt_m = lane <= last ? r.count : 0, bytes_m = lane <= last ? r.bytes : 0;
        const int incl_c = wave_scan_i32(cnt_m), incl_b = wave_scan_i32(bytes_m);
        const unsigned long long m_over = __ballot(lane <= last && out_pos + incl_b > out_cap);
        if (m_over) {
            const int over = __ffsll((long long)m_over) - 1;
            last = uni(over < last ? over : last);
        }
        SubResult e;
        e.end = 0;
        e.count = e.bytes = 0;
        e.flag = kSubNone;
        if (kept.keep != nullptr && m_over == 0ull && !__any(lane <= last && r.count > kSubKeep)) {
            // the output pass as a copy: what the lanes' last walks kept, to where it belongs
            uint32_t* to = tok + n_tok + incl_c - cnt_m;
            for (int k = 0; __any(k < cnt_m); ++k)
                if (k < cnt_m) to[k] = kept.keep[k * kWaveLanes + lane];
            if (lane <= last) e = r;
        } else if (lane <= last) {
            e = sub_emit(B, mem, x, stop, out_pos + incl_b - bytes_m, out_cap, tok + n_tok + incl_c - cnt_m);
        }
        const int flag = __builtin_amdgcn_readlane(e.flag, last);
        n_tok += __builtin_amdgcn_readlane(incl_c - cnt_m, last) + __builtin_amdgcn_readlane(e.count, last);
        progress.tokens(n_tok);
        out_pos += __builtin_amdgcn_readlane(incl_b - bytes_m, last) + __builtin_amdgcn_readlane(e.bytes, last);
        L.out_pos = out_pos;
        L.br.bp = first_dword * 32u + (uint32_t)__builtin_amdgcn_readlane((int)e.end, last);
        if (flag == kSubBad) lane_fail(L, kBadSymbol);
        else if (flag == kSubTrunc) lane_fail(L, kTruncated);
        else if (flag == kSubBeyond) L.state = kDone;
        else if (flag == kSubEnd) {
            L.state = kNeedBlock;
            if (uni(L.final_block)) {
                lane_ended(L);                   // (every lane the same)
            } else if (lane == 0) {
                L.br.seek(mem, L.br.bp);
            }
        }
    }
    StreamInfo rec;
    rec.status = L.status;
    rec.ended = L.ended;
    rec.adler = L.adler;
    rec.n_tokens = n_tok;
    rec.produced = L.out_pos;
    return rec;
}

__global__ __launch_bounds__(dbi::kWaveLanes) __attribute__((amdgpu_waves_per_eu(4, 8))) void inflate_tokens_wave_kernel(
    const uint8_t* __restrict__ comp, int64_t comp_total,
    const dbh_inflate_stream* __restrict__ streams, int n_streams, uint32_t* __restrict__ tokens,
    StreamInfo* __restrict__ info) {
    __shared__ __attribute__((aligned(16))) WaveLds lds;
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    if (i >= n_streams) return;
    const dbh_inflate_stream st = streams[i];
    const StreamInfo rec = tokens_wave_stream(lds, comp, comp_total, st, tokens, lane, NoProgress());
    if (lane == 0) info[i] = rec;
}

constexpr int kRing = dbi::kWindowRing;
static_assert(dbi::kStepTokens == 64, "one token per lane and step");
constexpr int kWaves2 = 5;                 // streams per workgroup of kernel 2: a 32 KiB ring each
// timing-only ablations of kernel 2 (wrong bytes): 1 no match copies, 2 no flush of the ring, 4 no
// literal stores
#ifndef DBI_K2_ABL
#define DBI_K2_ABL 0
#endif

// Wave-wide inclusive prefix sum on the DPP network (no LDS round trips): within rows of 16
// lanes by shifts, then each row's total handed on to the rows behind it.
__device__ __forceinline__ int wave_scan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);      // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);      // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);      // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);      // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Adler-32 without a reduction per piece: s1 = 1 + sum of the bytes, s2 = n + sum over the bytes
// of (n - position) * byte, n = the stream's length (known from kernel 1) - every lane keeps its
// own two sums, the wave adds them up once per stream.
struct AdlerLane {
    unsigned bytes;
    unsigned long long weighted;
    __device__ __forceinline__ void add4(unsigned w, unsigned left) {     // `left` = n - position
        const unsigned b0 = w & 255u, b1 = (w >> 8) & 255u, b2 = (w >> 16) & 255u, b3 = w >> 24;
        const unsigned s = b0 + b1 + b2 + b3;
        bytes += s;
        weighted += (unsigned long long)left * s - (b1 + 2u * b2 + 3u * b3);
    }
    __device__ __forceinline__ void add1(unsigned b, unsigned left) {
        bytes += b;
        weighted += (unsigned long long)left * b;
    }
};

__global__ __launch_bounds__(64 * kWaves2) void inflate_resolve_kernel(
    const uint8_t* __restrict__ comp, const dbh_inflate_stream* __restrict__ streams, int n_streams,
    const uint32_t* __restrict__ tokens, StreamInfo* __restrict__ info, uint8_t* __restrict__ out,
    int32_t* __restrict__ status_out) {
    __shared__ __attribute__((aligned(16))) uint8_t rings[kWaves2 * kRing];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint8_t* ring = rings + wave * kRing;
    for (int i = blockIdx.x * kWaves2 + wave; i < n_streams; i += gridDim.x * kWaves2) {
        const dbh_inflate_stream s = streams[i];
        uint8_t* dst = out + s.out_offset;
        const int64_t cap = s.out_bytes;
        if (s.mode != DBH_INFLATE_ZLIB) {
            // stored as it is (an unfiltered chunk, a contiguous dataset, or bytes the host has
            // inflated itself): copy, zero-extend
            const int64_t have = s.comp_bytes < cap ? s.comp_bytes : cap;
            const uint8_t* src = comp + s.comp_offset;
            const int64_t whole = have & ~(int64_t)7;
            for (int64_t k = 8 * (int64_t)lane; k < whole; k += 512) {
                uint64_t v;
                __builtin_memcpy(&v, src + k, 8);
                __builtin_memcpy(dst + k, &v, 8);
            }
            for (int64_t k = whole + lane; k < cap; k += 64) dst[k] = k < have ? src[k] : (uint8_t)0;
            if (lane == 0) status_out[i] = dbi::kOk;
            continue;
        }
        StreamInfo r = info[i];
        int status = r.status;
        const uint32_t* tok = tokens + s.out_offset;
        const int n_tok = r.n_tokens;
        const unsigned n_out = (unsigned)r.produced;
        int pos = 0, flushed = 0;                    // (a stream's output is far below 2^31 bytes)
        AdlerLane adler = {0u, 0ull};
        if (status == dbi::kOk) {
            uint32_t t_next = lane < n_tok ? tok[lane] : 0u;
            for (int t0 = 0; t0 < n_tok; t0 += 64) {
                const bool valid = t0 + lane < n_tok;
                const uint32_t t = t_next;
                // (the next step's tokens are on their way while this step's are resolved)
                t_next = t0 + 64 + lane < n_tok ? tok[t0 + 64 + lane] : 0u;
                const bool is_match = valid && (t & dbi::kMatchFlag);
                const int len = !valid ? 0 : is_match ? (int)(t & 0x1FFu) : 1;
                const int dist = (int)((t >> 9) & 0x7FFFu) + 1;
                const int incl = wave_scan_i32(len);
                const int total = __builtin_amdgcn_readlane(incl, 63);
                const int my = pos + incl - len;
                if (__any(is_match && dist > my)) {     // reaches before the start of the output
                    status = dbi::kBadDistance;
                    break;
                }
                const int src = my - dist;
                if (__any(is_match && dbi::ring_hazard(dist, my, pos + total))) {
                    // A match of this step reaches back so far that a write near the step's end
                    // would land on bytes it has yet to read (dbh_inflate_core.h: ring_hazard) -
                    // rare (distances beyond 16 K with long matches behind them): this step goes
                    // in strict token order, one lane at a time.
                    for (int l = 0; l < 64; ++l) {
                        lds_settle();
                        if (lane == l && valid) {
                            if (!is_match) {
                                ring[my & (kRing - 1)] = (uint8_t)t;
                            } else {
                                for (int k = 0; k < len; ++k)
                                    ring[(my + k) & (kRing - 1)] = ring[(src + k) & (kRing - 1)];
                            }
                        }
                    }
                    lds_settle();
                } else {
                if (!(DBI_K2_ABL & 4) && valid && !is_match) ring[my & (kRing - 1)] = (uint8_t)t;
                // A match repeats the `dist` bytes before it: byte k is byte k mod dist of them,
                // so everything it READS lies before its own start, in [src, src + min(len,
                // dist)) - it may go as soon as that is written, i.e. lies before the earliest
                // byte still to be written (the first waiting match's start: the positions ascend
                // with the lanes).  The first waiting match can always go.
                const int reach = src + (len < dist ? len : dist);
                bool waiting = is_match && !(DBI_K2_ABL & 1);
                unsigned long long mask = __ballot(waiting);
                while (mask != 0ull) {
                    lds_settle();
                    const int first = __builtin_amdgcn_readlane(my, __ffsll((long long)mask) - 1);
                    const bool go = waiting && reach <= first;
                    int k = 0, o = 0;                    // o = k mod dist
                    while (__any(go && k < len)) {
                        if (go && k < len) {
                            // four bytes at a time: the loads leave together, then the stores
                            int o1 = o + 1;
                            o1 = o1 == dist ? 0 : o1;
                            int o2 = o1 + 1;
                            o2 = o2 == dist ? 0 : o2;
                            int o3 = o2 + 1;
                            o3 = o3 == dist ? 0 : o3;
                            const uint8_t b0 = ring[(src + o) & (kRing - 1)];
                            const uint8_t b1 = ring[(src + o1) & (kRing - 1)];
                            const uint8_t b2 = ring[(src + o2) & (kRing - 1)];
                            const uint8_t b3 = ring[(src + o3) & (kRing - 1)];
                            ring[(my + k) & (kRing - 1)] = b0;
                            if (k + 1 < len) ring[(my + k + 1) & (kRing - 1)] = b1;
                            if (k + 2 < len) ring[(my + k + 2) & (kRing - 1)] = b2;
                            if (k + 3 < len) ring[(my + k + 3) & (kRing - 1)] = b3;
                            o = o3 + 1;
                            o = o == dist ? 0 : o;
                            k += 4;
                        }
                    }
                    waiting = waiting && !go;
                    mask = __ballot(waiting);
                }
                }
                pos += total;
                // whole 256-byte pieces out of the ring, the Adler-32 sums on the way
                if (!(DBI_K2_ABL & 2) && pos - flushed >= 256) {
                    lds_settle();
                    do {
                        const uint32_t w = *reinterpret_cast<const uint32_t*>(
                            ring + ((flushed + 4 * lane) & (kRing - 1)));
                        uint16_t* d16 = reinterpret_cast<uint16_t*>(dst + flushed + 4 * lane);
                        d16[0] = (uint16_t)w;             // (a read starts at an even byte, not
                        d16[1] = (uint16_t)(w >> 16);     //  necessarily at a multiple of four)
                        adler.add4(w, n_out - (unsigned)(flushed + 4 * lane));
                        flushed += 256;
                    } while (pos - flushed >= 256);
                }
            }
        }
        if (status == dbi::kOk) {
            lds_settle();
            const int rest = pos - flushed;           // < 256
            for (int k = lane; k < rest; k += 64) {
                const unsigned b = ring[(flushed + k) & (kRing - 1)];
                dst[flushed + k] = (uint8_t)b;
                adler.add1(b, n_out - (unsigned)(flushed + k));
            }
            if (r.ended) {
                const unsigned s1 = (1u + wave_sum_u32(adler.bytes)) % 65521u;
                const unsigned s2 =
                    (unsigned)(((unsigned long long)n_out + wave_sum_u64(adler.weighted)) % 65521ull);
                if (((s2 << 16) | s1) != r.adler) status = dbi::kBadChecksum;
            }
            // a stream that ends early (MinKNOW's short final chunk): libhdf5 zero-extends it
            for (int64_t k = pos + lane; k < cap; k += 64) dst[k] = 0;
        }
        if (status != dbi::kOk)                       // nothing of a damaged stream is handed on
            for (int64_t k = lane; k < cap; k += 64) dst[k] = 0;
        if (lane == 0) status_out[i] = status;
    }
}

// ---------------------------------------------------------------------------------------------
// Kernel 2, second form (what runs; DEEPBINNER_INFLATE_RESOLVE=rounds brings the first back).
// One wave per stream and up to 64 tokens per step as before - another schedule and a smaller
// ring (dbh_inflate_core.h: "Phase 2's second form").  In the first form every match went through
// the rounds: a step's ~38 matches of ~3.6 bytes took 3.65 rounds of 4.6 four-byte turns, each a
// dependent LDS load and store behind a drained queue - ~5 k cycles per step, 60 % of them waiting,
// with five waves per CU (the 32 KiB rings) and nothing to fill the waits.  Here
//   * a short match whose source is complete before its step ("pre", 82 %) reads its eight source
//     bytes at the boundary in front of the step - one unaligned ds_read_b64 from the ring, or one
//     unaligned global load from the stream's own output where the source lies further back than
//     the ring reaches - and is two overlapping four-byte stores (or two two-byte ones) at the top
//     of its step, together with the literals;
//   * the others ("late": ~3.9 per step) go in rounds by the exact rule (a scalar loop over the
//     waiting lanes): 1.36 rounds per step, eight bytes in the first turn; a match that overlaps
//     itself repeats its first dist bytes (k2_pattern8);
//   * the ring is 8 KiB: twenty streams per CU, one wave per workgroup (a CU's LDS comes back
//     stream by stream, not when the longest of five has ended).
// The CPU harness models this schedule byte for byte - which copy of a position (ring slot or
// flushed output) every read sees - and holds it against the tokens resolved in order.
constexpr int kRing3 = dbi::kSmallRing;
__device__ __forceinline__ uint64_t ring_read8(const uint8_t* ring, int p) {
    const int slot = p & (kRing3 - 1);
    uint64_t v;
    if (slot <= kRing3 - 8) {
        __builtin_memcpy(&v, ring + slot, 8);          // (byte-aligned: hipcc emits ds_read_b64)
    } else {                                           // over the end of the ring: byte by byte
        v = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v |= (uint64_t)ring[(slot + j) & (kRing3 - 1)] << (8 * j);
    }
    return v;
}
__device__ __forceinline__ uint64_t output_read8(const uint8_t* dst, int p) {
    uint64_t v;
    __builtin_memcpy(&v, dst + p, 8);                  // (one unaligned global_load_dwordx2)
    return v;
}
// the first `len` (1..8) bytes of v to positions p, p + 1, ...
__device__ __forceinline__ void ring_store_short(uint8_t* ring, int p, int len, uint64_t v) {
    const int slot = p & (kRing3 - 1);
    if (len >= 3 && slot + len <= kRing3) {
        if (len >= 4) {                                // two four-byte stores that may overlap
            const uint32_t a = (uint32_t)v, b = (uint32_t)(v >> (8 * (len - 4)));
            __builtin_memcpy(ring + slot, &a, 4);
            __builtin_memcpy(ring + slot + (len - 4), &b, 4);
        } else {                                       // three bytes: two two-byte stores
            const uint16_t a = (uint16_t)v, b = (uint16_t)(v >> 8);
            __builtin_memcpy(ring + slot, &a, 2);
            __builtin_memcpy(ring + slot + 1, &b, 2);
        }
    } else {                                           // a cut match of one or two bytes, or over
#pragma unroll                                         // the end of the ring
        for (int j = 0; j < 8; ++j)
            if (j < len) ring[(slot + j) & (kRing3 - 1)] = (uint8_t)(v >> (8 * j));
    }
}

// A step's tokens, decoded and placed: lane l holds token first + l if the step takes it.
struct StepTokens {
    uint32_t t;
    bool valid, is_match;
    int len, dist, my;
    int total, count;          // (uniform) bytes and tokens of the step
};
// `raw` = tok[first + lane] (0 behind the last token).  The step takes the tokens that together
// span at most kStepSpan bytes - all 64 unless there are long matches among them.
__device__ __forceinline__ StepTokens place_step(uint32_t raw, int first, int n_tok, int base, int lane) {
    StepTokens s;
    s.t = raw;
    s.valid = first + lane < n_tok;
    s.is_match = s.valid && (raw & dbi::kMatchFlag);
    s.len = !s.valid ? 0 : s.is_match ? (int)(raw & 0x1FFu) : 1;
    s.dist = (int)((raw >> 9) & 0x7FFFu) + 1;
    const int incl = wave_scan_i32(s.len);
    s.total = __builtin_amdgcn_readlane(incl, 63);
    s.count = n_tok - first < 64 ? n_tok - first : 64;
    if (s.total > dbi::kStepSpan) {                    // (rare: long matches)
        const bool keep = s.valid && incl <= dbi::kStepSpan;      // a prefix of the lanes, never empty
        s.count = __popcll(__ballot(keep));
        s.total = __builtin_amdgcn_readlane(incl, s.count - 1);
        s.valid = keep;
        s.is_match = s.is_match && keep;
        s.len = keep ? s.len : 0;
    }
    s.my = base + incl - s.len;
    return s;
}

// Adler-32 from sums that do not need the stream's length while they are taken (the pair kernel
// resolves a stream whose end kernel 1 has not reached yet): S = sum of the bytes, P = sum of
// position x byte; s1 = 1 + S, s2 = n + n S - P (mod 65521) once n is known.  Every lane keeps its
// own two sums, the wave adds them up once per stream.
struct AdlerSums {
    unsigned bytes;
    unsigned long long weighted;
    __device__ __forceinline__ void add4(unsigned w, unsigned position) {
        const unsigned b0 = w & 255u, b1 = (w >> 8) & 255u, b2 = (w >> 16) & 255u, b3 = w >> 24;
        const unsigned s = b0 + b1 + b2 + b3;
        bytes += s;
        weighted += (unsigned long long)position * s + (b1 + 2u * b2 + 3u * b3);
    }
    __device__ __forceinline__ void add1(unsigned b, unsigned position) {
        bytes += b;
        weighted += (unsigned long long)position * b;
    }
    // (wave-uniform) the stream's Adler-32, n = its length
    __device__ __forceinline__ unsigned finish(unsigned n) const {
        const unsigned long long S = wave_sum_u32(bytes) % 65521u, P = wave_sum_u64(weighted) % 65521ull;
        const unsigned long long nm = n % 65521u;
        const unsigned s1 = (unsigned)((1ull + S) % 65521ull);
        const unsigned s2 = (unsigned)((nm + nm * S + 65521ull - P) % 65521ull);
        return (s2 << 16) | s1;
    }
};

// how many of a stream's tokens are there: all of them (kernel 1 has ended: the two launches) ...
struct AllTokensThere {
    StreamInfo r;
    __device__ __forceinline__ void wait(int, int& limit, bool& done) const {
        limit = r.n_tokens;
        done = true;
    }
    __device__ __forceinline__ StreamInfo record() const { return r; }
};

// One stream through kernel 2's second form: its bytes to out + the stream's out_offset, the
// verdict to *status_slot.  `ring` (kSmallRing bytes) is this wave's own.  `there.wait(need, limit,
// done)` returns once tokens [0, need) are there or kernel 1 is done with the stream; `limit` = the
// number of tokens if it is done, INT_MAX otherwise.
template <class Tokens>
__device__ __forceinline__ void resolve_pre_stream(
    uint8_t* ring, const uint8_t* __restrict__ comp, const dbh_inflate_stream& s,
    const uint32_t* tokens, const Tokens& there, uint8_t* out, int32_t* status_slot, int lane) {
    {
        uint8_t* dst = out + s.out_offset;
        const int64_t cap = s.out_bytes;
        if (s.mode != DBH_INFLATE_ZLIB) {
            const int64_t have = s.comp_bytes < cap ? s.comp_bytes : cap;
            const uint8_t* src = comp + s.comp_offset;
            const int64_t whole = have & ~(int64_t)7;
            for (int64_t k = 8 * (int64_t)lane; k < whole; k += 512) {
                uint64_t v;
                __builtin_memcpy(&v, src + k, 8);
                __builtin_memcpy(dst + k, &v, 8);
            }
            for (int64_t k = whole + lane; k < cap; k += 64) dst[k] = k < have ? src[k] : (uint8_t)0;
            if (lane == 0) *status_slot = dbi::kOk;
            return;
        }
        int status = dbi::kOk;                       // (kernel 1's verdict joins at the end)
        const uint32_t* tok = tokens + s.out_offset;
        int n_tok;                                    // INT_MAX while kernel 1 is still at the stream
        bool all_there;
        // (a step needs its own tokens and the next step's, which are decoded while it runs)
        there.wait(128, n_tok, all_there);
        int pos = 0, flushed = 0;
        AdlerSums adler = {0u, 0ull};
        if (n_tok > 0) {
            int first = 0;                            // this step's first token
            StepTokens c = place_step(lane < n_tok ? tok[lane] : 0u, 0, n_tok, 0, lane);
            uint32_t raw_next = 64 + lane < n_tok ? tok[64 + lane] : 0u;      // (if this step takes 64)
            bool pre = false;                         // (nothing lies before the first step)
            uint64_t pv = 0ull;
            for (;;) {
                const int end = pos + c.total;
                if (__any(c.is_match && c.dist > c.my)) {   // reaches before the start of the output
                    status = dbi::kBadDistance;
                    break;
                }
                // the step behind this one is decoded and placed while this one's stores land
                const int first_n = first + c.count;
                // the next step's tokens are in registers; the ones behind them are requested now
                if (!all_there) there.wait(first_n + 128, n_tok, all_there);
                if (c.count != 64) raw_next = first_n + lane < n_tok ? tok[first_n + lane] : 0u;
                const uint32_t raw_after = first_n + 64 + lane < n_tok ? tok[first_n + 64 + lane] : 0u;
                const StepTokens n = place_step(raw_next, first_n, n_tok, end, lane);

                const int src = c.my - c.dist;
                const int reach = src + (c.len < c.dist ? c.len : c.dist);
                const int my_end = c.my + c.len;
                if (!(DBI_K2_ABL & 4) && c.valid && !c.is_match) ring[c.my & (kRing3 - 1)] = (uint8_t)c.t;
                if (pre) ring_store_short(ring, c.my, c.len, c.dist < c.len ? dbi::k2_pattern8(pv, c.dist) : pv);
                bool waiting = c.is_match && !pre && !(DBI_K2_ABL & 1);
                unsigned long long mask = __ballot(waiting);
                while (mask != 0ull) {
                    lds_settle();
                    bool blocked = false;
                    for (unsigned long long it = mask; it != 0ull; it &= it - 1ull) {
                        const int w = __ffsll((long long)it) - 1;
                        blocked = blocked || dbi::k2_blocks(__builtin_amdgcn_readlane(c.my, w),
                                                            __builtin_amdgcn_readlane(my_end, w), src, reach);
                    }
                    const bool go = waiting && !blocked;
                    if (go) {
                        uint64_t v = dbi::k2_in_ring(src, end) ? ring_read8(ring, src) : output_read8(dst, src);
                        if (c.dist < 8 && c.dist < c.len) v = dbi::k2_pattern8(v, c.dist);
                        ring_store_short(ring, c.my, c.len < 8 ? c.len : 8, v);
                    }
                    if (__any(go && c.len > 8)) {
                        // what is left of a long match: four bytes per turn, byte k = byte k mod
                        // dist of the dist bytes before the match
                        int k = 8, o = 8 % c.dist;
                        while (__any(go && k < c.len)) {
                            if (go && k < c.len) {
                                int o1 = o + 1;
                                o1 = o1 == c.dist ? 0 : o1;
                                int o2 = o1 + 1;
                                o2 = o2 == c.dist ? 0 : o2;
                                int o3 = o2 + 1;
                                o3 = o3 == c.dist ? 0 : o3;
                                const bool near = dbi::k2_in_ring(src, end);      // (then all of [src, my) is)
                                const uint8_t b0 = near ? ring[(src + o) & (kRing3 - 1)] : dst[src + o];
                                const uint8_t b1 = near ? ring[(src + o1) & (kRing3 - 1)] : dst[src + o1];
                                const uint8_t b2 = near ? ring[(src + o2) & (kRing3 - 1)] : dst[src + o2];
                                const uint8_t b3 = near ? ring[(src + o3) & (kRing3 - 1)] : dst[src + o3];
                                ring[(c.my + k) & (kRing3 - 1)] = b0;
                                if (k + 1 < c.len) ring[(c.my + k + 1) & (kRing3 - 1)] = b1;
                                if (k + 2 < c.len) ring[(c.my + k + 2) & (kRing3 - 1)] = b2;
                                if (k + 3 < c.len) ring[(c.my + k + 3) & (kRing3 - 1)] = b3;
                                o = o3 + 1;
                                o = o == c.dist ? 0 : o;
                                k += 4;
                            }
                        }
                    }
                    waiting = waiting && !go;
                    mask = __ballot(waiting);
                }
                lds_settle();                          // everything of this step is in the ring
                pos = end;
                // the boundary: the next step's short matches whose source is complete read it now -
                // from the ring, or from the flushed output where the ring no longer holds it
                const int src_n = n.my - n.dist;
                const bool pre_n = dbi::k2_pre(n.is_match, n.len,
                                               src_n + (n.len < n.dist ? n.len : n.dist), pos);
                uint64_t pv_n = 0ull;
                if (pre_n && src_n >= 0)
                    pv_n = dbi::k2_in_ring(src_n, pos) ? ring_read8(ring, src_n) : output_read8(dst, src_n);
                // whole 256-byte pieces out of the ring, the Adler-32 sums on the way
                if (!(DBI_K2_ABL & 2) && pos - flushed >= 256) {
                    do {
                        const uint32_t w = *reinterpret_cast<const uint32_t*>(
                            ring + ((flushed + 4 * lane) & (kRing3 - 1)));
                        uint16_t* d16 = reinterpret_cast<uint16_t*>(dst + flushed + 4 * lane);
                        d16[0] = (uint16_t)w;             // (a read starts at an even byte, not
                        d16[1] = (uint16_t)(w >> 16);     //  necessarily at a multiple of four)
                        adler.add4(w, (unsigned)(flushed + 4 * lane));
                        flushed += 256;
                    } while (pos - flushed >= 256);
                }
                if (first_n >= n_tok) break;
                first = first_n;
                c = n;
                raw_next = raw_after;
                pre = pre_n;
                pv = pv_n;
            }
        }
        // (the loop ends when kernel 1 has - its record is there - or on this wave's own error, with
        // the partner possibly still decoding: record() waits for its verdict, so that the code
        // handed back is the one the two-launch path gives)
        const StreamInfo r = there.record();
        if (r.status != dbi::kOk) status = r.status;
        if (status == dbi::kOk) {
            lds_settle();
            const int rest = pos - flushed;           // < 256
            for (int k = lane; k < rest; k += 64) {
                const unsigned b = ring[(flushed + k) & (kRing3 - 1)];
                dst[flushed + k] = (uint8_t)b;
                adler.add1(b, (unsigned)(flushed + k));
            }
            if (r.ended && adler.finish((unsigned)r.produced) != r.adler) status = dbi::kBadChecksum;
            // a stream that ends early (MinKNOW's short final chunk): libhdf5 zero-extends it
            for (int64_t k = pos + lane; k < cap; k += 64) dst[k] = 0;
        }
        if (status != dbi::kOk)                       // nothing of a damaged stream is handed on
            for (int64_t k = lane; k < cap; k += 64) dst[k] = 0;
        if (lane == 0) *status_slot = status;
    }
}

__global__ __launch_bounds__(64) void inflate_resolve_pre_kernel(
    const uint8_t* __restrict__ comp, const dbh_inflate_stream* __restrict__ streams, int n_streams,
    const uint32_t* __restrict__ tokens, StreamInfo* __restrict__ info, uint8_t* out,
    int32_t* __restrict__ status_out) {
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing3];
    const int lane = threadIdx.x;
    for (int i = blockIdx.x; i < n_streams; i += gridDim.x) {
        const dbh_inflate_stream s = streams[i];
        resolve_pre_stream(ring, comp, s, tokens, AllTokensThere{info[i]}, out, status_out + i, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// BOTH KERNELS AS A PAIR OF WAVES PER STREAM (what runs; DEEPBINNER_INFLATE_PAIR=0: two launches).
// A stream is one wave's sequential work in either kernel, and as two launches a container's
// inflating lasts as long as its longest stream takes in kernel 1 PLUS as long as it takes in
// kernel 2 - 4.8 + 3.9 ms for a read of 400 k samples, whatever else the launches hold; for
// containers of long reads (1,000 of ~100 k samples) that, not the kernels' CU time, set the
// streaming path's rate (profiles/r05_k2/long_reads_100k_samples.txt).  Here a workgroup is two
// waves on one stream: wave 0 is kernel 1 and says behind every chunk how many tokens are there
// (a word in LDS, behind a workgroup-scope release: the two waves share the CU's L1, the tokens
// need no more than to have left the wave), wave 1 is kernel 2 and resolves them as they come -
// a step at a time once its tokens, the next step's and the ones prefetched behind them are
// there, sleeping otherwise.  The same tokens, the same schedule, the same bytes; a stream lasts
// as long as the slower of its two halves.
struct PairWords {
    int committed;             // tokens kernel 1 has stored so far
    int done;                  // kernel 1 has ended; the record below is its verdict
    int status, ended, n_tokens, produced;
    unsigned adler;
};
struct PairProgress {
    PairWords* w;
    int lane;
    __device__ __forceinline__ void tokens(int n) const {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // (the tokens have left the wave)
        if (lane == 0) __hip_atomic_store(&w->committed, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
};
struct TokensFromPartner {
    PairWords* w;
    __device__ __forceinline__ void wait(int need, int& limit, bool& done) const {
        for (;;) {
            const int d = __hip_atomic_load(&w->done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const int c = __hip_atomic_load(&w->committed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (d != 0 || c >= need) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                done = d != 0;
                limit = done ? __hip_atomic_load(&w->n_tokens, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                             : 0x7FFFFFFF;
                return;
            }
            __builtin_amdgcn_s_sleep(16);
        }
    }
    __device__ __forceinline__ StreamInfo record() const {
        // (the resolver may get here on an error of its own while the decoder still runs: the words
        // below are written just before `done`)
        while (__hip_atomic_load(&w->done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0)
            __builtin_amdgcn_s_sleep(16);
        StreamInfo r;
        r.status = w->status;
        r.ended = w->ended;
        r.adler = w->adler;
        r.n_tokens = w->n_tokens;
        r.produced = w->produced;
        return r;
    }
};

__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 8))) void inflate_pair_kernel(
    const uint8_t* __restrict__ comp, int64_t comp_total,
    const dbh_inflate_stream* __restrict__ streams, int n_streams, uint32_t* tokens,
    StreamInfo* __restrict__ info, uint8_t* out, int32_t* __restrict__ status_out) {
    __shared__ __attribute__((aligned(16))) WaveLds decode;
    __shared__ __attribute__((aligned(16))) uint8_t ring[kRing3];
    __shared__ PairWords words;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i = blockIdx.x;
    if (i >= n_streams) return;
    if (threadIdx.x == 0) {
        words.committed = 0;
        words.done = 0;
    }
    __syncthreads();
    const dbh_inflate_stream st = streams[i];
    if (wave == 0) {
        const StreamInfo rec = tokens_wave_stream(decode, comp, comp_total, st, tokens, lane,
                                                  PairProgress{&words, lane});
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) {                                   // (lane 0's copy is the stream's state)
            info[i] = rec;
            words.status = rec.status;
            words.ended = rec.ended;
            words.adler = rec.adler;
            words.n_tokens = rec.n_tokens;
            words.produced = (int)rec.produced;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_store(&words.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    } else {
        resolve_pre_stream(ring, comp, st, tokens, TokensFromPartner{&words}, out, status_out + i, lane);
    }
}


thread_local char g_error[256];

// which form of kernel 1 runs: DEEPBINNER_INFLATE_KERNEL=wave (one wavefront per stream) | lane (one
// lane per stream); read at every call, so that a test can hold the two against each other
// which form of kernel 2 runs: DEEPBINNER_INFLATE_RESOLVE=pre (short matches with a complete source
// read at the step boundary, the rest by the exact rule) | rounds (every match through the rounds)
bool resolve_pre() {
    const char* v = std::getenv("DEEPBINNER_INFLATE_RESOLVE");
    if (v && std::strcmp(v, "rounds") == 0) return false;
    return true;
}

// both kernels as one launch, a pair of waves per stream (DEEPBINNER_INFLATE_PAIR=0: two launches)
bool pair_of_waves() {
    const char* v = std::getenv("DEEPBINNER_INFLATE_PAIR");
    return !(v && std::strcmp(v, "0") == 0);
}

bool wave_per_stream() {
    const char* v = std::getenv("DEEPBINNER_INFLATE_KERNEL");
    if (v && std::strcmp(v, "lane") == 0) return false;
    if (v && std::strcmp(v, "wave") == 0) return true;
    return DBI_DEFAULT_WAVE != 0;
}

int hip_failed(hipError_t e, const char* what) {
    std::snprintf(g_error, sizeof(g_error), "%s: %s", what, hipGetErrorString(e));
    return DBH_ERR_HIP;
}
#define DBI_HIP(call)                                          \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return hip_failed(e_, #call);    \
    } while (0)

}  // namespace dbh_inflate_detail

using namespace dbh_inflate_detail;

extern "C" {

const char* dbh_inflate_last_error(void) { return g_error; }

int dbh_inflate_workspace_bytes(int64_t total_out_bytes, int64_t n_streams, size_t* bytes) {
    if (!bytes || total_out_bytes < 0 || n_streams < 0) return DBH_ERR_INVALID_ARGUMENT;
    // one token slot per byte of output (a token yields at least one byte), then the per-stream
    // records of kernel 1
    *bytes = (size_t)total_out_bytes * sizeof(uint32_t) + 512 +
             (size_t)n_streams * sizeof(StreamInfo);
    return DBH_OK;
}

int dbh_inflate_dev(const uint8_t* comp_dev, int64_t comp_bytes,
                    const dbh_inflate_stream* streams_dev, int64_t n_streams,
                    int64_t total_out_bytes, uint8_t* out_dev, void* workspace_dev,
                    int32_t* status_dev, int streams_per_lane, dbh_stream stream) {
    if (n_streams < 0 || n_streams > 0x7FFFFFFF || total_out_bytes < 0 || comp_bytes < 0 ||
        streams_per_lane < 0)
        return DBH_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return DBH_OK;
    if (!comp_dev || !streams_dev || !out_dev || !workspace_dev || !status_dev)
        return DBH_ERR_INVALID_ARGUMENT;
    uint32_t* tokens = (uint32_t*)workspace_dev;
    char* behind = (char*)workspace_dev +
                   (((size_t)total_out_bytes * sizeof(uint32_t) + 255) & ~(size_t)255);
    int* counter = (int*)behind;
    StreamInfo* info = (StreamInfo*)(behind + 256);
    const int n = (int)n_streams;
    DBI_HIP(hipMemsetAsync(counter, 0, sizeof(int), (hipStream_t)stream));
    // the lanes take streams off a counter: with one stream per lane (the default) a launch is
    // as wide as it can be and lasts as long as its longest stream; with several, a fraction of
    // the CUs does the same work in the time the longest stream needs anyway
    if (pair_of_waves() && wave_per_stream() && resolve_pre()) {
        hipLaunchKernelGGL(inflate_pair_kernel, dim3((unsigned)n), dim3(128), 0, (hipStream_t)stream,
                           comp_dev, comp_bytes, streams_dev, n, tokens, info, out_dev, status_dev);
        DBI_HIP(hipGetLastError());
        return DBH_OK;
    }
    if (wave_per_stream()) {
        hipLaunchKernelGGL(inflate_tokens_wave_kernel, dim3((unsigned)n), dim3(dbi::kWaveLanes), 0,
                           (hipStream_t)stream, comp_dev, comp_bytes, streams_dev, n, tokens, info);
    } else {
        const int per_lane = (streams_per_lane > 0 ? streams_per_lane : 1) * kPerLane;
        const int64_t lanes = (n_streams + per_lane - 1) / per_lane;
        hipLaunchKernelGGL(inflate_tokens_kernel, dim3((unsigned)((lanes + kLanes - 1) / kLanes)),
                           dim3(kLanes), 0, (hipStream_t)stream, comp_dev, comp_bytes, streams_dev, n,
                           tokens, info, counter);
    }
    DBI_HIP(hipGetLastError());
    const int groups = (n + kWaves2 - 1) / kWaves2;
    const int blocks = groups < 1024 ? groups : 1024;
    if (resolve_pre())
        hipLaunchKernelGGL(inflate_resolve_pre_kernel, dim3((unsigned)(n < 65536 ? n : 65536)), dim3(64), 0,
                           (hipStream_t)stream, comp_dev, streams_dev, n, (const uint32_t*)tokens,
                           info, out_dev, status_dev);
    else
        hipLaunchKernelGGL(inflate_resolve_kernel, dim3((unsigned)blocks), dim3(64 * kWaves2), 0,
                           (hipStream_t)stream, comp_dev, streams_dev, n, (const uint32_t*)tokens,
                           info, out_dev, status_dev);
    DBI_HIP(hipGetLastError());
    return DBH_OK;
}

int dbh_inflate(const uint8_t* comp_host, size_t comp_bytes, const dbh_inflate_stream* streams_host,
                int64_t n_streams, uint8_t* out_host, size_t out_bytes, int32_t* status_host,
                int streams_per_lane, double* kernel_ms) {
    if (n_streams < 0 || streams_per_lane < 0) return DBH_ERR_INVALID_ARGUMENT;
    if (n_streams == 0) return DBH_OK;
    if (!comp_host || !streams_host || !out_host || !status_host) return DBH_ERR_INVALID_ARGUMENT;
    for (int64_t i = 0; i < n_streams; ++i) {
        const dbh_inflate_stream& s = streams_host[i];
        if (s.comp_offset < 0 || s.comp_bytes < 0 || s.out_offset < 0 || s.out_bytes < 0 ||
            (size_t)(s.comp_offset + s.comp_bytes) > comp_bytes ||
            (size_t)(s.out_offset + s.out_bytes) > out_bytes)
            return DBH_ERR_INVALID_ARGUMENT;
    }
    uint8_t *d_comp = nullptr, *d_out = nullptr;
    dbh_inflate_stream* d_streams = nullptr;
    void* d_work = nullptr;
    int32_t* d_status = nullptr;
    size_t work = 0;
    int st = dbh_inflate_workspace_bytes((int64_t)out_bytes, n_streams, &work);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto cleanup = [&] {
        if (d_comp) (void)hipFree(d_comp);
        if (d_out) (void)hipFree(d_out);
        if (d_streams) (void)hipFree(d_streams);
        if (d_work) (void)hipFree(d_work);
        if (d_status) (void)hipFree(d_status);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    };
    hipError_t e = hipMalloc((void**)&d_comp, comp_bytes + 64);        // (padded: see the header)
    if (e == hipSuccess) e = hipMemset(d_comp + comp_bytes, 0, 64);
    if (e == hipSuccess) e = hipMalloc((void**)&d_out, out_bytes ? out_bytes : 1);
    if (e == hipSuccess) e = hipMalloc((void**)&d_streams, (size_t)n_streams * sizeof(dbh_inflate_stream));
    if (e == hipSuccess) e = hipMalloc(&d_work, work);
    if (e == hipSuccess) e = hipMalloc((void**)&d_status, (size_t)n_streams * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(d_comp, comp_host, comp_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = hipMemcpy(d_streams, streams_host, (size_t)n_streams * sizeof(dbh_inflate_stream),
                      hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, nullptr);
    if (e == hipSuccess && st == DBH_OK)
        st = dbh_inflate_dev(d_comp, (int64_t)comp_bytes, d_streams, n_streams, (int64_t)out_bytes,
                             d_out, d_work, d_status, streams_per_lane, nullptr);
    if (e == hipSuccess) e = hipEventRecord(e1, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out_host, d_out, out_bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess)
        e = hipMemcpy(status_host, d_status, (size_t)n_streams * sizeof(int32_t),
                      hipMemcpyDeviceToHost);
    if (e == hipSuccess && kernel_ms) {
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, e0, e1);
        *kernel_ms = ms;
    }
    cleanup();
    if (e != hipSuccess) return hip_failed(e, "dbh_inflate");
    return st;
}

}  // extern "C"
