// dbh_inflate_core.h - RFC 1950 / 1951 decoding (zlib streams, the HDF5 "deflate" filter that
// fast5 Signal chunks are stored with), phase 1: Huffman decoding of ONE stream by ONE lane into a
// stream of tokens (literal byte | match {length, distance}).  Phase 2 (dbh_inflate.hip) resolves
// the tokens of a stream into bytes with a whole wavefront and the 32 KiB window in LDS.
// (Since round 5 what runs by default gives a stream a whole WAVEFRONT in phase 1 too:
// dbh_inflate_wave.h - the headers, code builds and per-token arithmetic are this header's.)
//
// What the reference does here: h5py -> libhdf5 -> zlib's inflate() on the host, one chunk after
// the other (deepbinner/load_fast5s.py:33-43 reads `Signal[:]`).  Inflating is ~85 % of what loading
// a read costs a CPU core (~100 us per 55 KB read), and a host hands out few cores
// (profiles/r03_cpu_capacity.txt); thousands of independent streams per container are what a GPU
// is good at.
//
// This header is compiled twice: by hipcc into the kernels, and by g++ into the CPU test harness
// (oracle/inflate_host_test.cpp via oracle/Makefile), which runs the SAME decoder lane by lane
// against zlib on the build box.  All per-lane memory is reached through a `Mem` accessor: LDS,
// interleaved by lane, on the device; plain arrays on the host.
//
// How a code is decoded - CANONICALLY, without decode tables: the codes of one length are
// consecutive numbers, and read as 16-bit numbers with the first bit on top (left-justified) the
// ranges of lengths 1, 2, .. 15 follow each other in ascending order.  So
//     length  = 1 + the number of range ENDS that are <= the next 16 bits   (15 compares against
//               per-lane registers: no memory, no branches)
//     symbol  = sorted[first index of that length + (bits - first code of that length)]
// and per lane only the symbols sorted by (length, symbol) live in LDS: 288 + 32 entries and two
// 16-entry {first code, first index} tables - 1.2 KB per lane against the 4.6 KB of zlib-style
// root and sub-tables, which is what decides how many streams a CU decodes at a time (LDS is the
// only thing this kernel runs out of).  The sorted entries of the literal/length code carry the
// decoded meaning (literal byte | end of block | length base and extra bits), so no
// per-symbol arithmetic is left in the loop.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DBI_HD __host__ __device__ __forceinline__
#else
#define DBI_HD inline
#endif

namespace dbi {

#if defined(__HIPCC__)
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
#else
struct U4 {
    uint32_t x, y, z, w;
};
#endif

constexpr int kMaxLens = 320;            // 288 literal/length + 32 distance code lengths
constexpr int kLitSyms = 288;
constexpr int kDistSyms = 32;
constexpr int kRingRows = 32;            // dwords of input a lane keeps in LDS (+ 2 mirror rows)
constexpr int kRingStore = kRingRows + 2;
constexpr int kFetchDwords = 8;          // dwords per request
// First-level decode tables (round 5): index = the next kLitBits / kDistBits bits of the stream
// (first bit lowest), entry = the code's length and what it means - one LDS read and a handful of
// instructions where the canonical method walks a 15-step compare chain and two dependent reads.
// A code longer than the index has no entry (length 0) and goes the canonical way (lockstep: the
// whole wave does, whenever one of its lanes meets one).
// Built, bit-exact and measured in round 5 - and NOT what ships (DBI_LIT_BITS 0 = no tables):
// with 16 lanes per wave and an 11-bit index (5 % of the rounds still go the canonical way) kernel
// 1 takes 11.1 ms instead of 13.0 for 4,000 streams, on four times the waves; with 32 lanes and 10
// bits (39 % of the rounds) 12.8 ms (profiles/r05_inflate/README.md).  The CPU harness compiles
// them in (oracle/inflate_host_test.cpp) and holds every answer against the canonical method.
#ifndef DBI_LIT_BITS
#define DBI_LIT_BITS 0
#endif
#ifndef DBI_DIST_BITS
#define DBI_DIST_BITS 8
#endif
constexpr bool kTables = DBI_LIT_BITS > 0;
constexpr int kLitBits = kTables ? DBI_LIT_BITS : 1, kDistBits = kTables ? DBI_DIST_BITS : 1;
static_assert(kLitBits >= 1 && kLitBits <= 15 && kDistBits >= 1 && kDistBits <= 15, "");
// token: literal = the byte; match = bit 31 | (distance - 1) << 9 | length
constexpr uint32_t kMatchFlag = 0x80000000u;
DBI_HD uint32_t match_token(uint32_t length, uint32_t distance) {
    return kMatchFlag | ((distance - 1u) << 9) | length;
}

enum Status : int {
    kOk = 0,
    kBadHeader = 1,        // not a zlib stream (CM, CINFO, FCHECK, FDICT)
    kBadBlock = 2,         // block type 3, stored LEN/NLEN mismatch
    kBadCodes = 3,         // over-subscribed / incomplete code lengths, bad repeat
    kBadSymbol = 4,        // a code that is not in the table, length/distance symbol out of range
    kTruncated = 5,        // ran out of input
    kTableSpace = 6,       // (unused since the decoder went canonical: every valid code fits)
    kTokenSpace = 7,       // more tokens than the caller's buffer holds
    kBadDistance = 8,      // (phase 2) distance reaches before the start of the output
    kBadChecksum = 9,      // (phase 2) Adler-32 mismatch
    kTooLong = 10,         // (phase 2, whole-stream mode) more output than announced
};

// Phase 2 keeps a stream's last 32 KiB of output in a ring of exactly that size (the largest
// distance deflate knows) and resolves kStepTokens tokens at a time, out of token order: all of a
// step's literals first, then its matches in rounds.  A byte written at position q takes the
// slot of byte q - kWindowRing; a step spans up to 64 x 258 bytes, so a write near the step's end
// can land on what a match near the step's start has yet to read - when that match reaches back
// far enough: its distance plus the bytes from its start to the step's end exceed the ring.  Such
// a step (zlib produces them: 32,000 random bytes twice in a row are 258-byte matches at distance
// 32,000 back to back) is resolved in strict token order instead, where every write follows the
// reads it could disturb.  Shared by the kernel and by the CPU harness's model of it.
constexpr int kWindowRing = 32768;
constexpr int kStepTokens = 64;
DBI_HD bool ring_hazard(int dist, int my, int step_end) {
    return dist + (step_end - my) > kWindowRing;
}

// Phase 2's second form (inflate_resolve_pre_kernel, what runs since the end of round 5).  What
// the tokens of a read look like (level-1 streams of squiggles, oracle/inflate_host_test with
// DBI_K2_STATS): distances are spread over the whole window (19 % within 256 bytes, 75 % within
// 8 K), 99.99 % of the matches are at most kShortMatch bytes long, and only one in ten reads
// anything its own step writes.  So:
//  * a SHORT match whose source lies wholly before its step - four fifths of all - is "pre"
//    (k2_pre): its eight source bytes are read at the boundary in front of the step, behind the
//    last write of the step before, and stored together with the step's literals;
//  * what is left ("late": matches that read what their own step writes, ~3.9 of 64 tokens, and
//    long ones) goes in rounds, by the exact rule: whoever reads nothing that a still-waiting
//    match writes (k2_blocks) - 1.36 rounds per step;
//  * the ring in LDS is kSmallRing bytes, not the window's 32 KiB: what lies further back is read
//    from the stream's OUTPUT in global memory, where the ring's 256-byte pieces go after every
//    step (a source before pos - kSmallRing was flushed long ago).  Twenty streams per CU instead
//    of five: the waits of one are the work of the others;
//  * a step takes as many of its (up to 64) tokens as span at most kStepSpan bytes, so that the
//    ring always holds a whole step and the unflushed bytes before it: during a step that ends at
//    `end` position p is in the ring if p >= end - kSmallRing, and in global memory otherwise
//    (k2_in_ring) - no write of the step can land on a byte that is still to be read from the
//    ring, and the first form's ring hazard (a step in token order) does not exist.
#ifndef DBI_SMALL_RING
#define DBI_SMALL_RING 8192
#endif
constexpr int kSmallRing = DBI_SMALL_RING;
constexpr int kStepSpan = DBI_SMALL_RING / 2;
static_assert(kSmallRing >= kStepSpan + 256 + 8 + 8, "a step, the unflushed bytes before it, an eight-byte read");
DBI_HD bool k2_in_ring(int p, int step_end) { return p >= step_end - kSmallRing; }
constexpr int kShortMatch = 8;
DBI_HD bool k2_pre(bool is_match, int len, int reach, int step_start) {
    return is_match && len <= kShortMatch && reach <= step_start;
}
// does the waiting match that writes [w_start, w_end) hold up the one that reads [src, reach)?
DBI_HD bool k2_blocks(int w_start, int w_end, int src, int reach) {
    return w_start < reach && w_end > src;
}
// the first eight bytes of the endless repetition of v's first `dist` bytes, 1 <= dist < 8 (a match
// that overlaps itself: byte k is byte k mod dist of the dist bytes before it)
DBI_HD uint64_t k2_pattern8(uint64_t v, int dist) {
    const int s1 = 8 * dist;
    v &= (1ull << s1) - 1ull;
    v |= v << s1;
    if (dist <= 3) v |= v << (2 * s1);
    if (dist == 1) v |= v << 32;
    return v;
}

// RFC 1951 section 3.2.5, as arithmetic (device code cannot index host-side constant arrays):
// length symbol 257 + c: c < 8: 3 + c; c < 28: e = (c - 4) / 4 extra bits, base 3 + ((4 + c % 4) << e);
// c = 28: 258.  Distance symbol d: d < 4: d + 1; else e = d / 2 - 1, base 1 + ((2 + d % 2) << e).
DBI_HD int len_extra(int c) { return (c < 8 || c == 28) ? 0 : (c - 4) >> 2; }
DBI_HD int len_base(int c) { return c < 8 ? 3 + c : c == 28 ? 258 : 3 + ((4 + (c & 3)) << len_extra(c)); }
// the order in which the code lengths of the code-length code are stored (section 3.2.7)
DBI_HD int cl_order(int i) {
    // 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15: five bits each in two words
    const uint64_t lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 |
                        9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
    const uint64_t hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 |
                        15ull << 30;
    return (int)((i < 12 ? lo >> (5 * i) : hi >> (5 * (i - 12))) & 31);
}

// what a sorted entry of the literal/length code says (16 bits)
constexpr uint32_t kEntryLength = 0x8000u;   // bits 0-7 base - 3, bits 8-10 extra bits
constexpr uint32_t kEntryEnd = 0x4000u;      // end of block
constexpr uint32_t kEntryBad = 0x2000u;      // symbols 286, 287: in the fixed code, never valid
DBI_HD uint32_t lit_entry(int s) {
    if (s < 256) return (uint32_t)s;
    if (s == 256) return kEntryEnd;
    if (s > 285) return kEntryBad;
    return kEntryLength | ((uint32_t)len_extra(s - 257) << 8) | (uint32_t)(len_base(s - 257) - 3);
}

// literal/length entry (16 bits): bits 0-3 code length (0 = no entry), 4-6 extra bits of a length
// symbol (6 = symbols 286 / 287: never valid, 7 = end of block), 7-14 the literal byte or the
// length's base - 3, bit 15 = not a literal
DBI_HD uint32_t lit_tab_entry(int s, int len) {
    uint32_t e;
    if (s < 256) e = (uint32_t)s << 7;
    else if (s == 256) e = 0x8000u | (7u << 4);
    else if (s > 285) e = 0x8000u | (6u << 4);
    else e = 0x8000u | ((uint32_t)len_extra(s - 257) << 4) | ((uint32_t)(len_base(s - 257) - 3) << 7);
    return e | (uint32_t)len;
}
// distance entry: bits 0-3 code length (0 = no entry), 4-8 the distance symbol
DBI_HD uint32_t dist_tab_entry(int s, int len) { return ((uint32_t)s << 4) | (uint32_t)len; }

DBI_HD uint32_t low_bits(uint32_t v, uint32_t n) {      // n <= 16
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(v, 0u, n);
#else
    return v & ((1u << n) - 1u);
#endif
}
// the next 16 bits of the stream as a number with the FIRST bit on top (Huffman codes are packed
// first bit first, everything else in deflate lowest bit first)
DBI_HD uint32_t first16(uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
#ifdef DBI_NO_SDWA
    uint32_t r = __builtin_bitreverse32(w) >> 16;
    asm volatile("" : "+v"(r));      // (a value of its own: not a sub-dword selection in every user)
    return r;
#endif
    return __builtin_bitreverse32(w) >> 16;
#else
    uint32_t r = 0;
    for (int k = 0; k < 16; ++k) r |= ((w >> k) & 1u) << (15 - k);
    return r;
#endif
}
DBI_HD uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }

// The input side of a lane: a ring of 32 dwords in LDS ([row][lane]: whatever row each lane
// wants, the lanes of a wave hit 64 different banks), filled eight dwords at a time by a request
// issued at one checkpoint and written to the ring at the next, so that nobody waits for global
// memory; rows 32 and 33 mirror rows 0 and 1, so that the 64-bit window at any bit position is
// three consecutive rows.  Nothing is consumed without being checked: a stream that runs beyond
// its last byte is told so (overrun), whatever the ring held there.
struct BitReader {
    const uint8_t* in;
    uint32_t limit_bits;   // bits of the stream (deflate data + trailer): streams are < 512 MB
    uint32_t bp;           // bits consumed so far = position of the next bit
    uint32_t wr;           // dwords written to the ring so far (a multiple of 8)
    uint32_t fetch_cap;    // a request may start no further than this many bytes behind `in`
    U4 pend0, pend1;       // a request in flight
    int pending;

    DBI_HD U4 load16(uint32_t at) const {
        U4 v;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_memcpy(&v, in + at, 16);     // (one unaligned 16-byte global load)
#else
        uint32_t w[4];
        for (int k = 0; k < 4; ++k) {
            const uint8_t* p = in + at + 4 * k;
            w[k] = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) |
                   ((uint32_t)p[3] << 24);
        }
        v.x = w[0];
        v.y = w[1];
        v.z = w[2];
        v.w = w[3];
#endif
        return v;
    }
    DBI_HD uint32_t level() const { return wr - (bp >> 5); }     // dwords not yet left behind
    DBI_HD void request() {
        // (beyond fetch_cap lies the end of the caller's buffer; what a lane reads instead is
        // never looked at: its stream has ended at least 32 bytes before)
        const uint32_t at = umin(wr * 4u, fetch_cap);
        pend0 = load16(at);
        pend1 = load16(at + 16u);
        pending = 1;
    }
    template <class Mem>
    DBI_HD void commit(Mem& mem) {
        const int row = (int)(wr & (uint32_t)(kRingRows - 1));
        mem.set_ring(row + 0, pend0.x);
        mem.set_ring(row + 1, pend0.y);
        mem.set_ring(row + 2, pend0.z);
        mem.set_ring(row + 3, pend0.w);
        mem.set_ring(row + 4, pend1.x);
        mem.set_ring(row + 5, pend1.y);
        mem.set_ring(row + 6, pend1.z);
        mem.set_ring(row + 7, pend1.w);
        if (row == 0) {
            mem.set_ring(kRingRows + 0, pend0.x);
            mem.set_ring(kRingRows + 1, pend0.y);
        }
        wr += (uint32_t)kFetchDwords;
        pending = 0;
    }
    // The checkpoint of the hot loop, every four tokens (<= 6 dwords consumed in between).  A
    // request goes out EARLY (as soon as the ring has room for it: level <= 24) and is written to
    // the ring LATE (when the level is down to 16 dwords): at ~1.6 dwords per checkpoint it has
    // some five checkpoints to make its way from HBM - with the round-4 policy (requested at one
    // checkpoint, committed at the next) a decoder faster than ~600 cycles per token waited for
    // memory at every fifth checkpoint, and the first-level tables bought nothing (round 5).
    // Level after a checkpoint >= 11 (17 with a request pending, minus 6), never above 24: the
    // ring neither runs dry nor is overwritten where it is still to be read.
    template <class Mem>
    DBI_HD void checkpoint(Mem& mem) {
        if (pending && level() <= 16u) commit(mem);
        if (!pending && level() <= 24u) request();
    }
    // The rare paths (headers, stored bytes) ask before every field instead: at least `need`
    // (<= 16) dwords in the ring.
    template <class Mem>
    DBI_HD void ensure(Mem& mem, uint32_t need) {
        while (level() < need) {
            if (!pending) request();
            commit(mem);
        }
    }
    template <class Mem>
    DBI_HD void start(Mem& mem, const uint8_t* data, int64_t n_bytes, int64_t readable_bytes) {
        in = data;
        limit_bits = (uint32_t)n_bytes * 8u;
        fetch_cap = (uint32_t)(readable_bytes - 32);
        bp = 0;
        wr = 0;
        pending = 0;
        ensure(mem, 16u);
    }
    // Go on at another bit of the stream (the one-wave-per-stream decoder - dbh_inflate_wave.h -
    // reads the Huffman blocks its own way and comes back here for the next block header): the
    // ring starts over at the block of eight dwords that bit lies in.
    template <class Mem>
    DBI_HD void seek(Mem& mem, uint32_t to_bit) {
        bp = to_bit;
        wr = (to_bit >> 5) & ~(uint32_t)(kFetchDwords - 1);
        request();          // (wr <= bp / 32 < wr + 8: level() is meaningful behind this block)
        commit(mem);
        ensure(mem, 16u);
    }
    // the next 64 bits
    template <class Mem>
    DBI_HD void window64(const Mem& mem, uint32_t& lo, uint32_t& hi) const {
        const int row = (int)((bp >> 5) & (uint32_t)(kRingRows - 1));
        const uint32_t sh = bp & 31u;
        const uint32_t d0 = mem.ring(row), d1 = mem.ring(row + 1), d2 = mem.ring(row + 2);
#if defined(__HIP_DEVICE_COMPILE__)
        lo = __builtin_amdgcn_alignbit(d1, d0, sh);
        hi = __builtin_amdgcn_alignbit(d2, d1, sh);
#else
        lo = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh);
        hi = (uint32_t)((((uint64_t)d2 << 32) | d1) >> sh);
#endif
    }
    template <class Mem>
    DBI_HD uint32_t take(Mem& mem, int n) {          // n <= 16; rare paths only
        ensure(mem, 4u);
        uint32_t lo, hi;
        window64(mem, lo, hi);
        bp += (uint32_t)n;
        return low_bits(lo, (uint32_t)n);
    }
    DBI_HD uint32_t consumed_bits() const { return bp; }
    DBI_HD bool overrun() const { return bp > limit_bits; }
    DBI_HD int64_t limit_bytes() const { return (int64_t)(limit_bits >> 3); }
};

// One lane's decoder state between iterations of the lockstep loop.
struct Lane {
    BitReader br;
    // left-justified ends of the code ranges of lengths 1 .. 15 (65,536 = a complete code's last)
    uint32_t lim_lit[15], lim_dist[15];
    int out_pos, out_cap;      // bytes produced / wanted (a stream's output is < 2 GB)
    int state;                 // see below
    int status;
    int final_block;
    int stored_left;
    int ended;                 // the deflate data ended (final end-of-block) with exactly the bytes
                               // produced: phase 2 may zero-extend and check the Adler-32
    uint32_t adler;            // ... against this, the four bytes behind the deflate data
};
// A lane does not stop AT the wanted number of bytes but at the first token BEYOND it: if the
// stream ends before any such token comes (end-of-block codes, empty blocks), the whole stream
// was decoded, and phase 2 can check its Adler-32.
enum LaneState : int { kNeedBlock = 0, kDecode = 1, kStored = 2, kDone = 3 };

// the first-level table of a code: every index whose low `len` bits are the code (read first bit
// lowest) gets the symbol's entry; a code longer than the index blanks the one index it begins with
template <int BITS, class Set>
DBI_HD void fill_table(uint32_t c16, int len, uint32_t entry, const Set& set) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t rev = __builtin_bitreverse32(c16) >> 16;      // the code's first bit lowest
#else
    uint32_t rev = 0;
    for (int k = 0; k < 16; ++k) rev |= ((c16 >> (15 - k)) & 1u) << k;
#endif
    if (len > BITS) {
        set((int)(rev & ((1u << BITS) - 1u)), 0u);
        return;
    }
    for (uint32_t k = rev & ((1u << len) - 1u); k < (1u << BITS); k += 1u << len) set((int)k, entry);
}
template <class Mem>
struct LitCode {
    Mem* m;
    static constexpr int kTableBits = kTables ? kLitBits : 0;
    DBI_HD void set_pair(int l, uint32_t v) const { m->set_lit_pair(l, v); }
    DBI_HD uint32_t pair(int l) const { return m->lit_pair(l); }
    DBI_HD void set_sym(int at, int s) const { m->set_lit_sym(at, lit_entry(s)); }
    DBI_HD void clear_table() const {
        for (int k = 0; k < (1 << kLitBits); ++k) m->set_lit_tab(k, 0u);
    }
    DBI_HD void fill(int s, int len, uint32_t c16) const {
        Mem* mm = m;
        fill_table<kLitBits>(c16, len, lit_tab_entry(s, len), [mm](int k, uint32_t v) { mm->set_lit_tab(k, v); });
    }
};
template <class Mem>
struct DistCode {
    Mem* m;
    static constexpr int kTableBits = kTables ? kDistBits : 0;
    DBI_HD void set_pair(int l, uint32_t v) const { m->set_dist_pair(l, v); }
    DBI_HD uint32_t pair(int l) const { return m->dist_pair(l); }
    DBI_HD void set_sym(int at, int s) const { m->set_dist_sym(at, (uint32_t)s); }
    DBI_HD void clear_table() const {
        for (int k = 0; k < (1 << kDistBits); ++k) m->set_dist_tab(k, 0u);
    }
    DBI_HD void fill(int s, int len, uint32_t c16) const {
        Mem* mm = m;
        fill_table<kDistBits>(c16, len, dist_tab_entry(s, len), [mm](int k, uint32_t v) { mm->set_dist_tab(k, v); });
    }
};
// The code-length code of a dynamic block: built where the distance code's sorted symbols will
// be.  Its codes are at most 7 bits long, so a table of 128 bytes {symbol << 3 | code length}
// answers every one of them - where the decoder's memory has room for it (Mem::kClTableBits = 7:
// the one-wavefront-per-stream form, whose block headers - ~300 code lengths each, decoded by
// lane 0 alone, one after the other - are a third of its instructions; 0: one lane per stream,
// 64 tables per wave).
template <class Mem>
struct CodeLengthCode {
    Mem* m;
    static constexpr int kTableBits = Mem::kClTableBits;
    DBI_HD void set_pair(int l, uint32_t v) const { m->set_dist_pair(l, v); }
    DBI_HD uint32_t pair(int l) const { return m->dist_pair(l); }
    DBI_HD void set_sym(int at, int s) const { m->set_dist_sym(at, (uint32_t)s); }
    DBI_HD void clear_table() const {
        for (int k = 0; k < 128; ++k) m->set_cl_tab(k, 0u);
    }
    DBI_HD void fill(int s, int len, uint32_t c16) const {
        Mem* mm = m;
        fill_table<7>(c16, len, ((uint32_t)s << 3) | (uint32_t)len, [mm](int k, uint32_t v) { mm->set_cl_tab(k, v); });
    }
};

// Prepares one code for decoding: `lens[first .. first+n)` are the code lengths.  Leaves the
// range ends in `lim`, {first code, first index} per length and the symbols sorted by (length,
// symbol) behind `code`.  -> kOk / kBadCodes.  An incomplete code is tolerated only where zlib
// tolerates it (inftrees.c): a literal/length or distance code made of a single code of length
// 1, never the code-length code; no codes at all is accepted (a block of literals only has no
// distance codes) and leaves a code nothing decodes with.
template <class Mem, class Code>
DBI_HD int build_code(Mem& mem, Code code, int first, int n, uint32_t (&lim)[15],
                      bool may_be_incomplete) {
    for (int l = 0; l < 16; ++l) mem.set_cnt(l, 0);
    for (int s = 0; s < n; ++s) {
        const int l = mem.len(first + s);
        mem.set_cnt(l, mem.cnt(l) + 1);
    }
    int left = 1, max = 0;
    uint32_t next_code = 0, index = 0;
    bool over = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int l = 1; l <= 15; ++l) {
        const uint32_t c = (uint32_t)mem.cnt(l);
        left = (left << 1) - (int)c;
        over = over || left < 0;
        if (c) max = l;
        code.set_pair(l, ((next_code << (16 - l)) & 0xFFFFu) | (index << 16));
        mem.set_cnt(l, (int)index);           // from here on: where the next symbol of length l goes
        next_code += c;
        index += c;
        lim[l - 1] = next_code << (16 - l);
        next_code <<= 1;
    }
    if (over) return kBadCodes;                                            // over-subscribed
    if (max != 0 && left > 0 && (!may_be_incomplete || max != 1)) return kBadCodes;   // incomplete
    // (a complete code writes every index of its first-level table; one that is not - a single
    // code of length 1, or none at all - leaves the rest without an entry)
    if (Code::kTableBits > 0 && (left > 0 || max == 0)) code.clear_table();
    for (int s = 0; s < n; ++s) {
        const int l = mem.len(first + s);
        if (l != 0) {
            const int at = mem.cnt(l);
            code.set_sym(at, s);
            mem.set_cnt(l, at + 1);
            if (Code::kTableBits > 0) {
                // its code, left-justified: the first code of its length + its place among them
                const uint32_t pair = code.pair(l);
                const uint32_t c16 = (pair & 0xFFFFu) + (((uint32_t)at - (pair >> 16)) << (16 - l));
                code.fill(s, l, c16);
            }
        }
    }
    return kOk;
}

// The length of the code the 16 bits `c` begin with: 1 .. N, or N + 1 = no such code.
// (c - end is negative exactly where c lies below the end of a range: the sign bits are shifted
// into one word and counted - two instructions per length, no condition codes)
template <int N>
DBI_HD uint32_t code_length(uint32_t c, const uint32_t (&lim)[15]) {
    uint32_t below = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int l = 0; l < N; ++l) below = __builtin_amdgcn_alignbit(below, c - lim[l], 31);
    return (uint32_t)(N + 1) - (uint32_t)__builtin_popcount(below);
#else
    for (int l = 0; l < N; ++l) below += (c - lim[l]) >> 31;
    return (uint32_t)(N + 1) - below;
#endif
}
// The same for N slots at once, the slots' chains interleaved instruction by instruction: a
// vector instruction that needs the result of the one before it issues later than one that does
// not, and the chain of one slot is 15 dependent instructions long.
template <int NL, int N>
DBI_HD void code_lengths(const uint32_t (&c)[N], const uint32_t* const (&lim)[N], uint32_t (&out)[N]) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t below[N];
#pragma unroll
    for (int s = 0; s < N; ++s) below[s] = 0;
#pragma unroll
    for (int l = 0; l < NL; ++l) {
#pragma unroll
        for (int s = 0; s < N; ++s) below[s] = __builtin_amdgcn_alignbit(below[s], c[s] - lim[s][l], 31);
    }
#pragma unroll
    for (int s = 0; s < N; ++s) out[s] = (uint32_t)(NL + 1) - (uint32_t)__builtin_popcount(below[s]);
#else
    for (int s = 0; s < N; ++s) {
        uint32_t below = 0;
        for (int l = 0; l < NL; ++l) below += (c[s] - lim[s][l]) >> 31;
        out[s] = (uint32_t)(NL + 1) - below;
    }
#endif
}
// index of its symbol among the sorted ones, from the {first code, first index} of its length
DBI_HD uint32_t sorted_index(uint32_t c, uint32_t len, uint32_t pair) {
    return (pair >> 16) + ((c - (pair & 0xFFFFu)) >> (16u - len));
}

DBI_HD void lane_fail(Lane& L, int status) {
    L.status = status;
    L.state = kDone;
}

// Zlib header -> lane ready for its first block.  `readable_bytes`: how far behind `data` the
// caller's buffer can be read (>= n_bytes + 32).
template <class Mem>
DBI_HD void lane_start(Lane& L, Mem& mem, const uint8_t* data, int64_t n_bytes, int64_t out_cap,
                       int64_t readable_bytes) {
    L.adler = 0;
    L.out_pos = 0;
    L.out_cap = (int)out_cap;
    L.state = kNeedBlock;
    L.status = kOk;
    L.final_block = 0;
    L.stored_left = 0;
    L.ended = 0;
    for (int l = 0; l < 15; ++l) L.lim_lit[l] = L.lim_dist[l] = 0;
    L.br.in = data;
    L.br.limit_bits = 0;
    L.br.bp = L.br.wr = 0;
    L.br.pending = 0;
    L.br.fetch_cap = 0;
    if (n_bytes < 6 || n_bytes >= (1 << 29) || out_cap >= (1ll << 31)) return lane_fail(L, kTruncated);
    L.br.start(mem, data, n_bytes, readable_bytes);
    const uint32_t cmf = L.br.take(mem, 8), flg = L.br.take(mem, 8);
    if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20))
        return lane_fail(L, kBadHeader);
    if (out_cap < 0) L.state = kDone;
}

// The deflate data has ended: the Adler-32 of the output follows at the next byte boundary.
DBI_HD void lane_ended(Lane& L) {
    const int64_t at = (int64_t)((L.br.consumed_bits() + 7u) >> 3);
    if (at + 4 > L.br.limit_bytes()) return lane_fail(L, kTruncated);
    const uint8_t* p = L.br.in + at;
    L.adler = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
    L.ended = 1;
    L.state = kDone;
}

// Block header (and, for a dynamic block, its code lengths and both codes).
template <class Mem>
DBI_HD void lane_block(Lane& L, Mem& mem) {
    BitReader& br = L.br;
    L.final_block = (int)br.take(mem, 1);
    const int type = (int)br.take(mem, 2);
    if (type == 3) return lane_fail(L, kBadBlock);
    if (type == 0) {
        br.bp += (8u - (br.bp & 7u)) & 7u;              // to the byte boundary
        const uint32_t len = br.take(mem, 16);
        const uint32_t nlen = br.take(mem, 16);
        if ((len ^ 0xFFFFu) != nlen) return lane_fail(L, kBadBlock);
        if (br.overrun()) return lane_fail(L, kTruncated);
        L.stored_left = (int)len;
        L.state = len ? kStored : kNeedBlock;
        if (!len && L.final_block) lane_ended(L);
        return;
    }
    int hlit = 288, hdist = 32;
    if (type == 1) {
        for (int s = 0; s < 288; ++s) mem.set_len(s, s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
        // (32 five-bit distance codes make the fixed code complete; 30 and 31 never occur in
        // valid data and are refused when decoded)
        for (int s = 0; s < 32; ++s) mem.set_len(288 + s, 5);
    } else {
        hlit = (int)br.take(mem, 5) + 257;
        hdist = (int)br.take(mem, 5) + 1;
        const int hclen = (int)br.take(mem, 4) + 4;
        if (hlit > 286 || hdist > 30) return lane_fail(L, kBadCodes);
        for (int i = 0; i < 19; ++i) mem.set_len(i, 0);
        for (int i = 0; i < hclen; ++i) mem.set_len(cl_order(i), (int)br.take(mem, 3));
        // the code-length code: sorted symbols and pairs where the distance code's will be
        uint32_t lim_cl[15];
        int st = build_code(mem, CodeLengthCode<Mem>{&mem}, 0, 19, lim_cl, false);
        if (st != kOk) return lane_fail(L, st);
        int have = 0, prev = 0;
        const int want = hlit + hdist;
        while (have < want) {
            // one 64-bit window of the stream serves as many code lengths as fit: a code and its
            // extra bits are at most 7 + 7
            br.ensure(mem, 4u);
            uint32_t lo, hi;
            br.window64(mem, lo, hi);
            const uint64_t w = ((uint64_t)hi << 32) | lo;
            uint32_t used = 0;
            while (have < want && used <= 50u) {
                const uint32_t v = (uint32_t)(w >> used);
                uint32_t cl;
                int s;
                if (Mem::kClTableBits > 0) {
                    const uint32_t e = mem.cl_tab((int)(v & 127u));
                    cl = e & 7u;
                    s = (int)(e >> 3);
                    if (cl == 0u) return lane_fail(L, kBadSymbol);
                } else {
                    const uint32_t c = first16(v);
                    cl = code_length<7>(c, lim_cl);
                    if (cl > 7u) return lane_fail(L, kBadSymbol);
                    s = (int)mem.dist_sym((int)umin(sorted_index(c, cl, mem.dist_pair((int)cl)), 18u));
                }
                used += cl;
                if (s < 16) {
                    mem.set_len(have++, s);
                    prev = s;
                    continue;
                }
                const uint32_t x = (uint32_t)(w >> used);
                int rep, val = 0;
                if (s == 16) {
                    if (have == 0) return lane_fail(L, kBadCodes);
                    val = prev;
                    rep = 3 + (int)(x & 3u);
                    used += 2u;
                } else if (s == 17) {
                    rep = 3 + (int)(x & 7u);
                    used += 3u;
                } else {
                    rep = 11 + (int)(x & 127u);
                    used += 7u;
                }
                if (have + rep > want) return lane_fail(L, kBadCodes);
                for (int k = 0; k < rep; ++k) mem.set_len(have++, val);
                if (s != 16) prev = 0;
            }
            br.bp += used;
        }
        if (br.overrun()) return lane_fail(L, kTruncated);
        if (mem.len(256) == 0) return lane_fail(L, kBadCodes);          // no end-of-block code
    }
    // lens[0 .. hlit) literal/length, lens[hlit .. hlit + hdist) distance
    int st = build_code(mem, LitCode<Mem>{&mem}, 0, hlit, L.lim_lit, true);
    if (st == kOk) st = build_code(mem, DistCode<Mem>{&mem}, hlit, hdist, L.lim_dist, true);
    if (st != kOk) return lane_fail(L, st);
    // (the hot loop's checkpoints keep the ring filled only if it is entered well filled)
    br.ensure(mem, 16u);
    L.state = kDecode;
}

// One byte of a stored block (rare: a chunk deflate could not shrink).  Returns true with the
// byte as a literal token in *token.
template <class Mem>
DBI_HD bool lane_stored(Lane& L, Mem& mem, uint32_t* token) {
    BitReader& br = L.br;
    if (L.out_pos >= L.out_cap) {          // more data than wanted
        L.state = kDone;
        return false;
    }
    const uint32_t byte = br.take(mem, 8);
    if (br.overrun()) {
        lane_fail(L, kTruncated);
        return false;
    }
    *token = byte;
    L.out_pos += 1;
    if (--L.stored_left == 0) {
        L.state = kNeedBlock;
        if (L.final_block) lane_ended(L);
    }
    return true;
}

// The hot path: ONE token of a lane that is inside a Huffman block - straight-line code, the
// same for a literal, a match and an end of block (all lanes of a wave run it together, so a
// branch would be taken by somebody every time): one 64-bit window (a token is at most 15 + 5 +
// 15 + 13 bits), both codes decoded from it, the results selected.
//   In two halves, because a token is ONE dependent chain - window, code length, two dependent
// LDS reads, selects, the second code the same again - and a wave that decodes alone on its SIMD
// (a container is 4,000 streams: 63 waves for 1,024 SIMDs) has nothing to fill the latencies with:
// lane_decode_front only LOOKS (no state changes), so a kernel whose lanes carry two streams each
// can run the two fronts side by side in one basic block - the compiler interleaves the two chains
// - and commit them one after the other (dbh_inflate.hip).
struct Decoded {
    uint32_t used, length, token;
    bool bad, fail, is_end, beyond;
};
// all ones if bit `bit` of v is set, else zero (selects below are AND / OR with such masks: a
// ternary may become a branch, and a branch ends the basic block the two chains share)
DBI_HD uint32_t bit_mask(uint32_t v, int bit) { return (uint32_t)((int32_t)(v << (31 - bit)) >> 31); }
DBI_HD uint32_t pick(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }

// LDS reads whose latency the front hides behind the OTHER slot's arithmetic: on the device they are
// issued as they are asked for and waited for where their value is needed, with a count of the
// requests that may still be in flight behind them (LDS answers in order) - the compiler's own
// wait would be for everything, right behind the request.  On the host they are plain reads.
template <class Mem>
DBI_HD void issue_ring3(const Mem& mem, int row, uint32_t& a, uint32_t& b, uint32_t& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    mem.ring3_issue(row, a, b, c);
#else
    a = mem.ring(row);
    b = mem.ring(row + 1);
    c = mem.ring(row + 2);
#endif
}
template <class Mem>
DBI_HD void issue_lit_pair(const Mem& mem, int l, uint32_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    mem.lit_pair_issue(l, v);
#else
    v = mem.lit_pair(l);
#endif
}
template <class Mem>
DBI_HD void issue_dist_pair(const Mem& mem, int l, uint32_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    mem.dist_pair_issue(l, v);
#else
    v = mem.dist_pair(l);
#endif
}
template <class Mem>
DBI_HD void issue_lit_sym(const Mem& mem, int i, uint32_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    mem.lit_sym_issue(i, v);
#else
    v = mem.lit_sym(i);
#endif
}
template <class Mem>
DBI_HD void issue_dist_sym(const Mem& mem, int i, uint32_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    mem.dist_sym_issue(i, v);
#else
    v = mem.dist_sym(i);
#endif
}
// everything but the PENDING youngest LDS requests has landed
template <int PENDING>
DBI_HD void lds_landed() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PENDING) : "memory");
#endif
}
DBI_HD void pin(uint32_t& v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(v));        // (uses of v stay behind the wait in front of this)
#else
    (void)v;
#endif
}

// The fronts of N slots (N = 1: the CPU harness and lane_decode; N = 2: the kernel), without a
// branch and in LOCKSTEP over the slots: every statement is made for all slots before the next
// one, so that the instruction stream alternates between the slots' (independent) chains - a
// vector instruction that depends on its predecessor issues later than one that does not - and
// each stage's LDS reads are in flight together.
template <int N, class Mem>
DBI_HD void lane_decode_fronts(const Lane (&L)[N], const Mem (&mem)[N], Decoded (&out)[N]) {
    uint32_t d0[N], d1[N], d2[N], lo[N], hi[N], sh[N];
    uint32_t c1[N], n1[N], l1[N], pair1[N], e[N], i1[N];
    uint32_t is_len[N], eb[N], length[N], c2[N], n2[N], l2[N], pair2[N], d[N], i2[N];
    uint32_t wl[N], wh[N], not_end[N], ext[N];
    const uint32_t* lim1[N];
    const uint32_t* lim2[N];
#if defined(__HIP_DEVICE_COMPILE__)
#define DBI_SLOTS _Pragma("unroll") for (int s = 0; s < N; ++s)
#else
#define DBI_SLOTS for (int s = 0; s < N; ++s)
#endif
    DBI_SLOTS {
        lim1[s] = L[s].lim_lit;
        lim2[s] = L[s].lim_dist;
    }
    DBI_SLOTS issue_ring3(mem[s], (int)((L[s].br.bp >> 5) & (uint32_t)(kRingRows - 1)), d0[s], d1[s], d2[s]);
    DBI_SLOTS sh[s] = L[s].br.bp & 31u;
    lds_landed<0>();
    DBI_SLOTS {
        pin(d0[s]);
        pin(d1[s]);
        pin(d2[s]);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    DBI_SLOTS lo[s] = __builtin_amdgcn_alignbit(d1[s], d0[s], sh[s]);
    DBI_SLOTS hi[s] = __builtin_amdgcn_alignbit(d2[s], d1[s], sh[s]);
#else
    DBI_SLOTS lo[s] = (uint32_t)((((uint64_t)d1[s] << 32) | d0[s]) >> sh[s]);
    DBI_SLOTS hi[s] = (uint32_t)((((uint64_t)d2[s] << 32) | d1[s]) >> sh[s]);
#endif
    // literal / length: the code's length, then {first code, first index} of that length ...
    DBI_SLOTS c1[s] = first16(lo[s]);
    code_lengths<15, N>(c1, lim1, n1);
    DBI_SLOTS l1[s] = umin(n1[s], 15u);
    DBI_SLOTS issue_lit_pair(mem[s], (int)l1[s], pair1[s]);
    lds_landed<0>();
    DBI_SLOTS pin(pair1[s]);
    // ... its entry among the sorted symbols
    DBI_SLOTS i1[s] = umin(sorted_index(c1[s], l1[s], pair1[s]), (uint32_t)(kLitSyms - 1));
    DBI_SLOTS issue_lit_sym(mem[s], (int)i1[s], e[s]);
    // (the window behind the code, while the entry travels)
    DBI_SLOTS {
        const uint64_t w = (((uint64_t)hi[s] << 32) | lo[s]) >> l1[s];
        wl[s] = (uint32_t)w;
        wh[s] = (uint32_t)(w >> 32);
    }
    lds_landed<0>();
    DBI_SLOTS pin(e[s]);
    DBI_SLOTS is_len[s] = bit_mask(e[s], 15);                          // kEntryLength
    DBI_SLOTS eb[s] = (e[s] >> 8) & 7u & is_len[s];
    DBI_SLOTS not_end[s] = ((e[s] >> 14) & 1u) ^ 1u;                   // kEntryEnd: length 0, a literal: 1
    DBI_SLOTS ext[s] = low_bits(wl[s], eb[s]);
    DBI_SLOTS length[s] = pick(is_len[s], 3u + (e[s] & 0xFFu) + ext[s], not_end[s]);
    DBI_SLOTS {
        const uint64_t w = (((uint64_t)wh[s] << 32) | wl[s]) >> eb[s];
        wl[s] = (uint32_t)w;
        wh[s] = (uint32_t)(w >> 32);
    }
    // distance (decoded whatever the symbol was; only looked at behind a length)
    DBI_SLOTS c2[s] = first16(wl[s]);
    code_lengths<15, N>(c2, lim2, n2);
    DBI_SLOTS l2[s] = umin(n2[s], 15u);
    DBI_SLOTS issue_dist_pair(mem[s], (int)l2[s], pair2[s]);
    lds_landed<0>();
    DBI_SLOTS pin(pair2[s]);
    DBI_SLOTS i2[s] = umin(sorted_index(c2[s], l2[s], pair2[s]), (uint32_t)(kDistSyms - 1));
    DBI_SLOTS issue_dist_sym(mem[s], (int)i2[s], d[s]);
    DBI_SLOTS {
        const uint64_t w = (((uint64_t)wh[s] << 32) | wl[s]) >> l2[s];
        wl[s] = (uint32_t)w;             // the distance's extra bits
    }
    lds_landed<0>();
    DBI_SLOTS pin(d[s]);
    DBI_SLOTS {
        const uint32_t half = d[s] >> 1;
        const uint32_t db = (half > 1u ? half : 1u) - 1u;
        const uint32_t small = (uint32_t)((int32_t)(d[s] - 4u) >> 31);          // d < 4
        const uint32_t dbase = pick(small, d[s] + 1u, 1u + ((2u | (d[s] & 1u)) << db));
        const uint32_t distance = dbase + low_bits(wl[s], db);
        const bool bad = n1[s] > 15u || (e[s] & kEntryBad) != 0 ||
                         (is_len[s] != 0u && (n2[s] > 15u || d[s] > 29u));
        Decoded& r = out[s];
        r.used = l1[s] + eb[s] + ((l2[s] + db) & is_len[s]);
        r.bad = bad;
        r.fail = bad || L[s].br.bp + r.used > L[s].br.limit_bits;
        // more data than wanted: a match keeps what is; the lane stops
        const int room = L[s].out_cap - L[s].out_pos;
        r.beyond = (int)length[s] > room;
        const uint32_t fits = umin(length[s], (uint32_t)(room > 0 ? room : 0));
        r.length = fits;
        r.is_end = (e[s] & kEntryEnd) != 0;
        r.token = pick(is_len[s], match_token(fits, distance), e[s] & 0xFFu);
    }
#undef DBI_SLOTS
}
template <class Mem>
DBI_HD Decoded lane_decode_front(const Lane& L, const Mem& mem) {
    // (one slot: references into one-element arrays)
    Decoded out[1];
    lane_decode_fronts<1, Mem>(reinterpret_cast<const Lane(&)[1]>(L),
                               reinterpret_cast<const Mem(&)[1]>(mem), out);
    return out[0];
}
// A lane in any state but kDecode passes through unchanged.  Returns true with *token set when
// a token was produced.
DBI_HD bool lane_decode_commit(Lane& L, const Decoded& r, uint32_t* token) {
    if (L.state != kDecode) return false;
    L.br.bp += r.used;
    if (r.fail) {
        lane_fail(L, r.bad ? kBadSymbol : kTruncated);
        return false;
    }
    L.out_pos += (int)r.length;
    if (r.beyond) {
        L.state = kDone;
    } else if (r.is_end) {
        L.state = kNeedBlock;
        if (L.final_block) lane_ended(L);
    }
    *token = r.token;
    return r.length > 0u;
}
// The front through the first-level tables.  Returns false where the token's literal/length code
// - or, behind a length, its distance code - has no entry (it is longer than the table's index):
// `r` is then not to be used and lane_decode_front decides.
template <class Mem>
DBI_HD bool lane_decode_fast(const Lane& L, const Mem& mem, Decoded& r) {
    const BitReader& br = L.br;
    uint32_t lo, hi;
    br.window64(mem, lo, hi);
    const uint32_t e = mem.lit_tab((int)(lo & ((1u << kLitBits) - 1u)));
    const uint32_t l1 = e & 15u;
    const uint32_t not_lit = bit_mask(e, 15);
    const uint32_t ebf = (e >> 4) & 7u;                        // 6: a bad symbol, 7: end of block
    const uint32_t is_len = not_lit & (uint32_t)((int32_t)(ebf - 6u) >> 31);       // ebf < 6
    const uint32_t eb = ebf & is_len;
    const uint32_t val = (e >> 7) & 0xFFu;
    uint64_t w = (((uint64_t)hi << 32) | lo) >> l1;
    // a literal: 1; end of block (or a bad symbol, refused below): 0
    const uint32_t length = pick(is_len, 3u + val + low_bits((uint32_t)w, eb), (~not_lit) & 1u);
    w >>= eb;
    const uint32_t e2 = mem.dist_tab((int)((uint32_t)w & ((1u << kDistBits) - 1u)));
    const uint32_t l2 = e2 & 15u;
    const uint32_t d = e2 >> 4;
    const uint32_t half = d >> 1;
    const uint32_t db = (half > 1u ? half : 1u) - 1u;
    const uint32_t small = (uint32_t)((int32_t)(d - 4u) >> 31);          // d < 4
    const uint32_t dbase = pick(small, d + 1u, 1u + ((2u | (d & 1u)) << db));
    const uint32_t distance = dbase + low_bits((uint32_t)(w >> l2), db);
    const bool bad = (not_lit != 0u && ebf == 6u) || (is_len != 0u && d > 29u);
    r.used = l1 + eb + ((l2 + db) & is_len);
    r.bad = bad;
    r.fail = bad || br.bp + r.used > br.limit_bits;
    const int room = L.out_cap - L.out_pos;
    r.beyond = (int)length > room;
    const uint32_t fits = umin(length, (uint32_t)(room > 0 ? room : 0));
    r.length = fits;
    r.is_end = not_lit != 0u && ebf == 7u;
    r.token = pick(is_len, match_token(fits, distance), val);
    return l1 != 0u && (is_len == 0u || l2 != 0u);
}
template <class Mem>
DBI_HD bool lane_decode(Lane& L, Mem& mem, uint32_t* token) {
    Decoded r;
    const bool fast = kTables && lane_decode_fast(L, mem, r);
#if defined(DBI_CHECK_TABLES)
    if (L.state == kDecode) dbi_table_answer(fast);
    // (CPU harness: whenever the tables answer, they must answer what the canonical method does)
    if (fast && L.state == kDecode) {
        const Decoded s = lane_decode_front(L, mem);
        // (a refused token's other fields are never looked at)
        if (s.bad != r.bad || s.fail != r.fail ||
            (!s.fail && (s.used != r.used || s.length != r.length || s.token != r.token ||
                         s.is_end != r.is_end || s.beyond != r.beyond)))
            dbi_table_mismatch();
    }
#endif
    if (!fast) r = lane_decode_front(L, mem);
    return lane_decode_commit(L, r, token);
}

}  // namespace dbi
