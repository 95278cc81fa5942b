// dbh_inflate_core.h - RFC 1950 / 1951 decoding (zlib streams, the HDF5 "deflate" filter that
// fast5 Signal chunks are stored with), phase 1: Huffman decoding of ONE stream by ONE lane into a
// stream of tokens (literal byte | match {length, distance}).  Phase 2 (dbh_inflate.hip) resolves
// the tokens of a stream into bytes with a whole wavefront and the 32 KiB window in LDS.
//
// What the reference does here: h5py -> libhdf5 -> zlib's inflate() on the host, one chunk after
// the other (deepbinner/load_fast5s.py:33-43 reads `Signal[:]`).  Inflating is ~85 % of what loading
// a read costs a CPU core (~100 us per 55 KB read), and a host hands out few cores
// (profiles/r03_cpu_capacity.txt); thousands of independent streams per container are what a GPU
// is good at.
//
// This header is compiled twice: by hipcc into the kernels, and by g++ into the CPU test harness
// (tests/inflate_host_test.cpp via oracle/Makefile), which runs the SAME decoder lane by lane
// against zlib on the build box.  All table memory is reached through a `Mem` accessor: LDS,
// interleaved by lane, on the device; plain arrays on the host.
//
// Decode tables (per lane): zlib's scheme - a root table indexed by the next ROOT bits whose
// entries either give {symbol, code length} or link to a sub-table for longer codes.
//   entry (16 bit): bit 15 = 0: bits 0-3 code length in this table (0 = no such code),
//                                bits 4-12 symbol
//                   bit 15 = 1: link: bits 0-3 index bits of the sub-table, bits 4-14 its offset
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

constexpr int kLitRoot = 10, kDistRoot = 8;
constexpr int kLitEntries = 1408;        // 1,024 root + sub-tables (more needed: stream refused)
constexpr int kDistEntries = 448;        // 256 root + sub-tables
constexpr int kMaxLens = 352;            // 19 + 13 spare, then up to 286 + 30 lengths as decoded
constexpr int kMaxSyms = 288;

// token: literal = the byte; match = bit 31 | (distance - 1) << 9 | length
constexpr uint32_t kMatchFlag = 0x80000000u;
DBI_HD uint32_t match_token(int length, int distance) {
    return kMatchFlag | ((uint32_t)(distance - 1) << 9) | (uint32_t)length;
}

enum Status : int {
    kOk = 0,
    kBadHeader = 1,        // not a zlib stream (CM, CINFO, FCHECK, FDICT)
    kBadBlock = 2,         // block type 3, stored LEN/NLEN mismatch
    kBadCodes = 3,         // over-subscribed / incomplete code lengths, bad repeat
    kBadSymbol = 4,        // a code that is not in the table, length/distance symbol out of range
    kTruncated = 5,        // ran out of input
    kTableSpace = 6,       // needs more sub-table space than this decoder carries (valid stream)
    kTokenSpace = 7,       // more tokens than the caller's buffer holds
    kBadDistance = 8,      // (phase 2) distance reaches before the start of the output
    kBadChecksum = 9,      // (phase 2) Adler-32 mismatch
    kTooLong = 10,         // (phase 2, whole-stream mode) more output than announced
};

// RFC 1951 section 3.2.5, as arithmetic (device code cannot index host-side constant arrays):
// length symbol 257 + c: c < 8: 3 + c; c < 28: e = (c - 4) / 4 extra bits, base 3 + ((4 + c % 4) << e);
// c = 28: 258.  Distance symbol d: d < 4: d + 1; else e = d / 2 - 1, base 1 + ((2 + d % 2) << e).
DBI_HD int len_extra(int c) { return (c < 8 || c == 28) ? 0 : (c - 4) >> 2; }
DBI_HD int len_base(int c) { return c < 8 ? 3 + c : c == 28 ? 258 : 3 + ((4 + (c & 3)) << len_extra(c)); }
DBI_HD int dist_extra(int d) { return d < 4 ? 0 : (d >> 1) - 1; }
DBI_HD int dist_base(int d) { return d < 4 ? d + 1 : 1 + ((2 + (d & 1)) << dist_extra(d)); }
// the order in which the code lengths of the code-length code are stored (section 3.2.7)
DBI_HD int cl_order(int i) {
    // 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15: five bits each in two words
    const uint64_t lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 |
                        9ull << 30 | 6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
    const uint64_t hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 |
                        15ull << 30;
    return (int)((i < 12 ? lo >> (5 * i) : hi >> (5 * (i - 12))) & 31);
}

// The input side of a lane: a 64-bit bit buffer refilled 32 bits at a time, the next dword
// always already requested (the load's latency hides behind the tokens the buffer still holds).
// The buffer behind a stream is readable (padding, or the next stream): the reader fetches up to
// 48 bytes beyond the stream's end without looking; a stream that CONSUMES bits from there is
// truncated and is told so (overrun).
struct BitReader {
    const uint8_t* in;
    uint32_t limit_bits;   // bits of the stream (deflate data + trailer): streams are < 512 MB
    uint32_t bp;           // bits consumed so far = position of the next bit
    uint32_t base;         // position of bit 0 of `lo` (a multiple of 32, <= bp)
    uint32_t lo, hi;       // the 64 bits from `base` on: everything is 32-bit arithmetic, the
                           // next 32 bits are ONE v_alignbit_b32 away
    // Sixteen bytes in hand behind those (q0 = next) and the sixteen after them already
    // requested: a lane needs a new 64-byte line of its stream every ~40 tokens, the lanes of a
    // wave step together, and a wave waits for its slowest lane.
    uint32_t q0, q1, q2, q3;
    U4 ahead;              // (kept as ONE 128-bit value until it is needed: taken apart earlier,
                           // the compiler waits for the load where it is issued)
    int left;              // dwords of q still unused
    uint32_t fetch;        // byte offset of the sixteen bytes to request next

    // (no bounds checks and no masking: the buffer is readable for 64 bytes beyond the stream -
    // the next stream, or the padding - and bits from beyond the stream are never CONSUMED
    // unnoticed: overrun() after every token)
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
    DBI_HD void start(const uint8_t* data, int64_t n_bytes) {
        in = data;
        limit_bits = (uint32_t)n_bytes * 8u;
        const U4 first = load16(0);
        lo = first.x;
        hi = first.y;
        q0 = first.z;
        q1 = first.w;
        ahead = load16(16);
        q2 = q3 = 0;
        left = 2;
        fetch = 32;
        bp = base = 0;
    }
    // `lo` leaves the window, the next dword enters it
    DBI_HD void shift() {
        lo = hi;
        hi = q0;
        q0 = q1;
        q1 = q2;
        q2 = q3;
        base += 32;
        if (--left == 0) {             // the sixteen bytes asked for a while ago; ask for more
            q0 = ahead.x;
            q1 = ahead.y;
            q2 = ahead.z;
            q3 = ahead.w;
            left = 4;
            ahead = load16(fetch);
            fetch += 16;
        }
    }
    // Normal form: the next bit lies in `lo`.  At most 32 bits may be consumed between two calls
    // (then one shift restores it); afterwards at least 33 bits are in the window.
    DBI_HD void refill() {
        if (bp - base >= 32u) shift();
    }
    // the next 32 bits, in normal form
    DBI_HD uint32_t window() const {
        const uint32_t sh = bp & 31u;
#if defined(__HIP_DEVICE_COMPILE__)
        return __builtin_amdgcn_alignbit(hi, lo, sh);
#else
        return (uint32_t)((((uint64_t)hi << 32) | lo) >> sh);
#endif
    }
    // the next n <= 16 bits, anywhere in the window (the header code's sequences of small fields)
    DBI_HD uint32_t peek(int n) const {
        const uint32_t sh = bp - base;       // 0 .. 63
        const uint32_t w = sh < 32u ? (uint32_t)((((uint64_t)hi << 32) | lo) >> sh) : hi >> (sh - 32u);
        return w & ((1u << n) - 1u);
    }
    DBI_HD void drop(int n) { bp += (uint32_t)n; }
    DBI_HD uint32_t take(int n) {
        const uint32_t v = peek(n);
        drop(n);
        return v;
    }
    DBI_HD uint32_t consumed_bits() const { return bp; }
    DBI_HD bool overrun() const { return bp > limit_bits; }
    DBI_HD int64_t limit_bytes() const { return (int64_t)(limit_bits >> 3); }
};

// Builds the decode table of one code (zlib's inflate_table, restated): `lens[first .. first+n)`
// are the code lengths; `Tab` reads and writes table entries; `work` holds n symbols.
//   -> kOk / kBadCodes / kTableSpace.  An incomplete code is tolerated only where zlib tolerates
// it: a literal/length or distance code made of a single code of length 1, never the code-length
// code; no codes at all is accepted (a block of literals only has no distance codes) and leaves a
// table without codes.
template <class Mem, class Tab>
DBI_HD int build_table(Mem& mem, Tab tab, int first, int n, int root, int capacity,
                       bool may_be_incomplete) {
    int count[16];
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int s = 0; s < n; ++s) count[mem.len(first + s)]++;
    int max = 15;
    while (max >= 1 && count[max] == 0) --max;
    for (int e = 0; e < (1 << root); ++e) tab.set(e, 0);      // "no such code" everywhere first
    if (max == 0) return kOk;                                   // no codes at all
    int min = 1;
    while (min < max && count[min] == 0) ++min;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return kBadCodes;                         // over-subscribed
    }
    if (left > 0 && (!may_be_incomplete || max != 1)) return kBadCodes;      // incomplete
    // symbols by (length, symbol)
    int offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
    for (int s = 0; s < n; ++s) {
        const int l = mem.len(first + s);
        if (l != 0) mem.set_work(offs[l]++, s);
    }
    unsigned huff = 0, low = ~0u;
    const unsigned mask = (1u << root) - 1u;
    int sym = 0, len = min, next = 0, curr = root, drop = 0, used = 1 << root;
    for (;;) {
        const int s = mem.work(sym);
        const uint16_t here = (uint16_t)((s << 4) | (len - drop));
        const unsigned incr = 1u << (len - drop);
        unsigned fill = 1u << curr;
        const unsigned span = fill;
        do {
            fill -= incr;
            tab.set(next + (int)((huff >> drop) + fill), here);
        } while (fill != 0);
        // backwards increment of the len-bit code
        unsigned inc = 1u << (len - 1);
        while (huff & inc) inc >>= 1;
        if (inc != 0) {
            huff &= inc - 1;
            huff += inc;
        } else {
            huff = 0;
        }
        ++sym;
        if (--count[len] == 0) {
            if (len == max) break;
            len = mem.len(first + mem.work(sym));
        }
        if (len > root && (huff & mask) != low) {              // a new sub-table
            if (drop == 0) drop = root;
            next += (int)span;
            curr = len - drop;
            int room = 1 << curr;
            while (curr + drop < max) {
                room -= count[curr + drop];
                if (room <= 0) break;
                ++curr;
                room <<= 1;
            }
            used += 1 << curr;
            if (used > capacity) return kTableSpace;
            low = huff & mask;
            tab.set((int)low, (uint16_t)(0x8000u | ((unsigned)next << 4) | (unsigned)curr));
        }
    }
    // an incomplete code (one code of length 1) leaves one entry without a code: already 0 in
    // the root; in zlib it is an "invalid code" marker too
    return kOk;
}

// One lane's decoder state between iterations of the lockstep loop.
struct Lane {
    BitReader br;
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

template <class Mem>
struct LitTab {
    Mem* m;
    DBI_HD void set(int e, uint16_t v) const { m->set_lit(e, v); }
};
template <class Mem>
struct DistTab {
    Mem* m;
    DBI_HD void set(int e, uint16_t v) const { m->set_dist(e, v); }
};

// Decodes one symbol of the code whose table `get(e)` reads: -> symbol, or -1 (no such code).
// The reader is in normal form (refill() since the last 32 bits were consumed).
template <class Get>
DBI_HD int decode_symbol(BitReader& br, int root, const Get& get) {
    const uint32_t w = br.window();
    uint16_t e = get((int)(w & ((1u << root) - 1u)));
    int base_bits = 0;
    if (e & 0x8000u) {
        const int sub_bits = e & 15, off = (e >> 4) & 0x7FF;
        e = get(off + (int)((w >> root) & ((1u << sub_bits) - 1u)));
        base_bits = root;
        if (e & 0x8000u) return -1;
    }
    const int len = e & 15;
    if (len == 0) return -1;
    br.drop(base_bits + len);
    return (e >> 4) & 0x1FF;
}

// Zlib header -> lane ready for its first block.
DBI_HD void lane_start(Lane& L, const uint8_t* data, int64_t n_bytes, int64_t out_cap) {
    L.br.start(data, n_bytes);
    L.adler = 0;
    L.out_pos = 0;
    L.out_cap = (int)out_cap;
    L.state = kNeedBlock;
    L.status = kOk;
    L.final_block = 0;
    L.stored_left = 0;
    L.ended = 0;
    if (n_bytes < 6 || n_bytes >= (1 << 29) || out_cap >= (1ll << 31)) {
        L.status = kTruncated;
        L.state = kDone;
        return;
    }
    const uint32_t cmf = L.br.take(8), flg = L.br.take(8);
    if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) {
        L.status = kBadHeader;
        L.state = kDone;
    }
    if (out_cap < 0) L.state = kDone;
}

DBI_HD void lane_fail(Lane& L, int status) {
    L.status = status;
    L.state = kDone;
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

// Block header (and, for a dynamic block, its code lengths and both tables).
template <class Mem>
DBI_HD void lane_block(Lane& L, Mem& mem) {
    BitReader& br = L.br;
    br.refill();
    L.final_block = (int)br.take(1);
    const int type = (int)br.take(2);
    if (type == 3) return lane_fail(L, kBadBlock);
    if (type == 0) {
        br.drop((int)((8u - (br.bp & 7u)) & 7u));       // to the byte boundary
        br.refill();
        const uint32_t len = br.take(16);
        br.refill();
        const uint32_t nlen = br.take(16);
        if ((len ^ 0xFFFFu) != nlen) return lane_fail(L, kBadBlock);
        if (br.overrun()) return lane_fail(L, kTruncated);
        L.stored_left = (int)len;
        L.state = len ? kStored : kNeedBlock;
        if (!len && L.final_block) lane_ended(L);
        return;
    }
    int hlit = 288, hdist = 30;
    if (type == 1) {
        for (int s = 0; s < 288; ++s) mem.set_len(s, s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8);
        // (32 five-bit distance codes make the fixed code complete; 30 and 31 never occur in
        // valid data and are refused when decoded)
        hdist = 32;
        for (int s = 0; s < 32; ++s) mem.set_len(288 + s, 5);
    } else {
        hlit = (int)br.take(5) + 257;
        hdist = (int)br.take(5) + 1;
        const int hclen = (int)br.take(4) + 4;
        if (hlit > 286 || hdist > 30) return lane_fail(L, kBadCodes);
        for (int i = 0; i < 19; ++i) mem.set_len(i, 0);
        for (int i = 0; i < hclen; ++i) {
            br.refill();
            mem.set_len(cl_order(i), (int)br.take(3));
        }
        // the code-length code: its table in the distance table's place (7-bit root, no links)
        int st = build_table(mem, DistTab<Mem>{&mem}, 0, 19, 7, 128, false);
        if (st != kOk) return lane_fail(L, st);
        int have = 0, prev = 0;
        const int want = hlit + hdist;
        while (have < want) {
            br.refill();
            const int s = decode_symbol(br, 7, [&](int e) { return mem.dist(e); });
            if (s < 0) return lane_fail(L, kBadSymbol);
            if (s < 16) {
                // (the code-length table is read from the distance area; the lengths themselves
                // go to a second run of the length scratch, behind the 19 already used)
                mem.set_len(32 + have++, s);
                prev = s;
                continue;
            }
            int rep, val = 0;
            if (s == 16) {
                if (have == 0) return lane_fail(L, kBadCodes);
                val = prev;
                rep = 3 + (int)br.take(2);
            } else if (s == 17) {
                rep = 3 + (int)br.take(3);
            } else {
                rep = 11 + (int)br.take(7);
            }
            if (have + rep > want) return lane_fail(L, kBadCodes);
            for (int k = 0; k < rep; ++k) mem.set_len(32 + have++, val);
            if (s != 16) prev = 0;
        }
        if (br.overrun()) return lane_fail(L, kTruncated);
        if (mem.len(32 + 256) == 0) return lane_fail(L, kBadCodes);      // no end-of-block code
        // move to the front: lens[0 .. hlit) literal/length, lens[288 .. 288 + hdist) distance
        for (int s = 0; s < hlit; ++s) mem.set_len(s, mem.len(32 + s));
        // (both moves go DOWN in the scratch - 288 + s < 32 + hlit + s as hlit >= 257 - so copying
        // in ascending order never overwrites what is still to be read)
        for (int s = 0; s < hdist; ++s) mem.set_len(288 + s, mem.len(32 + hlit + s));
    }
    int st = build_table(mem, LitTab<Mem>{&mem}, 0, hlit, kLitRoot, kLitEntries, true);
    if (st == kOk)
        st = build_table(mem, DistTab<Mem>{&mem}, 288, hdist, kDistRoot, kDistEntries, true);
    if (st != kOk) return lane_fail(L, st);
    L.state = kDecode;
}

// One token (or one state transition) of a lane in the lockstep loop.  Returns the token in
// `*token` with true, or false when this step produced none.
template <class Mem>
DBI_HD bool lane_step(Lane& L, Mem& mem, uint32_t* token) {
    BitReader& br = L.br;
    if (L.state == kStored) {
        if (L.out_pos >= L.out_cap) {          // more data than wanted
            L.state = kDone;
            return false;
        }
        br.refill();
        const uint32_t byte = br.take(8);
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
    // kDecode
    br.refill();
    const int s = decode_symbol(br, kLitRoot, [&](int e) { return mem.lit(e); });
    if (s < 0 || s > 285) {
        lane_fail(L, kBadSymbol);
        return false;
    }
    if (s < 256) {
        if (br.overrun()) {
            lane_fail(L, kTruncated);
            return false;
        }
        if (L.out_pos >= L.out_cap) {          // more data than wanted
            L.state = kDone;
            return false;
        }
        *token = (uint32_t)s;
        L.out_pos += 1;
        return true;
    }
    if (s == 256) {
        if (br.overrun()) {
            lane_fail(L, kTruncated);
            return false;
        }
        L.state = kNeedBlock;
        if (L.final_block) lane_ended(L);
        return false;
    }
    int length = len_base(s - 257) + (int)br.take(len_extra(s - 257));
    br.refill();
    const int d = decode_symbol(br, kDistRoot, [&](int e) { return mem.dist(e); });
    if (d < 0 || d > 29) {
        lane_fail(L, kBadSymbol);
        return false;
    }
    const int distance = dist_base(d) + (int)br.take(dist_extra(d));
    if (br.overrun()) {
        lane_fail(L, kTruncated);
        return false;
    }
    if (L.out_pos + length > L.out_cap) {      // more data than wanted: keep what is
        length = L.out_cap - L.out_pos;
        L.state = kDone;
        if (length == 0) return false;
    }
    *token = match_token(length, distance);
    L.out_pos += length;
    return true;
}

// The hot path: ONE token of a lane that is inside a Huffman block (state kDecode), written to be
// cheap when 32 lanes run it together - one pass over straight-line code with two conditional
// regions (a linked sub-table; the distance half of a match) instead of a state machine: the
// general lane_step costs ~300 instructions per token and wave, this ~120.
// Returns true with *token set when a token was produced; the lane's state, status, position
// and reader are updated as lane_step would.
template <class Mem>
DBI_HD bool lane_decode(Lane& L, Mem& mem, uint32_t* token) {
    BitReader& br = L.br;
    br.refill();
    const uint32_t w = br.window();
    uint32_t e = mem.lit((int)(w & ((1u << kLitRoot) - 1u)));
    uint32_t used = 0;
    if (e & 0x8000u) {
        e = mem.lit((int)(((e >> 4) & 0x7FFu) + ((w >> kLitRoot) & ((1u << (e & 15u)) - 1u))));
        used = kLitRoot;
    }
    const uint32_t len = e & 15u, sym = (e >> 4) & 0x1FFu;
    bool bad = len == 0 || (e & 0x8000u) || sym > 285u;
    used += len;
    uint32_t tk = sym;
    int produced = 1;
    if (sym > 256u && !bad) {
        const int c = (int)sym - 257;
        const int eb = len_extra(c);
        int length = len_base(c) + (int)((w >> used) & ((1u << eb) - 1u));
        br.bp += used + (uint32_t)eb;
        br.refill();
        const uint32_t w1 = br.window();
        uint32_t e2 = mem.dist((int)(w1 & ((1u << kDistRoot) - 1u)));
        uint32_t used2 = 0;
        if (e2 & 0x8000u) {
            e2 = mem.dist((int)(((e2 >> 4) & 0x7FFu) +
                                ((w1 >> kDistRoot) & ((1u << (e2 & 15u)) - 1u))));
            used2 = kDistRoot;
        }
        const uint32_t dlen = e2 & 15u, d = (e2 >> 4) & 0x1FFu;
        bad = dlen == 0 || (e2 & 0x8000u) || d > 29u;
        used2 += dlen;
        const int db = dist_extra((int)d);
        const int distance = dist_base((int)d) + (int)((w1 >> used2) & ((1u << db) - 1u));
        br.bp += used2 + (uint32_t)db;
        if (L.out_pos + length > L.out_cap) {          // more data than wanted: keep what is
            length = L.out_cap - L.out_pos;
            L.state = kDone;
        }
        tk = match_token(length, distance);
        produced = length;
    } else {
        br.bp += used;
        if (sym == 256u && !bad) {
            produced = 0;
            L.state = kNeedBlock;
        } else if (L.out_pos >= L.out_cap) {           // a literal beyond what is wanted
            produced = 0;
            L.state = kDone;
        }
    }
    if (bad || br.overrun()) {
        lane_fail(L, bad ? kBadSymbol : kTruncated);
        return false;
    }
    if (sym == 256u && L.final_block) lane_ended(L);
    L.out_pos += produced;
    *token = tk;
    return produced > 0;
}

}  // namespace dbi
