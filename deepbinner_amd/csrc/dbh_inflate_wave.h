// dbh_inflate_wave.h - phase 1 of the GPU inflate with ONE WAVEFRONT PER STREAM (round 5): the
// 64 lanes of a wave decode 64 consecutive pieces ("homes", kSubBits bits each) of the SAME Huffman
// block at the same time, although only the first of them knows where its first token begins.
//
// Why it works: a prefix code re-synchronises.  A decoder started at an arbitrary bit of a block
// reads garbage for a few tokens and then - almost always within a few dozen bits - lands on a
// true token boundary, after which it decodes exactly what a decoder started at the right place
// decodes.  So (after Weissenberger & Schmidt, "Massively parallel Huffman decoding on GPUs", ICPP
// 2018 - there for a single code; a deflate token is a literal/length code, extra bits, a distance
// code, extra bits, and re-synchronises the same way):
//   round 1   every lane decodes from the first bit of its home until it has left the home, and
//             notes where it ended (= where the next home's first token begins, if the lane
//             itself began at a true boundary);
//   round 2.. every lane whose predecessor ended somewhere else than the lane began decodes
//             again, from there.  Lane 0 always began at a true boundary, so after round k lanes
//             0 .. k-1 are right; in practice all are after round 2, because their round-1 garbage
//             had re-synchronised inside its home and their END did not move.  The loop stops as
//             soon as every lane up to the first one that met the end of the block (or bad data)
//             began where its predecessor ended;
//   output    token counts and byte counts are prefix-summed over the lanes, and every lane
//             decodes its home a last time, writing its tokens where they belong.
// Three decodes per token instead of one - on 64 lanes instead of one: the one-lane-per-stream
// kernel (dbh_inflate_core.h) gives a container's 4,000 streams to 63 wavefronts for as long as the
// LONGEST stream lasts (a 400 k-sample read: ~90 ms); here a stream is a wave's work for 3/64 of
// that, and 4,000 waves fill the GPU.
//
// The block headers and the two code builds are the serial code of the core, run by lane 0 (a
// fifth of the kernel's instructions, one lane wide: the next thing to spread over the lanes);
// the bytes of a stored block become literal tokens 64 at a time; the codes are decoded
// canonically as in the core - but the range ends are wave-uniform now (scalar registers), and
// the sorted symbols live in LDS once per wave.
//
// Compiled twice like the core: by hipcc into inflate_tokens_wave_kernel (dbh_inflate.hip) and by
// g++ into the CPU harness (oracle/inflate_host_test.cpp), which runs the rounds lane after lane
// and must produce exactly the tokens of the one-lane decoder.
#pragma once
#include "dbh_inflate_core.h"

namespace dbi {

constexpr int kWaveLanes = 64;
// a lane's home: 17 dwords - an ODD number, so that the lanes' first reads (and, as long as they
// advance alike, all their reads) fall into different LDS banks
#ifndef DBI_SUB_DWORDS
#define DBI_SUB_DWORDS 17
#endif
constexpr int kSubDwords = DBI_SUB_DWORDS;
constexpr uint32_t kSubBits = 32u * kSubDwords;
constexpr int kChunkDwords = kWaveLanes * kSubDwords;
// a chunk begins at any bit of its first dword, and a lane's last token may begin at the last bit
// of the last home: its 64-bit window reaches into the third dword behind the chunk
constexpr int kStageDwords = (kChunkDwords + 1 + 2 + 3) & ~3;

// how a lane's walk through its home ended
enum SubFlag : int { kSubNone = 0, kSubEnd = 1, kSubBad = 2, kSubTrunc = 3, kSubBeyond = 4 };

struct Tok {
    uint32_t used, length, is_len, distance, lit;
    bool bad, is_end;
};

// One token from the 64 bits (lo, hi) it begins with: the arithmetic of lane_decode_fronts
// (dbh_inflate_core.h), without a lane's state.
template <class Mem>
DBI_HD Tok token_decode(uint32_t lo, uint32_t hi, const uint32_t (&lim_lit)[15],
                        const uint32_t (&lim_dist)[15], const Mem& mem) {
    const uint32_t c1 = first16(lo);
    const uint32_t n1 = code_length<15>(c1, lim_lit);
    const uint32_t l1 = umin(n1, 15u);
    const uint32_t pair1 = mem.lit_pair((int)l1);
    const uint32_t i1 = umin(sorted_index(c1, l1, pair1), (uint32_t)(kLitSyms - 1));
    const uint32_t e = mem.lit_sym((int)i1);
    uint64_t w = (((uint64_t)hi << 32) | lo) >> l1;
    const uint32_t is_len = bit_mask(e, 15);                       // kEntryLength
    const uint32_t eb = (e >> 8) & 7u & is_len;
    const uint32_t not_end = ((e >> 14) & 1u) ^ 1u;                // kEntryEnd: length 0, a literal: 1
    const uint32_t length = pick(is_len, 3u + (e & 0xFFu) + low_bits((uint32_t)w, eb), not_end);
    w >>= eb;
    const uint32_t c2 = first16((uint32_t)w);
    const uint32_t n2 = code_length<15>(c2, lim_dist);
    const uint32_t l2 = umin(n2, 15u);
    const uint32_t pair2 = mem.dist_pair((int)l2);
    const uint32_t i2 = umin(sorted_index(c2, l2, pair2), (uint32_t)(kDistSyms - 1));
    const uint32_t d = mem.dist_sym((int)i2);
    w >>= l2;
    const uint32_t half = d >> 1;
    const uint32_t db = (half > 1u ? half : 1u) - 1u;
    const uint32_t small = (uint32_t)((int32_t)(d - 4u) >> 31);          // d < 4
    const uint32_t dbase = pick(small, d + 1u, 1u + ((2u | (d & 1u)) << db));
    Tok t;
    t.distance = dbase + low_bits((uint32_t)w, db);
    t.bad = n1 > 15u || (e & kEntryBad) != 0 || (is_len != 0u && (n2 > 15u || d > 29u));
    t.used = l1 + eb + ((l2 + db) & is_len);
    t.length = length;
    t.is_len = is_len;
    t.lit = e & 0xFFu;
    t.is_end = (e & kEntryEnd) != 0;
    return t;
}

// FIRST-LEVEL DECODE TABLES for the one-wavefront-per-stream form (the second session of round 5).
// The canonical decode above is ~60 vector instructions and two dependent LDS reads per code,
// twice per token; a table indexed by the next kWaveLitBits / kWaveDistBits bits of the stream
// answers in one read - {code length, meaning} - whenever the code is no longer than the index.
// With one LANE per stream the tables did not pay (dbh_inflate_core.h: 64 copies of them fill a
// CU's LDS, and one lane with a longer code sends its whole wave the canonical way); here a wave
// has ONE pair of tables, 4.5 KB, and its lanes fill them together behind every block header:
// lane l decodes indices l, l + 64, .. the canonical way - no serial code at all.  A lane whose
// code is longer than the index (kWaveLitBits = 11: about one token in 300 of a level-1 squiggle
// stream) decodes canonically, the others wait for it: the same tokens either way, and the CPU
// harness holds every table answer against the canonical one.
#ifndef DBI_WAVE_LIT_BITS
#define DBI_WAVE_LIT_BITS 11
#endif
#ifndef DBI_WAVE_DIST_BITS
#define DBI_WAVE_DIST_BITS 8
#endif
constexpr int kWaveLitBits = DBI_WAVE_LIT_BITS, kWaveDistBits = DBI_WAVE_DIST_BITS;
constexpr bool kWaveTables = DBI_WAVE_LIT_BITS > 0;
static_assert(kWaveLitBits <= 15 && kWaveDistBits >= 1 && kWaveDistBits <= 15, "");
constexpr int kWaveLitEntries = kWaveTables ? 1 << kWaveLitBits : 1;
constexpr int kWaveDistEntries = kWaveTables ? 1 << kWaveDistBits : 1;

// the entry of the literal/length table for the index k (the stream's next bits, lowest first):
// lit_tab_entry's format (dbh_inflate_core.h), 0 = the code that begins so is longer than the index
// (or does not exist: the canonical decoder refuses it)
template <class Mem>
DBI_HD uint32_t wave_lit_entry(uint32_t k, const uint32_t (&lim_lit)[15], const Mem& mem) {
    const uint32_t c1 = first16(k);
    const uint32_t n1 = code_length<15>(c1, lim_lit);
    if (n1 > (uint32_t)kWaveLitBits) return 0u;
    const uint32_t i1 = umin(sorted_index(c1, n1, mem.lit_pair((int)n1)), (uint32_t)(kLitSyms - 1));
    const uint32_t e = mem.lit_sym((int)i1);
    uint32_t t;
    if (e & kEntryLength) t = 0x8000u | (((e >> 8) & 7u) << 4) | ((e & 0xFFu) << 7);
    else if (e & kEntryEnd) t = 0x8000u | (7u << 4);
    else if (e & kEntryBad) t = 0x8000u | (6u << 4);
    else t = (e & 0xFFu) << 7;
    return t | n1;
}
template <class Mem>
DBI_HD uint32_t wave_dist_entry(uint32_t k, const uint32_t (&lim_dist)[15], const Mem& mem) {
    const uint32_t c2 = first16(k);
    const uint32_t n2 = code_length<15>(c2, lim_dist);
    if (n2 > (uint32_t)kWaveDistBits) return 0u;
    const uint32_t i2 = umin(sorted_index(c2, n2, mem.dist_pair((int)n2)), (uint32_t)(kDistSyms - 1));
    return dist_tab_entry((int)mem.dist_sym((int)i2), (int)n2);
}

// One token through the tables (the arithmetic of lane_decode_fast, without a lane's state).
// Returns false where a code has no entry: `t` is then not to be used.
template <class Mem>
DBI_HD bool token_decode_tables(uint32_t lo, uint32_t hi, const Mem& mem, Tok& t) {
    const uint32_t e = mem.wave_lit_tab((int)(lo & (uint32_t)(kWaveLitEntries - 1)));
    const uint32_t l1 = e & 15u;
    const uint32_t not_lit = bit_mask(e, 15);
    const uint32_t ebf = (e >> 4) & 7u;                        // 6: a bad symbol, 7: end of block
    const uint32_t is_len = not_lit & (uint32_t)((int32_t)(ebf - 6u) >> 31);       // ebf < 6
    const uint32_t eb = ebf & is_len;
    const uint32_t val = (e >> 7) & 0xFFu;
    uint64_t w = (((uint64_t)hi << 32) | lo) >> l1;
    // a literal: 1; end of block (or a bad symbol, refused below): 0
    t.length = pick(is_len, 3u + val + low_bits((uint32_t)w, eb), (~not_lit) & 1u);
    w >>= eb;
    const uint32_t e2 = mem.wave_dist_tab((int)((uint32_t)w & (uint32_t)(kWaveDistEntries - 1)));
    const uint32_t l2 = e2 & 15u;
    const uint32_t d = e2 >> 4;
    const uint32_t half = d >> 1;
    const uint32_t db = (half > 1u ? half : 1u) - 1u;
    const uint32_t small = (uint32_t)((int32_t)(d - 4u) >> 31);          // d < 4
    const uint32_t dbase = pick(small, d + 1u, 1u + ((2u | (d & 1u)) << db));
    t.distance = dbase + low_bits((uint32_t)(w >> l2), db);
    t.bad = (not_lit != 0u && ebf == 6u) || (is_len != 0u && d > 29u);
    t.used = l1 + eb + ((l2 + db) & is_len);
    t.is_len = is_len;
    t.lit = val;
    t.is_end = not_lit != 0u && ebf == 7u;
    return l1 != 0u && (is_len == 0u || l2 != 0u);
}

// the token that begins with the 64 bits (lo, hi): through the tables where they answer
template <class Mem>
DBI_HD Tok token_of(uint32_t lo, uint32_t hi, const uint32_t (&lim_lit)[15],
                    const uint32_t (&lim_dist)[15], const Mem& mem) {
    Tok t;
    const bool answered = kWaveTables && token_decode_tables(lo, hi, mem, t);
#if defined(DBI_CHECK_TABLES)
    dbi_wave_table_answer(answered);
    if (answered) {      // (CPU harness: a table answer must be the canonical one)
        const Tok c = token_decode(lo, hi, lim_lit, lim_dist, mem);
        // (a refused token's other fields are never looked at; a literal's distance neither)
        if (c.bad != t.bad || (!c.bad && (c.used != t.used || c.length != t.length ||
                                          (c.is_len != 0u) != (t.is_len != 0u) || c.is_end != t.is_end ||
                                          (c.is_len != 0u && c.distance != t.distance) ||
                                          (c.is_len == 0u && !c.is_end && c.lit != t.lit))))
            dbi_table_mismatch();
    }
#endif
    if (!answered) t = token_decode(lo, hi, lim_lit, lim_dist, mem);
    return t;
}

// the 64 bits at bit x of the staged chunk
template <class Mem>
DBI_HD void stage_window(const Mem& mem, uint32_t x, uint32_t& lo, uint32_t& hi) {
    const int at = (int)(x >> 5);
    const uint32_t sh = x & 31u;
    const uint32_t d0 = mem.stage(at), d1 = mem.stage(at + 1), d2 = mem.stage(at + 2);
#if defined(__HIP_DEVICE_COMPILE__)
    lo = __builtin_amdgcn_alignbit(d1, d0, sh);
    hi = __builtin_amdgcn_alignbit(d2, d1, sh);
#else
    lo = (uint32_t)((((uint64_t)d1 << 32) | d0) >> sh);
    hi = (uint32_t)((((uint64_t)d2 << 32) | d1) >> sh);
#endif
}

// What the whole wave knows about the block it is in.  Bit positions with the suffix _rel count
// from the first bit of the staged chunk's first dword.
struct WaveBlock {
    uint32_t lim_lit[15], lim_dist[15];
    uint32_t limit_rel;        // the stream's last bit (deflate data + trailer)
};

struct SubResult {
    uint32_t end;              // where the walk stopped (behind the last token it took)
    int count, bytes;          // tokens that yield bytes (all but the end-of-block code), their bytes
    int flag;                  // SubFlag: why it stopped before leaving the home, if it did
};

// THE TOKENS OF A LANE'S LAST WALK ARE KEPT (second session of round 5).  The output pass used to
// be a walk of its own - every token decoded once more, a quarter of the kernel's decodes.  Now a
// walk of the rounds stores its tokens as it goes, token k of lane l at keep[k * 64 + l] (the
// lanes of a wave store side by side), and once the rounds agree the output pass is a copy to
// where the prefix sums put them.  What a lane keeps is its LAST walk's - it walks again only from
// another start, and then from token 0.  The kept tokens are uncut, so a chunk in which the wanted
// number of bytes is exceeded still takes the walking output pass, and so does one in which a
// lane has more than kSubKeep tokens (a home is 544 bits).  `keep` is the END of the stream's own
// token region (one slot per byte of output: far more than a stream's tokens, unless it is nearly
// all literals or short) - keep_room says whether the chunk's tokens stay clear of it.
#ifndef DBI_SUB_KEEP
#define DBI_SUB_KEEP 96
#endif
constexpr int kSubKeep = DBI_SUB_KEEP;
constexpr int kKeepSlots = kSubKeep * kWaveLanes;
// may the chunk that begins with n_tok tokens stored use the last kKeepSlots of cap_slots?
DBI_HD bool keep_room(int n_tok, int64_t cap_slots) {
    return kSubKeep > 0 && (int64_t)n_tok + 2 * kKeepSlots <= cap_slots;
}
struct KeepNothing {
    DBI_HD void put(int, uint32_t) const {}
};
struct KeepTokens {
    uint32_t* keep;            // null: nothing is kept
    int lane;
    DBI_HD void put(int k, uint32_t token) const {
        if (keep != nullptr && k < kSubKeep) keep[k * kWaveLanes + lane] = token;
    }
};

// A lane's walk through its home in rounds 1, 2, ..: from x until the home [.., stop) is left.
// Nothing of the stream's output is written, and nothing depends on how many bytes are wanted
// (that is the output pass's business); the tokens go to `kept`.
template <class Mem, class Keep>
DBI_HD SubResult sub_decode(const WaveBlock& B, const Mem& mem, uint32_t x, uint32_t stop,
                            const Keep& kept) {
    SubResult r;
    r.count = 0;
    r.bytes = 0;
    r.flag = kSubNone;
    while (x < stop) {
        uint32_t lo, hi;
        stage_window(mem, x, lo, hi);
        const Tok t = token_of(lo, hi, B.lim_lit, B.lim_dist, mem);
        if (t.bad) {
            r.flag = kSubBad;
            break;
        }
        if (x + t.used > B.limit_rel) {
            r.flag = kSubTrunc;
            break;
        }
        x += t.used;
        if (t.is_end) {
            r.flag = kSubEnd;
            break;
        }
        kept.put(r.count, pick(t.is_len, match_token(t.length, t.distance), t.lit));
        r.count += 1;
        r.bytes += (int)t.length;
    }
    r.end = x;
    return r;
}

// The output pass of a lane whose start is known to be right: the walk again, token by token
// what lane_decode_commit (dbh_inflate_core.h) does - the token that goes BEYOND the wanted bytes
// is cut and ends the stream's decoding, a refused one fails it, the end-of-block code ends the
// block.  `tok` is where the lane's first token goes, out_pos the bytes in front of it.
template <class Mem>
DBI_HD SubResult sub_emit(const WaveBlock& B, const Mem& mem, uint32_t x, uint32_t stop,
                          int out_pos, int out_cap, uint32_t* tok) {
    SubResult r;
    r.count = 0;
    r.bytes = 0;
    r.flag = kSubNone;
    while (x < stop) {
        uint32_t lo, hi;
        stage_window(mem, x, lo, hi);
        const Tok t = token_of(lo, hi, B.lim_lit, B.lim_dist, mem);
        if (t.bad) {
            r.flag = kSubBad;
            break;
        }
        if (x + t.used > B.limit_rel) {
            r.flag = kSubTrunc;
            break;
        }
        x += t.used;
        const int room = out_cap - (out_pos + r.bytes);
        const uint32_t fits = umin(t.length, (uint32_t)(room > 0 ? room : 0));
        if (fits > 0u) tok[r.count++] = pick(t.is_len, match_token(fits, t.distance), t.lit);
        r.bytes += (int)fits;
        if ((int)t.length > room) {
            r.flag = kSubBeyond;
            break;
        }
        if (t.is_end) {
            r.flag = kSubEnd;
            break;
        }
    }
    r.end = x;
    return r;
}

// A stored block (rare: bytes deflate could not shrink) is the 64 lanes' work too: the number of
// its bytes that become literal tokens now - all that are left of it, unless the wanted number
// of bytes or the end of the stream comes first - and what lane_stored (dbh_inflate_core.h)
// would have found at the byte behind them.
DBI_HD int stored_run(const Lane& L) {
    const int room = L.out_cap > L.out_pos ? L.out_cap - L.out_pos : 0;
    const int avail = (int)((L.br.limit_bits - L.br.bp) >> 3);      // (bp <= limit_bits: lane_block)
    const int n = L.stored_left < room ? L.stored_left : room;
    return n < avail ? n : avail;
}
// ... the state behind a run of n bytes (their tokens are written by the caller).  Returns true
// if the block has ended and the next block header is to be read.
DBI_HD bool stored_advance(Lane& L, int n) {
    L.stored_left -= n;
    L.out_pos += n;
    L.br.bp += 8u * (uint32_t)n;
    if (L.stored_left == 0) {
        L.state = kNeedBlock;
        if (L.final_block) lane_ended(L);
        return L.state == kNeedBlock;
    }
    if (L.out_pos >= L.out_cap) L.state = kDone;       // more data than wanted
    else lane_fail(L, kTruncated);                     // the stream ends inside the block
    return false;
}

// where lane i of a chunk that begins at bit rel0 (< 32) of the stage begins / stops
DBI_HD uint32_t sub_start(uint32_t rel0, int lane) { return rel0 + (uint32_t)lane * kSubBits; }

// the byte (from the stream's first) at which the 16-byte piece k of a chunk's stage is read -
// never beyond fetch_cap (BitReader::request: what lies there instead is behind the stream's end
// and never looked at)
DBI_HD uint32_t stage_piece_at(uint32_t first_dword, int piece, uint32_t fetch_cap) {
    return umin((first_dword + 4u * (uint32_t)piece) * 4u, fetch_cap);
}

}  // namespace dbi
