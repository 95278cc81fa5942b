// TEST INFRASTRUCTURE (oracle/): the GPU inflate's phase-1 decoder (deepbinner_amd/csrc/
// dbh_inflate_core.h, the very header the kernels are compiled from) run on the CPU, one lane at a
// time, with a sequential stand-in for phase 2 - so that tests/test_inflate.py can hold it to
// zlib (Python's zlib module = the library libhdf5 inflates Signal chunks with) on the build box,
// without a GPU.  Not part of the product; nothing in deepbinner_amd/ calls it.
//   cases file: u32 n; per case: u32 comp_bytes, u32 out_cap, bytes
//   result file: per case: i32 status, i32 ended, i32 adler_ok, u32 n_tokens, u32 out_bytes, bytes
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// the first-level decode tables (an experiment of round 5 the product does not ship: see
// dbh_inflate_core.h) are compiled in here and every answer of theirs is held against the
// canonical method's
#define DBI_CHECK_TABLES 1
#ifndef DBI_LIT_BITS
#define DBI_LIT_BITS 10
#endif
static void dbi_table_mismatch();
static long g_answers[2];
static void dbi_table_answer(bool fast) { ++g_answers[fast ? 1 : 0]; }
static long g_wave_answers[2];
static void dbi_wave_table_answer(bool fast) { ++g_wave_answers[fast ? 1 : 0]; }
#include "../deepbinner_amd/csrc/dbh_inflate_core.h"
#include "../deepbinner_amd/csrc/dbh_inflate_wave.h"
static void dbi_table_mismatch() {
    std::fprintf(stderr, "a first-level table entry disagrees with the canonical decoder\n");
    std::abort();
}

struct HostMem {
    static constexpr int kClTableBits = 7;            // (the code-length code's table: as the wave form has it)
    uint8_t cl_tab_[128];
    uint32_t cl_tab(int i) const { return cl_tab_[check(i, 128)]; }
    void set_cl_tab(int i, uint32_t v) { cl_tab_[check(i, 128)] = (uint8_t)v; }
    uint32_t ring_[dbi::kRingStore], lit_pair_[16], dist_pair_[16];
    uint16_t lit_sym_[dbi::kLitSyms], cnt_[16];
    uint8_t dist_sym_[dbi::kDistSyms], lens_[dbi::kMaxLens];
    uint16_t lit_tab_[1 << dbi::kLitBits], dist_tab_[1 << dbi::kDistBits];
    uint32_t stage_[dbi::kStageDwords];      // (the one-wave-per-stream decoder's chunk)
    uint16_t wave_lit_tab_[dbi::kWaveLitEntries], wave_dist_tab_[dbi::kWaveDistEntries];   // (its tables)
    uint32_t wave_lit_tab(int i) const { return wave_lit_tab_[check(i, dbi::kWaveLitEntries)]; }
    uint32_t wave_dist_tab(int i) const { return wave_dist_tab_[check(i, dbi::kWaveDistEntries)]; }
    uint32_t stage(int i) const { return stage_[check(i, dbi::kStageDwords)]; }
    void set_stage(int i, uint32_t v) { stage_[check(i, dbi::kStageDwords)] = v; }
    uint32_t lit_tab(int i) const { return lit_tab_[check(i, 1 << dbi::kLitBits)]; }
    void set_lit_tab(int i, uint32_t v) { lit_tab_[check(i, 1 << dbi::kLitBits)] = (uint16_t)v; }
    uint32_t dist_tab(int i) const { return dist_tab_[check(i, 1 << dbi::kDistBits)]; }
    void set_dist_tab(int i, uint32_t v) { dist_tab_[check(i, 1 << dbi::kDistBits)] = (uint16_t)v; }
    uint32_t ring(int r) const { return ring_[check(r, dbi::kRingStore)]; }
    void set_ring(int r, uint32_t v) { ring_[check(r, dbi::kRingStore)] = v; }
    int len(int i) const { return lens_[check(i, dbi::kMaxLens)]; }
    void set_len(int i, int v) { lens_[check(i, dbi::kMaxLens)] = (uint8_t)v; }
    int cnt(int l) const { return cnt_[check(l, 16)]; }
    void set_cnt(int l, int v) { cnt_[check(l, 16)] = (uint16_t)v; }
    uint32_t lit_pair(int l) const { return lit_pair_[check(l, 16)]; }
    void set_lit_pair(int l, uint32_t v) { lit_pair_[check(l, 16)] = v; }
    uint32_t dist_pair(int l) const { return dist_pair_[check(l, 16)]; }
    void set_dist_pair(int l, uint32_t v) { dist_pair_[check(l, 16)] = v; }
    uint32_t lit_sym(int i) const { return lit_sym_[check(i, dbi::kLitSyms)]; }
    void set_lit_sym(int i, uint32_t v) { lit_sym_[check(i, dbi::kLitSyms)] = (uint16_t)v; }
    uint32_t dist_sym(int i) const { return dist_sym_[check(i, dbi::kDistSyms)]; }
    void set_dist_sym(int i, uint32_t v) { dist_sym_[check(i, dbi::kDistSyms)] = (uint8_t)v; }
    static int check(int i, int n) {
        if (i < 0 || i >= n) {
            std::fprintf(stderr, "index %d outside [0, %d)\n", i, n);
            std::abort();
        }
        return i;
    }
};

// The one-wave-per-stream decoder (dbh_inflate_wave.h) as inflate_tokens_wave_kernel runs it, the
// 64 lanes one after the other: serial code (headers, code builds, stored blocks) as lane 0 runs
// it, then chunk after chunk of the Huffman block - round 1 from every home's first bit, further
// rounds for the lanes whose predecessor ended elsewhere, the output pass.  Must leave the very
// tokens and the very lane state of the one-lane decoder.
static long g_wave_chunks, g_wave_rounds, g_wave_walks, g_wave_tokens, g_wave_kept_chunks;
static long g_k2_steps, g_k2_rounds, g_k2_turns, g_k2_matches, g_k2_match_bytes, g_k2_tokens;
static long g_k2p_steps, g_k2p_rounds, g_k2p_late, g_k2p_pre, g_k2p_far, g_k2p_short_steps;
static void run_wave(const std::vector<uint8_t>& comp, uint32_t comp_bytes, uint32_t out_cap,
                     HostMem& mem, dbi::Lane& L, std::vector<uint32_t>& tokens) {
    using namespace dbi;
    lane_start(L, mem, comp.data(), (int64_t)comp_bytes, (int64_t)out_cap, (int64_t)comp.size());
    long guard = 0;
    std::vector<uint32_t> lane_tokens[kWaveLanes];
    while (L.state != kDone) {
        if (++guard > 100000000L) std::abort();
        if (L.state == kNeedBlock) {
            lane_block(L, mem);
            if (kWaveTables && L.state == kDecode) {
                // the first-level tables, as the wave's lanes fill them behind a block header
                for (int k = 0; k < kWaveLitEntries; ++k)
                    mem.wave_lit_tab_[k] = (uint16_t)wave_lit_entry((uint32_t)k, L.lim_lit, mem);
                for (int k = 0; k < kWaveDistEntries; ++k)
                    mem.wave_dist_tab_[k] = (uint16_t)wave_dist_entry((uint32_t)k, L.lim_dist, mem);
            }
            continue;
        }
        if (L.state == kStored) {
            const int n = stored_run(L);
            for (int k = 0; k < n; ++k) tokens.push_back(comp[(L.br.bp >> 3) + (size_t)k]);
            if (stored_advance(L, n)) L.br.seek(mem, L.br.bp);
            continue;
        }
        // kDecode: one chunk
        const uint32_t first_dword = L.br.bp >> 5, rel0 = L.br.bp & 31u;
        for (int piece = 0; piece < kStageDwords / 4; ++piece) {
            const U4 v = L.br.load16(stage_piece_at(first_dword, piece, L.br.fetch_cap));
            mem.set_stage(4 * piece + 0, v.x);
            mem.set_stage(4 * piece + 1, v.y);
            mem.set_stage(4 * piece + 2, v.z);
            mem.set_stage(4 * piece + 3, v.w);
        }
        WaveBlock B;
        for (int l = 0; l < 15; ++l) {
            B.lim_lit[l] = L.lim_lit[l];
            B.lim_dist[l] = L.lim_dist[l];
        }
        B.limit_rel = L.br.limit_bits - first_dword * 32u;
        uint32_t x[kWaveLanes];
        SubResult r[kWaveLanes];
        // (the walks keep their tokens where the kernel's do: at the end of the stream's token
        // region, one slot per byte of output - here a buffer of that shape)
        static std::vector<uint32_t> keep_buffer;
        keep_buffer.assign((size_t)kKeepSlots + 1, 0xDEADBEEFu);
        const bool room = keep_room((int)tokens.size(), (int64_t)out_cap);
        for (int i = 0; i < kWaveLanes; ++i) {
            x[i] = sub_start(rel0, i);
            r[i] = sub_decode(B, mem, x[i], sub_start(rel0, i + 1), KeepTokens{room ? keep_buffer.data() : nullptr, i});
        }
        ++g_wave_chunks;
        ++g_wave_rounds;
        g_wave_walks += kWaveLanes;
        int last;                       // the lane whose walk ends the chunk
        for (;;) {
            uint32_t want[kWaveLanes];
            int first_moved = kWaveLanes, first_flag = kWaveLanes;
            for (int i = 0; i < kWaveLanes; ++i) {
                want[i] = i == 0 ? rel0 : r[i - 1].flag != kSubNone ? sub_start(rel0, i) : r[i - 1].end;
                if (want[i] != x[i] && first_moved == kWaveLanes) first_moved = i;
                if (r[i].flag != kSubNone && first_flag == kWaveLanes) first_flag = i;
            }
            if (first_moved > first_flag || first_moved == kWaveLanes) {
                last = first_flag < kWaveLanes ? first_flag : kWaveLanes - 1;
                break;
            }
            ++g_wave_rounds;
            for (int i = 0; i < kWaveLanes; ++i)
                if (want[i] != x[i]) {
                    x[i] = want[i];
                    r[i] = sub_decode(B, mem, x[i], sub_start(rel0, i + 1),
                                      KeepTokens{room ? keep_buffer.data() : nullptr, i});
                    ++g_wave_walks;
                }
        }
        // the lane in which the wanted number of bytes is exceeded, if that comes first
        int before = L.out_pos;
        bool over = false;
        for (int i = 0; i <= last; ++i) {
            if (before + r[i].bytes > L.out_cap) {
                last = i;
                over = true;
                break;
            }
            before += r[i].bytes;
        }
        bool many = false;
        for (int i = 0; i <= last; ++i) many = many || r[i].count > kSubKeep;
        int out_pos = L.out_pos;
        SubResult e{};
        if (room && !over && !many) {
            // the output pass as a copy of what the lanes' last walks kept
            ++g_wave_kept_chunks;
            for (int i = 0; i <= last; ++i) {
                for (int k = 0; k < r[i].count; ++k) tokens.push_back(keep_buffer[(size_t)k * kWaveLanes + i]);
                out_pos += r[i].bytes;
                g_wave_tokens += r[i].count;
            }
            e = r[last];
        } else
        for (int i = 0; i <= last; ++i) {
            lane_tokens[i].assign((size_t)r[i].count + 1, 0u);
            e = sub_emit(B, mem, x[i], sub_start(rel0, i + 1), out_pos, L.out_cap, lane_tokens[i].data());
            if (i < last && (e.flag != kSubNone || e.count != r[i].count || e.bytes != r[i].bytes ||
                             e.end != r[i].end)) {
                std::fprintf(stderr, "wave model: the output pass disagrees with the rounds\n");
                std::abort();
            }
            tokens.insert(tokens.end(), lane_tokens[i].begin(), lane_tokens[i].begin() + e.count);
            out_pos += e.bytes;
            g_wave_tokens += e.count;
        }
        L.out_pos = out_pos;
        L.br.bp = first_dword * 32u + e.end;
        if (e.flag == kSubBad) lane_fail(L, kBadSymbol);
        else if (e.flag == kSubTrunc) lane_fail(L, kTruncated);
        else if (e.flag == kSubBeyond) L.state = kDone;
        else if (e.flag == kSubEnd) {
            L.state = kNeedBlock;
            if (L.final_block) lane_ended(L);
            else L.br.seek(mem, L.br.bp);
        }
    }
}

int main(int argc, char** argv) {
    if (argc != 3) return 2;
    FILE* in = std::fopen(argv[1], "rb");
    FILE* out = std::fopen(argv[2], "wb");
    if (!in || !out) return 2;
    uint32_t n_cases = 0;
    if (std::fread(&n_cases, 4, 1, in) != 1) return 2;
    if (std::getenv("DBI_TABLE_STATS")) std::atexit([] {
        std::fprintf(stderr, "tokens answered by the tables: %ld, sent the canonical way: %ld\n", g_answers[1], g_answers[0]);
    });
    if (std::getenv("DBI_WAVE_STATS")) std::atexit([] {
        std::fprintf(stderr, "one wave per stream: tokens answered by the tables: %ld, sent the canonical way: %ld\n",
                     g_wave_answers[1], g_wave_answers[0]);
        std::fprintf(stderr, "one wave per stream: the output pass a copy of kept tokens in %ld of %ld chunks\n",
                     g_wave_kept_chunks, g_wave_chunks);
        std::fprintf(stderr, "one wave per stream: %ld chunks, %.3f rounds per chunk, %.3f walks per lane and chunk, "
                             "%.1f tokens per lane and chunk\n", g_wave_chunks,
                     (double)g_wave_rounds / (double)std::max(1L, g_wave_chunks),
                     (double)g_wave_walks / (64.0 * (double)std::max(1L, g_wave_chunks)),
                     (double)g_wave_tokens / (64.0 * (double)std::max(1L, g_wave_chunks)));
    });
    if (std::getenv("DBI_K2_STATS")) std::atexit([] {
        std::fprintf(stderr, "resolve schedule: %ld steps, %.1f tokens, %.1f matches (%.1f bytes each), %.2f rounds, "
                             "%.2f four-byte turns per step\n", g_k2_steps, (double)g_k2_tokens / g_k2_steps,
                     (double)g_k2_matches / g_k2_steps, (double)g_k2_match_bytes / std::max(1L, g_k2_matches),
                     (double)g_k2_rounds / g_k2_steps, (double)g_k2_turns / g_k2_steps);
        std::fprintf(stderr, "second form: %ld steps (%ld cut short by their span), %.1f matches per step read at the "
                             "boundary (%.1f of them from the flushed output), %.2f late ones in %.2f rounds\n",
                     g_k2p_steps, g_k2p_short_steps, (double)g_k2p_pre / std::max(1L, g_k2p_steps),
                     (double)g_k2p_far / std::max(1L, g_k2p_steps), (double)g_k2p_late / std::max(1L, g_k2p_steps),
                     (double)g_k2p_rounds / std::max(1L, g_k2p_steps));
    });
    for (uint32_t c = 0; c < n_cases; ++c) {
        uint32_t comp_bytes = 0, out_cap = 0;
        if (std::fread(&comp_bytes, 4, 1, in) != 1 || std::fread(&out_cap, 4, 1, in) != 1) return 2;
        std::vector<uint8_t> comp((size_t)comp_bytes + 64, 0);       // padded like the device buffer
        if (comp_bytes && std::fread(comp.data(), 1, comp_bytes, in) != comp_bytes) return 2;
        HostMem mem;
        std::memset(&mem, 0, sizeof(mem));
        dbi::Lane L;
        dbi::lane_start(L, mem, comp.data(), (int64_t)comp_bytes, (int64_t)out_cap,
                        (int64_t)comp.size());
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
            if (L.state == dbi::kStored) {
                if (dbi::lane_stored(L, mem, &token)) tokens.push_back(token);
                continue;
            }
            // (kDecode: the hot path as the kernel runs it - four tokens, then the checkpoint
            // that keeps the input ring filled)
            for (int k = 0; k < 4; ++k)
                if (dbi::lane_decode(L, mem, &token)) tokens.push_back(token);
            L.br.checkpoint(mem);
        }
        // the one-wave-per-stream decoder must leave the same tokens and the same record
        {
            HostMem wmem;
            std::memset(&wmem, 0, sizeof(wmem));
            dbi::Lane W;
            std::vector<uint32_t> wave_tokens;
            run_wave(comp, comp_bytes, out_cap, wmem, W, wave_tokens);
            if (W.status != L.status || W.ended != L.ended || W.adler != L.adler ||
                (L.status == dbi::kOk && (W.out_pos != L.out_pos || wave_tokens != tokens))) {
                std::fprintf(stderr, "case %u: one wave per stream: status %d ended %d bytes %d tokens %zu, "
                                     "one lane per stream: status %d ended %d bytes %d tokens %zu\n", c,
                             W.status, W.ended, W.out_pos, wave_tokens.size(), L.status, L.ended,
                             L.out_pos, tokens.size());
                return 5;
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
        // ... and as the resolve kernel SCHEDULES it (dbh_inflate.hip: inflate_resolve_kernel):
        // a ring of exactly 32 KiB, 64 tokens per step, all literals of a step first, then the
        // matches in rounds (whoever reads nothing that is still to be written goes, four bytes
        // per lockstep turn, the loads of a turn before its stores), steps with a ring hazard in
        // token order, whole 256-byte pieces flushed after every step.  Must give the same bytes.
        if (status == dbi::kOk) {
            std::vector<uint8_t> ring(dbi::kWindowRing, 0), model;
            auto at = [&](long p) -> uint8_t& { return ring[(size_t)(p & (dbi::kWindowRing - 1))]; };
            long pos = 0, flushed = 0;
            const int n_tok = (int)tokens.size();
            for (int t0 = 0; t0 < n_tok; t0 += dbi::kStepTokens) {
                const int lanes = std::min(dbi::kStepTokens, n_tok - t0);
                long my[64];
                int len[64], dist[64];
                bool is_match[64];
                long end = pos;
                for (int l = 0; l < lanes; ++l) {
                    const uint32_t t = tokens[t0 + l];
                    is_match[l] = (t & dbi::kMatchFlag) != 0;
                    len[l] = is_match[l] ? (int)(t & 0x1FF) : 1;
                    dist[l] = (int)((t >> 9) & 0x7FFF) + 1;
                    my[l] = end;
                    end += len[l];
                }
                ++g_k2_steps;
                g_k2_tokens += lanes;
                for (int l = 0; l < lanes; ++l)
                    if (is_match[l]) {
                        ++g_k2_matches;
                        g_k2_match_bytes += len[l];
                    }
                bool hazard = false;
                for (int l = 0; l < lanes; ++l)
                    hazard = hazard || (is_match[l] && dbi::ring_hazard(dist[l], (int)my[l], (int)end));
                if (hazard) {
                    for (int l = 0; l < lanes; ++l) {
                        if (!is_match[l]) at(my[l]) = (uint8_t)tokens[t0 + l];
                        else for (int k = 0; k < len[l]; ++k) at(my[l] + k) = at(my[l] - dist[l] + k);
                    }
                } else {
                    for (int l = 0; l < lanes; ++l)
                        if (!is_match[l]) at(my[l]) = (uint8_t)tokens[t0 + l];
                    bool waiting[64];
                    for (int l = 0; l < lanes; ++l) waiting[l] = is_match[l];
                    for (;;) {
                        int first_lane = -1;
                        for (int l = 0; l < lanes && first_lane < 0; ++l)
                            if (waiting[l]) first_lane = l;
                        if (first_lane < 0) break;
                        const long first = my[first_lane];
                        bool go[64];
                        int longest = 0;
                        for (int l = 0; l < lanes; ++l) {
                            const long src = my[l] - dist[l];
                            go[l] = waiting[l] && src + std::min(len[l], dist[l]) <= first;
                            if (go[l]) longest = std::max(longest, len[l]);
                        }
                        ++g_k2_rounds;
                        g_k2_turns += (longest + 3) / 4;
                        for (int k = 0; k < longest; k += 4) {
                            uint8_t b[64][4];
                            for (int l = 0; l < lanes; ++l)
                                if (go[l] && k < len[l])
                                    for (int j = 0; j < 4; ++j)
                                        b[l][j] = at(my[l] - dist[l] + (k + j) % dist[l]);
                            for (int l = 0; l < lanes; ++l)
                                if (go[l] && k < len[l])
                                    for (int j = 0; j < 4 && k + j < len[l]; ++j)
                                        at(my[l] + k + j) = b[l][j];
                        }
                        for (int l = 0; l < lanes; ++l) waiting[l] = waiting[l] && !go[l];
                    }
                }
                pos = end;
                while (pos - flushed >= 256) {
                    for (int k = 0; k < 256; ++k) model.push_back(at(flushed + k));
                    flushed += 256;
                }
            }
            for (long k = flushed; k < pos; ++k) model.push_back(at(k));
            if (model != bytes) {
                std::fprintf(stderr, "case %u: the resolve kernel's schedule gives other bytes than "
                                     "the tokens in order\n", c);
                return 4;
            }
        }
        // ... and as the SECOND form schedules it (inflate_resolve_pre_kernel, what runs; dbh_inflate_
        // core.h, "Phase 2's second form"): a ring of kSmallRing bytes and the flushed output behind
        // it; steps of up to 64 tokens that span at most kStepSpan bytes; a short match whose source
        // lies wholly before its step reads eight raw bytes at the boundary in front of the step
        // (ring or flushed output), and is stored with the step's literals; the late matches go in
        // rounds by the exact rule, eight bytes in the first turn (self-overlapping ones expanded by
        // k2_pattern8), four per further turn.  The model knows which POSITION every ring slot
        // holds and how far the output is flushed: a read of anything else is poisoned, so a read
        // the schedule may not rely on shows up as wrong bytes.
        if (status == dbi::kOk) {
            const int R = dbi::kSmallRing;
            std::vector<uint8_t> ring((size_t)R, 0), model;       // model = the flushed output
            std::vector<long> holds((size_t)R, -1);
            auto put = [&](long p, uint8_t v) {
                ring[(size_t)(p & (R - 1))] = v;
                holds[(size_t)(p & (R - 1))] = p;
            };
            auto ring_byte = [&](long p, int j) -> uint8_t {      // position p as the ring has it
                return holds[(size_t)(p & (R - 1))] == p ? ring[(size_t)(p & (R - 1))] : (uint8_t)(0xA5 ^ (j * 37));
            };
            auto out_byte = [&](long p, int j) -> uint8_t {       // ... as the flushed output has it
                return p >= 0 && p < (long)model.size() ? model[(size_t)p] : (uint8_t)(0x5A ^ (j * 41));
            };
            auto read8 = [&](long p, bool from_ring) {
                uint64_t v = 0;
                for (int j = 0; j < 8; ++j)
                    v |= (uint64_t)(from_ring ? ring_byte(p + j, j) : out_byte(p + j, j)) << (8 * j);
                return v;
            };
            auto store_short = [&](long p, int n, uint64_t v) {
                for (int j = 0; j < n; ++j) put(p + j, (uint8_t)(v >> (8 * j)));
            };
            long pos = 0, flushed = 0;
            const int n_tok = (int)tokens.size();
            bool pre[64] = {};
            uint64_t pv[64] = {};
            // the tokens a step takes: up to 64, spanning at most kStepSpan bytes
            auto step_count = [&](int first) {
                int n = 0;
                long span = 0;
                while (n < dbi::kStepTokens && first + n < n_tok) {
                    const uint32_t t = tokens[(size_t)(first + n)];
                    const int ln = (t & dbi::kMatchFlag) ? (int)(t & 0x1FF) : 1;
                    if (n > 0 && span + ln > dbi::kStepSpan) break;
                    span += ln;
                    ++n;
                }
                return n;
            };
            for (int t0 = 0; t0 < n_tok;) {
                const int lanes = step_count(t0);
                long my[64];
                int len[64], dist[64];
                bool is_match[64];
                long end = pos;
                for (int l = 0; l < lanes; ++l) {
                    const uint32_t t = tokens[t0 + l];
                    is_match[l] = (t & dbi::kMatchFlag) != 0;
                    len[l] = is_match[l] ? (int)(t & 0x1FF) : 1;
                    dist[l] = (int)((t >> 9) & 0x7FFF) + 1;
                    my[l] = end;
                    end += len[l];
                }
                ++g_k2p_steps;
                if (lanes < dbi::kStepTokens && t0 + lanes < n_tok) ++g_k2p_short_steps;
                auto pre_value = [&](int l) { return dist[l] < len[l] ? dbi::k2_pattern8(pv[l], dist[l]) : pv[l]; };
                for (int l = 0; l < lanes; ++l) {
                    if (!is_match[l]) put(my[l], (uint8_t)tokens[t0 + l]);
                    else if (pre[l]) store_short(my[l], len[l], pre_value(l));
                }
                bool waiting[64];
                for (int l = 0; l < lanes; ++l) {
                    waiting[l] = is_match[l] && !pre[l];
                    if (waiting[l]) ++g_k2p_late;
                }
                for (;;) {
                    bool any = false;
                    for (int l = 0; l < lanes; ++l) any = any || waiting[l];
                    if (!any) break;
                    ++g_k2p_rounds;
                    bool go[64];
                    int longest = 0;
                    for (int l = 0; l < lanes; ++l) {
                        const long src = my[l] - dist[l], reach = src + std::min(len[l], dist[l]);
                        bool blocked = false;
                        for (int w = 0; w < lanes; ++w)
                            blocked = blocked || (waiting[w] && dbi::k2_blocks((int)my[w], (int)my[w] + len[w],
                                                                                (int)src, (int)reach));
                        go[l] = waiting[l] && !blocked;
                        if (go[l]) longest = std::max(longest, len[l]);
                    }
                    if (longest == 0) {
                        std::fprintf(stderr, "case %u: a round of the second form lets nobody go\n", c);
                        return 4;
                    }
                    // first turn: every lane's read, then every lane's stores
                    uint64_t v[64];
                    for (int l = 0; l < lanes; ++l)
                        if (go[l]) {
                            const long src = my[l] - dist[l];
                            v[l] = read8(src, dbi::k2_in_ring((int)src, (int)end));
                            if (dist[l] < 8 && dist[l] < len[l]) v[l] = dbi::k2_pattern8(v[l], dist[l]);
                        }
                    for (int l = 0; l < lanes; ++l)
                        if (go[l]) store_short(my[l], std::min(len[l], 8), v[l]);
                    for (int k = 8; k < longest; k += 4) {
                        uint8_t b[64][4];
                        for (int l = 0; l < lanes; ++l)
                            if (go[l] && k < len[l]) {
                                const long src = my[l] - dist[l];
                                const bool near = dbi::k2_in_ring((int)src, (int)end);
                                for (int j = 0; j < 4; ++j) {
                                    const long q = src + (k + j) % dist[l];
                                    b[l][j] = near ? ring_byte(q, j) : out_byte(q, j);
                                }
                            }
                        for (int l = 0; l < lanes; ++l)
                            if (go[l] && k < len[l])
                                for (int j = 0; j < 4 && k + j < len[l]; ++j) put(my[l] + k + j, b[l][j]);
                    }
                    for (int l = 0; l < lanes; ++l) waiting[l] = waiting[l] && !go[l];
                }
                pos = end;
                t0 += lanes;
                // the boundary: the next step's pre matches read their source, then the ring's
                // whole 256-byte pieces leave
                {
                    const int lanes_n = step_count(t0);
                    long my_n = pos;
                    for (int l = 0; l < 64; ++l) pre[l] = false;
                    for (int l = 0; l < lanes_n; ++l) {
                        const uint32_t t = tokens[t0 + l];
                        const bool m = (t & dbi::kMatchFlag) != 0;
                        const int ln = m ? (int)(t & 0x1FF) : 1, d = (int)((t >> 9) & 0x7FFF) + 1;
                        const long src = my_n - d;
                        pre[l] = dbi::k2_pre(m, ln, (int)(src + std::min(ln, d)), (int)pos);
                        if (pre[l]) {
                            pv[l] = read8(src, dbi::k2_in_ring((int)src, (int)pos));
                            ++g_k2p_pre;
                            if (!dbi::k2_in_ring((int)src, (int)pos)) ++g_k2p_far;
                        }
                        my_n += ln;
                    }
                }
                while (pos - flushed >= 256) {
                    for (int k = 0; k < 256; ++k) model.push_back(ring_byte(flushed + k, k));
                    flushed += 256;
                }
            }
            for (long k = flushed; k < pos; ++k) model.push_back(ring_byte(k, 0));
            if (model != bytes) {
                std::fprintf(stderr, "case %u: the second resolve kernel's schedule gives other bytes "
                                     "than the tokens in order\n", c);
                return 4;
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
