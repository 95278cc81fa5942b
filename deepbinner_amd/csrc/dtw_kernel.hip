// libdeepbinner_dtw.so - semi-global dynamic time warping for MI355X (gfx950 only).
// C ABI: include/deepbinner_dtw.h (replaces the reference's deepbinner/dtw/dtw.cpp:58-151).
//
// One wavefront per (reference signal, query signal) pair.  The query runs across the lanes, C
// consecutive columns per lane; lane l works on reference row s - l at step s, so that the value it
// needs from its left neighbour (row i of column chunk l - 1) was produced one step earlier and
// moves over with one DPP wave shift - no LDS, no barrier.  The C costs of a lane's chunk stay in
// registers from row to row.  Queries longer than 64 * 16 columns are done in panels, the last
// column of a panel handed to the next through a per-pair edge buffer.
//
// What goes to HBM is the direction of every cell, 2 bits, packed per lane and step into one
// word ([panel][step][lane], 4 bytes, one coalesced store per step): 0.25 - 1 byte per cell,
// written once.  The walk back
// runs in the same kernel: the wave fetches the words of 64 rows of the current column chunk at
// once (one per lane) and steps through them with v_readlane until the path leaves that tile.
//
// Arithmetic as in dtw.cpp, in fp64 and without contraction (-ffp-contract=off): cost = best of
// (diagonal, left, up) + (ref[i] - query[j])^2, the diagonal winning ties, an exact left/up tie
// going LEFT (the reference draws rand() there, dtw.cpp:40-45).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/deepbinner_dtw.h"

namespace {

constexpr int kLanes = 64;
enum : unsigned { NIL = 0, DIAGONAL = 1, LEFT = 2, UP = 3 };

struct PairJob {
    int64_t ref_off, query_off;   // first sample in refs / queries
    int64_t path_off;             // first direction word of this pair (in words of the launch)
    int64_t edge_off;             // first double of its 2 * ref_len edge buffer (multi-panel only)
    int64_t align_off;            // first int of its alignment
    int32_t ref_len, query_len;
    int32_t pair, pad;
};

using Word = uint32_t;   // 2 bits per cell, at most 16 cells per lane

// lane l <- lane l - 1 (DPP wave_shr:1); lane 0 receives 0
__device__ __forceinline__ double from_left_lane(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x138, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x138, 0xF, 0xF, false);
    return __builtin_bit_cast(double,
                              (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// v_min_f64 as it is: llvm.minnum would first canonicalise both inputs (IEEE mode, signalling
// NaNs), two more instructions on the dependent chain of every cell
__device__ __forceinline__ double min_f64(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// the double held by lane `lane` (uniform), for every lane
__device__ __forceinline__ double lane_value(double v, int lane) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)b, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, (long long)(((unsigned long long)hi << 32) | lo));
}

__device__ __forceinline__ Word word_of_lane(Word w, int lane) {
    return (Word)__builtin_amdgcn_readlane((int)w, lane);
}

template <int C>
__global__ __launch_bounds__(kLanes) void dtw_kernel(
    const PairJob* __restrict__ jobs, const double* __restrict__ refs,
    const double* __restrict__ queries, Word* __restrict__ path,
    double* __restrict__ edges, double* __restrict__ distances, int32_t* __restrict__ positions,
    int32_t* __restrict__ path_lengths, int32_t* __restrict__ alignment) {
    constexpr int kPanel = kLanes * C;
    // only the widest variant ever sees a query of more than one panel (cells_per_lane): the
    // narrower ones drop the edge-column hand-over from their step altogether
    constexpr bool kMultiPanel = (C == 16);
    const PairJob job = jobs[blockIdx.x];
    const int lane = threadIdx.x;
    const int R = job.ref_len, Q = job.query_len;
    const double* ref = refs + job.ref_off;
    const double* query = queries + job.query_off;
    Word* words = path + job.path_off;
    double* edge = edges + job.edge_off;
    const int panels = (Q + kPanel - 1) / kPanel;
    // direction words are stored by STEP, [panel][s][lane] with s = row + lane: the 64 lanes of a
    // step then write 64 consecutive words (by row they would hit 64 different cache lines)
    const size_t kSteps = (size_t)R + kLanes - 1;
    const int last_lane = ((Q - 1) % kPanel) / C;
    const int last_c = __builtin_amdgcn_readfirstlane((Q - 1) % C);

    // ---- the fill (dtw.cpp:68-111) ----------------------------------------------------------
    double best = DBL_MAX;
    int best_i = 0;
    for (int p = 0; p < panels; ++p) {
        const int j0 = p * kPanel + lane * C;
        // cost[] holds the row above; +inf above row 0 (and in diag_in) makes the general rule
        // pick LEFT along the top row, which is what dtw.cpp:81-87 fills in there
        double q[C], cost[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            q[c] = (j0 + c < Q) ? query[j0 + c] : 0.0;
            cost[c] = HUGE_VAL;
        }
        const double* edge_in = edge + (size_t)((p + 1) & 1) * R;   // what panel p - 1 wrote
        double* edge_out = edge + (size_t)(p & 1) * R;
        const bool tracks_end = (p == panels - 1) && (lane == last_lane);
        const bool first_column = (j0 == 0);
        double newest_last = 0.0;    // cost of my last column in my newest row
        double diag_in = HUGE_VAL;   // left neighbour's last column one row up
        // Lane l needs ref[s - l] at step s - what lane l - 1 had one step earlier, so the
        // reference value travels along the lanes like the costs do and only lane 0 takes in a new
        // one per step.  Those come 64 at a time: one coalesced load per 64 steps, issued 64 steps
        // ahead, handed out with v_readlane - a load per step would put its latency on every
        // step.  The edge column of the previous panel reaches lane 0 the same way.
        auto block_at = [&](const double* base, int first) {
            const int at = first + lane;
            return (at < R) ? base[at] : 0.0;
        };
        double ref_block = block_at(ref, 0), ref_ahead = 0.0;
        double edge_block = 0.0, edge_ahead = 0.0;
        if constexpr (kMultiPanel) edge_block = (p > 0) ? block_at(edge_in, 0) : 0.0;
        double r = 0.0;
        for (int s = 0; s < R + kLanes - 1; ++s) {
            const int i = s - lane;
            const int k = s % kLanes;
            if (k == 0) {
                ref_ahead = block_at(ref, s + kLanes);
                if constexpr (kMultiPanel)
                    if (p > 0) edge_ahead = block_at(edge_in, s + kLanes);
            }
            double left_in = from_left_lane(newest_last);
            r = from_left_lane(r);
            const double ref_s = lane_value(ref_block, k);
            if (lane == 0) r = ref_s;
            if constexpr (kMultiPanel) {
                const double edge_s = lane_value(edge_block, k);
                if (lane == 0 && p > 0) left_in = edge_s;
            }
            if (k == kLanes - 1) {
                ref_block = ref_ahead;
                if constexpr (kMultiPanel) edge_block = edge_ahead;
            }
            if (i < 0 || i >= R) continue;
            double left = left_in, diag = diag_in;
            Word word = 0;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                // dtw.cpp:29-46 without branches: the cost to come from is min(diag, left, top);
                // DIAGONAL iff that is the diagonal (it wins ties), else UP iff top < left
                const double top = cost[c];
                const double d = r - q[c];
                const double from = min_f64(left, min_f64(diag, top));
                unsigned dir = (from == diag) ? DIAGONAL : ((top < left) ? UP : LEFT);
                double value = from + d * d;
                if (c == 0 && first_column) {   // dtw.cpp:68-78: the query may start anywhere
                    value = 0.0;
                    dir = NIL;
                }
                word |= dir << (2 * c);
                diag = top;
                left = value;
                cost[c] = value;
            }
            words[((size_t)p * kSteps + s) * kLanes + lane] = word;
            newest_last = cost[C - 1];
            diag_in = left_in;
            if constexpr (kMultiPanel)
                if (lane == kLanes - 1 && p + 1 < panels) edge_out[i] = newest_last;
            if (tracks_end && i >= 1) {    // dtw.cpp:113-122: first smallest over rows 1..R-1
                const double v = cost[last_c];   // uniform index: s_set_gpr_idx + v_mov
                if (v < best) {
                    best = v;
                    best_i = i;
                }
            }
        }
        __threadfence();    // the edge column (and, last time round, the words) for other lanes
    }
    best = __shfl(best, last_lane);
    best_i = __shfl(best_i, last_lane);

    // ---- the walk back (dtw.cpp:124-145) ----------------------------------------------------
    int32_t* out = alignment ? alignment + job.align_off : nullptr;
    int i = best_i, j = Q - 1, n = 0;
    for (;;) {
        const int p = j / kPanel, chunk = (j % kPanel) / C;
        const int tile_top = i;
        const int row = tile_top - lane;
        Word mine = 0;
        if (row >= 0) mine = words[((size_t)p * kSteps + row + chunk) * kLanes + chunk];
        bool done = false;
        for (;;) {
            if (out && lane == 0) {
                out[2 * n] = i;
                out[2 * n + 1] = j;
            }
            ++n;
            if (j == 0) {
                done = true;
                break;
            }
            const Word w = word_of_lane(mine, __builtin_amdgcn_readfirstlane(tile_top - i));
            const unsigned dir = (unsigned)(w >> (2 * (j % C))) & 3u;
            if (dir != LEFT) --i;
            if (dir != UP) --j;
            if (tile_top - i >= kLanes || j / C != p * kLanes + chunk) break;
        }
        if (done) break;
    }
    if (lane == 0) {
        distances[job.pair] = best;
        positions[2 * job.pair] = i;
        positions[2 * job.pair + 1] = best_i;
        path_lengths[job.pair] = n;
    }
}

// ------------------------------------------------------------------------------------------------
thread_local std::string g_error;

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    hipError_t reserve(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
        const hipError_t e = hipMalloc(&ptr, need);
        if (e == hipSuccess) bytes = need;
        return e;
    }
};

struct Context {
    std::mutex lock;
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t begin = nullptr, end = nullptr;
    DeviceBuffer refs, queries, jobs, path, edges, distances, positions, lengths, alignment;
    double last_ms = 0.0;
    int64_t last_cells = 0;
};
Context g_ctx;

#define DTW_HIP(call)                                                                    \
    do {                                                                                 \
        const hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) {                                                          \
            g_error = std::string(#call) + ": " + hipGetErrorString(e_);                 \
            return DTW_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

int prepare(Context& ctx) {
    int device = 0, count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
        g_error = "no HIP device";
        return DTW_ERR_NO_DEVICE;
    }
    DTW_HIP(hipGetDevice(&device));
    if (ctx.device != device) {
        hipDeviceProp_t prop;
        DTW_HIP(hipGetDeviceProperties(&prop, device));
        if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
            g_error = std::string("built for gfx950, device is ") + prop.gcnArchName;
            return DTW_ERR_NO_DEVICE;
        }
        // first use, or the caller switched devices: buffers of the other device are dropped
        for (DeviceBuffer* b : {&ctx.refs, &ctx.queries, &ctx.jobs, &ctx.path, &ctx.edges,
                                &ctx.distances, &ctx.positions, &ctx.lengths, &ctx.alignment})
            *b = DeviceBuffer{};
        ctx.device = device;
        DTW_HIP(hipStreamCreateWithFlags(&ctx.stream, hipStreamNonBlocking));
        DTW_HIP(hipEventCreate(&ctx.begin));
        DTW_HIP(hipEventCreate(&ctx.end));
    }
    return DTW_OK;
}

size_t path_budget_bytes() {
    if (const char* text = std::getenv("DEEPBINNER_DTW_PATH_BYTES")) {
        const long long v = std::atoll(text);
        if (v > 0) return (size_t)v;
    }
    return (size_t)8 << 30;
}

int cells_per_lane(int query_len) {
    // longer queries than 64 * 16 columns go in panels of that width: measured faster than 32
    // columns per lane, which leaves room for two waves per SIMD only
    for (int c : {4, 8})
        if (query_len <= kLanes * c) return c;
    return 16;
}

template <int C>
void launch(Context& ctx, int n, int32_t* alignment) {
    hipLaunchKernelGGL(dtw_kernel<C>, dim3(n), dim3(kLanes), 0, ctx.stream,
                       (const PairJob*)ctx.jobs.ptr, (const double*)ctx.refs.ptr,
                       (const double*)ctx.queries.ptr, (Word*)ctx.path.ptr,
                       (double*)ctx.edges.ptr, (double*)ctx.distances.ptr,
                       (int32_t*)ctx.positions.ptr, (int32_t*)ctx.lengths.ptr, alignment);
}

// One launch: jobs of one lane width whose direction words fit the budget.
int run_group(Context& ctx, int c, std::vector<PairJob>& jobs, bool want_alignment) {
    const int panel = kLanes * c;
    // the longest first: the tail of the launch is then made of the short ones
    std::sort(jobs.begin(), jobs.end(), [](const PairJob& a, const PairJob& b) {
        return (int64_t)a.ref_len * a.query_len > (int64_t)b.ref_len * b.query_len;
    });
    size_t words = 0, edge_doubles = 0;
    for (PairJob& job : jobs) {
        const int panels = (job.query_len + panel - 1) / panel;
        job.path_off = (int64_t)words;
        words += (size_t)panels * ((size_t)job.ref_len + kLanes - 1) * kLanes;
        job.edge_off = (int64_t)edge_doubles;
        if (panels > 1) edge_doubles += (size_t)2 * job.ref_len;
    }
    DTW_HIP(ctx.path.reserve(std::max<size_t>(words * sizeof(Word), 8)));
    DTW_HIP(ctx.edges.reserve(std::max<size_t>(edge_doubles * sizeof(double), 8)));
    DTW_HIP(ctx.jobs.reserve(jobs.size() * sizeof(PairJob)));
    DTW_HIP(hipMemcpyAsync(ctx.jobs.ptr, jobs.data(), jobs.size() * sizeof(PairJob),
                           hipMemcpyHostToDevice, ctx.stream));
    int32_t* alignment = want_alignment ? (int32_t*)ctx.alignment.ptr : nullptr;
    DTW_HIP(hipEventRecord(ctx.begin, ctx.stream));
    switch (c) {
        case 4: launch<4>(ctx, (int)jobs.size(), alignment); break;
        case 8: launch<8>(ctx, (int)jobs.size(), alignment); break;
        default: launch<16>(ctx, (int)jobs.size(), alignment); break;
    }
    DTW_HIP(hipGetLastError());
    DTW_HIP(hipEventRecord(ctx.end, ctx.stream));
    DTW_HIP(hipStreamSynchronize(ctx.stream));
    float ms = 0.f;
    DTW_HIP(hipEventElapsedTime(&ms, ctx.begin, ctx.end));
    ctx.last_ms += ms;
    for (const PairJob& job : jobs) ctx.last_cells += (int64_t)job.ref_len * job.query_len;
    return DTW_OK;
}

int run_batch(Context& ctx, const double* refs, const int64_t* ref_offsets, const double* queries,
              const int64_t* query_offsets, int64_t n_pairs, double* distances, int32_t* positions,
              int32_t* path_lengths, int32_t* alignment) {
    const int64_t n_ref = ref_offsets[n_pairs] - ref_offsets[0];
    const int64_t n_query = query_offsets[n_pairs] - query_offsets[0];
    const size_t align_ints = (size_t)2 * (size_t)(ref_offsets[n_pairs] + query_offsets[n_pairs]);
    ctx.last_ms = 0.0;
    ctx.last_cells = 0;
    DTW_HIP(ctx.refs.reserve((size_t)n_ref * sizeof(double)));
    DTW_HIP(ctx.queries.reserve((size_t)n_query * sizeof(double)));
    DTW_HIP(ctx.distances.reserve((size_t)n_pairs * sizeof(double)));
    DTW_HIP(ctx.positions.reserve((size_t)n_pairs * 2 * sizeof(int32_t)));
    DTW_HIP(ctx.lengths.reserve((size_t)n_pairs * sizeof(int32_t)));
    if (alignment) DTW_HIP(ctx.alignment.reserve(align_ints * sizeof(int32_t)));
    DTW_HIP(hipMemcpyAsync(ctx.refs.ptr, refs + ref_offsets[0], (size_t)n_ref * sizeof(double),
                           hipMemcpyHostToDevice, ctx.stream));
    DTW_HIP(hipMemcpyAsync(ctx.queries.ptr, queries + query_offsets[0],
                           (size_t)n_query * sizeof(double), hipMemcpyHostToDevice, ctx.stream));

    const size_t budget = path_budget_bytes();
    for (int c : {4, 8, 16}) {
        std::vector<PairJob> group;
        size_t bytes = 0;
        for (int64_t p = 0; p <= n_pairs; ++p) {
            size_t need = 0;
            if (p < n_pairs) {
                const int64_t r = ref_offsets[p + 1] - ref_offsets[p];
                const int64_t q = query_offsets[p + 1] - query_offsets[p];
                if (cells_per_lane((int)q) != c) continue;
                const int64_t panels = (q + kLanes * c - 1) / (kLanes * c);
                need = (size_t)panels * (size_t)(r + kLanes - 1) * kLanes * sizeof(Word);
                if (!group.empty() && bytes + need > budget) {
                    const int st = run_group(ctx, c, group, alignment != nullptr);
                    if (st != DTW_OK) return st;
                    group.clear();
                    bytes = 0;
                }
                PairJob job{};
                job.ref_off = ref_offsets[p] - ref_offsets[0];
                job.query_off = query_offsets[p] - query_offsets[0];
                job.align_off = 2 * (ref_offsets[p] + query_offsets[p]);
                job.ref_len = (int32_t)r;
                job.query_len = (int32_t)q;
                job.pair = (int32_t)p;
                group.push_back(job);
                bytes += need;
            } else if (!group.empty()) {
                const int st = run_group(ctx, c, group, alignment != nullptr);
                if (st != DTW_OK) return st;
            }
        }
    }
    DTW_HIP(hipMemcpyAsync(distances, ctx.distances.ptr, (size_t)n_pairs * sizeof(double),
                           hipMemcpyDeviceToHost, ctx.stream));
    DTW_HIP(hipMemcpyAsync(positions, ctx.positions.ptr, (size_t)n_pairs * 2 * sizeof(int32_t),
                           hipMemcpyDeviceToHost, ctx.stream));
    DTW_HIP(hipMemcpyAsync(path_lengths, ctx.lengths.ptr, (size_t)n_pairs * sizeof(int32_t),
                           hipMemcpyDeviceToHost, ctx.stream));
    DTW_HIP(hipStreamSynchronize(ctx.stream));
    if (alignment) {
        const size_t first = (size_t)2 * (size_t)(ref_offsets[0] + query_offsets[0]);
        DTW_HIP(hipMemcpyAsync(alignment + first, (const int32_t*)ctx.alignment.ptr + first,
                               (align_ints - first) * sizeof(int32_t), hipMemcpyDeviceToHost,
                               ctx.stream));
        DTW_HIP(hipStreamSynchronize(ctx.stream));
    }
    return DTW_OK;
}

}  // namespace

extern "C" {

const char* dtw_version(void) { return "deepbinner_dtw 1 (gfx950)"; }

const char* dtw_status_string(int status) {
    switch (status) {
        case DTW_OK: return "ok";
        case DTW_ERR_ARGUMENT: return "invalid argument";
        case DTW_ERR_NO_DEVICE: return "no gfx950 device";
        case DTW_ERR_HIP: return "HIP error";
        default: return "unknown status";
    }
}

const char* dtw_last_error(void) { return g_error.c_str(); }

int dtw_semi_global_batch(const double* refs, const int64_t* ref_offsets, const double* queries,
                          const int64_t* query_offsets, int64_t n_pairs, double* distances,
                          int32_t* positions, int32_t* path_lengths, int32_t* alignment) {
    if (n_pairs < 0 || (n_pairs > 0 && (!refs || !ref_offsets || !queries || !query_offsets ||
                                        !distances || !positions || !path_lengths))) {
        g_error = "null pointer or negative pair count";
        return DTW_ERR_ARGUMENT;
    }
    if (n_pairs == 0) return DTW_OK;
    if (ref_offsets[0] < 0 || query_offsets[0] < 0) {
        g_error = "negative offset";
        return DTW_ERR_ARGUMENT;
    }
    for (int64_t p = 0; p < n_pairs; ++p) {
        const int64_t r = ref_offsets[p + 1] - ref_offsets[p];
        const int64_t q = query_offsets[p + 1] - query_offsets[p];
        if (r < 1 || q < 1 || r > (1 << 30) || q > (1 << 30)) {
            g_error = "every pair needs 1 .. 2^30 samples of reference and of query";
            return DTW_ERR_ARGUMENT;
        }
    }
    std::lock_guard<std::mutex> guard(g_ctx.lock);
    const int st = prepare(g_ctx);
    if (st != DTW_OK) return st;
    return run_batch(g_ctx, refs, ref_offsets, queries, query_offsets, n_pairs, distances,
                     positions, path_lengths, alignment);
}

double semi_global_dtw(const double* ref, const double* query, int ref_len, int query_len,
                       int* alignment, int* positions, int* path_length) {
    if (path_length) path_length[0] = 0;
    if (!positions || !path_length) return std::nan("");
    const int64_t ref_offsets[2] = {0, ref_len}, query_offsets[2] = {0, query_len};
    double distance = 0.0;
    int32_t n = 0;
    const int st = dtw_semi_global_batch(ref, ref_offsets, query, query_offsets, 1, &distance,
                                         positions, &n, alignment);
    if (st != DTW_OK) return std::nan("");
    path_length[0] = n;
    return distance;
}

int dtw_last_kernel_time(double* milliseconds, int64_t* cells) {
    std::lock_guard<std::mutex> guard(g_ctx.lock);
    if (milliseconds) *milliseconds = g_ctx.last_ms;
    if (cells) *cells = g_ctx.last_cells;
    return DTW_OK;
}

}  // extern "C"
