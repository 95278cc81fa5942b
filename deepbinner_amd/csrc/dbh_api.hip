// dbh_api.hip — host side of libdeepbinner_hip.so: the C ABI declared in
// include/deepbinner_hip.h.  Packs the canonical weight blob into the kernel's fragment order,
// owns the device buffers, launches the kernels of dbh_forward.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <utility>
#include <mutex>
#include <vector>

#include "../../include/deepbinner_hip.h"
#define DBH_FORWARD_NS dbh
#define DBH_TIMELINE 0
#include "dbh_forward.hip"
#undef DBH_FORWARD_NS
#undef DBH_TIMELINE
#define DBH_FORWARD_NS dbh_timeline      // the same kernels with the cycle stamps compiled in
#define DBH_TIMELINE 1
#include "dbh_forward.hip"
#undef DBH_FORWARD_NS
#undef DBH_TIMELINE

namespace {

thread_local std::string g_last_error;

int hip_fail(hipError_t e, const char* what) {
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? DBH_ERR_OUT_OF_MEMORY : DBH_ERR_HIP;
}

#define DBH_HIP(call)                                         \
    do {                                                      \
        hipError_t e_ = (call);                               \
        if (e_ != hipSuccess) return hip_fail(e_, #call);     \
    } while (0)

// canonical (Keras) shapes of the 20 convolutions for a given class count
struct CanonConv { int k, cin, cout; };
void canon_convs(int n_classes, CanonConv out[dbh::kNumConvs]) {
    for (int i = 0; i < dbh::kNumConvs; ++i) {
        out[i].k = dbh::kConv[i].taps;
        out[i].cin = dbh::kConv[i].cin;
        out[i].cout = (i == dbh::kNumConvs - 1) ? n_classes : dbh::kConv[i].cout_pad;
    }
}

int64_t canon_param_count(int n_classes) {
    CanonConv cc[dbh::kNumConvs];
    canon_convs(n_classes, cc);
    int64_t n = 0;
    for (int i = 0; i < dbh::kNumConvs; ++i) n += (int64_t)cc[i].k * cc[i].cin * cc[i].cout + cc[i].cout;
    for (int i = 0; i < dbh::kNumBn; ++i) n += 4 * dbh::kBnChannels[i];
    return n;
}

// canonical blob -> packed buffer (layout: dbh_layout.h)
void pack_weights(const float* w, int n_classes, std::vector<float>& packed) {
    using namespace dbh;
    packed.assign(kPackedFloats, 0.f);
    CanonConv cc[kNumConvs];
    canon_convs(n_classes, cc);
    // BN2 (scale s, shift t per channel of conv1d_4's pooled output) is folded into conv1d_5, a 1x1
    // convolution with no padding to get in the way: W5'[c][o] = W5[c][o] s[c], b5'[o] = b5[o] +
    // sum_c W5[c][o] t[c] - exact algebra, in fp64 here; the forward kernel feeds conv1d_5 the
    // pooled values as they are (dbh_forward.hip: stage_b_chain)
    const float* bn_first = w;
    for (int i = 0; i < kNumConvs; ++i) bn_first += (size_t)cc[i].k * cc[i].cin * cc[i].cout + cc[i].cout;
    std::vector<double> fold_scale(48, 1.0), fold_shift(48, 0.0);
    {
        const float* q = bn_first + 4 * kBnChannels[0];      // BN2 = the second batch normalisation
        const float *gamma = q, *beta = q + 48, *mean = q + 96, *var = q + 144;
        for (int c = 0; c < 48; ++c) {
            fold_scale[c] = (double)gamma[c] / std::sqrt((double)var[c] + 1e-3);
            fold_shift[c] = (double)beta[c] - (double)mean[c] * fold_scale[c];
        }
    }
    static_assert(kBnChannels[1] == 48 && kConv[4].cin == 48 && kConv[4].taps == 1, "");
    const float* p = w;
    for (int i = 0; i < kNumConvs; ++i) {
        const int k = cc[i].k, cin = cc[i].cin, cout = cc[i].cout;
        const float* kernel = p;                 // [k][cin][cout]
        const float* bias = p + (size_t)k * cin * cout;
        p = bias + cout;
        float* dst = packed.data() + weight_offset(i);
        if (i == 0) {
            // conv1d_1 stays [tap][cout]: one value per lane and channel group, read once per workgroup
            for (int tap = 0; tap < 3; ++tap)
                for (int c = 0; c < cout; ++c) dst[tap * 48 + c] = kernel[(tap * cin) * cout + c];
        } else if (kConv[i].wino) {
            // Winograd F(2,3) / F(4,3): the transformed matrices V = G g, computed in fp64, each
            // in fragment order
            const int sp_n = cin / 8, nt = kConv[i].cout_pad / 16;
            const int n_xi = kConv[i].wino == 4 ? 6 : 4;
            static const double G23[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
            static const double G43[6][3] = {{1. / 4, 0, 0},
                                             {-1. / 6, -1. / 6, -1. / 6},
                                             {-1. / 6, 1. / 6, -1. / 6},
                                             {1. / 24, 1. / 12, 1. / 6},
                                             {1. / 24, -1. / 12, 1. / 6},
                                             {0, 0, 1}};
            for (int xi = 0; xi < n_xi; ++xi) {
                const double* G = kConv[i].wino == 4 ? G43[xi] : G23[xi];
                for (int sp = 0; sp < sp_n; ++sp)
                    for (int t = 0; t < nt; ++t)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 2; ++e) {
                                // (conv1d_2's k-steps walk the channels in the order conv1d_1's
                                // transposed MFMAs leave them in registers: dbh_layout.h)
                                const int ci = frag_cin(i, sp, lane >> 4, e);
                                const int co = 16 * t + (lane & 15);
                                double v = 0.0;
                                if (co < cout)
                                    for (int tap = 0; tap < 3; ++tap)
                                        v += G[tap] * (double)kernel[((size_t)tap * cin + ci) * cout + co];
                                // F(2,3): matrix-major [xi][sp][t][lane][e].  F(4,3): by N tile,
                                // [t][sp][xi >> 1][lane][xi & 1][e], so that one 16-byte LDS read
                                // fetches the fragments of two matrices for two k-steps.
                                const size_t idx =
                                    kConv[i].wino == 4
                                        ? (((((size_t)t * sp_n + sp) * 3 + (xi >> 1)) * 64 + lane) * 2 + (xi & 1)) * 2 + e
                                        : wino2_by_tile(i)
                                              ? (((((size_t)t * sp_n + sp) * 2 + (xi >> 1)) * 64 + lane) * 2 + (xi & 1)) * 2 + e
                                              : ((((size_t)xi * sp_n + sp) * nt + t) * 64 + lane) * 2 + e;
                                dst[idx] = (float)v;
                            }
            }
        } else {
            const int sp_n = cin / 8, nt = kConv[i].cout_pad / 16;
            for (int tap = 0; tap < k; ++tap)
                for (int sp = 0; sp < sp_n; ++sp)
                    for (int t = 0; t < nt; ++t)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int e = 0; e < 2; ++e) {
                                const int ci = frag_cin(i, sp, lane >> 4, e);
                                const int co = 16 * t + (lane & 15);
                                float v = co < cout ? kernel[((size_t)tap * cin + ci) * cout + co] : 0.f;
                                if (i == 4 && DBH_FOLD_BN2) v = (float)((double)v * fold_scale[ci]);   // BN2's scale
                                dst[((((size_t)tap * sp_n + sp) * nt + t) * 64 + lane) * 2 + e] = v;
                            }
        }
        float* bdst = packed.data() + bias_offset(i);
        for (int c = 0; c < cout; ++c) bdst[c] = bias[c] * kActScale;      // (exact: a power of two)
        if (i == 4 && DBH_FOLD_BN2)       // ... and BN2's shift, through conv1d_5's weights, in its bias
            for (int c = 0; c < cout; ++c) {
                double extra = 0.0;
                for (int ci = 0; ci < cin; ++ci) extra += (double)kernel[(size_t)ci * cout + c] * fold_shift[ci];
                bdst[c] = (float)(((double)bias[c] + extra) * (double)kActScale);
            }
    }
    for (int i = 0; i < kNumBn; ++i) {
        const int c_n = kBnChannels[i];
        const float *gamma = p, *beta = p + c_n, *mean = p + 2 * c_n, *var = p + 3 * c_n;
        p += 4 * c_n;
        float* sc = packed.data() + bn_scale_offset(i);
        float* sh = packed.data() + bn_shift_offset(i);
        for (int c = 0; c < c_n; ++c) {
            const double scale = (double)gamma[c] / std::sqrt((double)var[c] + 1e-3);
            sc[c] = (float)scale;
            sh[c] = (float)((double)beta[c] - (double)mean[c] * scale) * kActScale;
        }
    }
}

}  // namespace

struct dbh_model {
    int n_classes = 0;
    int device = 0;
    int cus = 256;               // workgroups of a persistent forward launch (one per CU)
    int cus_total = 256;         // ... before dbh_model_reserve_cus took some away
    bool windows_by_counter = true;        // DEEPBINNER_STATIC_WINDOWS=1: fixed shares (A/B)
    int inflate_streams_per_lane = 0;      // dbh_classify_pair_deflated -> dbh_inflate_dev; 0 = by
                                           // the lengths of the streams
    bool launch_per_batch = false;   // DEEPBINNER_LAUNCH_PER_BATCH=1: one launch per batch (A/B)
    // The end of a persistent launch: with fewer than chunk4_rounds x grid windows not yet handed
    // out a workgroup asks for groups of 2 instead of 4, below chunk2_rounds x grid for single
    // windows (DEEPBINNER_CHUNK4_ROUNDS / DEEPBINNER_CHUNK2_ROUNDS: A/B)
    // (measured on configs[1] one launch per step, 10,000 windows: 8 / 3: 5.44 M reads/s, 4 / 2: 5.75 M,
    // 2 / 1: 5.74 M, 0 / 0 - always four - 5.74 M: a small group pays the whole stage D-E-F chain for
    // one or two windows, so they are kept to the launch's very end)
    int chunk4_rounds = 2, chunk2_rounds = 1;
    float* d_packed = nullptr;
    // workspace for the host-pointer entry points, grown on demand
    void* d_in = nullptr;      size_t in_bytes = 0;
    void* d_work = nullptr;    size_t work_bytes = 0;
    void* d_out = nullptr;     size_t out_bytes = 0;
    // conv17 outputs parked per workgroup until the batched tail runs (dbh_forward.hip); one
    // buffer per stream a model launches on - launches on one stream follow each other, launches
    // on two streams may overlap: `tails` for the streams callers of the *_dev entry points
    // bring, one per staging slot for the host-buffer pipeline
    struct Tail { hipStream_t stream; void* ptr; size_t bytes; };
    std::vector<Tail> tails;
    void* d_tail = nullptr;    size_t tail_bytes = 0;      // the timeline entry points' (null stream)
    void* d_clock = nullptr;   size_t clock_bytes = 0;     // dbh_forward_clock_enable
    bool clock_probe = false;  unsigned clock_grid = 0;
    bool phase_probe = false;      // dbh_forward_phases_enable
    // staging slots of the host-buffer entry points (classify_host: overlapped H2D / kernels / D2H)
    static constexpr int kSlots = 3;
    static constexpr int64_t kDefaultGroup = 32768;
    int64_t host_group_windows = kDefaultGroup;     // dbh_model_set_host_group
    bool host_zero_copy = true;      // DEEPBINNER_HOST_ZERO_COPY=0: copy host samples to HBM first
    struct Slot {
        hipStream_t stream = nullptr;
        void* h_in = nullptr;   size_t h_in_bytes = 0;     // pinned
        void* h_out = nullptr;  size_t h_out_bytes = 0;    // pinned
        void* d_in = nullptr;   size_t d_in_bytes = 0;
        void* d_out = nullptr;  size_t d_out_bytes = 0;
        void* d_work = nullptr; size_t d_work_bytes = 0;
        void* d_tail = nullptr; size_t d_tail_bytes = 0;
    } slot[kSlots];
    // buffers of dbh_classify_pair_deflated (the start model's, or the only model's, are used)
    struct Deflated {
        hipStream_t stream = nullptr;
        hipEvent_t inflated = nullptr, classified = nullptr;    // hand-over to the forward stream
        void* h_small = nullptr;  size_t h_small_bytes = 0;     // pinned: records, offsets, results
        void* h_comp = nullptr;   size_t h_comp_bytes = 0;      // pinned staging for pageable input
        void* d_comp = nullptr;   size_t d_comp_bytes = 0;
        void* d_small = nullptr;  size_t d_small_bytes = 0;
        void* d_samples = nullptr; size_t d_samples_bytes = 0;
        void* d_tokens = nullptr; size_t d_tokens_bytes = 0;
        void* d_out = nullptr;    size_t d_out_bytes = 0;
        void* d_work = nullptr;   size_t d_work_bytes = 0;
        void* d_tail = nullptr;   size_t d_tail_bytes = 0;
        std::vector<int32_t> order;                             // the records as they were sent
    } deflated;
    // live timing of the forward kernel (dbh_forward_timing_*)
    int64_t hint_len = 0, hint_cap = 0;   // dbh_model_set_read_length_hint
    int timing = 0;              // 0 = off, n = open an event bracket at every n-th forward launch
    int timing_span = 1;         // consecutive launches one bracket covers (<= timing)
    hipEvent_t open_stop = nullptr;   // stop event of the bracket being filled
    int64_t open_windows = 0;
    int64_t timed_launches = 0;
    int64_t launch_counter = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
    size_t events_used = 0;
    int64_t timed_windows = 0;
};

namespace {

int ensure(void** ptr, size_t* have, size_t need) {
    if (*have >= need) return DBH_OK;
    if (*ptr) {
        DBH_HIP(hipFree(*ptr));
        *ptr = nullptr;
        *have = 0;
    }
    DBH_HIP(hipMalloc(ptr, need));
    *have = need;
    return DBH_OK;
}

struct FusedInput {          // seam-b2 mode of the forward kernel (all null/zero = seam b1)
    const int16_t* samples = nullptr;
    const int64_t* offsets = nullptr;
    int steps = 1;
    int side = 0;
    double score_diff = 0.0;
    int32_t* calls = nullptr;
    // "every read is len_hint samples long": offsets[0] belongs to read number read0 of the
    // sample buffer, which holds at least hint_cap samples (see dbh_model_set_read_length_hint)
    int64_t read0 = 0, len_hint = 0, hint_cap = 0;
    // where this launch parks its conv17 outputs (null = the model's own buffer)
    void** tail = nullptr;
    size_t* tail_bytes = nullptr;
};

int ensure_host(void** ptr, size_t* have, size_t need) {
    if (*have >= need) return DBH_OK;
    if (*ptr) {
        DBH_HIP(hipHostFree(*ptr));
        *ptr = nullptr;
        *have = 0;
    }
    DBH_HIP(hipHostMalloc(ptr, need, hipHostMallocDefault));
    *have = need;
    return DBH_OK;
}

int launch_forward(dbh_model* m, const float* x_dev, int64_t n, float* probs_dev,
                   int debug_stage, float* debug_dev, hipStream_t stream,
                   const FusedInput& in = FusedInput()) {
    if (n == 0) return DBH_OK;
    // grid.x limit is 2^31-1 blocks; chunk anyway to keep launches bounded
    const int64_t kChunk = (int64_t)(1 << 20) * in.steps;   // whole reads per launch
    for (int64_t off = 0; off < n; off += kChunk) {
        const int64_t cnt = (n - off < kChunk) ? (n - off) : kChunk;
        hipEvent_t ev_stop = nullptr;
        if (m->timing > 0 && (debug_stage < 0 || debug_stage >= 100)) {
            const int64_t pos = m->launch_counter++ % m->timing;
            if (pos == 0) {
                if (m->events_used == m->events.size()) {
                    hipEvent_t a, b;
                    DBH_HIP(hipEventCreate(&a));
                    DBH_HIP(hipEventCreate(&b));
                    m->events.emplace_back(a, b);
                }
                DBH_HIP(hipEventRecord(m->events[m->events_used].first, stream));
                m->open_stop = m->events[m->events_used].second;
                m->open_windows = 0;
            }
            if (m->open_stop && pos < m->timing_span) {
                m->open_windows += cnt;
                if (pos == m->timing_span - 1) {      // last launch of the bracket
                    ev_stop = m->open_stop;
                    m->open_stop = nullptr;
                }
            }
        }
        // production launches are persistent: at most one workgroup per CU, each walking its
        // share of the windows; the debug / timeline modes keep one workgroup per window
        // (a workgroup takes its windows in groups of dbh::kGroup: dbh_forward.hip)
        const bool one_per_group = debug_stage >= 0 && debug_stage != 301;
        const int64_t groups = (cnt + dbh::kGroup - 1) / dbh::kGroup;
        const unsigned grid = (unsigned)(one_per_group || groups < m->cus ? groups : m->cus);
        void** tail = in.tail;
        size_t* tail_bytes = in.tail_bytes;
        if (!tail) {
            size_t k = 0;
            while (k < m->tails.size() && m->tails[k].stream != stream) ++k;
            if (k == m->tails.size()) m->tails.push_back(dbh_model::Tail{stream, nullptr, 0});
            tail = &m->tails[k].ptr;
            tail_bytes = &m->tails[k].bytes;
        }
        // (the first 256 bytes: the counter the launch's workgroups take their windows off -
        // zero between launches, so zeroed here only when the buffer is new)
        {
            void* before = *tail;
            const int st = ensure(tail, tail_bytes, 256 + (size_t)grid * dbh::kWgScratchFloats * sizeof(float));
            if (st != DBH_OK) return st;
            if (*tail != before) DBH_HIP(hipMemsetAsync(*tail, 0, 256, stream));
        }
        dbh::ForwardArgs a;
        a.packed = m->d_packed;
        a.x = x_dev ? x_dev + off * dbh::kWindow : nullptr;
        a.probs = probs_dev ? probs_dev + off * m->n_classes : nullptr;
        a.debug_out = debug_dev ? debug_dev + off * dbh::kStageFloats[debug_stage < 0 || debug_stage > 7 ? 0 : debug_stage]
                                : nullptr;
        a.samples = in.samples;
        a.offsets = in.offsets ? (const long long*)(in.offsets + off / in.steps) : nullptr;
        a.calls = in.calls ? (int*)(in.calls + off / in.steps) : nullptr;
        a.tail_scratch = (float*)((char*)*tail + 256);
        // production launches hand their windows out by counter; the debug and timeline modes
        // (which may stop half-way, and index their output by window) keep fixed shares
        a.win_counter = (debug_stage < 0 && m->windows_by_counter) ? (int*)*tail : nullptr;
        a.clock_out = nullptr;
        if (m->clock_probe && debug_stage < 0) {
            const int st = ensure(&m->d_clock, &m->clock_bytes,
                                  (size_t)grid * (4 + dbh::kPhaseMarks * dbh::kPhaseGroups) * sizeof(int64_t));
            if (st != DBH_OK) return st;
            a.clock_out = (long long*)m->d_clock;
            m->clock_grid = grid;
        }
        a.score_diff = in.score_diff;
        a.read0 = (long long)(in.read0 + off / in.steps);
        a.len_hint = (long long)in.len_hint;
        a.hint_cap = (long long)in.hint_cap;
        a.n_windows = (long long)cnt;
        a.n_classes = m->n_classes;
        a.debug_stage = debug_stage;
        a.steps = in.steps;
        a.side = in.side;
        a.phases = m->phase_probe ? 1 : 0;
        a.chunk4_min_left = m->chunk4_rounds * (int)grid;
        a.chunk2_min_left = m->chunk2_rounds * (int)grid;
        if (debug_stage == 300)
            hipLaunchKernelGGL(dbh_timeline::dbh_forward_kernel, dim3(grid), dim3(dbh::kThreads),
                               0, stream, reinterpret_cast<dbh_timeline::ForwardArgs&>(a));
        else
            hipLaunchKernelGGL(dbh::dbh_forward_kernel, dim3(grid), dim3(dbh::kThreads), 0, stream,
                               a);
        DBH_HIP(hipGetLastError());
        if (ev_stop) {
            DBH_HIP(hipEventRecord(ev_stop, stream));
            ++m->events_used;                          // only closed brackets count
            m->timed_windows += m->open_windows;
            m->timed_launches += m->timing_span;
        }
    }
    return DBH_OK;
}

inline int steps_for(int scan_size) { return scan_size / (dbh::kWindow / 2); }

}  // namespace

extern "C" {

const char* dbh_version(void) { return "deepbinner_hip 0.1 (gfx950)"; }

const char* dbh_status_string(int status) {
    switch (status) {
        case DBH_OK: return "ok";
        case DBH_ERR_INVALID_ARGUMENT: return "invalid argument";
        case DBH_ERR_NO_DEVICE: return "no HIP device available";
        case DBH_ERR_HIP: return "HIP runtime error";
        case DBH_ERR_BAD_WEIGHTS: return "weight blob does not match the Deepbinner architecture";
        case DBH_ERR_UNSUPPORTED: return "unsupported model geometry";
        case DBH_ERR_OUT_OF_MEMORY: return "out of device memory";
        case DBH_ERR_COMM: return "multi-device exchange failed";
        default: return "unknown status";
    }
}

const char* dbh_last_error(void) { return g_last_error.c_str(); }

int dbh_device_count(int* count) {
    if (!count) return DBH_ERR_INVALID_ARGUMENT;
    int n = 0;
    // A host thread that waits for the GPU sleeps instead of spinning: the hosts this runs on
    // have few cores per GPU (profiles/r03_cpu_capacity.txt) and the loader wants them.  Must be
    // said before the runtime creates its context; DEEPBINNER_SPIN_WAIT=1 keeps the default.
    static const bool once = [] {
        const char* spin = std::getenv("DEEPBINNER_SPIN_WAIT");
        if (!(spin && spin[0] == '1')) {
            int devices = 0;
            if (hipGetDeviceCount(&devices) == hipSuccess)
                for (int d = 0; d < devices; ++d)
                    if (hipSetDevice(d) == hipSuccess)
                        (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
            if (devices > 0) (void)hipSetDevice(0);
            (void)hipGetLastError();
        }
        return true;
    }();
    (void)once;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        hip_fail(e, "hipGetDeviceCount");
        return DBH_ERR_NO_DEVICE;
    }
    *count = n;
    return n > 0 ? DBH_OK : DBH_ERR_NO_DEVICE;
}

int dbh_set_device(int ordinal) {
    DBH_HIP(hipSetDevice(ordinal));
    return DBH_OK;
}

int dbh_get_device(int* ordinal) {
    if (!ordinal) return DBH_ERR_INVALID_ARGUMENT;
    DBH_HIP(hipGetDevice(ordinal));
    return DBH_OK;
}

int dbh_device_name(int ordinal, char* buf, int buf_len) {
    if (!buf || buf_len <= 0) return DBH_ERR_INVALID_ARGUMENT;
    hipDeviceProp_t prop;
    DBH_HIP(hipGetDeviceProperties(&prop, ordinal));
    std::snprintf(buf, (size_t)buf_len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName,
                  prop.multiProcessorCount);
    return DBH_OK;
}

int dbh_device_synchronize(void) {
    DBH_HIP(hipDeviceSynchronize());
    return DBH_OK;
}

int dbh_malloc(void** dev_ptr, size_t bytes) {
    if (!dev_ptr) return DBH_ERR_INVALID_ARGUMENT;
    DBH_HIP(hipMalloc(dev_ptr, bytes ? bytes : 1));
    return DBH_OK;
}
int dbh_free(void* dev_ptr) {
    if (dev_ptr) DBH_HIP(hipFree(dev_ptr));
    return DBH_OK;
}
int dbh_malloc_host(void** host_ptr, size_t bytes) {
    if (!host_ptr) return DBH_ERR_INVALID_ARGUMENT;
    DBH_HIP(hipHostMalloc(host_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return DBH_OK;
}
int dbh_free_host(void* host_ptr) {
    if (host_ptr) DBH_HIP(hipHostFree(host_ptr));
    return DBH_OK;
}
int dbh_host_device_pointer(void* host_ptr, void** dev_ptr) {
    if (!host_ptr || !dev_ptr) return DBH_ERR_INVALID_ARGUMENT;
    DBH_HIP(hipHostGetDevicePointer(dev_ptr, host_ptr, 0));
    return DBH_OK;
}
int dbh_memcpy_h2d(void* dst, const void* src, size_t bytes, dbh_stream stream) {
    DBH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return DBH_OK;
}
int dbh_memcpy_d2h(void* dst, const void* src, size_t bytes, dbh_stream stream) {
    DBH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return DBH_OK;
}
int dbh_memcpy_d2d(void* dst, const void* src, size_t bytes, dbh_stream stream) {
    DBH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DBH_OK;
}
int dbh_stream_create(dbh_stream* stream) {
    if (!stream) return DBH_ERR_INVALID_ARGUMENT;
    hipStream_t s;
    DBH_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (dbh_stream)s;
    return DBH_OK;
}
int dbh_stream_destroy(dbh_stream stream) {
    DBH_HIP(hipStreamDestroy((hipStream_t)stream));
    return DBH_OK;
}
int dbh_stream_synchronize(dbh_stream stream) {
    DBH_HIP(hipStreamSynchronize((hipStream_t)stream));
    return DBH_OK;
}
int dbh_event_create(dbh_event* event) {
    if (!event) return DBH_ERR_INVALID_ARGUMENT;
    hipEvent_t e;
    DBH_HIP(hipEventCreate(&e));
    *event = (dbh_event)e;
    return DBH_OK;
}
int dbh_event_destroy(dbh_event event) {
    DBH_HIP(hipEventDestroy((hipEvent_t)event));
    return DBH_OK;
}
int dbh_event_record(dbh_event event, dbh_stream stream) {
    DBH_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return DBH_OK;
}
int dbh_event_synchronize(dbh_event event) {
    DBH_HIP(hipEventSynchronize((hipEvent_t)event));
    return DBH_OK;
}
int dbh_stream_wait_event(dbh_stream stream, dbh_event event) {
    DBH_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
    return DBH_OK;
}
int dbh_event_elapsed_ms(dbh_event start, dbh_event stop, float* ms) {
    if (!ms) return DBH_ERR_INVALID_ARGUMENT;
    DBH_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return DBH_OK;
}

int dbh_model_create(const float* weights, int64_t n_floats, int n_classes, int input_size,
                     dbh_model** model) {
    if (!weights || !model) return DBH_ERR_INVALID_ARGUMENT;
    *model = nullptr;
    if (input_size != dbh::kWindow || n_classes < 2 || n_classes > dbh::kMaxClasses)
        return DBH_ERR_UNSUPPORTED;
    if (n_floats != canon_param_count(n_classes)) return DBH_ERR_BAD_WEIGHTS;
    int count = 0;
    int st = dbh_device_count(&count);
    if (st != DBH_OK) return st;
    std::vector<float> packed;
    pack_weights(weights, n_classes, packed);
    dbh_model* m = new (std::nothrow) dbh_model();
    if (!m) return DBH_ERR_OUT_OF_MEMORY;
    m->n_classes = n_classes;
    hipError_t e = hipGetDevice(&m->device);
    if (e == hipSuccess) {
        int cus = 0;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, m->device);
        if (e == hipSuccess && cus > 0) m->cus = cus;
        // DEEPBINNER_GRID_CAP=n: at most n workgroups per launch (tests: long shares of windows
        // per workgroup - prefetch across windows, full and partial tail batches - out of a few
        // hundred windows)
        if (const char* cap = std::getenv("DEEPBINNER_GRID_CAP")) {
            const int c = std::atoi(cap);
            if (c > 0 && c < m->cus) m->cus = c;
        }
        m->cus_total = m->cus;
    }
    {
        const char* knob = std::getenv("DEEPBINNER_LAUNCH_PER_BATCH");
        m->launch_per_batch = knob && knob[0] == '1';
        const char* zc = std::getenv("DEEPBINNER_HOST_ZERO_COPY");
        m->host_zero_copy = !(zc && zc[0] == '0');
        const char* sw = std::getenv("DEEPBINNER_STATIC_WINDOWS");
        m->windows_by_counter = !(sw && sw[0] == '1');
        if (const char* c4 = std::getenv("DEEPBINNER_CHUNK4_ROUNDS")) m->chunk4_rounds = std::atoi(c4);
        if (const char* c2 = std::getenv("DEEPBINNER_CHUNK2_ROUNDS")) m->chunk2_rounds = std::atoi(c2);
    }
    if (e == hipSuccess) e = hipMalloc((void**)&m->d_packed, packed.size() * sizeof(float));
    if (e == hipSuccess)
        e = hipMemcpy(m->d_packed, packed.data(), packed.size() * sizeof(float),
                      hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (m->d_packed) (void)hipFree(m->d_packed);
        delete m;
        return hip_fail(e, "dbh_model_create");
    }
    *model = m;
    return DBH_OK;
}

int dbh_model_destroy(dbh_model* m) {
    if (!m) return DBH_OK;
    if (m->d_packed) (void)hipFree(m->d_packed);
    if (m->d_in) (void)hipFree(m->d_in);
    if (m->d_work) (void)hipFree(m->d_work);
    if (m->d_out) (void)hipFree(m->d_out);
    for (auto& t : m->tails)
        if (t.ptr) (void)hipFree(t.ptr);
    if (m->d_tail) (void)hipFree(m->d_tail);
    if (m->d_clock) (void)hipFree(m->d_clock);
    for (auto& ev : m->events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    {
        dbh_model::Deflated& d = m->deflated;
        if (d.stream) (void)hipStreamDestroy(d.stream);
        if (d.inflated) (void)hipEventDestroy(d.inflated);
        if (d.classified) (void)hipEventDestroy(d.classified);
        if (d.h_small) (void)hipHostFree(d.h_small);
        if (d.h_comp) (void)hipHostFree(d.h_comp);
        for (void* p : {d.d_comp, d.d_small, d.d_samples, d.d_tokens, d.d_out, d.d_work, d.d_tail})
            if (p) (void)hipFree(p);
    }
    for (auto& sl : m->slot) {
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
        if (sl.h_in) (void)hipHostFree(sl.h_in);
        if (sl.h_out) (void)hipHostFree(sl.h_out);
        if (sl.d_in) (void)hipFree(sl.d_in);
        if (sl.d_out) (void)hipFree(sl.d_out);
        if (sl.d_work) (void)hipFree(sl.d_work);
        if (sl.d_tail) (void)hipFree(sl.d_tail);
    }
    delete m;
    return DBH_OK;
}

int dbh_model_input_size(const dbh_model* m, int* input_size) {
    if (!m || !input_size) return DBH_ERR_INVALID_ARGUMENT;
    *input_size = dbh::kWindow;
    return DBH_OK;
}
int dbh_model_output_size(const dbh_model* m, int* n_classes) {
    if (!m || !n_classes) return DBH_ERR_INVALID_ARGUMENT;
    *n_classes = m->n_classes;
    return DBH_OK;
}

int dbh_predict_dev(dbh_model* m, const float* x_dev, int64_t n, float* probs_dev,
                    dbh_stream stream) {
    if (!m || n < 0 || (n > 0 && (!x_dev || !probs_dev))) return DBH_ERR_INVALID_ARGUMENT;
    return launch_forward(m, x_dev, n, probs_dev, -1, nullptr, (hipStream_t)stream);
}

int dbh_predict(dbh_model* m, const float* x_host, int64_t n, float* probs_host) {
    if (!m || n < 0 || (n > 0 && (!x_host || !probs_host))) return DBH_ERR_INVALID_ARGUMENT;
    if (n == 0) return DBH_OK;
    DBH_HIP(hipSetDevice(m->device));      // (see dbh_classify_i16)
    // bounded staging: 65,536 windows (256 MiB of fp32 input) per round trip
    const int64_t kChunk = 65536;
    const int64_t cap = n < kChunk ? n : kChunk;
    int st = ensure(&m->d_in, &m->in_bytes, (size_t)cap * dbh::kWindow * sizeof(float));
    if (st != DBH_OK) return st;
    st = ensure(&m->d_out, &m->out_bytes, (size_t)cap * m->n_classes * sizeof(float));
    if (st != DBH_OK) return st;
    for (int64_t off = 0; off < n; off += kChunk) {
        const int64_t cnt = (n - off < kChunk) ? (n - off) : kChunk;
        DBH_HIP(hipMemcpyAsync(m->d_in, x_host + off * dbh::kWindow,
                               (size_t)cnt * dbh::kWindow * sizeof(float), hipMemcpyHostToDevice, 0));
        st = launch_forward(m, (const float*)m->d_in, cnt, (float*)m->d_out, -1, nullptr, 0);
        if (st != DBH_OK) return st;
        DBH_HIP(hipMemcpyAsync(probs_host + off * m->n_classes, m->d_out,
                               (size_t)cnt * m->n_classes * sizeof(float), hipMemcpyDeviceToHost, 0));
        DBH_HIP(hipStreamSynchronize(0));
    }
    return DBH_OK;
}

int dbh_normalise_windows_dev(const int16_t* samples_dev, const int64_t* offsets_dev,
                              int64_t n_reads, int side, int scan_size, float* windows_dev,
                              dbh_stream stream) {
    const int steps = steps_for(scan_size);
    if (n_reads < 0 || steps <= 0 || steps * (dbh::kWindow / 2) != scan_size ||
        (side != DBH_SIDE_START && side != DBH_SIDE_END))
        return DBH_ERR_INVALID_ARGUMENT;
    if (n_reads == 0) return DBH_OK;
    if (!samples_dev || !offsets_dev || !windows_dev) return DBH_ERR_INVALID_ARGUMENT;
    const int64_t kChunkReads = (1 << 24) / steps;
    for (int64_t r0 = 0; r0 < n_reads; r0 += kChunkReads) {
        const int64_t cnt = (n_reads - r0 < kChunkReads) ? (n_reads - r0) : kChunkReads;
        hipLaunchKernelGGL(dbh::dbh_normalise_kernel, dim3((unsigned)(cnt * steps)), dim3(256), 0,
                           (hipStream_t)stream, samples_dev,
                           (const long long*)(offsets_dev + r0), steps, side,
                           windows_dev + r0 * steps * dbh::kWindow);
        DBH_HIP(hipGetLastError());
    }
    return DBH_OK;
}

int dbh_merge_calls_dev(const float* window_probs_dev, int64_t n_reads, int steps, int n_classes,
                        double score_diff, float* probs_dev, int32_t* calls_dev,
                        dbh_stream stream) {
    if (n_reads < 0 || steps <= 0 || n_classes < 2 || n_classes > dbh::kMaxClasses)
        return DBH_ERR_INVALID_ARGUMENT;
    if (n_reads == 0) return DBH_OK;
    if (!window_probs_dev || !probs_dev || !calls_dev) return DBH_ERR_INVALID_ARGUMENT;
    const int64_t blocks = (n_reads + 7) / 8;
    hipLaunchKernelGGL(dbh::dbh_merge_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, window_probs_dev, (long long)n_reads, steps,
                       n_classes, score_diff, probs_dev, (int*)calls_dev);
    DBH_HIP(hipGetLastError());
    return DBH_OK;
}

namespace {
// classify.py:298-322 on call numbers (0 = 'none'); one read per thread, 12 bytes of traffic each
__global__ void combine_calls_kernel(const int32_t* __restrict__ start_calls,
                                     const int32_t* __restrict__ end_calls, long long n, int mode,
                                     int32_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t s = start_calls[i], e = end_calls[i];
    int32_t call;
    if (s == e) call = s;
    else if (mode == DBH_REQUIRE_BOTH) call = DBH_CALL_NONE;
    else if (e == DBH_CALL_NONE) call = s;
    else if (mode == DBH_REQUIRE_START) call = DBH_CALL_NONE;
    else call = (s == DBH_CALL_NONE) ? e : DBH_CALL_NONE;
    out[i] = call;
}
}  // namespace

int dbh_combine_calls_dev(const int32_t* start_calls_dev, const int32_t* end_calls_dev,
                          int64_t n_reads, int mode, int32_t* out_dev, dbh_stream stream) {
    if (n_reads < 0 || mode < DBH_REQUIRE_EITHER || mode > DBH_REQUIRE_BOTH)
        return DBH_ERR_INVALID_ARGUMENT;
    if (n_reads == 0) return DBH_OK;
    if (!start_calls_dev || !end_calls_dev || !out_dev) return DBH_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(combine_calls_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, start_calls_dev, end_calls_dev, (long long)n_reads,
                       mode, out_dev);
    DBH_HIP(hipGetLastError());
    return DBH_OK;
}

int dbh_classify_workspace_bytes(const dbh_model* m, int64_t n_reads, int scan_size,
                                 size_t* bytes) {
    if (!m || !bytes || n_reads < 0) return DBH_ERR_INVALID_ARGUMENT;
    const int steps = steps_for(scan_size);
    if (steps <= 0 || steps * (dbh::kWindow / 2) != scan_size) return DBH_ERR_INVALID_ARGUMENT;
    // the per-window probabilities the merge kernel reads; one scan step needs nothing (the
    // forward kernel finishes the read itself), and the windows themselves never leave LDS
    const size_t windows = (size_t)n_reads * steps;
    *bytes = steps == 1 ? 0 : windows * m->n_classes * sizeof(float) + 256;
    return DBH_OK;
}

}  // extern "C"

namespace {
int classify_i16_dev(dbh_model* m, const int16_t* samples_dev, const int64_t* offsets_dev,
                     int64_t n_reads, int side, int scan_size, double score_diff,
                     float* probs_dev, int32_t* calls_dev, void* workspace_dev, dbh_stream stream,
                     int64_t read0, int64_t len_hint, int64_t hint_cap, void** tail = nullptr,
                     size_t* tail_bytes = nullptr) {
    if (!m || n_reads < 0) return DBH_ERR_INVALID_ARGUMENT;
    if (n_reads == 0) return DBH_OK;
    const int steps = steps_for(scan_size);
    if (steps <= 0 || steps * (dbh::kWindow / 2) != scan_size ||
        (side != DBH_SIDE_START && side != DBH_SIDE_END) || !samples_dev || !offsets_dev ||
        !probs_dev || !calls_dev)
        return DBH_ERR_INVALID_ARGUMENT;
    FusedInput in;
    in.samples = samples_dev;
    in.offsets = offsets_dev;
    in.steps = steps;
    in.side = side;
    in.score_diff = score_diff;
    in.read0 = read0;
    in.len_hint = len_hint;
    in.hint_cap = hint_cap;
    in.tail = tail;
    in.tail_bytes = tail_bytes;
    if (steps == 1) {
        // one window per read: slice + normalise + CNN + renormalise + call in ONE launch
        in.calls = calls_dev;
        return launch_forward(m, nullptr, n_reads, probs_dev, -1, nullptr, (hipStream_t)stream, in);
    }
    // several scan steps per read: the CNN kernel slices and normalises its own windows and
    // leaves per-window probabilities in the workspace; the merge kernel finishes each read
    if (!workspace_dev) return DBH_ERR_INVALID_ARGUMENT;
    float* wprobs = (float*)workspace_dev;
    int st = launch_forward(m, nullptr, n_reads * steps, wprobs, -1, nullptr, (hipStream_t)stream,
                            in);
    if (st != DBH_OK) return st;
    return dbh_merge_calls_dev(wprobs, n_reads, steps, m->n_classes, score_diff, probs_dev,
                               calls_dev, stream);
}
}  // namespace

extern "C" {

int dbh_model_set_read_length_hint(dbh_model* m, int64_t read_length, int64_t capacity_samples) {
    if (!m || read_length < 0 || capacity_samples < 0) return DBH_ERR_INVALID_ARGUMENT;
    m->hint_len = read_length;
    m->hint_cap = read_length > 0 ? capacity_samples : 0;
    return DBH_OK;
}

int dbh_classify_i16_dev(dbh_model* m, const int16_t* samples_dev, const int64_t* offsets_dev,
                         int64_t n_reads, int side, int scan_size, double score_diff,
                         float* probs_dev, int32_t* calls_dev, void* workspace_dev,
                         dbh_stream stream) {
    if (!m) return DBH_ERR_INVALID_ARGUMENT;
    return classify_i16_dev(m, samples_dev, offsets_dev, n_reads, side, scan_size, score_diff,
                            probs_dev, calls_dev, workspace_dev, stream, 0, m->hint_len,
                            m->hint_cap);
}

int dbh_classify_i16_batched_dev(dbh_model* m, const int16_t* samples_dev,
                                 const int64_t* offsets_dev, int64_t n_reads, int batch_size,
                                 int side, int scan_size, double score_diff, float* probs_dev,
                                 int32_t* calls_dev, dbh_stream stream) {
    if (!m || n_reads < 0 || batch_size <= 0) return DBH_ERR_INVALID_ARGUMENT;
    if (n_reads == 0) return DBH_OK;
    // The reference walks its reads batch by batch because model.predict does (classify.py:130,
    // 361); a read's result does not depend on its batch.  Here the forward kernel is
    // persistent - one workgroup per CU walking windows in read order - so the batches flow
    // through ONE launch without a boundary between them (DEEPBINNER_LAUNCH_PER_BATCH=1 brings
    // the launch per batch back, for comparison).  Reads with several scan steps go in chunks
    // whose per-window probabilities fit a bounded workspace, one merge launch per chunk.
    const int steps = steps_for(scan_size);
    if (steps <= 0) return DBH_ERR_INVALID_ARGUMENT;
    const int64_t chunk = m->launch_per_batch ? (int64_t)batch_size
                                              : (steps == 1 ? n_reads : ((int64_t)1 << 20) / steps);
    size_t work = 0;
    int st = dbh_classify_workspace_bytes(m, chunk < n_reads ? chunk : n_reads, scan_size, &work);
    if (st != DBH_OK) return st;
    if (work) {
        st = ensure(&m->d_work, &m->work_bytes, work);
        if (st != DBH_OK) return st;
    }
    // in-order on the caller's stream, so the workspace can be reused chunk after chunk
    for (int64_t r0 = 0; r0 < n_reads; r0 += chunk) {
        const int64_t cnt = (n_reads - r0 < chunk) ? (n_reads - r0) : chunk;
        st = classify_i16_dev(m, samples_dev, offsets_dev + r0, cnt, side, scan_size, score_diff,
                              probs_dev + r0 * m->n_classes, calls_dev + r0, m->d_work, stream, r0,
                              m->hint_len, m->hint_cap);
        if (st != DBH_OK) return st;
    }
    return DBH_OK;
}

}  // extern "C"

namespace {

// Is [p, p + bytes) host memory the GPU's DMA engine can read in place (hipHostMalloc /
// hipHostRegister: what dbh_malloc_host and dbh_host_alloc hand out)?
bool is_pinned(const void* p, size_t bytes) {
    if (!p || bytes == 0) return false;
    hipPointerAttribute_t a0, a1;
    if (hipPointerGetAttributes(&a0, p) != hipSuccess ||
        hipPointerGetAttributes(&a1, (const char*)p + bytes - 1) != hipSuccess) {
        (void)hipGetLastError();      // "not a registered pointer" is an answer, not a failure
        return false;
    }
    return a0.type == hipMemoryTypeHost && a1.type == hipMemoryTypeHost;
}

// Pageable memory -> a pinned staging slot.  One thread moves ~10 GB/s; PCIe takes 50+.  Large
// copies are cut up between the calling thread and a few helpers.
void staged_copy(void* dst, const void* src, size_t bytes) {
    constexpr size_t kPiece = 8u << 20;
    const int helpers = bytes >= 4 * kPiece ? 3 : (bytes >= 2 * kPiece ? 1 : 0);
    if (helpers == 0) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t share = ((bytes / (size_t)(helpers + 1)) + 4095) & ~(size_t)4095;
    std::vector<std::thread> team;
    for (int t = 1; t <= helpers; ++t) {
        const size_t lo = std::min(bytes, share * (size_t)t), hi = std::min(bytes, lo + share);
        if (hi > lo)
            team.emplace_back([=] { std::memcpy((char*)dst + lo, (const char*)src + lo, hi - lo); });
    }
    std::memcpy(dst, src, std::min(bytes, share));
    for (std::thread& t : team) t.join();
}

struct HostJob {
    dbh_model* model[2] = {nullptr, nullptr};      // start model, end model (either may be null)
    int side[2] = {DBH_SIDE_START, DBH_SIDE_END};
    float* probs_host[2] = {nullptr, nullptr};     // per model: n_reads x C, or null
    int32_t* calls_host[2] = {nullptr, nullptr};   // per model: n_reads, or null
    int32_t* final_host = nullptr;                 // combine_calls of the two, or null
    int combine_mode = DBH_REQUIRE_EITHER;
};

// The host-buffer pipeline behind dbh_classify_i16 and dbh_classify_pair_i16.  Reads travel in
// groups of at most ~32k windows per model through kSlots staging slots, each with its own
// stream: while the GPU works on group g, group g+1 is uploaded (straight from the caller's
// buffer when that is pinned, through a pinned staging copy otherwise) and the results of group
// g-1 come back.  Both models read the SAME uploaded samples; their kernels, the merge kernels and
// the combine kernel of a group follow each other on the group's stream.
int classify_host(const HostJob& job, const int16_t* samples_host, const int64_t* offsets_host,
                  int64_t n_reads, int scan_size, double score_diff) {
    dbh_model* m = job.model[0] ? job.model[0] : job.model[1];      // owns the slots
    if (!m || n_reads < 0) return DBH_ERR_INVALID_ARGUMENT;
    if (n_reads == 0) return DBH_OK;
    if (!offsets_host) return DBH_ERR_INVALID_ARGUMENT;
    const int steps = steps_for(scan_size);
    if (steps <= 0 || steps * (dbh::kWindow / 2) != scan_size) return DBH_ERR_INVALID_ARGUMENT;
    const bool both = job.model[0] && job.model[1];
    if (both && (job.model[0]->device != job.model[1]->device ||
                 job.model[0]->n_classes != job.model[1]->n_classes))
        return DBH_ERR_INVALID_ARGUMENT;
    if (job.final_host && !both) return DBH_ERR_INVALID_ARGUMENT;
    // the current device is a per-thread setting: a caller on another thread than the one that
    // created the model gets the model's device
    DBH_HIP(hipSetDevice(m->device));
    const int C = m->n_classes;
    const int64_t total_samples = offsets_host[n_reads] - offsets_host[0];
    if (total_samples < 0 || (total_samples > 0 && !samples_host)) return DBH_ERR_INVALID_ARGUMENT;
    const bool pinned = total_samples > 0 &&
                        is_pinned(samples_host + offsets_host[0], (size_t)total_samples * 2);
    const int64_t group = std::max<int64_t>(1, m->host_group_windows / steps);
    constexpr int kSlots = dbh_model::kSlots;
    struct Pending { int64_t r0 = 0, cnt = 0; bool live = false; } pending[kSlots];
    // what comes back per read of a group, in this order
    const size_t probs_bytes = (size_t)C * sizeof(float);
    auto out_layout = [&](int64_t cnt, size_t (&off)[5]) -> size_t {
        size_t at = 0;
        for (int k = 0; k < 2; ++k) {
            off[k] = at;
            if (job.model[k]) at += (size_t)cnt * probs_bytes;
        }
        for (int k = 0; k < 2; ++k) {
            off[2 + k] = at;
            if (job.model[k]) at += (size_t)cnt * sizeof(int32_t);
        }
        off[4] = at;
        if (both) at += (size_t)cnt * sizeof(int32_t);
        return at;
    };
    auto drain = [&](int k) -> int {
        dbh_model::Slot& sl = m->slot[k];
        if (!pending[k].live) return DBH_OK;
        DBH_HIP(hipStreamSynchronize(sl.stream));
        const int64_t cnt = pending[k].cnt, r0 = pending[k].r0;
        size_t off[5];
        out_layout(cnt, off);
        const char* h = (const char*)sl.h_out;
        for (int j = 0; j < 2; ++j) {
            if (job.model[j] && job.probs_host[j])
                std::memcpy(job.probs_host[j] + r0 * C, h + off[j], (size_t)cnt * probs_bytes);
            if (job.model[j] && job.calls_host[j])
                std::memcpy(job.calls_host[j] + r0, h + off[2 + j], (size_t)cnt * sizeof(int32_t));
        }
        if (both && job.final_host)
            std::memcpy(job.final_host + r0, h + off[4], (size_t)cnt * sizeof(int32_t));
        pending[k].live = false;
        return DBH_OK;
    };
    // an error in the middle of the loop must not leave another slot's copies in flight: its
    // pinned buffers belong to the next call
    auto fail = [&](int status) -> int {
        for (auto& sl : m->slot)
            if (sl.stream) (void)hipStreamSynchronize(sl.stream);
        return status;
    };
    int64_t g = 0;
    for (int64_t r0 = 0; r0 < n_reads; r0 += group, ++g) {
        const int k = (int)(g % kSlots);
        dbh_model::Slot& sl = m->slot[k];
        int st = drain(k);
        if (st != DBH_OK) return fail(st);
        if (!sl.stream) {
            hipError_t e = hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking);
            if (e != hipSuccess) return fail(hip_fail(e, "hipStreamCreateWithFlags"));
        }
        const int64_t cnt = (n_reads - r0 < group) ? (n_reads - r0) : group;
        const int64_t s0 = offsets_host[r0], s1 = offsets_host[r0 + cnt];
        if (s1 < s0) return fail(DBH_ERR_INVALID_ARGUMENT);
        const size_t raw_bytes = (size_t)(s1 - s0) * sizeof(int16_t);
        const size_t sample_bytes = (raw_bytes + 255) & ~(size_t)255;
        const size_t off_bytes = (size_t)(cnt + 1) * sizeof(int64_t);
        size_t out_off[5];
        const size_t out_bytes = out_layout(cnt, out_off);
        size_t work = 0;
        st = dbh_classify_workspace_bytes(m, cnt, scan_size, &work);
        // host side: the relative offsets always, the samples only when they have to be staged
        if (st == DBH_OK)
            st = ensure_host(&sl.h_in, &sl.h_in_bytes, off_bytes + (pinned ? 0 : sample_bytes));
        if (st == DBH_OK) st = ensure_host(&sl.h_out, &sl.h_out_bytes, out_bytes);
        if (st == DBH_OK) st = ensure(&sl.d_in, &sl.d_in_bytes, sample_bytes + off_bytes);
        if (st == DBH_OK) st = ensure(&sl.d_out, &sl.d_out_bytes, out_bytes);
        if (st == DBH_OK && work) st = ensure(&sl.d_work, &sl.d_work_bytes, work);
        if (st != DBH_OK) return fail(st);

        int64_t* rel = (int64_t*)sl.h_in;
        for (int64_t i = 0; i <= cnt; ++i) rel[i] = offsets_host[r0 + i] - s0;
        // all reads of the group equally long?  then the kernel need not wait for the offsets
        int64_t uniform = rel[1];
        for (int64_t i = 1; i <= cnt && uniform > 0; ++i)
            if (rel[i] != i * uniform) uniform = 0;
        const void* src = samples_host + s0;
        if (raw_bytes && !pinned) {
            staged_copy((char*)sl.h_in + off_bytes, src, raw_bytes);
            src = (char*)sl.h_in + off_bytes;
        }
        hipError_t e = hipSuccess;
        // The samples reach the kernel either by a copy into HBM first, or not at all: the
        // forward kernel asks for a window's 2 KB ~25,000 cycles before it needs them, which
        // hides a PCIe round trip as well as an HBM one - so it can read pinned host memory in
        // place.  (A copy between two launches cannot overlap them here: the persistent kernel
        // fills every CU's registers and LDS, and the copy of the next group only starts when
        // it has drained - 1.2 ms of idle GPU per 6 ms group, profiles/r03_host_path_trace.txt.)
        const int16_t* d_samples = (const int16_t*)sl.d_in;
        if (raw_bytes && m->host_zero_copy) {
            void* mapped = nullptr;
            e = hipHostGetDevicePointer(&mapped, const_cast<void*>(src), 0);
            d_samples = (const int16_t*)mapped;
        } else if (raw_bytes) {
            e = hipMemcpyAsync(sl.d_in, src, raw_bytes, hipMemcpyHostToDevice, sl.stream);
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync((char*)sl.d_in + sample_bytes, rel, off_bytes, hipMemcpyHostToDevice,
                               sl.stream);
        if (e != hipSuccess) return fail(hip_fail(e, "hipMemcpyAsync H2D"));
        int32_t* d_calls[2] = {nullptr, nullptr};
        for (int j = 0; j < 2; ++j) {
            if (!job.model[j]) continue;
            d_calls[j] = (int32_t*)((char*)sl.d_out + out_off[2 + j]);
            st = classify_i16_dev(job.model[j], d_samples,
                                  (const int64_t*)((char*)sl.d_in + sample_bytes), cnt, job.side[j],
                                  scan_size, score_diff, (float*)((char*)sl.d_out + out_off[j]),
                                  d_calls[j], sl.d_work, (dbh_stream)sl.stream, 0, uniform, s1 - s0,
                                  &sl.d_tail, &sl.d_tail_bytes);
            if (st != DBH_OK) return fail(st);
        }
        if (both) {
            st = dbh_combine_calls_dev(d_calls[0], d_calls[1], cnt, job.combine_mode,
                                       (int32_t*)((char*)sl.d_out + out_off[4]),
                                       (dbh_stream)sl.stream);
            if (st != DBH_OK) return fail(st);
        }
        e = hipMemcpyAsync(sl.h_out, sl.d_out, out_bytes, hipMemcpyDeviceToHost, sl.stream);
        if (e != hipSuccess) return fail(hip_fail(e, "hipMemcpyAsync D2H"));
        pending[k].r0 = r0;
        pending[k].cnt = cnt;
        pending[k].live = true;
    }
    for (int i = 0; i < kSlots; ++i) {       // oldest first
        const int st = drain((int)((g + i) % kSlots));
        if (st != DBH_OK) return fail(st);
    }
    return DBH_OK;
}

}  // namespace

extern "C" {

int dbh_classify_i16(dbh_model* m, const int16_t* samples_host, const int64_t* offsets_host,
                     int64_t n_reads, int side, int scan_size, double score_diff,
                     float* probs_host, int32_t* calls_host) {
    if (!m || n_reads < 0) return DBH_ERR_INVALID_ARGUMENT;
    if (n_reads == 0) return DBH_OK;
    if (!offsets_host || !probs_host || !calls_host ||
        (side != DBH_SIDE_START && side != DBH_SIDE_END))
        return DBH_ERR_INVALID_ARGUMENT;
    HostJob job;
    job.model[0] = m;
    job.side[0] = side;
    job.probs_host[0] = probs_host;
    job.calls_host[0] = calls_host;
    return classify_host(job, samples_host, offsets_host, n_reads, scan_size, score_diff);
}

int dbh_classify_pair_i16(dbh_model* start_model, dbh_model* end_model,
                          const int16_t* samples_host, const int64_t* offsets_host,
                          int64_t n_reads, int scan_size, double score_diff, int combine_mode,
                          int32_t* calls_host, int32_t* start_calls_host, int32_t* end_calls_host,
                          float* start_probs_host, float* end_probs_host) {
    if ((!start_model && !end_model) || n_reads < 0 || combine_mode < DBH_REQUIRE_EITHER ||
        combine_mode > DBH_REQUIRE_BOTH)
        return DBH_ERR_INVALID_ARGUMENT;
    if (n_reads == 0) return DBH_OK;
    if (!offsets_host || !calls_host) return DBH_ERR_INVALID_ARGUMENT;
    HostJob job;
    job.model[0] = start_model;
    job.model[1] = end_model;
    job.probs_host[0] = start_probs_host;
    job.probs_host[1] = end_probs_host;
    job.calls_host[0] = start_calls_host;
    job.calls_host[1] = end_calls_host;
    job.combine_mode = combine_mode;
    if (start_model && end_model) job.final_host = calls_host;
    else if (start_model && !start_calls_host) job.calls_host[0] = calls_host;
    else if (end_model && !end_calls_host) job.calls_host[1] = calls_host;
    const int st = classify_host(job, samples_host, offsets_host, n_reads, scan_size, score_diff);
    // one model and the caller wanted its calls twice (final + per side)
    if (st == DBH_OK && !(start_model && end_model)) {
        int32_t* side_calls = start_model ? start_calls_host : end_calls_host;
        if (side_calls && side_calls != calls_host)
            std::memcpy(calls_host, side_calls, (size_t)n_reads * sizeof(int32_t));
    }
    return st;
}

namespace {
// THE FORWARD STREAM of a device: every queue's forward launches (and the small kernels between
// them) go through this one stream, in the order the queues arrive, so that at most one forward
// kernel is on the GPU at a time.  The forward kernel is persistent - one workgroup per CU for as
// long as the launch lasts: with every queue launching on its own stream two or three of them
// share the CUs, each stretched, the small kernels behind them (merge, combine, the next
// container's fill) wait for whichever ends last, and the queues fall into step: all inflating,
// then all classifying (profiles/r05_k2/steady_state_trace_share0.txt: 16 of every 68 ms with no
// forward kernel on the GPU).  One at a time on 256 - n CUs (dbh_model_reserve_cus), the inflate
// kernels of the containers behind keep the other n CUs busy all the time.
// DEEPBINNER_FORWARD_STREAM=own: every queue on its own stream again (A/B).
// (the streams live as long as the process - the queues of any model may use them - and are left to
// the runtime's teardown on purpose; a device ordinal of kMaxDevices or more keeps every queue on
// its own stream, which is correct, only slower)
constexpr int kMaxDevices = 64;
std::mutex g_forward_mutex[kMaxDevices];
hipStream_t g_forward_stream[kMaxDevices] = {};
bool shared_forward_stream() {
    const char* v = std::getenv("DEEPBINNER_FORWARD_STREAM");
    return !(v && std::strcmp(v, "own") == 0);
}
}  // namespace

int dbh_classify_pair_deflated(dbh_model* start_model, dbh_model* end_model,
                               const uint8_t* comp_host, int64_t comp_bytes,
                               const dbh_inflate_stream* streams_host, int64_t n_streams,
                               const int64_t* offsets_host, int64_t n_reads, int scan_size,
                               double score_diff, int combine_mode, int32_t* calls_host,
                               int32_t* stream_status_host, int16_t* samples_host,
                               double* stage_ms) {
    return dbh_classify_pair_deflated_verbose(start_model, end_model, comp_host, comp_bytes,
                                              streams_host, n_streams, offsets_host, n_reads,
                                              scan_size, score_diff, combine_mode, calls_host,
                                              stream_status_host, samples_host, stage_ms, nullptr,
                                              nullptr, nullptr, nullptr);
}

int dbh_classify_pair_deflated_verbose(dbh_model* start_model, dbh_model* end_model,
                                       const uint8_t* comp_host, int64_t comp_bytes,
                                       const dbh_inflate_stream* streams_host, int64_t n_streams,
                                       const int64_t* offsets_host, int64_t n_reads, int scan_size,
                                       double score_diff, int combine_mode, int32_t* calls_host,
                                       int32_t* stream_status_host, int16_t* samples_host,
                                       double* stage_ms, int32_t* start_calls_host,
                                       int32_t* end_calls_host, float* start_probs_host,
                                       float* end_probs_host) {
    dbh_model* m = start_model ? start_model : end_model;
    if (!m || n_reads < 0 || n_streams < 0 || comp_bytes < 0 ||
        combine_mode < DBH_REQUIRE_EITHER || combine_mode > DBH_REQUIRE_BOTH)
        return DBH_ERR_INVALID_ARGUMENT;
    if (n_reads == 0) return DBH_OK;
    if (!offsets_host || !calls_host || (n_streams > 0 && (!streams_host || !comp_host)))
        return DBH_ERR_INVALID_ARGUMENT;
    const bool both = start_model && end_model;
    if (both && (start_model->device != end_model->device ||
                 start_model->n_classes != end_model->n_classes))
        return DBH_ERR_INVALID_ARGUMENT;
    const int steps = steps_for(scan_size);
    if (steps <= 0 || steps * (dbh::kWindow / 2) != scan_size) return DBH_ERR_INVALID_ARGUMENT;
    const int64_t total_samples = offsets_host[n_reads] - offsets_host[0];
    if (offsets_host[0] != 0 || total_samples < 0) return DBH_ERR_INVALID_ARGUMENT;
    const int64_t out_bytes = total_samples * 2;
    for (int64_t i = 0; i < n_streams; ++i) {
        const dbh_inflate_stream& r = streams_host[i];
        if (r.comp_offset < 0 || r.comp_bytes < 0 || r.out_offset < 0 || r.out_bytes < 0 ||
            (r.out_offset & 1) || r.comp_offset + r.comp_bytes > comp_bytes ||
            r.out_offset + r.out_bytes > out_bytes)
            return DBH_ERR_INVALID_ARGUMENT;
    }
    DBH_HIP(hipSetDevice(m->device));
    dbh_model::Deflated& d = m->deflated;
    if (!d.stream) DBH_HIP(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    const int C = m->n_classes;
    // small things travel together: [records | offsets] in, [status | final calls | side calls] out
    const size_t rec_bytes = ((size_t)n_streams * sizeof(dbh_inflate_stream) + 255) & ~(size_t)255;
    const size_t off_bytes = ((size_t)(n_reads + 1) * sizeof(int64_t) + 255) & ~(size_t)255;
    const size_t status_bytes = ((size_t)n_streams * sizeof(int32_t) + 255) & ~(size_t)255;
    const size_t calls_bytes = ((size_t)n_reads * sizeof(int32_t) + 255) & ~(size_t)255;
    const size_t in_small = rec_bytes + off_bytes;
    const size_t out_small = status_bytes + 3 * calls_bytes;
    const size_t probs_bytes = (size_t)n_reads * C * sizeof(float);
    size_t token_bytes = 0, work = 0;
    int st = dbh_inflate_workspace_bytes(out_bytes, n_streams, &token_bytes);
    if (st == DBH_OK) st = dbh_classify_workspace_bytes(m, n_reads, scan_size, &work);
    if (st == DBH_OK) st = ensure_host(&d.h_small, &d.h_small_bytes, in_small + out_small);
    if (st == DBH_OK) st = ensure(&d.d_small, &d.d_small_bytes, in_small + out_small);
    if (st == DBH_OK) st = ensure(&d.d_comp, &d.d_comp_bytes, (size_t)comp_bytes + 64);
    if (st == DBH_OK) st = ensure(&d.d_samples, &d.d_samples_bytes, (size_t)out_bytes + 256);
    if (st == DBH_OK) st = ensure(&d.d_tokens, &d.d_tokens_bytes, token_bytes);
    if (st == DBH_OK) st = ensure(&d.d_out, &d.d_out_bytes, 2 * probs_bytes + 256);
    if (st == DBH_OK && work) st = ensure(&d.d_work, &d.d_work_bytes, work);
    if (st != DBH_OK) return st;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    if (stage_ms)
        for (hipEvent_t& e : ev) DBH_HIP(hipEventCreate(&e));
    auto done = [&](int status) -> int {
        (void)hipStreamSynchronize(d.stream);
        for (hipEvent_t e : ev)
            if (e) (void)hipEventDestroy(e);
        return status;
    };
    char* h = (char*)d.h_small;
    // the records go over longest stream first (a lane of the decoder takes streams off a counter
    // in this order: what is long starts early, what is short fills the gaps; streams that need no
    // decoding last); the status comes back in the same order and is put back below
    std::vector<int32_t>& order = d.order;
    order.resize((size_t)n_streams);
    for (int64_t i = 0; i < n_streams; ++i) order[(size_t)i] = (int32_t)i;
    std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        const dbh_inflate_stream &x = streams_host[a], &y = streams_host[b];
        const int64_t kx = x.mode == DBH_INFLATE_ZLIB ? x.comp_bytes : -1;
        const int64_t ky = y.mode == DBH_INFLATE_ZLIB ? y.comp_bytes : -1;
        return kx != ky ? kx > ky : a < b;
    });
    {
        dbh_inflate_stream* staged = (dbh_inflate_stream*)h;
        for (int64_t i = 0; i < n_streams; ++i) staged[i] = streams_host[order[(size_t)i]];
    }
    std::memcpy(h + rec_bytes, offsets_host, (size_t)(n_reads + 1) * sizeof(int64_t));
    char* ds = (char*)d.d_small;
    const dbh_inflate_stream* d_records = (const dbh_inflate_stream*)ds;
    const int64_t* d_offsets = (const int64_t*)(ds + rec_bytes);
    int32_t* d_status = (int32_t*)(ds + in_small);
    int32_t* d_final = (int32_t*)(ds + in_small + status_bytes);
    int32_t* d_side[2] = {(int32_t*)(ds + in_small + status_bytes + calls_bytes),
                          (int32_t*)(ds + in_small + status_bytes + 2 * calls_bytes)};
    if (ev[0]) (void)hipEventRecord(ev[0], d.stream);
    hipError_t e = hipMemcpyAsync(ds, h, in_small, hipMemcpyHostToDevice, d.stream);
    if (e == hipSuccess && comp_bytes > 0) {
        const void* src = comp_host;
        // the decoder fetches up to 64 bytes beyond the last stream: the loader's buffers carry
        // them; a pageable buffer is staged (and padded) first
        if (!is_pinned(comp_host, (size_t)comp_bytes + 64)) {
            st = ensure_host(&d.h_comp, &d.h_comp_bytes, (size_t)comp_bytes + 64);
            if (st != DBH_OK) return done(st);
            staged_copy(d.h_comp, comp_host, (size_t)comp_bytes);
            std::memset((char*)d.h_comp + comp_bytes, 0, 64);
            src = d.h_comp;
        }
        e = hipMemcpyAsync(d.d_comp, src, (size_t)comp_bytes + 64, hipMemcpyHostToDevice, d.stream);
    }
    if (e != hipSuccess) return done(hip_fail(e, "hipMemcpyAsync H2D"));
    if (ev[1]) (void)hipEventRecord(ev[1], d.stream);
    // a read without a piece (nothing stored, or nothing readable) must still be zeros
    if (out_bytes > 0) {
        e = hipMemsetAsync(d.d_samples, 0, (size_t)out_bytes, d.stream);
        if (e != hipSuccess) return done(hip_fail(e, "hipMemsetAsync"));
    }
    if (n_streams > 0) {
        // How wide kernel 1 is launched: it lasts as long as its longest stream whatever the
        // width, so the lanes may as well take as many streams one after the other as fit into
        // that time - this call is one of several in flight, and what it leaves free the others
        // use.  (A lane takes a new stream only at a block boundary, ~1.35 blocks to a mean
        // read: hence the margin.)
        int per_lane = m->inflate_streams_per_lane;
        if (per_lane <= 0) {
            int64_t sum = 0, count = 0;
            for (int64_t i = 0; i < n_streams; ++i)
                if (streams_host[i].mode == DBH_INFLATE_ZLIB) {
                    sum += streams_host[i].comp_bytes;
                    ++count;
                }
            const int64_t longest = count ? streams_host[order[0]].comp_bytes : 0;
            per_lane = sum > 0 ? (int)(longest * count * 2 / (sum * 3)) : 1;
            per_lane = per_lane < 1 ? 1 : per_lane > 8 ? 8 : per_lane;
        }
        st = dbh_inflate_dev((const uint8_t*)d.d_comp, (int64_t)comp_bytes, d_records, n_streams,
                             out_bytes, (uint8_t*)d.d_samples, d.d_tokens, d_status, per_lane,
                             (dbh_stream)d.stream);
        if (st != DBH_OK) return done(st);
    }
    if (ev[2]) (void)hipEventRecord(ev[2], d.stream);
    dbh_model* models[2] = {start_model, end_model};
    const int32_t* d_calls = both ? d_final : d_side[start_model ? 0 : 1];
    {
        // both models' launches and combine_calls: on the device's forward stream, as one block
        const bool shared = shared_forward_stream() && m->device >= 0 && m->device < kMaxDevices;
        std::unique_lock<std::mutex> block;
        hipStream_t on = d.stream;
        if (shared) {
            e = hipSuccess;
            if (!d.inflated) e = hipEventCreateWithFlags(&d.inflated, hipEventDisableTiming);
            if (e == hipSuccess && !d.classified)
                e = hipEventCreateWithFlags(&d.classified, hipEventDisableTiming);
            if (e != hipSuccess) return done(hip_fail(e, "hipEventCreateWithFlags"));
            // The forward kernel's per-workgroup scratch is sized BEFORE the device's lock is taken
            // (ADVICE round 5): growing it means hipFree / hipMalloc, which synchronise the device,
            // and under the lock that would stall every other queue's launch.  (Its first 256
            // bytes - the window counter - are zeroed here when it is new, on this queue's stream,
            // which the forward stream waits for below.)
            {
                void* before = d.d_tail;
                int cus = 0;
                for (int j = 0; j < 2; ++j)
                    if (models[j] && models[j]->cus > cus) cus = models[j]->cus;
                const int stt = ensure(&d.d_tail, &d.d_tail_bytes,
                                       256 + (size_t)cus * dbh::kWgScratchFloats * sizeof(float));
                if (stt != DBH_OK) return done(stt);
                if (d.d_tail != before) {
                    e = hipMemsetAsync(d.d_tail, 0, 256, d.stream);
                    if (e != hipSuccess) return done(hip_fail(e, "hipMemsetAsync"));
                }
            }
            block = std::unique_lock<std::mutex>(g_forward_mutex[m->device]);
            hipStream_t& fs = g_forward_stream[m->device];
            if (!fs) {
                int least = 0, greatest = 0;
                (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
                e = hipStreamCreateWithPriority(&fs, hipStreamNonBlocking, greatest);
                if (e != hipSuccess) return done(hip_fail(e, "hipStreamCreateWithPriority"));
            }
            on = fs;
            e = hipEventRecord(d.inflated, d.stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(on, d.inflated, 0);
            if (e != hipSuccess) return done(hip_fail(e, "handing over to the forward stream"));
        }
        for (int j = 0; j < 2; ++j) {
            if (!models[j]) continue;
            st = classify_i16_dev(models[j], (const int16_t*)d.d_samples, d_offsets, n_reads,
                                  j == 0 ? DBH_SIDE_START : DBH_SIDE_END, scan_size, score_diff,
                                  (float*)((char*)d.d_out + (size_t)j * probs_bytes), d_side[j],
                                  d.d_work, (dbh_stream)on, 0, 0, 0, &d.d_tail, &d.d_tail_bytes);
            if (st != DBH_OK) break;
        }
        if (st == DBH_OK && both)
            st = dbh_combine_calls_dev(d_side[0], d_side[1], n_reads, combine_mode, d_final,
                                       (dbh_stream)on);
        if (shared) {
            // (whatever was launched must be waited for, also behind an error)
            e = hipEventRecord(d.classified, on);
            if (e == hipSuccess) e = hipStreamWaitEvent(d.stream, d.classified, 0);
            if (e != hipSuccess) {
                block.unlock();        // (the wait below must not hold up the other queues' launches)
                (void)hipStreamSynchronize(on);
                return done(hip_fail(e, "taking over from the forward stream"));
            }
        }
        if (st != DBH_OK) return done(st);
    }
    if (ev[3]) (void)hipEventRecord(ev[3], d.stream);
    e = hipMemcpyAsync(h + in_small, ds + in_small, out_small, hipMemcpyDeviceToHost, d.stream);
    if (e == hipSuccess && samples_host && out_bytes > 0)
        e = hipMemcpyAsync(samples_host, d.d_samples, (size_t)out_bytes, hipMemcpyDeviceToHost,
                           d.stream);
    // what --verbose prints beside the final call (classify.py:157-171): the sides' probabilities
    float* side_probs_host[2] = {start_probs_host, end_probs_host};
    for (int j = 0; j < 2; ++j)
        if (e == hipSuccess && side_probs_host[j] && models[j])
            e = hipMemcpyAsync(side_probs_host[j], (const char*)d.d_out + (size_t)j * probs_bytes,
                               probs_bytes, hipMemcpyDeviceToHost, d.stream);
    if (e == hipSuccess) e = hipStreamSynchronize(d.stream);
    if (e != hipSuccess) return done(hip_fail(e, "dbh_classify_pair_deflated"));
    std::memcpy(calls_host, h + in_small + ((const char*)d_calls - (const char*)d_status),
                (size_t)n_reads * sizeof(int32_t));
    // ... and the sides' own calls
    int32_t* side_calls_host[2] = {start_calls_host, end_calls_host};
    for (int j = 0; j < 2; ++j)
        if (side_calls_host[j] && models[j])
            std::memcpy(side_calls_host[j],
                        h + in_small + ((const char*)d_side[j] - (const char*)d_status),
                        (size_t)n_reads * sizeof(int32_t));
    if (stream_status_host && n_streams) {
        const int32_t* sorted_status = (const int32_t*)(h + in_small);
        for (int64_t i = 0; i < n_streams; ++i)
            stream_status_host[order[(size_t)i]] = sorted_status[i];
    }
    if (stage_ms) {
        float ms = 0.f;
        for (int k = 0; k < 3; ++k) {
            stage_ms[k] = hipEventElapsedTime(&ms, ev[k], ev[k + 1]) == hipSuccess ? ms : -1.0;
        }
    }
    return done(DBH_OK);
}

void* dbh_host_alloc(size_t bytes, void* user) {
    (void)user;
    void* p = nullptr;
    // portable: the loader's threads allocate with no device of their own chosen, and with
    // several GPUs any of them may be the one that reads the batch
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocPortable) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void dbh_host_release(void* ptr, void* user) {
    (void)user;
    if (ptr) (void)hipHostFree(ptr);
}

int dbh_host_is_pinned(const void* ptr, size_t bytes, int* pinned) {
    if (!pinned) return DBH_ERR_INVALID_ARGUMENT;
    *pinned = is_pinned(ptr, bytes) ? 1 : 0;
    return DBH_OK;
}

int dbh_model_set_host_group(dbh_model* m, int64_t windows_per_group) {
    if (!m || windows_per_group < 0) return DBH_ERR_INVALID_ARGUMENT;
    m->host_group_windows = windows_per_group > 0 ? windows_per_group : dbh_model::kDefaultGroup;
    return DBH_OK;
}

int dbh_model_reserve_cus(dbh_model* m, int n_cus) {
    if (!m || n_cus < 0) return DBH_ERR_INVALID_ARGUMENT;
    m->cus = m->cus_total - n_cus > 1 ? m->cus_total - n_cus : 1;
    return DBH_OK;
}

int dbh_stage_floats(int stage, int64_t* floats_per_window) {
    if (!floats_per_window || stage < 0 || stage > 7) return DBH_ERR_INVALID_ARGUMENT;
    *floats_per_window = dbh::kStageFloats[stage];
    return DBH_OK;
}

int dbh_debug_forward(dbh_model* m, const float* x_host, int64_t n, int stage, float* out_host) {
    if (!m || n <= 0 || !x_host || !out_host || stage < 0 || stage > 7)
        return DBH_ERR_INVALID_ARGUMENT;
    const size_t per = (size_t)dbh::kStageFloats[stage];
    int st = ensure(&m->d_in, &m->in_bytes, (size_t)n * dbh::kWindow * sizeof(float));
    if (st != DBH_OK) return st;
    st = ensure(&m->d_work, &m->work_bytes, (size_t)n * per * sizeof(float));
    if (st != DBH_OK) return st;
    DBH_HIP(hipMemcpyAsync(m->d_in, x_host, (size_t)n * dbh::kWindow * sizeof(float),
                           hipMemcpyHostToDevice, 0));
    DBH_HIP(hipMemsetAsync(m->d_work, 0, (size_t)n * per * sizeof(float), 0));
    st = launch_forward(m, (const float*)m->d_in, n, nullptr, stage, (float*)m->d_work, 0);
    if (st != DBH_OK) return st;
    DBH_HIP(hipMemcpyAsync(out_host, m->d_work, (size_t)n * per * sizeof(float),
                           hipMemcpyDeviceToHost, 0));
    DBH_HIP(hipStreamSynchronize(0));
    return DBH_OK;
}

int dbh_forward_kernel_info(int* threads_per_block, int* lds_bytes, int* vgprs) {
    hipFuncAttributes attr;
    DBH_HIP(hipFuncGetAttributes(&attr, (const void*)dbh::dbh_forward_kernel));
    if (threads_per_block) *threads_per_block = dbh::kThreads;
    if (lds_bytes) *lds_bytes = (int)attr.sharedSizeBytes;
    if (vgprs) *vgprs = attr.numRegs;
    return DBH_OK;
}

int dbh_forward_executed_mfmas(int n_classes, int64_t* mfmas_per_window,
                                int64_t* flop_per_window) {
    if (n_classes < 2 || n_classes > dbh::kMaxClasses) return DBH_ERR_INVALID_ARGUMENT;
    const int64_t n = dbh::forward_mfmas(n_classes);
    if (mfmas_per_window) *mfmas_per_window = n;
    if (flop_per_window) *flop_per_window = n * 2048;      // 16 x 16 x 4 multiply-adds
    return DBH_OK;
}

int dbh_forward_truncated_dev(dbh_model* m, const float* x_dev, int64_t n, int last_stage,
                              dbh_stream stream) {
    if (!m || n <= 0 || !x_dev || last_stage < 0 || last_stage > 6) return DBH_ERR_INVALID_ARGUMENT;
    return launch_forward(m, x_dev, n, nullptr, 100 + last_stage, nullptr, (hipStream_t)stream);
}

int dbh_forward_timeline(dbh_model* m, const float* x_host, int64_t n, int64_t* stamps_host) {
    if (!m || n <= 0 || !x_host || !stamps_host) return DBH_ERR_INVALID_ARGUMENT;
    const size_t stamp_bytes = (size_t)n * dbh::kWaves * 64 * sizeof(int64_t);
    int st = ensure(&m->d_in, &m->in_bytes, (size_t)n * dbh::kWindow * sizeof(float));
    if (st != DBH_OK) return st;
    st = ensure(&m->d_work, &m->work_bytes, stamp_bytes);
    if (st != DBH_OK) return st;
    st = ensure(&m->d_out, &m->out_bytes, (size_t)n * m->n_classes * sizeof(float));
    if (st != DBH_OK) return st;
    DBH_HIP(hipMemcpyAsync(m->d_in, x_host, (size_t)n * dbh::kWindow * sizeof(float),
                           hipMemcpyHostToDevice, 0));
    DBH_HIP(hipMemsetAsync(m->d_work, 0, stamp_bytes, 0));
    {
        dbh_timeline::ForwardArgs a = {};
        a.packed = m->d_packed;
        a.x = (const float*)m->d_in;
        a.probs = (float*)m->d_out;
        a.debug_out = (float*)m->d_work;
        a.n_windows = (long long)n;
        a.n_classes = m->n_classes;
        a.debug_stage = 300;
        a.steps = 1;
        const unsigned grid = (unsigned)((n + dbh::kGroup - 1) / dbh::kGroup);
        st = ensure(&m->d_tail, &m->tail_bytes, (size_t)grid * dbh::kWgScratchFloats * sizeof(float));
        if (st != DBH_OK) return st;
        a.tail_scratch = (float*)m->d_tail;
        a.win_counter = nullptr;
        hipLaunchKernelGGL(dbh_timeline::dbh_forward_kernel, dim3(grid), dim3(dbh::kThreads), 0, 0, a);
    }
    DBH_HIP(hipGetLastError());
    DBH_HIP(hipMemcpyAsync(stamps_host, m->d_work, stamp_bytes, hipMemcpyDeviceToHost, 0));
    DBH_HIP(hipStreamSynchronize(0));
    return DBH_OK;
}

int dbh_forward_timeline_i16(dbh_model* m, const int16_t* samples_host, int64_t n,
                             int64_t* stamps_host) {
    if (!m || n <= 0 || !samples_host || !stamps_host) return DBH_ERR_INVALID_ARGUMENT;
    const size_t stamp_bytes = (size_t)n * dbh::kWaves * 64 * sizeof(int64_t);
    const size_t sample_bytes = (size_t)n * dbh::kWindow * sizeof(int16_t);
    const size_t offset_bytes = (size_t)(n + 1) * sizeof(int64_t);
    int st = ensure(&m->d_in, &m->in_bytes, sample_bytes + offset_bytes + 16);
    if (st != DBH_OK) return st;
    st = ensure(&m->d_work, &m->work_bytes, stamp_bytes);
    if (st != DBH_OK) return st;
    st = ensure(&m->d_out, &m->out_bytes, (size_t)n * (m->n_classes + 1) * sizeof(float));
    if (st != DBH_OK) return st;
    std::vector<int64_t> offsets((size_t)n + 1);
    for (int64_t i = 0; i <= n; ++i) offsets[(size_t)i] = i * dbh::kWindow;
    char* d_offsets = (char*)m->d_in + ((sample_bytes + 7) & ~(size_t)7);
    DBH_HIP(hipMemcpyAsync(m->d_in, samples_host, sample_bytes, hipMemcpyHostToDevice, 0));
    DBH_HIP(hipMemcpyAsync(d_offsets, offsets.data(), offset_bytes, hipMemcpyHostToDevice, 0));
    DBH_HIP(hipMemsetAsync(m->d_work, 0, stamp_bytes, 0));
    DBH_HIP(hipStreamSynchronize(0));
    {
        dbh_timeline::ForwardArgs a = {};
        a.packed = m->d_packed;
        a.probs = (float*)m->d_out;
        a.debug_out = (float*)m->d_work;
        a.samples = (const int16_t*)m->d_in;
        a.offsets = (const long long*)d_offsets;
        a.calls = (int*)((float*)m->d_out + n * m->n_classes);
        a.score_diff = 0.5;
        a.len_hint = (long long)m->hint_len;
        a.hint_cap = (long long)m->hint_cap;
        a.n_windows = (long long)n;
        a.n_classes = m->n_classes;
        // more windows than CUs: a persistent launch, as in production (stamps per window)
        a.debug_stage = n > m->cus ? 301 : 300;
        a.steps = 1;
        const int64_t groups = (n + dbh::kGroup - 1) / dbh::kGroup;
        const unsigned grid = (unsigned)(n > m->cus ? (groups < m->cus ? groups : m->cus) : groups);
        st = ensure(&m->d_tail, &m->tail_bytes, (size_t)grid * dbh::kWgScratchFloats * sizeof(float));
        if (st != DBH_OK) return st;
        a.tail_scratch = (float*)m->d_tail;
        a.win_counter = nullptr;
        hipLaunchKernelGGL(dbh_timeline::dbh_forward_kernel, dim3(grid), dim3(dbh::kThreads), 0, 0, a);
    }
    DBH_HIP(hipGetLastError());
    DBH_HIP(hipMemcpyAsync(stamps_host, m->d_work, stamp_bytes, hipMemcpyDeviceToHost, 0));
    DBH_HIP(hipStreamSynchronize(0));
    return DBH_OK;
}

int dbh_forward_timing_enable_span(dbh_model* m, int every_nth, int span) {
    if (!m || span < 1 || (every_nth > 0 && span > every_nth)) return DBH_ERR_INVALID_ARGUMENT;
    m->timing = every_nth > 0 ? every_nth : 0;
    m->timing_span = span;
    m->open_stop = nullptr;
    m->open_windows = 0;
    m->launch_counter = 0;
    m->events_used = 0;
    m->timed_windows = 0;
    m->timed_launches = 0;
    return DBH_OK;
}

int dbh_forward_timing_enable(dbh_model* m, int enable) {
    return dbh_forward_timing_enable_span(m, enable, 1);
}

int dbh_forward_timing_read(dbh_model* m, double* total_ms, int64_t* launches, int64_t* windows) {
    if (!m || !total_ms || !launches || !windows) return DBH_ERR_INVALID_ARGUMENT;
    double sum = 0.0;
    for (size_t i = 0; i < m->events_used; ++i) {
        DBH_HIP(hipEventSynchronize(m->events[i].second));
        float ms = 0.f;
        DBH_HIP(hipEventElapsedTime(&ms, m->events[i].first, m->events[i].second));
        sum += ms;
    }
    *total_ms = sum;
    *launches = m->timed_launches;
    *windows = m->timed_windows;
    m->events_used = 0;
    m->timed_windows = 0;
    m->timed_launches = 0;
    m->open_stop = nullptr;
    return DBH_OK;
}

int dbh_forward_clock_enable(dbh_model* m, int enable) {
    if (!m) return DBH_ERR_INVALID_ARGUMENT;
    m->clock_probe = enable != 0;
    if (!enable) m->clock_grid = 0;
    return DBH_OK;
}

int dbh_forward_clock_read(dbh_model* m, double* shader_ghz) {
    if (!m || !shader_ghz) return DBH_ERR_INVALID_ARGUMENT;
    *shader_ghz = 0.0;
    if (!m->clock_probe || m->clock_grid == 0 || !m->d_clock) return DBH_ERR_INVALID_ARGUMENT;
    DBH_HIP(hipSetDevice(m->device));
    DBH_HIP(hipDeviceSynchronize());
    const size_t per_wg = 4 + dbh::kPhaseMarks * dbh::kPhaseGroups;
    std::vector<int64_t> c((size_t)m->clock_grid * per_wg);
    DBH_HIP(hipMemcpy(c.data(), m->d_clock, c.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    int wall_khz = 0;      // the rate of s_memrealtime (100 MHz on this hardware)
    DBH_HIP(hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, m->device));
    if (wall_khz <= 0) return DBH_ERR_HIP;
    std::vector<double> ratios;
    for (unsigned b = 0; b < m->clock_grid; ++b) {
        const double shader = (double)(c[b * per_wg + 2] - c[b * per_wg]);
        const double wall = (double)(c[b * per_wg + 3] - c[b * per_wg + 1]);
        if (shader > 0 && wall > 0) ratios.push_back(shader / wall);
    }
    if (ratios.empty()) return DBH_ERR_HIP;
    std::nth_element(ratios.begin(), ratios.begin() + ratios.size() / 2, ratios.end());
    *shader_ghz = ratios[ratios.size() / 2] * (double)wall_khz * 1e-6;
    return DBH_OK;
}

int dbh_forward_phases_enable(dbh_model* m, int enable) {
    if (!m) return DBH_ERR_INVALID_ARGUMENT;
    m->phase_probe = enable != 0;
    return DBH_OK;
}

int dbh_forward_phases_read(dbh_model* m, double* mean_cycles, int64_t* groups) {
    if (!m || !mean_cycles || !groups) return DBH_ERR_INVALID_ARGUMENT;
    if (!m->clock_probe || !m->phase_probe || m->clock_grid == 0 || !m->d_clock) return DBH_ERR_INVALID_ARGUMENT;
    DBH_HIP(hipSetDevice(m->device));
    DBH_HIP(hipDeviceSynchronize());
    const size_t per_wg = 4 + dbh::kPhaseMarks * dbh::kPhaseGroups;
    std::vector<int64_t> c((size_t)m->clock_grid * per_wg);
    DBH_HIP(hipMemcpy(c.data(), m->d_clock, c.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    // steady state: not a workgroup's first group (cold), and only groups followed by another one
    // (the last interval runs to the next group's first stamp)
    double sum[dbh::kPhaseMarks] = {};
    int64_t n = 0;
    for (unsigned b = 0; b < m->clock_grid; ++b) {
        const int64_t* s = c.data() + b * per_wg + 4;
        for (int g = 1; g + 1 < dbh::kPhaseGroups; ++g) {
            const int64_t* a = s + g * dbh::kPhaseMarks;
            if (a[0] < 0 || a[dbh::kPhaseMarks] < 0 || a[dbh::kPhaseMarks + 1] < 0) break;
            bool ok = true;
            double d[dbh::kPhaseMarks];
            for (int i = 0; i < 5; ++i) {
                // (the fifth interval runs to the next group's first stamp)
                const int64_t from = a[i], to = i < 4 ? a[i + 1] : a[dbh::kPhaseMarks];
                d[i] = (double)(uint32_t)((uint32_t)to - (uint32_t)from);
                if (d[i] > 4e6) ok = false;                      // (a group that skipped a phase)
            }
            for (int i = 5; i < dbh::kPhaseMarks; ++i) d[i] = (double)a[i];      // (sums of intervals)
            if (!ok) continue;
            for (int i = 0; i < dbh::kPhaseMarks; ++i) sum[i] += d[i];
            ++n;
        }
    }
    for (int i = 0; i < dbh::kPhaseMarks; ++i) mean_cycles[i] = n ? sum[i] / (double)n : 0.0;
    *groups = n;
    return DBH_OK;
}

}  // extern "C"
