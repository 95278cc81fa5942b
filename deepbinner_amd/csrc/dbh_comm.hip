// dbh_comm.hip — the multi-GPU exchange of the classify path behind the C ABI
// (include/deepbinner_hip.h, "multi-device" section): reads shard over the GPUs of one node with
// no data-path collective; the only exchange is an all-gather of per-read int32 barcode calls
// (SURVEY.md section 8e).  The reference has nothing here - its one device knob is
// deepbinner/classify.py:416-423.
//
// Two host models over the same entry points:
//   - ONE process driving n devices (dbh_comm_init_all -> ncclCommInitAll, one stream per device,
//     the all-gather issued for all devices inside one ncclGroupStart/End);
//   - one process per GPU (dbh_comm_unique_id on rank 0, the 128 bytes shipped to the other ranks
//     over any host channel, dbh_comm_init_rank everywhere).
// RCCL (librccl.so.1) is looked up with dlopen at the first use, so the library loads - and the
// single-GPU path runs - on a box without it.  DBH_COMM_COPY is the same all-gather made of
// hipMemcpyPeerAsync copies (single-process form only): what a test uses when two "devices" are
// the same physical GPU, which RCCL refuses.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/deepbinner_hip.h"

namespace dbh_comm_detail {

thread_local std::string g_comm_error;

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool ok = false;
};

Rccl* rccl() {
    static Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* name : names) {
            x.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.handle) break;
        }
        if (!x.handle) return x;
#define DBH_SYM(field, sym) x.field = (decltype(x.field))dlsym(x.handle, sym)
        DBH_SYM(GetUniqueId, "ncclGetUniqueId");
        DBH_SYM(CommInitRank, "ncclCommInitRank");
        DBH_SYM(CommInitAll, "ncclCommInitAll");
        DBH_SYM(CommDestroy, "ncclCommDestroy");
        DBH_SYM(AllGather, "ncclAllGather");
        DBH_SYM(GroupStart, "ncclGroupStart");
        DBH_SYM(GroupEnd, "ncclGroupEnd");
        DBH_SYM(GetErrorString, "ncclGetErrorString");
#undef DBH_SYM
        x.ok = x.GetUniqueId && x.CommInitRank && x.CommInitAll && x.CommDestroy && x.AllGather &&
               x.GroupStart && x.GroupEnd && x.GetErrorString;
        return x;
    }();
    return &r;
}

int rccl_fail(ncclResult_t e, const char* what) {
    g_comm_error = std::string(what) + ": " + (rccl()->ok ? rccl()->GetErrorString(e) : "?");
    return DBH_ERR_COMM;
}
// (DBH_ERR_COMM for HIP failures too: the text is in dbh_comm_last_error(), which is where a
// caller looks after that status)
int hip_fail(hipError_t e, const char* what) {
    g_comm_error = std::string(what) + ": " + hipGetErrorString(e);
    return DBH_ERR_COMM;
}

// Puts the calling thread's current device back when a function that switches devices returns -
// on every path, the early error returns included.
struct DeviceRestore {
    int prev = -1;
    DeviceRestore() {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    }
    ~DeviceRestore() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

}  // namespace dbh_comm_detail

using namespace dbh_comm_detail;

#define DBH_NCCL(call)                                              \
    do {                                                            \
        ncclResult_t e_ = (call);                                   \
        if (e_ != ncclSuccess) return rccl_fail(e_, #call);         \
    } while (0)
#define DBH_CHIP(call)                                              \
    do {                                                            \
        hipError_t e_ = (call);                                     \
        if (e_ != hipSuccess) return hip_fail(e_, #call);           \
    } while (0)

struct dbh_comm {
    int transport = DBH_COMM_RCCL;
    int n_ranks = 0;                       // size of the communicator
    int rank0 = 0;                         // rank of local device 0
    std::vector<int> devices;              // HIP ordinals of the local devices
    std::vector<ncclComm_t> comms;         // one per local device (RCCL transport)
    std::vector<hipEvent_t> ready;         // COPY transport: "send buffer of device s is final"
    std::vector<hipEvent_t> copied;        //                 "device d has pulled every block"
};

extern "C" {

const char* dbh_comm_last_error(void) { return g_comm_error.c_str(); }

int dbh_comm_available(void) { return rccl()->ok ? 1 : 0; }

int dbh_comm_init_all(int n_devices, const int* ordinals, int transport, dbh_comm** comm) {
    if (!comm || n_devices < 1 || (transport != DBH_COMM_RCCL && transport != DBH_COMM_COPY))
        return DBH_ERR_INVALID_ARGUMENT;
    *comm = nullptr;
    dbh_comm* c = new (std::nothrow) dbh_comm();
    if (!c) return DBH_ERR_OUT_OF_MEMORY;
    c->transport = transport;
    c->n_ranks = n_devices;
    for (int i = 0; i < n_devices; ++i) c->devices.push_back(ordinals ? ordinals[i] : i);
    DeviceRestore restore;
    int st = DBH_OK;
    if (transport == DBH_COMM_RCCL) {
        if (!rccl()->ok) {
            g_comm_error = "librccl.so.1 could not be loaded";
            st = DBH_ERR_COMM;
        } else {
            c->comms.assign((size_t)n_devices, nullptr);
            ncclResult_t e = rccl()->CommInitAll(c->comms.data(), n_devices, c->devices.data());
            if (e != ncclSuccess) {
                c->comms.clear();
                st = rccl_fail(e, "ncclCommInitAll");
            }
        }
    } else {
        for (int i = 0; i < n_devices && st == DBH_OK; ++i) {
            hipEvent_t ev = nullptr, ev2 = nullptr;
            hipError_t e = hipSetDevice(c->devices[(size_t)i]);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev2, hipEventDisableTiming);
            if (ev) c->ready.push_back(ev);
            if (ev2) c->copied.push_back(ev2);
            if (e != hipSuccess) st = hip_fail(e, "dbh_comm_init_all");
        }
    }
    if (st != DBH_OK) {
        for (hipEvent_t ev : c->ready) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : c->copied) (void)hipEventDestroy(ev);
        delete c;
        return st;
    }
    *comm = c;
    return DBH_OK;
}

int dbh_comm_unique_id(void* id_out) {
    if (!id_out) return DBH_ERR_INVALID_ARGUMENT;
    static_assert(sizeof(ncclUniqueId) == DBH_COMM_ID_BYTES, "unique id size");
    if (!rccl()->ok) {
        g_comm_error = "librccl.so.1 could not be loaded";
        return DBH_ERR_COMM;
    }
    ncclUniqueId id;
    DBH_NCCL(rccl()->GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return DBH_OK;
}

int dbh_comm_init_rank(const void* id_in, int n_ranks, int rank, dbh_comm** comm) {
    if (!comm || !id_in || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return DBH_ERR_INVALID_ARGUMENT;
    *comm = nullptr;
    if (!rccl()->ok) {
        g_comm_error = "librccl.so.1 could not be loaded";
        return DBH_ERR_COMM;
    }
    dbh_comm* c = new (std::nothrow) dbh_comm();
    if (!c) return DBH_ERR_OUT_OF_MEMORY;
    c->transport = DBH_COMM_RCCL;
    c->n_ranks = n_ranks;
    c->rank0 = rank;
    int dev = 0;
    hipError_t he = hipGetDevice(&dev);
    if (he != hipSuccess) {
        delete c;
        return hip_fail(he, "hipGetDevice");
    }
    c->devices.push_back(dev);
    ncclUniqueId id;
    std::memcpy(&id, id_in, sizeof(id));
    ncclComm_t nc = nullptr;
    ncclResult_t e = rccl()->CommInitRank(&nc, n_ranks, id, rank);
    if (e != ncclSuccess) {
        delete c;
        return rccl_fail(e, "ncclCommInitRank");
    }
    c->comms.push_back(nc);
    *comm = c;
    return DBH_OK;
}

int dbh_comm_info(const dbh_comm* c, int* n_ranks, int* n_local, int* transport) {
    if (!c) return DBH_ERR_INVALID_ARGUMENT;
    if (n_ranks) *n_ranks = c->n_ranks;
    if (n_local) *n_local = (int)c->devices.size();
    if (transport) *transport = c->transport;
    return DBH_OK;
}

int dbh_comm_all_gather_i32(dbh_comm* c, const int32_t* const* send_dev, int32_t* const* recv_dev,
                            int64_t count, const dbh_stream* streams) {
    if (!c || !send_dev || !recv_dev || !streams || count < 0) return DBH_ERR_INVALID_ARGUMENT;
    if (count == 0) return DBH_OK;
    const int n_local = (int)c->devices.size();
    for (int i = 0; i < n_local; ++i)
        if (!send_dev[i] || !recv_dev[i]) return DBH_ERR_INVALID_ARGUMENT;
    if (c->transport == DBH_COMM_RCCL) {
        // one group: with several communicators in one thread the calls must not block on each
        // other (and a single-communicator group costs nothing)
        DBH_NCCL(rccl()->GroupStart());
        ncclResult_t first = ncclSuccess;
        for (int i = 0; i < n_local; ++i) {
            ncclResult_t e = rccl()->AllGather(send_dev[i], recv_dev[i], (size_t)count, ncclInt32,
                                               c->comms[(size_t)i], (hipStream_t)streams[i]);
            if (e != ncclSuccess && first == ncclSuccess) first = e;
        }
        ncclResult_t end = rccl()->GroupEnd();
        if (first != ncclSuccess) return rccl_fail(first, "ncclAllGather");
        if (end != ncclSuccess) return rccl_fail(end, "ncclGroupEnd");
        return DBH_OK;
    }
    // COPY transport: device d's stream waits until every send buffer is final, then pulls the
    // n blocks into its receive buffer (peer copies; the same device twice is a plain D2D copy).
    // And the back edge: whatever device s queues NEXT on its stream (the next step's
    // classification overwrites its send buffer) waits until every other device has pulled.
    DeviceRestore restore;
    const size_t bytes = (size_t)count * sizeof(int32_t);
    for (int s = 0; s < n_local; ++s) {
        DBH_CHIP(hipSetDevice(c->devices[(size_t)s]));
        DBH_CHIP(hipEventRecord(c->ready[(size_t)s], (hipStream_t)streams[s]));
    }
    for (int d = 0; d < n_local; ++d) {
        DBH_CHIP(hipSetDevice(c->devices[(size_t)d]));
        hipStream_t st = (hipStream_t)streams[d];
        for (int s = 0; s < n_local; ++s) {
            if (s != d) DBH_CHIP(hipStreamWaitEvent(st, c->ready[(size_t)s], 0));
            int32_t* dst = recv_dev[d] + (size_t)s * (size_t)count;
            if (c->devices[(size_t)s] == c->devices[(size_t)d])
                DBH_CHIP(hipMemcpyAsync(dst, send_dev[s], bytes, hipMemcpyDeviceToDevice, st));
            else
                DBH_CHIP(hipMemcpyPeerAsync(dst, c->devices[(size_t)d], send_dev[s],
                                            c->devices[(size_t)s], bytes, st));
        }
        DBH_CHIP(hipEventRecord(c->copied[(size_t)d], st));
    }
    for (int s = 0; s < n_local; ++s) {
        DBH_CHIP(hipSetDevice(c->devices[(size_t)s]));
        for (int d = 0; d < n_local; ++d)
            if (d != s) DBH_CHIP(hipStreamWaitEvent((hipStream_t)streams[s], c->copied[(size_t)d], 0));
    }
    return DBH_OK;
}

int dbh_comm_destroy(dbh_comm* c) {
    if (!c) return DBH_OK;
    for (ncclComm_t nc : c->comms)
        if (nc && rccl()->ok) (void)rccl()->CommDestroy(nc);
    for (hipEvent_t ev : c->ready) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : c->copied) (void)hipEventDestroy(ev);
    delete c;
    return DBH_OK;
}

}  // extern "C"
