// Data parallel inside the C ABI (include/awr_hip.h, "Data-parallel API"): RCCL communicators for a host that does not bring its own.
//
// The reference is single-GPU (train.py:29,:233); SURVEY 8b / 8e ask for one process per GPU with an all-reduce of the gradient arena over
// xGMI, overlapped with the backward.  The Python host does that through torch.distributed and the plan's bucket callback; a non-Python
// host gets the same thing from these entry points: the library opens librccl.so at run time (dlopen -- libawr_hip.so keeps no link-time
// dependency on RCCL, and a process that exchanges gradients itself never loads it), creates one communicator per rank, and
// awr_plan_set_dp makes the backward replay issue `ncclAllReduce(SUM)` for every gradient bucket the moment it is final, on a stream of
// the communicator's own (an event orders it behind the bucket's scatter; the caller's stream waits for all of them at the end of the
// backward).  Host code only.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <map>
#include <set>
#include <vector>

#include <rccl/rccl.h>      // types and prototypes only: every call goes through the pointers below

#include "awr_common.h"

namespace awrdp {

using awr::set_error;

struct Api {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    char path[256] = "";
    char err[256] = "";       // why the library could not be opened (dlerror() right after the failed call: the next dl* call clears it)
};

static Api g_api;
static Api* api() {
    Api& a = g_api;
    static std::once_flag once;
    static bool ok = false;
    std::call_once(once, [&a]() {
        // the copy the process already holds (a framework's) first, then the system's; AWR_RCCL_LIB overrides
        const char* env = getenv("AWR_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (int pass = 0; pass < 2 && !a.handle; ++pass)
            for (const char* n : names) {
                if (!n) continue;
                a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
                if (a.handle) {
                    strncpy(a.path, n, sizeof a.path - 1);
                    break;
                }
                if (pass == 1) {
                    const char* e = dlerror();
                    if (e) strncpy(a.err, e, sizeof a.err - 1);
                }
            }
        if (!a.handle) return;
#define AWR_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, name))
        AWR_SYM(GetUniqueId, "ncclGetUniqueId");
        AWR_SYM(CommInitRank, "ncclCommInitRank");
        AWR_SYM(CommDestroy, "ncclCommDestroy");
        AWR_SYM(AllReduce, "ncclAllReduce");
        AWR_SYM(Broadcast, "ncclBroadcast");
        AWR_SYM(GetErrorString, "ncclGetErrorString");
        AWR_SYM(GetVersion, "ncclGetVersion");
#undef AWR_SYM
        ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.AllReduce && a.Broadcast && a.GetErrorString;
        if (!ok) snprintf(a.err, sizeof a.err, "%s lacks one of ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce / Broadcast / GetErrorString", a.path);
    });
    return ok ? &a : nullptr;
}

}  // namespace awrdp

struct awr_dp {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;      // the collectives' own stream: they never queue in front of the plan's weight-gradient launches
    // ordering events: one per hand-over since the last awr_dp_wait, created on demand (a host that exchanges tensor by tensor -- 458 for
    // Hourglass-2 -- must not wrap a fixed ring onto an event whose wait has not been enqueued yet); awr_dp_wait starts the list over
    // (hipStreamWaitEvent captures the record it waits for when it is called: re-recording afterwards is safe)
    std::vector<hipEvent_t> ev;
    size_t ev_next = 0;
};

// live communicators: a plan keeps a raw pointer (awr_plan_set_dp) and must not call into one its host has destroyed
// (keyed by address, valued by a process-wide generation number: a NEW communicator that the allocator places at a destroyed one's address
// is a different generation, so a plan still holding the old pointer does not mistake it for its own)
static std::mutex g_live_mu;
static std::map<const awr_dp*, unsigned long long> g_live;
static unsigned long long g_next_gen = 1;
extern "C" unsigned long long awr_dp_generation(const awr_dp* d) {      // 0 = not a live communicator
    std::lock_guard<std::mutex> lk(g_live_mu);
    auto it = g_live.find(d);
    return it == g_live.end() ? 0ull : it->second;
}

using namespace awrdp;

#define DP_API()                                                                                                             \
    Api* A = api();                                                                                                          \
    if (!A) {                                                                                                                \
        set_error("data parallel: librccl.so could not be used (set AWR_RCCL_LIB to its path): %s", awrdp::g_api.err[0] ? awrdp::g_api.err : "not found"); \
        return AWR_ERR_UNSUPPORTED;                                                                                          \
    }
#define DP_TRY(call)                                                        \
    do {                                                                    \
        ncclResult_t r__ = (call);                                          \
        if (r__ != ncclSuccess) {                                           \
            set_error("%s: %s", #call, A->GetErrorString(r__));             \
            return AWR_ERR_HIP;                                             \
        }                                                                   \
    } while (0)
#define DP_HIP(call)                                                        \
    do {                                                                    \
        hipError_t e__ = (call);                                            \
        if (e__ != hipSuccess) {                                            \
            set_error("%s: %s", #call, hipGetErrorString(e__));             \
            return AWR_ERR_HIP;                                             \
        }                                                                   \
    } while (0)

extern "C" {

int awr_dp_available(int* version, const char** path) {
    Api* A = api();
    if (version) {
        *version = 0;
        if (A && A->GetVersion) (void)A->GetVersion(version);
    }
    if (path) *path = A ? A->path : "";
    return A ? AWR_OK : AWR_ERR_UNSUPPORTED;
}

int awr_dp_unique_id(void* id128) {
    AWR_REQUIRE(id128, "dp_unique_id: null pointer");
    DP_API();
    static_assert(sizeof(ncclUniqueId) == AWR_DP_ID_BYTES, "unique id size");
    ncclUniqueId id;
    DP_TRY(A->GetUniqueId(&id));
    memcpy(id128, &id, sizeof id);
    return AWR_OK;
}

int awr_dp_init(int rank, int world, const void* id128, awr_dp** out) {
    AWR_REQUIRE(out && id128 && world >= 1 && rank >= 0 && rank < world, "dp_init: rank %d of %d", rank, world);
    DP_API();
    awr_dp* d = new awr_dp();
    d->rank = rank;
    d->world = world;
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    if (hipGetDevice(&d->device) != hipSuccess || hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking) != hipSuccess) {
        set_error("dp_init: could not create the communication stream");
        delete d;
        return AWR_ERR_HIP;
    }
    const ncclResult_t r = A->CommInitRank(&d->comm, world, id, rank);      // collective: every rank of the job calls it (current device)
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, A->GetErrorString(r));
        (void)hipStreamDestroy(d->stream);
        delete d;
        return AWR_ERR_HIP;
    }
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live[d] = g_next_gen++;
    }
    *out = d;
    return AWR_OK;
}

int awr_dp_destroy(awr_dp* d) {
    if (!d) return AWR_OK;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        if (!g_live.erase(d)) {
            set_error("dp_destroy: not a live communicator (destroyed twice?)");
            return AWR_ERR_ARG;
        }
    }
    Api* A = api();
    (void)hipStreamSynchronize(d->stream);
    if (A && d->comm) (void)A->CommDestroy(d->comm);
    for (auto& e : d->ev) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(d->stream);
    delete d;
    return AWR_OK;
}

int awr_dp_info(const awr_dp* d, int* rank, int* world, int* device) {
    AWR_REQUIRE(d, "dp_info: null pointer");
    if (rank) *rank = d->rank;
    if (world) *world = d->world;
    if (device) *device = d->device;
    return AWR_OK;
}

// `buf` is final in `stream` order -> collective on the communicator's stream behind an event; awr_dp_wait orders a stream after it
static int next_event(awr_dp* d, hipEvent_t* out) {
    if (d->ev_next == d->ev.size()) {
        hipEvent_t e;
        DP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        d->ev.push_back(e);
    }
    *out = d->ev[d->ev_next++];
    return AWR_OK;
}
static int order_behind(awr_dp* d, hipStream_t stream) {
    hipEvent_t e;
    if (int rc = next_event(d, &e)) return rc;
    DP_HIP(hipEventRecord(e, stream));
    DP_HIP(hipStreamWaitEvent(d->stream, e, 0));
    return AWR_OK;
}

int awr_dp_allreduce(awr_dp* d, float* buf, int64_t n, void* stream) {
    AWR_REQUIRE(d && buf && n > 0, "dp_allreduce: bad arguments");
    DP_API();
    if (int e = order_behind(d, awr::as_stream(stream))) return e;
    DP_TRY(A->AllReduce(buf, buf, (size_t)n, ncclFloat32, ncclSum, d->comm, d->stream));
    return AWR_OK;
}

int awr_dp_broadcast(awr_dp* d, float* buf, int64_t n, int root, void* stream) {
    AWR_REQUIRE(d && buf && n > 0 && root >= 0 && root < d->world, "dp_broadcast: bad arguments");
    DP_API();
    if (int e = order_behind(d, awr::as_stream(stream))) return e;
    DP_TRY(A->Broadcast(buf, buf, (size_t)n, ncclFloat32, root, d->comm, d->stream));
    return AWR_OK;
}

int awr_dp_wait(awr_dp* d, void* stream) {
    AWR_REQUIRE(d, "dp_wait: null pointer");
    hipEvent_t e;
    if (int rc = next_event(d, &e)) return rc;
    DP_HIP(hipEventRecord(e, d->stream));
    DP_HIP(hipStreamWaitEvent(awr::as_stream(stream), e, 0));
    d->ev_next = 0;      // every wait on the events handed out so far has been enqueued: the list starts over
    return AWR_OK;
}

}  // extern "C"
