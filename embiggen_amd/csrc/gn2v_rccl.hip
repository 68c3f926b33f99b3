// gn2v_rccl_*: include/gn2v_rccl.h -- the communicator of gn2v_train_world over RCCL.
// One process per GPU; xGMI is point to point, and the schedule's only bulk traffic is a part
// to the ring neighbour per episode (one link each way) plus the round's walks all-gathered.
// RCCL is reached through dlopen / dlsym: no link-time dependency (see the header).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/gn2v_rccl.h"
#include "handle.h"

static_assert(sizeof(ncclUniqueId) == GN2V_RCCL_ID_BYTES, "GN2V_RCCL_ID_BYTES != sizeof(ncclUniqueId)");

namespace {

using gn2v_host::fail;

struct Rccl {
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    bool ok = false;
    std::string why;
};

const Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *name = getenv("GN2V_RCCL_LIB");
        void *h = dlopen(name && *name ? name : "librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h && !(name && *name)) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            const char *e = dlerror();
            r.why = std::string("cannot load RCCL: ") + (e ? e : "dlopen failed");
            return;
        }
        bool all = true;
        auto sym = [&](auto &fn, const char *s) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, s));
            if (!fn) {
                all = false;
                r.why = std::string("RCCL lacks ") + s;
            }
        };
        sym(r.GetUniqueId, "ncclGetUniqueId");
        sym(r.CommInitRank, "ncclCommInitRank");
        sym(r.CommDestroy, "ncclCommDestroy");
        sym(r.GetErrorString, "ncclGetErrorString");
        sym(r.AllGather, "ncclAllGather");
        sym(r.Broadcast, "ncclBroadcast");
        sym(r.Send, "ncclSend");
        sym(r.Recv, "ncclRecv");
        sym(r.GroupStart, "ncclGroupStart");
        sym(r.GroupEnd, "ncclGroupEnd");
        r.ok = all;
    });
    return r;
}

int nccl_fail(const char *what, ncclResult_t rc) {
    const Rccl &r = rccl();
    return fail(std::string(what) + ": " + (r.GetErrorString ? r.GetErrorString(rc) : "RCCL error"));
}

#define NCCL_TRY(expr)                                             \
    do {                                                           \
        ncclResult_t rc_ = (expr);                                 \
        if (rc_ != ncclSuccess) return nccl_fail(#expr, rc_);      \
    } while (0)

struct Fabric {
    ncclComm_t nccl = nullptr;
    hipStream_t side = nullptr;  // the part exchanges' own stream
    hipEvent_t ev = nullptr;
    int device = 0;
};

int all_gather(void *ctx, const void *send, void *recv, uint64_t bytes, void *stream) {
    Fabric *f = (Fabric *)ctx;  // on the caller's stream: the library reads `recv` right after
    NCCL_TRY(rccl().AllGather(send, recv, bytes, ncclUint8, f->nccl, (hipStream_t)stream));
    return 0;
}

int sendrecv_start(void *ctx, const void *send, uint64_t send_bytes, uint32_t dst, void *recv,
                   uint64_t recv_bytes, uint32_t src, void *stream, void **handle) {
    Fabric *f = (Fabric *)ctx;
    // after what the caller's stream holds now (the kernel that trained the part that leaves)
    HIP_TRY(hipEventRecord(f->ev, (hipStream_t)stream));
    HIP_TRY(hipStreamWaitEvent(f->side, f->ev, 0));
    const Rccl &r = rccl();
    NCCL_TRY(r.GroupStart());
    ncclResult_t a = r.Send(send, send_bytes, ncclUint8, (int)dst, f->nccl, f->side);
    ncclResult_t b = r.Recv(recv, recv_bytes, ncclUint8, (int)src, f->nccl, f->side);
    ncclResult_t c = r.GroupEnd();
    if (a != ncclSuccess) return nccl_fail("ncclSend", a);
    if (b != ncclSuccess) return nccl_fail("ncclRecv", b);
    if (c != ncclSuccess) return nccl_fail("ncclGroupEnd", c);
    if (handle) *handle = f;
    return 0;
}

int sendrecv_wait(void *ctx, void *handle, void *stream) {
    Fabric *f = (Fabric *)(handle ? handle : ctx);
    HIP_TRY(hipEventRecord(f->ev, f->side));
    HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, f->ev, 0));
    return 0;
}

int broadcast(void *ctx, void *buf, uint64_t bytes, uint32_t root, void *stream) {
    Fabric *f = (Fabric *)ctx;
    NCCL_TRY(rccl().Broadcast(buf, buf, bytes, ncclUint8, (int)root, f->nccl, (hipStream_t)stream));
    return 0;
}

}  // namespace

extern "C" int gn2v_rccl_unique_id(void *out) {
    if (!out) return fail("NULL id buffer");
    const Rccl &r = rccl();
    if (!r.ok) return fail(r.why);
    ncclUniqueId id;
    NCCL_TRY(r.GetUniqueId(&id));
    memcpy(out, &id, sizeof(id));
    return 0;
}

extern "C" int gn2v_rccl_comm_create(const void *unique_id, uint32_t rank, uint32_t world,
                                     int device, gn2v_comm *comm) {
    if (!unique_id || !comm) return fail("NULL id / communicator");
    if (world < 1 || rank >= world) return fail("need rank < world");
    const Rccl &r = rccl();
    if (!r.ok) return fail(r.why);
    gn2v_host::DeviceGuard guard(device);
    if (!guard.ok()) return fail("cannot select the HIP device");
    Fabric *f = new Fabric;
    f->device = device;
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclResult_t rc = r.CommInitRank(&f->nccl, (int)world, id, (int)rank);
    if (rc != ncclSuccess) {
        delete f;
        return nccl_fail("ncclCommInitRank", rc);
    }
    if (hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&f->ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (f->side) (void)hipStreamDestroy(f->side);
        (void)r.CommDestroy(f->nccl);
        delete f;
        return fail("cannot create the exchange stream / event");
    }
    comm->ctx = f;
    comm->rank = rank;
    comm->world = world;
    comm->all_gather = all_gather;
    comm->sendrecv_start = sendrecv_start;
    comm->sendrecv_wait = sendrecv_wait;
    comm->broadcast = broadcast;
    return 0;
}

extern "C" int gn2v_rccl_comm_destroy(gn2v_comm *comm) {
    if (!comm || !comm->ctx) return 0;
    Fabric *f = (Fabric *)comm->ctx;
    gn2v_host::DeviceGuard guard(f->device);
    (void)hipStreamSynchronize(f->side);
    (void)hipEventDestroy(f->ev);
    (void)hipStreamDestroy(f->side);
    const Rccl &r = rccl();
    int bad = 0;
    if (r.ok && r.CommDestroy(f->nccl) != ncclSuccess) bad = fail("ncclCommDestroy failed");
    delete f;
    memset(comm, 0, sizeof(*comm));
    return bad;
}
