// stx_comm.cpp — RCCL point-to-point exchange of contribution strips over xGMI.
// librccl is loaded lazily (dlopen) by stx_comm_unique_id / stx_comm_create: single-GPU users never
// touch it.  All sends and receives of one exchange are issued as ONE RCCL group on the communicator's
// own HIP stream, ordered by events after the kernels that produced the strips (context stream) and
// before the gather kernels that consume them — no host synchronisation in between, and kernels that
// do not depend on the strips overlap with the transfer.
#include <dlfcn.h>

#include <cstring>

#include <map>

#include "stx_internal.h"

namespace {

typedef struct { char internal[128]; } RcclUniqueId;  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef void* RcclComm;
enum { RCCL_UINT8 = 1 };  // ncclUint8

struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*Send)(const void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    // optional (stx_comm_info): what the library itself says about the communicator
    int (*CommCount)(RcclComm, int*) = nullptr;
    int (*CommUserRank)(RcclComm, int*) = nullptr;
    int (*GetVersion)(int*) = nullptr;
};
RcclApi g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return STX_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return stx_fail(STX_ERR_UNSUPPORTED, "cannot load librccl: %s", dlerror());
    RcclApi a;
    a.handle = h;
#define SYM(field, name)                                                                     \
    *(void**)(&a.field) = dlsym(h, name);                                                    \
    if (!a.field) return stx_fail(STX_ERR_UNSUPPORTED, "librccl lacks the symbol %s", name);
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
    *(void**)(&a.CommCount) = dlsym(h, "ncclCommCount");
    *(void**)(&a.CommUserRank) = dlsym(h, "ncclCommUserRank");
    *(void**)(&a.GetVersion) = dlsym(h, "ncclGetVersion");
    g_rccl = a;
    return STX_OK;
}

#define STX_RCCL(call)                                                                                   \
    do {                                                                                                 \
        int r_ = (call);                                                                                 \
        if (r_ != 0) return stx_fail(STX_ERR_HIP, "%s failed: %s", #call, g_rccl.GetErrorString(r_));    \
    } while (0)

}  // namespace

struct stx_comm {
    stx_ctx* ctx;
    RcclComm comm;
    int nranks, rank;
    // exchanges run on their own stream so that independent kernels queued on the context stream after
    // stx_comm_exchange_begin overlap with the transfer
    hipStream_t stream = nullptr;
    // one (ready, done) event pair per context that exchanges through this communicator: several panoramas in flight
    // each wait for their OWN transfer only
    struct Slot { hipEvent_t ready = nullptr, done = nullptr; bool in_flight = false; };
    std::map<stx_ctx*, Slot> slots;
};

STX_EXPORT int stx_comm_unique_id(unsigned char out[128])
{
    if (!out) return stx_fail(STX_ERR_INVALID, "null argument");
    STX_TRY(load_rccl());
    RcclUniqueId id;
    STX_RCCL(g_rccl.GetUniqueId(&id));
    memcpy(out, id.internal, 128);
    return STX_OK;
}

STX_EXPORT int stx_comm_create(stx_ctx* ctx, int nranks, int rank, const unsigned char id[128], stx_comm** out)
{
    if (!ctx || !id || !out) return stx_fail(STX_ERR_INVALID, "null argument");
    if (nranks < 1 || rank < 0 || rank >= nranks) return stx_fail(STX_ERR_INVALID, "rank %d of %d", rank, nranks);
    STX_TRY(load_rccl());
    STX_TRY(stx_set_device(ctx));
    RcclUniqueId uid;
    memcpy(uid.internal, id, 128);
    RcclComm c = nullptr;
    STX_RCCL(g_rccl.CommInitRank(&c, nranks, uid, rank));
    stx_comm* k = new stx_comm();
    k->ctx = ctx; k->comm = c; k->nranks = nranks; k->rank = rank;
    *out = k;
    return STX_OK;
}

// {ranks the library counts in the communicator (ncclCommCount), this rank by the library (ncclCommUserRank), library version
// (ncclGetVersion), device}; -1 where the loaded library lacks the call
STX_EXPORT int stx_comm_info(const stx_comm* comm, int out[4])
{
    if (!comm || !out) return stx_fail(STX_ERR_INVALID, "null argument");
    out[0] = out[1] = out[2] = -1;
    out[3] = comm->ctx->device;
    if (g_rccl.CommCount) STX_RCCL(g_rccl.CommCount(comm->comm, &out[0]));
    if (g_rccl.CommUserRank) STX_RCCL(g_rccl.CommUserRank(comm->comm, &out[1]));
    if (g_rccl.GetVersion) STX_RCCL(g_rccl.GetVersion(&out[2]));
    return STX_OK;
}

STX_EXPORT int stx_comm_exchange_begin(stx_comm* comm, int n_ops, const int* peers, const int* is_send, void* const* dev_ptrs,
                                       const size_t* bytes)
{
    return stx_comm_exchange_begin_on(comm, comm ? comm->ctx : nullptr, n_ops, peers, is_send, dev_ptrs, bytes);
}

STX_EXPORT int stx_comm_exchange_end(stx_comm* comm) { return stx_comm_exchange_end_on(comm, comm ? comm->ctx : nullptr); }

// `on`: the context (same device) whose stream produced the send buffers and will consume the receive buffers.
// One communicator serves several contexts (panoramas in flight on different streams): their exchanges run one
// after the other on the communicator's stream, in the order the begin calls were made — the same on every rank.
STX_EXPORT int stx_comm_exchange_begin_on(stx_comm* comm, stx_ctx* on, int n_ops, const int* peers, const int* is_send,
                                          void* const* dev_ptrs, const size_t* bytes)
{
    if (!comm || !on || n_ops < 0 || (n_ops && (!peers || !is_send || !dev_ptrs || !bytes)))
        return stx_fail(STX_ERR_INVALID, "bad argument");
    if (on->device != comm->ctx->device) return stx_fail(STX_ERR_INVALID, "context lives on another device than the communicator");
    stx_comm::Slot& slot = comm->slots[on];
    if (slot.in_flight) return stx_fail(STX_ERR_STATE, "an exchange of this context is already in flight on this communicator");
    STX_TRY(stx_set_device(comm->ctx));
    for (int i = 0; i < n_ops; i++)
        if (peers[i] < 0 || peers[i] >= comm->nranks || !dev_ptrs[i])
            return stx_fail(STX_ERR_INVALID, "exchange op %d: peer %d / null buffer", i, peers[i]);
    if (!comm->stream) STX_HIP(hipStreamCreateWithFlags(&comm->stream, hipStreamNonBlocking));
    if (!slot.ready) {
        STX_HIP(hipEventCreateWithFlags(&slot.ready, hipEventDisableTiming));
        STX_HIP(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
    }
    // the transfer starts after everything queued so far on the context stream (the kernels that filled the
    // send buffers, the last users of the memory behind the receive buffers) ...
    STX_HIP(hipEventRecord(slot.ready, on->stream));
    STX_HIP(hipStreamWaitEvent(comm->stream, slot.ready, 0));
    if (n_ops > 0) {
        STX_RCCL(g_rccl.GroupStart());
        for (int i = 0; i < n_ops; i++) {
            int r = is_send[i] ? g_rccl.Send(dev_ptrs[i], bytes[i], RCCL_UINT8, peers[i], comm->comm, comm->stream)
                               : g_rccl.Recv(dev_ptrs[i], bytes[i], RCCL_UINT8, peers[i], comm->comm, comm->stream);
            if (r != 0) {
                g_rccl.GroupEnd();
                return stx_fail(STX_ERR_HIP, "rccl %s to/from rank %d failed: %s", is_send[i] ? "send" : "recv", peers[i],
                                g_rccl.GetErrorString(r));
            }
        }
        STX_RCCL(g_rccl.GroupEnd());
    }
    STX_HIP(hipEventRecord(slot.done, comm->stream));
    slot.in_flight = true;
    return STX_OK;
}

// ... and everything queued on the context stream after this call sees the received strips
STX_EXPORT int stx_comm_exchange_end_on(stx_comm* comm, stx_ctx* on)
{
    if (!comm || !on) return stx_fail(STX_ERR_INVALID, "null argument");
    auto it = comm->slots.find(on);
    if (it == comm->slots.end() || !it->second.in_flight) return STX_OK;
    STX_TRY(stx_set_device(comm->ctx));
    STX_HIP(hipStreamWaitEvent(on->stream, it->second.done, 0));
    it->second.in_flight = false;
    return STX_OK;
}

STX_EXPORT int stx_comm_exchange(stx_comm* comm, int n_ops, const int* peers, const int* is_send, void* const* dev_ptrs,
                                 const size_t* bytes)
{
    STX_TRY(stx_comm_exchange_begin(comm, n_ops, peers, is_send, dev_ptrs, bytes));
    return stx_comm_exchange_end(comm);
}

STX_EXPORT int stx_comm_destroy(stx_comm* comm)
{
    if (!comm) return STX_OK;
    hipSetDevice(comm->ctx->device);
    if (comm->stream) hipStreamSynchronize(comm->stream);
    if (g_rccl.handle && comm->comm) {
        hipStreamSynchronize(comm->ctx->stream);
        g_rccl.CommDestroy(comm->comm);
    }
    for (auto& kv : comm->slots) {
        if (kv.second.ready) hipEventDestroy(kv.second.ready);
        if (kv.second.done) hipEventDestroy(kv.second.done);
    }
    if (comm->stream) hipStreamDestroy(comm->stream);
    delete comm;
    return STX_OK;
}
