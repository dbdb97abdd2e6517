// TEST DOUBLE for librccl (tests/test_gpu_two_process.py): the eight RCCL entry points stitching_amd/csrc/stx_comm.cpp binds, carried over
// UNIX sockets and host staging, so that the product's RCCL code path — dlopen, unique id over the control plane, ncclCommInitRank, one
// ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd per exchange on the communicator's stream, the event ordering around it — runs on a
// box with ONE GPU, where the real library refuses two ranks on a device.  Not a transport: a group is executed synchronously inside
// ncclGroupEnd (stream synchronised, device -> host -> socket -> host -> device).  What it checks on the way: every send meets a
// receive of the same size from the peer it names, in the same order on both sides (a mismatch is an error, not a hang past the timeout).
// Built by the test: g++ -shared -fPIC fake_rccl.cpp -o librccl.so.1 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L/opt/rocm/lib -lamdhip64
#ifdef FAKE_RCCL_NO_HIP  // the socket protocol alone, on host memory (tests/test_host_logic.py, no GPU)
#include <cstring>
typedef void* hipStream_t;
enum { hipSuccess = 0, hipMemcpyDeviceToHost = 2, hipMemcpyHostToDevice = 1 };
static int hipStreamSynchronize(hipStream_t) { return 0; }
static int hipMemcpy(void* d, const void* s, size_t n, int) { memcpy(d, s, n); return 0; }
#else
#include <hip/hip_runtime_api.h>
#endif
#include <poll.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace {

enum { OK = 0, ERR_SYSTEM = 2, ERR_INTERNAL = 3, ERR_ARGUMENT = 4 };
const int TIMEOUT_MS = 120000;

struct Comm {
    int nranks = 0, rank = 0;
    std::vector<int> fd;  // one stream socket per peer
    std::string dir;      // rendezvous directory (rank 0 removes it)
};
struct Op { bool send; void* ptr; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local std::vector<Op> g_ops;
thread_local int g_depth = 0;
std::string g_err = "no error";

int fail(int code, const std::string& what)
{
    g_err = what;
    fprintf(stderr, "[fake rccl] %s\n", what.c_str());
    return code;
}

bool write_all(int fd, const void* p, size_t n)
{
    const char* c = static_cast<const char*>(p);
    while (n) {
        ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
        if (k <= 0) return false;
        c += k; n -= (size_t)k;
    }
    return true;
}

bool read_all(int fd, void* p, size_t n)
{
    char* c = static_cast<char*>(p);
    while (n) {
        pollfd pf{fd, POLLIN, 0};
        if (poll(&pf, 1, TIMEOUT_MS) <= 0) return false;
        ssize_t k = ::recv(fd, c, n, 0);
        if (k <= 0) return false;
        c += k; n -= (size_t)k;
    }
    return true;
}

size_t type_size(int t)
{
    switch (t) {
    case 0: case 1: return 1;            // int8, uint8
    case 6: case 9: return 2;            // half, bfloat16
    case 2: case 3: case 7: return 4;    // int32, uint32, float
    case 4: case 5: case 8: return 8;    // int64, uint64, double
    default: return 0;
    }
}

int run_group(std::vector<Op>& ops)
{
    // the group sits on the streams the caller named: everything queued there before it has to be finished first
    std::vector<hipStream_t> streams;
    for (const Op& o : ops) {
        bool seen = false;
        for (hipStream_t s : streams) seen = seen || s == o.stream;
        if (!seen) streams.push_back(o.stream);
    }
    for (hipStream_t s : streams)
        if (hipStreamSynchronize(s) != hipSuccess) return fail(ERR_INTERNAL, "hipStreamSynchronize failed");
    // sends: staged to the host, then one writer thread per peer (both sides may owe each other more than a socket buffer holds)
    std::map<int, std::vector<std::vector<char>>> out;
    Comm* comm = ops.empty() ? nullptr : ops[0].comm;
    for (const Op& o : ops) {
        if (o.comm != comm) return fail(ERR_ARGUMENT, "one group, two communicators");
        if (o.peer < 0 || o.peer >= comm->nranks || o.peer == comm->rank) return fail(ERR_ARGUMENT, "bad peer " + std::to_string(o.peer));
        if (!o.send) continue;
        std::vector<char> h(o.bytes);
        if (o.bytes && hipMemcpy(h.data(), o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(ERR_INTERNAL, "D2H copy of a send buffer failed");
        out[o.peer].push_back(std::move(h));
    }
    std::vector<std::thread> writers;
    std::vector<int> wrc(out.size(), 0);
    int wi = 0;
    for (auto& kv : out) {
        const int fd = comm->fd[kv.first];
        std::vector<std::vector<char>>* msgs = &kv.second;
        int* rc = &wrc[wi++];
        writers.emplace_back([fd, msgs, rc] {
            for (auto& m : *msgs) {
                uint64_t n = m.size();
                if (!write_all(fd, &n, 8) || !write_all(fd, m.data(), m.size())) { *rc = 1; return; }
            }
        });
    }
    int rc = OK;
    for (const Op& o : ops) {
        if (o.send || rc != OK) continue;
        uint64_t n = 0;
        if (!read_all(comm->fd[o.peer], &n, 8)) { rc = fail(ERR_SYSTEM, "rank " + std::to_string(comm->rank) + ": no message from rank " + std::to_string(o.peer)); break; }
        if (n != o.bytes) { rc = fail(ERR_ARGUMENT, "rank " + std::to_string(comm->rank) + " receives " + std::to_string(o.bytes) + " bytes from rank " + std::to_string(o.peer) + " which sends " + std::to_string(n)); break; }
        std::vector<char> h(o.bytes);
        if (!read_all(comm->fd[o.peer], h.data(), o.bytes)) { rc = fail(ERR_SYSTEM, "short message"); break; }
        // FAKE_RCCL_CORRUPT (fault injection): a transport that initialises and then delivers wrong bytes — what the ring probe exists for
        if (o.bytes && getenv("FAKE_RCCL_CORRUPT")) h[o.bytes / 2] ^= 0x5a;
        if (o.bytes && hipMemcpy(o.ptr, h.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(ERR_INTERNAL, "H2D copy into a receive buffer failed"); break; }
    }
    for (auto& t : writers) t.join();
    for (int w : wrc)
        if (w && rc == OK) rc = fail(ERR_SYSTEM, "a send failed");
    return rc;
}

int post(const Op& o)
{
    g_ops.push_back(o);
    if (g_depth > 0) return OK;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_group(ops);
}

}  // namespace

#define FAKE_API extern "C" __attribute__((visibility("default")))

FAKE_API int ncclGetUniqueId(char* id)  // ncclUniqueId*: 128 bytes
{
    memset(id, 0, 128);
    snprintf(id, 128, "/tmp/fake_rccl_%d_%lld", (int)getpid(), (long long)std::chrono::steady_clock::now().time_since_epoch().count());
    if (mkdir(id, 0700) != 0) return fail(ERR_SYSTEM, std::string("cannot create ") + id);
    return OK;
}

struct FakeId { char internal[128]; };

FAKE_API int ncclCommInitRank(void** out, int nranks, FakeId id, int rank)
{
    if (!out || nranks < 1 || rank < 0 || rank >= nranks) return fail(ERR_ARGUMENT, "ncclCommInitRank: bad argument");
    id.internal[127] = 0;
    const std::string dir = id.internal;
    Comm* c = new Comm();
    c->nranks = nranks; c->rank = rank; c->fd.assign(nranks, -1);
    auto addr_of = [&](int r) {
        sockaddr_un a{};
        a.sun_family = AF_UNIX;
        snprintf(a.sun_path, sizeof(a.sun_path), "%s/r%d", dir.c_str(), r);
        return a;
    };
    int ls = socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un me = addr_of(rank);
    if (ls < 0 || bind(ls, (sockaddr*)&me, sizeof(me)) != 0 || listen(ls, nranks) != 0) return fail(ERR_SYSTEM, "cannot listen in " + dir);
    for (int p = 0; p < rank; p++) {  // connect to every lower rank (its socket may not exist yet)
        sockaddr_un a = addr_of(p);
        int fd = -1;
        for (int tries = 0; tries < TIMEOUT_MS / 20; tries++) {
            fd = socket(AF_UNIX, SOCK_STREAM, 0);
            if (connect(fd, (sockaddr*)&a, sizeof(a)) == 0) break;
            close(fd); fd = -1;
            usleep(20000);
        }
        if (fd < 0) return fail(ERR_SYSTEM, "rank " + std::to_string(rank) + " cannot reach rank " + std::to_string(p));
        int32_t r32 = rank;
        if (!write_all(fd, &r32, 4)) return fail(ERR_SYSTEM, "handshake failed");
        c->fd[p] = fd;
    }
    for (int k = 0; k < nranks - 1 - rank; k++) {  // ... and take the higher ones
        pollfd pf{ls, POLLIN, 0};
        if (poll(&pf, 1, TIMEOUT_MS) <= 0) return fail(ERR_SYSTEM, "rank " + std::to_string(rank) + ": a higher rank never connected");
        int fd = accept(ls, nullptr, nullptr);
        int32_t r32 = -1;
        if (fd < 0 || !read_all(fd, &r32, 4) || r32 <= rank || r32 >= nranks || c->fd[r32] >= 0) return fail(ERR_SYSTEM, "bad handshake");
        c->fd[r32] = fd;
    }
    close(ls);
    unlink(me.sun_path);  // every higher rank is connected: the name is not needed any more
    c->dir = dir;
    *out = c;
    return OK;
}

FAKE_API int ncclCommDestroy(void* comm)
{
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return OK;
    for (int fd : c->fd)
        if (fd >= 0) close(fd);
    if (c->rank == 0) rmdir(c->dir.c_str());  // empty once every rank has unlinked its socket; otherwise it stays, harmlessly
    delete c;
    return OK;
}

FAKE_API int ncclCommCount(void* comm, int* count) { if (!comm || !count) return fail(ERR_ARGUMENT, "ncclCommCount: bad argument"); *count = static_cast<Comm*>(comm)->nranks; return OK; }
FAKE_API int ncclCommUserRank(void* comm, int* rank) { if (!comm || !rank) return fail(ERR_ARGUMENT, "ncclCommUserRank: bad argument"); *rank = static_cast<Comm*>(comm)->rank; return OK; }
FAKE_API int ncclGetVersion(int* v) { if (!v) return fail(ERR_ARGUMENT, "ncclGetVersion: bad argument"); *v = 0; return OK; }  // 0: the test double

FAKE_API int ncclGroupStart() { g_depth++; return OK; }

FAKE_API int ncclGroupEnd()
{
    if (g_depth <= 0) return fail(ERR_ARGUMENT, "ncclGroupEnd without ncclGroupStart");
    if (--g_depth > 0) return OK;
    std::vector<Op> ops;
    ops.swap(g_ops);
    return run_group(ops);
}

FAKE_API int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream)
{
    if (!comm || (!buf && count) || !type_size(dtype)) return fail(ERR_ARGUMENT, "ncclSend: bad argument");
    return post(Op{true, const_cast<void*>(buf), count * type_size(dtype), peer, static_cast<Comm*>(comm), stream});
}

FAKE_API int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream)
{
    if (!comm || (!buf && count) || !type_size(dtype)) return fail(ERR_ARGUMENT, "ncclRecv: bad argument");
    return post(Op{false, buf, count * type_size(dtype), peer, static_cast<Comm*>(comm), stream});
}

FAKE_API const char* ncclGetErrorString(int) { return g_err.c_str(); }
