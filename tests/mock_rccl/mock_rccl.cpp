// mock_rccl.cpp -- TEST DOUBLE of the nine RCCL entry points the slab runtime uses, for rehearsing a
// real multi-PROCESS run (one rank per process, FS_RCCL_LIBRARY=libmock_rccl.so) on a box with ONE GPU,
// where RCCL itself refuses several ranks on one device.  Test infrastructure only: nothing in the
// product loads it unless that environment variable names it.
//
// Semantics kept from NCCL: point-to-point messages between a pair of ranks match in FIFO order;
// a send and its receive must agree on the element count (checked: a mismatch is an error here, a hang
// or corruption with the real library); every rank must issue collectives in the same order (a rank
// that does not gets a timeout here instead of a hang).  NOT kept: asynchrony -- every call completes
// on the host before it returns (the stream is drained first), so overlap is not rehearsed here (the
// in-process asynchronous group does that).
//
// Transport: POSIX shared memory.  A header object holds one mailbox ring per (src, dst) pair and the
// all-reduce scratch; each message's payload is its own short-lived shared-memory object.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>

namespace {

constexpr int RING = 256;
constexpr int MAX_RANKS = 64;
constexpr int MAX_AR = 1024;
constexpr double TIMEOUT_S = 60.0;

struct Slot {
    uint64_t count;
    uint64_t bytes;
};
struct Mailbox {
    std::atomic<uint64_t> head;   // messages published by the sender
    std::atomic<uint64_t> tail;   // messages consumed by the receiver
    Slot slots[RING];
};
struct Header {
    std::atomic<int> joined, left;
    std::atomic<uint64_t> ar_arrived, ar_done;
    float ar_vals[2][MAX_RANKS][MAX_AR];
    Mailbox box[1];               // nranks * nranks, index src * nranks + dst
};

struct Comm {
    Header *hdr = nullptr;
    size_t bytes = 0;
    int rank = 0, nranks = 1;
    std::string base;
    uint64_t ar_gen = 0;
};

struct Op {
    bool send;
    void *buf;
    size_t count;
    int peer;
    Comm *comm;
    hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
template <class F>
bool wait_for(F cond, const char *what, const Comm *c)
{
    const double t0 = now();
    while (!cond()) {
        if (now() - t0 > TIMEOUT_S) {
            fprintf(stderr, "[mock_rccl] rank %d/%d: timed out waiting for %s\n", c->rank, c->nranks, what);
            return false;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return true;
}

std::string payload_name(const Comm *c, int src, int dst, uint64_t idx)
{
    return c->base + "_" + std::to_string(src) + "_" + std::to_string(dst) + "_" + std::to_string(idx);
}

ncclResult_t do_send(const Op &op)
{
    Comm *c = op.comm;
    Mailbox &mb = c->hdr->box[c->rank * c->nranks + op.peer];
    const uint64_t idx = mb.head.load(std::memory_order_relaxed);
    if (!wait_for([&] { return idx - mb.tail.load(std::memory_order_acquire) < RING; }, "a free mailbox slot", c))
        return ncclSystemError;
    const size_t bytes = op.count * sizeof(float);
    const std::string name = payload_name(c, c->rank, op.peer, idx);
    int fd = shm_open(name.c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)(bytes ? bytes : 1)) != 0) { perror("[mock_rccl] shm payload"); return ncclSystemError; }
    void *m = mmap(nullptr, bytes ? bytes : 1, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { perror("[mock_rccl] mmap payload"); return ncclSystemError; }
    if (bytes && hipMemcpy(m, op.buf, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    munmap(m, bytes ? bytes : 1);
    mb.slots[idx % RING].count = op.count;
    mb.slots[idx % RING].bytes = bytes;
    mb.head.store(idx + 1, std::memory_order_release);
    return ncclSuccess;
}

ncclResult_t do_recv(const Op &op)
{
    Comm *c = op.comm;
    Mailbox &mb = c->hdr->box[op.peer * c->nranks + c->rank];
    const uint64_t idx = mb.tail.load(std::memory_order_relaxed);
    if (!wait_for([&] { return mb.head.load(std::memory_order_acquire) > idx; }, "a message (peer sent fewer than I receive?)", c))
        return ncclSystemError;
    const Slot s = mb.slots[idx % RING];
    const std::string name = payload_name(c, op.peer, c->rank, idx);
    ncclResult_t rc = ncclSuccess;
    if (s.count != op.count) {
        fprintf(stderr, "[mock_rccl] rank %d: message %llu from rank %d has %llu elements, the receive expects %zu\n",
                c->rank, (unsigned long long)idx, op.peer, (unsigned long long)s.count, op.count);
        rc = ncclInvalidArgument;
    } else {
        int fd = shm_open(name.c_str(), O_RDWR, 0600);
        if (fd < 0) { perror("[mock_rccl] open payload"); return ncclSystemError; }
        void *m = mmap(nullptr, s.bytes ? s.bytes : 1, PROT_READ, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { perror("[mock_rccl] mmap payload"); return ncclSystemError; }
        if (s.bytes && hipMemcpy(op.buf, m, s.bytes, hipMemcpyHostToDevice) != hipSuccess) rc = ncclUnhandledCudaError;
        munmap(m, s.bytes ? s.bytes : 1);
    }
    shm_unlink(name.c_str());
    mb.tail.store(idx + 1, std::memory_order_release);
    return rc;
}

ncclResult_t flush()
{
    std::vector<Op> ops;
    ops.swap(g_ops);
    for (const Op &op : ops)
        if (hipStreamSynchronize(op.stream) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t rc = ncclSuccess;
    for (const Op &op : ops)
        if (op.send && rc == ncclSuccess) rc = do_send(op);
    for (const Op &op : ops)
        if (!op.send && rc == ncclSuccess) rc = do_recv(op);
    return rc;
}

ncclResult_t enqueue(bool send, const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || dt != ncclFloat || peer < 0 || peer >= c->nranks) return ncclInvalidArgument;
    g_ops.push_back(Op{send, const_cast<void *>(buf), count, peer, c, st});
    return g_depth == 0 ? flush() : ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    snprintf(id->internal, sizeof(id->internal), "/fsmock_%d_%llx", (int)getpid(),
             (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm *c = new Comm();
    c->rank = rank; c->nranks = nranks;
    c->base = std::string(id.internal, strnlen(id.internal, sizeof(id.internal)));
    c->bytes = sizeof(Header) + sizeof(Mailbox) * (size_t)nranks * nranks;
    int fd = shm_open((c->base + "_hdr").c_str(), O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { perror("[mock_rccl] shm header"); delete c; return ncclSystemError; }
    void *m = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { perror("[mock_rccl] mmap header"); delete c; return ncclSystemError; }
    c->hdr = reinterpret_cast<Header *>(m);   // a fresh object is zero-filled: all counters start at 0
    c->hdr->joined.fetch_add(1);
    if (!wait_for([&] { return c->hdr->joined.load() >= nranks; }, "all ranks to join", c)) { delete c; return ncclSystemError; }
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c) return ncclSuccess;
    const bool last = c->hdr->left.fetch_add(1) == c->nranks - 1;
    munmap(c->hdr, c->bytes);
    if (last) shm_unlink((c->base + "_hdr").c_str());
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd()
{
    if (g_depth <= 0) return ncclInvalidUsage;
    return --g_depth == 0 ? flush() : ncclSuccess;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st)
{
    return enqueue(true, buf, count, dt, peer, comm, st);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st)
{
    return enqueue(false, buf, count, dt, peer, comm, st);
}

ncclResult_t ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t dt, ncclRedOp_t op,
                           ncclComm_t comm, hipStream_t st)
{
    Comm *c = reinterpret_cast<Comm *>(comm);
    if (!c || (dt != ncclFloat && dt != ncclInt32) || count > MAX_AR || (op != ncclMax && op != ncclSum && op != ncclMin)) return ncclInvalidArgument;
    if (dt == ncclInt32 && op != ncclSum) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    const uint64_t g = c->ar_gen++, n = (uint64_t)c->nranks;
    Header *h = c->hdr;
    // the buffer of generation g was last used by generation g-2: everyone must have read that one
    if (g >= 2 && !wait_for([&] { return h->ar_done.load(std::memory_order_acquire) >= (g - 1) * n; }, "all-reduce buffer", c))
        return ncclSystemError;
    float *mine = h->ar_vals[g & 1][c->rank];
    if (count && hipMemcpy(mine, sendbuf, count * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    h->ar_arrived.fetch_add(1, std::memory_order_release);
    if (!wait_for([&] { return h->ar_arrived.load(std::memory_order_acquire) >= (g + 1) * n; },
                  "the other ranks' all-reduce (collectives issued in a different order?)", c))
        return ncclSystemError;
    std::vector<float> r(count);
    for (size_t i = 0; i < count && dt == ncclInt32; ++i) {   // 32-bit integer sum: the same 4-byte slots, read as ints
        int acc = 0;
        for (int q = 0; q < c->nranks; ++q) {
            int v;
            memcpy(&v, &h->ar_vals[g & 1][q][i], sizeof(v));
            acc += v;
        }
        memcpy(&r[i], &acc, sizeof(acc));
    }
    for (size_t i = 0; i < count && dt == ncclFloat; ++i) {
        float acc = h->ar_vals[g & 1][0][i];
        for (int q = 1; q < c->nranks; ++q) {
            const float v = h->ar_vals[g & 1][q][i];
            if (op == ncclSum) acc += v;
            else if (op == ncclMax) acc = (v > acc || v != v) ? v : acc;
            else acc = (v < acc || v != v) ? v : acc;
        }
        r[i] = acc;
    }
    h->ar_done.fetch_add(1, std::memory_order_release);
    if (count && hipMemcpy(recvbuf, r.data(), count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "success";
    case ncclUnhandledCudaError: return "HIP error";
    case ncclSystemError: return "system error / timeout";
    case ncclInvalidArgument: return "invalid argument (count mismatch?)";
    case ncclInvalidUsage: return "invalid usage";
    default: return "error";
    }
}

}  // extern "C"
