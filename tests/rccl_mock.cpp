/*
 * rccl_mock.cpp - TEST INFRASTRUCTURE (never shipped, never linked by the product): a stand-in for librccl with the nine entry points the
 * product's opt-in RCCL exchange resolves through dlopen (criteria3d_amd/csrc/sf3d_host_build.inc: load_rccl) - ncclGetUniqueId,
 * ncclCommInitRank, ncclCommDestroy, ncclSend, ncclRecv, ncclAllGather, ncclAllReduce, ncclGroupStart, ncclGroupEnd - carried by POSIX
 * shared memory + hipMemcpyAsync, so that SF3D_EXCHANGE=rccl (pack / unpack kernels, the host's sequencing of halo exchanges and
 * all-gathers between the kernels, the separate decision kernels) can be run with the ranks of a test SHARING ONE GPU, which the real
 * library refuses.  tests/test_gpu_multirank.py holds such runs bit for bit against the window transport.
 *
 * Semantics kept: every call is asynchronous and ordered on the caller's stream (the data of a send is read when the stream reaches
 * it, a receive's buffer is valid for whatever the stream runs next); sends never wait for the matching receive of the same round
 * (two message slots per ordered pair of ranks), so the product's grouped send / recv loops cannot deadlock; an all-gather is complete
 * on a rank when every rank's contribution of that round has arrived.  Waits are bounded (60 s, then abort with a message).
 *
 * Build: hipcc -shared -fPIC tests/rccl_mock.cpp -o tests/librccl_mock.so (tests/conftest.py / the multirank test do it on demand).
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

constexpr int kMaxRanks = 8;
constexpr size_t kSlotBytes = 1u << 20;          /* one message: up to 128 Ki doubles (C4 in two strips sends 164 KB) */
constexpr size_t kGatherDoubles = 64;            /* per rank and round */

struct Channel {                                  /* messages src -> dst */
    std::atomic<uint64_t> published;              /* messages written so far */
    std::atomic<uint64_t> consumed;               /* messages read so far */
    char pad[48];
};
struct Shared {
    std::atomic<uint32_t> ready;                  /* set by the creator once the header is initialised */
    std::atomic<uint32_t> attached;
    uint32_t world, pad0;
    Channel ch[kMaxRanks][kMaxRanks];
    std::atomic<uint64_t> gatherRound[kMaxRanks]; /* all-gather rounds a rank has contributed to */
    double gather[2][kMaxRanks][kGatherDoubles];
    /* followed by world * world * 2 message slots of kSlotBytes */
};

struct Comm {
    Shared* sh = nullptr;
    char* slots = nullptr;
    size_t bytes = 0;
    int rank = 0, world = 1;
    char name[64] = {0};
    uint64_t sent[kMaxRanks] = {0}, received[kMaxRanks] = {0}, gathers = 0;
    bool registered = false;
};

char* slot_of(Comm* c, int src, int dst, uint64_t seq)
{
    return c->slots + (((size_t)src * c->world + dst) * 2 + (seq & 1u)) * kSlotBytes;
}

[[noreturn]] void die(const char* what)
{
    std::fprintf(stderr, "rccl_mock: %s\n", what);
    std::fflush(stderr);
    std::_Exit(97);
}

template <class Pred> void wait_for(Pred ok, const char* what)
{
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (!ok()) {
        if ((++spins & 63u) == 0) {
            sched_yield();
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) die(what);
        }
    }
}

/* host functions run by the stream: they only touch the shared header (no HIP calls inside) */
struct Op { Comm* c; int peer; uint64_t seq; };
void cb_wait_slot_free(void* p)   { Op* o = static_cast<Op*>(p); Channel& ch = o->c->sh->ch[o->c->rank][o->peer]; const uint64_t s = o->seq; wait_for([&] { return ch.consumed.load(std::memory_order_acquire) + 2 > s; }, "send: the peer did not consume the message before the last within 60 s"); delete o; }
void cb_publish(void* p)          { Op* o = static_cast<Op*>(p); o->c->sh->ch[o->c->rank][o->peer].published.store(o->seq + 1, std::memory_order_release); delete o; }
void cb_wait_published(void* p)   { Op* o = static_cast<Op*>(p); Channel& ch = o->c->sh->ch[o->peer][o->c->rank]; const uint64_t s = o->seq; wait_for([&] { return ch.published.load(std::memory_order_acquire) > s; }, "recv: no message from the peer within 60 s"); delete o; }
void cb_consumed(void* p)         { Op* o = static_cast<Op*>(p); o->c->sh->ch[o->peer][o->c->rank].consumed.store(o->seq + 1, std::memory_order_release); delete o; }
void cb_gather_publish(void* p)   { Op* o = static_cast<Op*>(p); o->c->sh->gatherRound[o->c->rank].store(o->seq + 1, std::memory_order_release); delete o; }
void cb_gather_wait(void* p)
{
    Op* o = static_cast<Op*>(p);
    Comm* c = o->c; const uint64_t s = o->seq;
    for (int r = 0; r < c->world; ++r) wait_for([&] { return c->sh->gatherRound[r].load(std::memory_order_acquire) > s; }, "all-gather: a rank did not contribute within 60 s");
    delete o;
}

size_t type_bytes(ncclDataType_t t) { return (t == ncclDouble || t == ncclInt64 || t == ncclUint64) ? 8 : ((t == ncclFloat || t == ncclInt32 || t == ncclUint32) ? 4 : (t == ncclHalf ? 2 : 1)); }

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id)
{
    static std::atomic<unsigned> counter{0};
    std::memset(id, 0, sizeof(*id));
    std::snprintf(id->internal, sizeof(id->internal), "/sf3d_rccl_mock_%d_%u", (int)getpid(), counter.fetch_add(1));
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    Comm* c = new Comm();
    c->rank = rank; c->world = nranks;
    std::snprintf(c->name, sizeof(c->name), "%.60s", id.internal);
    c->bytes = sizeof(Shared) + (size_t)nranks * nranks * 2 * kSlotBytes;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(c->name);
        fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { std::perror("rccl_mock: shm_open"); delete c; return ncclSystemError; }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        struct stat stt;
        while ((fd = shm_open(c->name, O_RDWR, 0600)) < 0 || fstat(fd, &stt) != 0 || (size_t)stt.st_size < c->bytes) {
            if (fd >= 0) { close(fd); fd = -1; }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) { delete c; return ncclSystemError; }
            usleep(1000);
        }
    }
    void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return ncclSystemError; }
    c->sh = static_cast<Shared*>(p);
    c->slots = static_cast<char*>(p) + sizeof(Shared);
    if (rank == 0) {      /* (a fresh object is zero-filled: counters start at 0) */
        c->sh->world = (uint32_t)nranks;
        c->sh->ready.store(1, std::memory_order_release);
    } else {
        wait_for([&] { return c->sh->ready.load(std::memory_order_acquire) == 1; }, "CommInitRank: rank 0 did not initialise the segment within 60 s");
    }
    /* pinned + mapped: the copies below are truly asynchronous */
    if (hipHostRegister(p, c->bytes, hipHostRegisterPortable) == hipSuccess) c->registered = true; else (void)hipGetLastError();
    if (c->sh->attached.fetch_add(1) + 1 == (uint32_t)nranks) shm_unlink(c->name);      /* everyone is in: the name can go (the mappings stay) */
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c) return ncclSuccess;
    (void)hipDeviceSynchronize();
    if (c->registered) (void)hipHostUnregister(c->sh);
    munmap(c->sh, c->bytes);
    shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }
ncclResult_t ncclGroupEnd() { return ncclSuccess; }

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = count * type_bytes(type);
    if (!c || peer < 0 || peer >= c->world || peer == c->rank || bytes > kSlotBytes) return ncclInvalidArgument;
    const uint64_t seq = c->sent[peer]++;
    if (hipLaunchHostFunc(stream, cb_wait_slot_free, new Op{c, peer, seq}) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes && hipMemcpyAsync(slot_of(c, c->rank, peer, seq), sendbuff, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(stream, cb_publish, new Op{c, peer, seq}) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = count * type_bytes(type);
    if (!c || peer < 0 || peer >= c->world || peer == c->rank || bytes > kSlotBytes) return ncclInvalidArgument;
    const uint64_t seq = c->received[peer]++;
    if (hipLaunchHostFunc(stream, cb_wait_published, new Op{c, peer, seq}) != hipSuccess) return ncclUnhandledCudaError;
    if (bytes && hipMemcpyAsync(recvbuff, slot_of(c, peer, c->rank, seq), bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(stream, cb_consumed, new Op{c, peer, seq}) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

/* round n lives in parity n & 1: a rank can only write round n + 2 after every rank has written n + 1, i.e. after every rank has read n */
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t type, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    const size_t bytes = sendcount * type_bytes(type);
    if (!c || bytes > kGatherDoubles * sizeof(double)) return ncclInvalidArgument;
    const uint64_t seq = c->gathers++;
    double (*g)[kGatherDoubles] = c->sh->gather[seq & 1u];
    if (bytes && hipMemcpyAsync(g[c->rank], sendbuff, bytes, hipMemcpyDeviceToHost, stream) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(stream, cb_gather_publish, new Op{c, 0, seq}) != hipSuccess) return ncclUnhandledCudaError;
    if (hipLaunchHostFunc(stream, cb_gather_wait, new Op{c, 0, seq}) != hipSuccess) return ncclUnhandledCudaError;
    for (int r = 0; r < c->world && bytes; ++r)
        if (hipMemcpyAsync(static_cast<char*>(recvbuff) + (size_t)r * bytes, g[r], bytes, hipMemcpyHostToDevice, stream) != hipSuccess) return ncclUnhandledCudaError;
    return ncclSuccess;
}

/* (resolved by the product's loader, not called by it: sums of doubles only, through the all-gather) */
ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm, hipStream_t stream)
{
    Comm* c = reinterpret_cast<Comm*>(comm);
    if (!c || type != ncclDouble || op != ncclSum || count > kGatherDoubles) return ncclInvalidArgument;
    double* tmp = nullptr;
    if (hipMalloc(&tmp, sizeof(double) * count * c->world) != hipSuccess) return ncclUnhandledCudaError;
    ncclResult_t r = ncclAllGather(sendbuff, tmp, count, type, comm, stream);
    if (r != ncclSuccess) { (void)hipFree(tmp); return r; }
    if (hipStreamSynchronize(stream) != hipSuccess) { (void)hipFree(tmp); return ncclUnhandledCudaError; }
    double host[kGatherDoubles * kMaxRanks], sum[kGatherDoubles] = {0};
    (void)hipMemcpy(host, tmp, sizeof(double) * count * c->world, hipMemcpyDeviceToHost);
    for (int rk = 0; rk < c->world; ++rk) for (size_t k = 0; k < count; ++k) sum[k] += host[(size_t)rk * count + k];
    (void)hipMemcpy(recvbuff, sum, sizeof(double) * count, hipMemcpyHostToDevice);
    (void)hipFree(tmp);
    return ncclSuccess;
}

}  /* extern "C" */
