// fdw_comm.cpp -- communicators of the multi-GPU paths of libfdwave.so (include/fdwave.h, "multi-GPU").
//
// The reference has no multi-GPU code; what travels here are the halo rows of the x-slab decomposition
// (fdw_slabs.cpp).  Two backends behind one interface:
//
//   RCCL   one rank per process (or per host thread) and per GPU: ncclSend / ncclRecv of the halo blocks inside ONE
//          ncclGroupStart / ncclGroupEnd, enqueued on the caller's communication stream -- neighbour-only traffic, one xGMI link per
//          pair.  librccl is opened lazily (dlopen "librccl.so.1") the first time a communicator is asked for, so programs that stay on
//          one GPU never load it, and inside a PyTorch process the copy PyTorch already loaded is the one that is used.
//   local  all ranks inside one process, a host thread each: a halo transfer is a device-to-device copy enqueued on the RECEIVER's
//          communication stream, ordered after the sender's stream by an event and held against the sender's next write by a second
//          event -- the same dependencies a send in flight has.  Ranks may share one device (tests and rehearsals on a one-GPU box: RCCL
//          refuses two ranks on one device) or sit on different devices of the node (peer copies over xGMI, no RCCL involved).
//   shm    one rank per PROCESS without RCCL: halo blocks are staged through a POSIX shared-memory segment.  The GPU only ever touches
//          process-private pinned buffers: device -> private buffer on the sender's stream, then a host function behind that copy moves the
//          block into the segment and publishes its sequence number; the receiver waits for the number, copies the block out of the segment
//          with the CPU, publishes "taken" and enqueues private buffer -> device on its stream.  (A first version let both processes' copy
//          engines work on the segment itself, registered as pinned memory in each; a three-process run then produced a wrong image once in
//          a few dozen runs -- two GPU mappings of one set of pages is not something to build a checker on.)  A TEST TRANSPORT: it lets the slab drivers of fdw_slabs.cpp run as real processes whose streams know
//          nothing of each other beyond message arrival -- which is what RCCL gives -- with all ranks on ONE GPU (tests/test_slabs_gpu.py,
//          bench.py --backend shm).  Slow by design (two PCIe crossings per block); never chosen automatically.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>

#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <vector>

#include "fdw_internal.h"
#include "fdwave.h"

// ------------------------------------------------------------------------------------------------
// RCCL through dlopen: just the entry points the halo exchange needs (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220,
// ncclCommDestroy :260, ncclSend :700, ncclRecv :722, ncclGroupStart :923, ncclGroupEnd :933, ncclAllReduce :611)
// ------------------------------------------------------------------------------------------------
namespace {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[FDW_COMM_ID_BYTES]; } ncclUniqueId;
typedef int ncclResult_t;                       // ncclSuccess == 0
enum { kNcclFloat = 7, kNcclDouble = 8, kNcclSum = 0, kNcclMax = 2 };   // ncclFloat32, ncclFloat64, ncclSum, ncclMax (rccl.h:448-467)

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    char why[256] = "";
};

Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names)
            if ((r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;      // (a copy already in the process -- PyTorch's -- is found by its SONAME and reused)
        if (!r.so) {
            snprintf(r.why, sizeof r.why, "librccl.so.1 cannot be loaded: %s", dlerror());
            return;
        }
#define FDW_SYM(field, name)                                                            \
    do {                                                                                \
        *(void**)(&r.field) = dlsym(r.so, name);                                        \
        if (!r.field && !r.why[0]) snprintf(r.why, sizeof r.why, "librccl has no %s", name); \
    } while (0)
        FDW_SYM(GetUniqueId, "ncclGetUniqueId");
        FDW_SYM(CommInitRank, "ncclCommInitRank");
        FDW_SYM(CommDestroy, "ncclCommDestroy");
        FDW_SYM(Send, "ncclSend");
        FDW_SYM(Recv, "ncclRecv");
        FDW_SYM(AllReduce, "ncclAllReduce");
        FDW_SYM(GroupStart, "ncclGroupStart");
        FDW_SYM(GroupEnd, "ncclGroupEnd");
        FDW_SYM(GetErrorString, "ncclGetErrorString");
#undef FDW_SYM
    });
    return &r;
}

#define NCCL_TRY(call)                                                                                                   \
    do {                                                                                                                 \
        ncclResult_t r_ = (call);                                                                                        \
        if (r_ != 0) return fdw_fail(FDW_ECOMM, "%s failed: %s", #call, rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "?"); \
    } while (0)
#define HIP_TRY(call)                                                                                            \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess) return fdw_fail(FDW_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// A rank that dies (an error on one host thread) must not leave its neighbours waiting for ever
constexpr std::chrono::seconds kRendezvousTimeout(120);

// ---- the local backend's rendezvous: one mailbox per rank and direction ----
struct Mailbox {
    long posted = 0, taken = 0;                 // sequence numbers of the last message posted by the owner / consumed by the neighbour
    const float* ptr[FDW_COMM_MAX_FIELDS] = {};
    size_t count = 0;
    int nfields = 0, device = 0;
    hipEvent_t ready = nullptr;                 // recorded by the sender on its stream when the rows to send are final
    hipEvent_t done = nullptr;                  // recorded by the receiver on its stream when the copy has been enqueued behind `ready`
};
struct LocalGroup {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    std::vector<Mailbox> box;                   // [rank * 2 + dir], dir 0 = towards rank - 1, 1 = towards rank + 1
    std::vector<long> seq;                      // exchanges each rank has started
    int barrier_count = 0;
    long barrier_gen = 0;
    std::vector<double> red;                    // all-reduce scratch
    int red_count = 0;
    long red_gen = 0;
    double red_result = 0.0;
};

// ---- the process transport: one segment per communicator ----
constexpr unsigned kShmMagic = 0xFD3A5EEDu;
constexpr int kShmMaxWorld = 64;
struct ShmBox {                                 // [rank * 2 + dir]: what `rank` sends towards rank - 1 (dir 0) / rank + 1 (dir 1)
    std::atomic<long> posted;                   // sequence number of the last block the owner's stream has finished writing
    std::atomic<long> taken;                    // ... the neighbour's stream has finished reading
    size_t count;
    int nfields;
    char pad[256 - 2 * sizeof(std::atomic<long>) - sizeof(size_t) - sizeof(int)];
};
struct ShmHeader {
    std::atomic<unsigned> magic;                // written last by rank 0
    int world;
    size_t cap_bytes;                           // payload capacity of one box
    std::atomic<int> attached, detached;
    std::atomic<int> red_count;
    std::atomic<long> red_gen;
    double red[kShmMaxWorld];
    double red_result;
};
static_assert(sizeof(ShmBox) == 256, "box header");
struct ShmSeg {
    char* base = nullptr;
    size_t bytes = 0;
    std::vector<long> seq;                      // exchanges this rank has started
    // process-private pinned staging (hipHostMalloc): what the copy engines of THIS process read and write
    char* send_buf[2] = {nullptr, nullptr};     // [dir]
    char* recv_buf[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};      // [dir][exchange parity]
    hipEvent_t recv_done[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};   // the upload out of recv_buf[dir][parity] has run
    ShmHeader* hdr() const { return reinterpret_cast<ShmHeader*>(base); }
    static size_t hdr_bytes() { return (sizeof(ShmHeader) + 4095) / 4096 * 4096; }
    ShmBox* box(int rank, int dir) const { return reinterpret_cast<ShmBox*>(base + hdr_bytes() + (size_t)(rank * 2 + dir) * (sizeof(ShmBox) + hdr()->cap_bytes)); }
    float* data(int rank, int dir) const { return reinterpret_cast<float*>(reinterpret_cast<char*>(box(rank, dir)) + sizeof(ShmBox)); }
};
struct ShmPost {                                // argument of the host function that runs behind the device -> private-buffer copies of one block
    std::atomic<long>* flag;
    long seq;
    char* dst;                                  // the block's place in the segment
    const char* src;                            // the private buffer the copies filled
    size_t bytes;
};
void shm_post(void* p)
{
    ShmPost* f = static_cast<ShmPost*>(p);
    memcpy(f->dst, f->src, f->bytes);
    f->flag->store(f->seq, std::memory_order_release);
    delete f;
}
template <class Pred>
bool shm_wait(Pred&& ok)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0; !ok(); spins++) {
        if (spins > 200) usleep(50);
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > kRendezvousTimeout) return false;
    }
    return true;
}
}  // namespace

struct fdw_comm {
    int rank = 0, world = 1, device = 0;
    bool stub = false;                          // fdw_comm_init_stub: nothing travels (timing experiments)
    ncclComm_t nccl = nullptr;                  // RCCL backend
    std::shared_ptr<LocalGroup> grp;            // local backend
    std::unique_ptr<ShmSeg> shm;                // process transport
};

extern "C" int fdw_comm_rank(const fdw_comm* c) { return c ? c->rank : 0; }
extern "C" int fdw_comm_world(const fdw_comm* c) { return c ? c->world : 1; }
extern "C" int fdw_comm_device(const fdw_comm* c) { return c ? c->device : 0; }
extern "C" int fdw_comm_is_local(const fdw_comm* c) { return c && c->grp ? 1 : 0; }
extern "C" int fdw_comm_kind(const fdw_comm* c) { return !c ? 0 : (c->nccl ? FDW_COMM_RCCL : (c->grp ? FDW_COMM_LOCAL : (c->shm ? FDW_COMM_SHM : 0))); }

extern "C" int fdw_comm_get_unique_id(char id[FDW_COMM_ID_BYTES])
{
    if (!id) return fdw_fail(FDW_EINVAL, "id is NULL");
    Rccl* r = rccl();
    if (r->why[0]) return fdw_fail(FDW_ECOMM, "%s", r->why);
    ncclUniqueId u;
    NCCL_TRY(r->GetUniqueId(&u));
    memcpy(id, u.internal, FDW_COMM_ID_BYTES);
    return FDW_OK;
}

extern "C" int fdw_comm_init_rank(const char id[FDW_COMM_ID_BYTES], int rank, int world, int device, fdw_comm** out)
{
    if (!out) return fdw_fail(FDW_EINVAL, "out is NULL");
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return fdw_fail(FDW_EINVAL, "comm_init_rank: rank %d of %d", rank, world);
    Rccl* r = rccl();
    if (r->why[0]) return fdw_fail(FDW_ECOMM, "%s", r->why);
    HIP_TRY(hipSetDevice(device));
    fdw_comm* c = new (std::nothrow) fdw_comm();
    if (!c) return fdw_fail(FDW_ENOMEM, "out of host memory");
    c->rank = rank; c->world = world; c->device = device;
    ncclUniqueId u;
    memcpy(u.internal, id, FDW_COMM_ID_BYTES);
    ncclResult_t rc = r->CommInitRank(&c->nccl, world, u, rank);
    if (rc != 0) {
        delete c;
        return fdw_fail(FDW_ECOMM, "ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, r->GetErrorString(rc));
    }
    *out = c;
    return FDW_OK;
}

extern "C" int fdw_comm_init_stub(int rank, int world, int device, fdw_comm** out)
{
    if (!out || world < 1 || rank < 0 || rank >= world) return fdw_fail(FDW_EINVAL, "comm_init_stub: rank %d of %d", rank, world);
    fdw_comm* c = new (std::nothrow) fdw_comm();
    if (!c) return fdw_fail(FDW_ENOMEM, "out of host memory");
    c->rank = rank; c->world = world; c->device = device; c->stub = true;
    *out = c;
    return FDW_OK;
}

// Rank 0 creates the segment, the others attach (whoever comes first waits for the other); once every rank holds a mapping the name is
// unlinked, so nothing is left in /dev/shm whatever happens to the processes afterwards.
extern "C" int fdw_comm_init_shm(const char* name, int rank, int world, int device, size_t box_bytes, fdw_comm** out)
{
    if (!out) return fdw_fail(FDW_EINVAL, "out is NULL");
    *out = nullptr;
    if (!name || name[0] != '/' || world < 1 || world > kShmMaxWorld || rank < 0 || rank >= world || box_bytes == 0)
        return fdw_fail(FDW_EINVAL, "comm_init_shm: name must start with '/', rank %d of %d (at most %d), box_bytes %zu", rank, world, kShmMaxWorld, box_bytes);
    HIP_TRY(hipSetDevice(device));
    const size_t cap = (box_bytes + 4095) / 4096 * 4096;
    const size_t bytes = ShmSeg::hdr_bytes() + (size_t)world * 2 * (sizeof(ShmBox) + cap);
    int fd = -1;
    if (rank == 0) {
        (void)shm_unlink(name);      // a stale segment of a run that died
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0) return fdw_fail(FDW_ECOMM, "shm_open(%s, create) failed: %s", name, strerror(errno));
        if (ftruncate(fd, (off_t)bytes) != 0) {
            const int e = errno;
            close(fd);
            (void)shm_unlink(name);
            return fdw_fail(FDW_ECOMM, "ftruncate(%s, %zu) failed: %s", name, bytes, strerror(e));
        }
    } else {
        const bool ok = shm_wait([&] {
            fd = shm_open(name, O_RDWR, 0600);
            if (fd < 0) return false;
            struct stat st;
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) return true;
            close(fd);
            fd = -1;
            return false;
        });
        if (!ok) return fdw_fail(FDW_ECOMM, "rank %d waited %lds for rank 0 to create %s", rank, (long)kRendezvousTimeout.count(), name);
    }
    void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) {
        if (rank == 0) (void)shm_unlink(name);
        return fdw_fail(FDW_ECOMM, "mmap(%s, %zu) failed: %s", name, bytes, strerror(errno));
    }
    fdw_comm* c = new (std::nothrow) fdw_comm();
    if (!c) {
        munmap(m, bytes);
        return fdw_fail(FDW_ENOMEM, "out of host memory");
    }
    c->rank = rank; c->world = world; c->device = device;
    c->shm.reset(new ShmSeg());
    c->shm->base = static_cast<char*>(m);
    c->shm->bytes = bytes;
    c->shm->seq.assign(1, 0);
    ShmHeader* h = c->shm->hdr();
    if (rank == 0) {      // a fresh segment is zero-filled: counters and sequence numbers start at 0
        h->world = world;
        h->cap_bytes = cap;
        h->magic.store(kShmMagic, std::memory_order_release);
    } else if (!shm_wait([&] { return h->magic.load(std::memory_order_acquire) == kShmMagic; }) || h->world != world || h->cap_bytes != cap) {
        munmap(m, bytes);
        c->shm->base = nullptr;
        delete c;
        return fdw_fail(FDW_ECOMM, "rank %d: segment %s was not initialised by rank 0 for %d ranks x %zu bytes", rank, name, world, cap);
    }
    // private pinned staging: two send buffers, two pairs of receive buffers (a block is uploaded while the next one is copied out)
    {
        hipError_t e = hipSuccess;
        for (int d = 0; d < 2 && e == hipSuccess; d++) {
            e = hipHostMalloc((void**)&c->shm->send_buf[d], cap, hipHostMallocDefault);
            for (int q = 0; q < 2 && e == hipSuccess; q++) {
                e = hipHostMalloc((void**)&c->shm->recv_buf[d][q], cap, hipHostMallocDefault);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&c->shm->recv_done[d][q], hipEventDisableTiming);
            }
        }
        if (e != hipSuccess) {
            h->attached.fetch_add(1);           // the others must not wait for this rank for ever; they fail in their first exchange instead
            fdw_comm_destroy(c);
            return fdw_fail(FDW_ENOMEM, "comm_init_shm: pinned staging buffers (%zu bytes each): %s", cap, hipGetErrorString(e));
        }
    }
    h->attached.fetch_add(1);
    const bool all = shm_wait([&] { return h->attached.load() >= world; });
    if (rank == 0) (void)shm_unlink(name);
    if (!all) {
        fdw_comm_destroy(c);
        return fdw_fail(FDW_ECOMM, "rank %d waited %lds for all %d ranks to attach to %s", rank, (long)kRendezvousTimeout.count(), world, name);
    }
    *out = c;
    return FDW_OK;
}

static int init_local_impl(int world, const int* devices, fdw_comm** out)
{
    auto g = std::make_shared<LocalGroup>();
    g->world = world;
    g->box.resize((size_t)world * 2);
    g->seq.assign(world, 0);
    g->red.assign(world, 0.0);
    for (int r = 0; r < world; r++) {
        fdw_comm* c = new (std::nothrow) fdw_comm();
        if (!c) return fdw_fail(FDW_ENOMEM, "out of host memory");
        c->rank = r; c->world = world; c->device = devices ? devices[r] : 0; c->grp = g;
        out[r] = c;      // from here on the caller's cleanup finds it (and the events created so far)
        HIP_TRY(hipSetDevice(c->device));
        for (int d = 0; d < 2; d++) {
            Mailbox& b = g->box[(size_t)r * 2 + d];
            b.device = c->device;
            HIP_TRY(hipEventCreateWithFlags(&b.ready, hipEventDisableTiming));
        }
    }
    // `done` events live on the RECEIVER's device: box (r, dir) is consumed by rank r -/+ 1
    for (int r = 0; r < world; r++)
        for (int d = 0; d < 2; d++) {
            const int nb = d == 0 ? r - 1 : r + 1;
            if (nb < 0 || nb >= world) continue;
            HIP_TRY(hipSetDevice(out[nb]->device));
            HIP_TRY(hipEventCreateWithFlags(&g->box[(size_t)r * 2 + d].done, hipEventDisableTiming));
        }
    return FDW_OK;
}

extern "C" int fdw_comm_init_local(int world, const int* devices, fdw_comm** out)
{
    if (!out || world < 1) return fdw_fail(FDW_EINVAL, "comm_init_local: world=%d", world);
    for (int r = 0; r < world; r++) out[r] = nullptr;
    int prev = -1;
    (void)hipGetDevice(&prev);
    const int rc = init_local_impl(world, devices, out);
    if (rc != FDW_OK) {      // nothing half-built is handed back: the ranks created so far go, with their events
        for (int r = 0; r < world; r++) {
            if (out[r]) fdw_comm_destroy(out[r]);
            out[r] = nullptr;
        }
    }
    if (prev >= 0) (void)hipSetDevice(prev);      // the calling thread keeps the device it had
    return rc;
}

extern "C" void fdw_comm_destroy(fdw_comm* c)
{
    if (!c) return;
    if (c->nccl && rccl()->CommDestroy) (void)rccl()->CommDestroy(c->nccl);
    if (c->shm && c->shm->base) {
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();           // host functions that publish into the segment have run
        for (int d = 0; d < 2; d++) {
            if (c->shm->send_buf[d]) (void)hipHostFree(c->shm->send_buf[d]);
            for (int q = 0; q < 2; q++) {
                if (c->shm->recv_buf[d][q]) (void)hipHostFree(c->shm->recv_buf[d][q]);
                if (c->shm->recv_done[d][q]) (void)hipEventDestroy(c->shm->recv_done[d][q]);
            }
        }
        munmap(c->shm->base, c->shm->bytes);
        c->shm->base = nullptr;
    }
    if (c->grp) {      // the events of this rank's mailboxes go with it (the group itself lives until its last rank is gone)
        for (int d = 0; d < 2; d++) {
            Mailbox& b = c->grp->box[(size_t)c->rank * 2 + d];
            if (b.ready) (void)hipEventDestroy(b.ready);
            if (b.done) (void)hipEventDestroy(b.done);
            b.ready = b.done = nullptr;
        }
    }
    delete c;
}

// Halo exchange: for every field, `count` floats starting at send_lo go to rank - 1 and the same amount arrives from it at recv_lo;
// likewise send_hi / recv_hi with rank + 1 (element offsets into each field).  Everything is enqueued on `stream` and nothing blocks the
// host beyond the rendezvous of the local backend.  Fields are listed by ROLE, so every rank issues the same message sequence.
int fdw_comm_exchange(fdw_comm* c, int nfields, float* const* fields, size_t send_lo, size_t recv_lo, size_t send_hi, size_t recv_hi,
                      size_t count, hipStream_t stream)
{
    if (!c || c->world == 1 || c->stub || count == 0 || nfields == 0) return FDW_OK;
    if (nfields > FDW_COMM_MAX_FIELDS) return fdw_fail(FDW_EINVAL, "exchange: %d fields", nfields);
    const bool has_lo = c->rank > 0, has_hi = c->rank < c->world - 1;
    if (c->nccl) {
        Rccl* r = rccl();
        NCCL_TRY(r->GroupStart());
        for (int f = 0; f < nfields; f++) {
            if (has_lo) {
                NCCL_TRY(r->Send(fields[f] + send_lo, count, kNcclFloat, c->rank - 1, c->nccl, stream));
                NCCL_TRY(r->Recv(fields[f] + recv_lo, count, kNcclFloat, c->rank - 1, c->nccl, stream));
            }
            if (has_hi) {
                NCCL_TRY(r->Send(fields[f] + send_hi, count, kNcclFloat, c->rank + 1, c->nccl, stream));
                NCCL_TRY(r->Recv(fields[f] + recv_hi, count, kNcclFloat, c->rank + 1, c->nccl, stream));
            }
        }
        NCCL_TRY(r->GroupEnd());
        return FDW_OK;
    }
    if (c->shm) {
        ShmSeg& sg = *c->shm;
        const size_t bytes = count * sizeof(float);
        if ((size_t)nfields * bytes > sg.hdr()->cap_bytes)
            return fdw_fail(FDW_ECOMM, "exchange: %d fields x %zu bytes exceed the segment's box capacity %zu (fdw_comm_init_shm box_bytes)", nfields, bytes, sg.hdr()->cap_bytes);
        const long seq = ++sg.seq[0];
        const int par = (int)(seq & 1);
        // 1. my blocks: device -> private send buffer behind everything queued on `stream`; the host function behind the copies moves the block
        //    into the segment and publishes the sequence number.  The send buffer and the box are free again once the previous block has been
        //    posted (my own host function has run) and taken (the neighbour has copied it out).
        for (int d = 0; d < 2; d++) {
            if (!(d == 0 ? has_lo : has_hi)) continue;
            ShmBox* b = sg.box(c->rank, d);
            if (!shm_wait([&] { return b->posted.load(std::memory_order_acquire) >= seq - 1 && b->taken.load(std::memory_order_acquire) >= seq - 1; }))
                return fdw_fail(FDW_ECOMM, "exchange %ld: rank %d waited %lds for its previous halo rows to be taken", seq, c->rank, (long)kRendezvousTimeout.count());
            b->count = count; b->nfields = nfields;
            for (int f = 0; f < nfields; f++)
                HIP_TRY(hipMemcpyAsync(sg.send_buf[d] + (size_t)f * bytes, fields[f] + (d == 0 ? send_lo : send_hi), bytes, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipLaunchHostFunc(stream, shm_post, new ShmPost{&b->posted, seq, reinterpret_cast<char*>(sg.data(c->rank, d)), sg.send_buf[d], (size_t)nfields * bytes}));
        }
        // 2. the neighbours' blocks: wait (host) for their arrival in the segment, copy them out with the CPU, say so, upload from the private buffer
        for (int d = 0; d < 2; d++) {
            if (!(d == 0 ? has_lo : has_hi)) continue;
            const int nb = d == 0 ? c->rank - 1 : c->rank + 1;
            ShmBox* b = sg.box(nb, 1 - d);
            if (!shm_wait([&] { return b->posted.load(std::memory_order_acquire) >= seq; }))
                return fdw_fail(FDW_ECOMM, "exchange %ld: rank %d waited %lds for the halo rows of rank %d", seq, c->rank, (long)kRendezvousTimeout.count(), nb);
            if (b->posted.load() != seq || b->nfields != nfields || b->count != count)
                return fdw_fail(FDW_ECOMM, "exchange %ld: rank %d and rank %d disagree (their message %ld: %d fields x %zu, mine %d x %zu)", seq, c->rank, nb,
                                b->posted.load(), b->nfields, b->count, nfields, count);
            HIP_TRY(hipEventSynchronize(sg.recv_done[d][par]));      // the upload of exchange seq - 2 out of this buffer has run
            memcpy(sg.recv_buf[d][par], sg.data(nb, 1 - d), (size_t)nfields * bytes);
            b->taken.store(seq, std::memory_order_release);
            for (int f = 0; f < nfields; f++)
                HIP_TRY(hipMemcpyAsync(fields[f] + (d == 0 ? recv_lo : recv_hi), sg.recv_buf[d][par] + (size_t)f * bytes, bytes, hipMemcpyHostToDevice, stream));
            HIP_TRY(hipEventRecord(sg.recv_done[d][par], stream));
        }
        return FDW_OK;
    }
    LocalGroup& g = *c->grp;
    const long seq = ++g.seq[c->rank];
    // 1. post what the neighbours may take, once everything queued on `stream` so far has run
    for (int d = 0; d < 2; d++) {
        if (!(d == 0 ? has_lo : has_hi)) continue;
        Mailbox& b = g.box[(size_t)c->rank * 2 + d];
        HIP_TRY(hipEventRecord(b.ready, stream));
        std::lock_guard<std::mutex> lk(g.m);
        for (int f = 0; f < nfields; f++) b.ptr[f] = fields[f] + (d == 0 ? send_lo : send_hi);
        b.count = count; b.nfields = nfields; b.posted = seq;
    }
    g.cv.notify_all();
    // 2. take the neighbours' rows: the copy runs on MY stream, behind the sender's `ready`
    for (int d = 0; d < 2; d++) {
        if (!(d == 0 ? has_lo : has_hi)) continue;
        const int nb = d == 0 ? c->rank - 1 : c->rank + 1;
        Mailbox& b = g.box[(size_t)nb * 2 + (1 - d)];      // the neighbour's box that points at me
        {
            std::unique_lock<std::mutex> lk(g.m);
            if (!g.cv.wait_for(lk, kRendezvousTimeout, [&] { return b.posted >= seq; }))
                return fdw_fail(FDW_ECOMM, "exchange %ld: rank %d waited %lds for rank %d to post its halo rows", seq, c->rank, (long)kRendezvousTimeout.count(), nb);
            if (b.posted != seq || b.nfields != nfields || b.count != count)
                return fdw_fail(FDW_ECOMM, "exchange %ld: rank %d and rank %d disagree (their message %ld: %d fields x %zu, mine %d x %zu)", seq, c->rank, nb,
                                b.posted, b.nfields, b.count, nfields, count);
        }
        HIP_TRY(hipStreamWaitEvent(stream, b.ready, 0));
        for (int f = 0; f < nfields; f++) {
            float* dst = fields[f] + (d == 0 ? recv_lo : recv_hi);
            if (b.device == c->device) HIP_TRY(hipMemcpyAsync(dst, b.ptr[f], count * sizeof(float), hipMemcpyDeviceToDevice, stream));
            else HIP_TRY(hipMemcpyPeerAsync(dst, c->device, b.ptr[f], b.device, count * sizeof(float), stream));
        }
        HIP_TRY(hipEventRecord(b.done, stream));
        {
            std::lock_guard<std::mutex> lk(g.m);
            b.taken = seq;
        }
        g.cv.notify_all();
    }
    // 3. a send is in flight until its rows have been taken: later work on `stream` must not overwrite them before
    for (int d = 0; d < 2; d++) {
        if (!(d == 0 ? has_lo : has_hi)) continue;
        Mailbox& b = g.box[(size_t)c->rank * 2 + d];
        {
            std::unique_lock<std::mutex> lk(g.m);
            if (!g.cv.wait_for(lk, kRendezvousTimeout, [&] { return b.taken >= seq; }))
                return fdw_fail(FDW_ECOMM, "exchange %ld: rank %d waited %lds for its halo rows to be taken", seq, c->rank, (long)kRendezvousTimeout.count());
        }
        HIP_TRY(hipStreamWaitEvent(stream, b.done, 0));
    }
    return FDW_OK;
}

// Host-side barrier (local backend) / device-side all-reduce of one double per rank: MAX when op = 1, SUM when op = 0.
extern "C" int fdw_comm_allreduce(fdw_comm* c, double* value, int op_max)
{
    if (!c || !value) return fdw_fail(FDW_EINVAL, "allreduce: NULL argument");
    if (c->stub || (c->world == 1 && !c->nccl)) return FDW_OK;      // (a one-rank RCCL communicator still goes through ncclAllReduce: the only way a one-GPU box can exercise that call)
    if (c->nccl) {
        // RCCL reduces device memory: one double per rank through a device word, ncclSum / ncclMax on ncclFloat64 (any value survives)
        HIP_TRY(hipSetDevice(c->device));
        double* d = nullptr;
        HIP_TRY(hipMalloc((void**)&d, sizeof(double)));
        hipError_t e = hipMemcpy(d, value, sizeof(double), hipMemcpyHostToDevice);
        ncclResult_t rc = 0;
        if (e == hipSuccess) rc = rccl()->AllReduce(d, d, 1, kNcclDouble, op_max ? kNcclMax : kNcclSum, c->nccl, nullptr);
        if (e == hipSuccess && rc == 0) e = hipStreamSynchronize(nullptr);
        if (e == hipSuccess && rc == 0) e = hipMemcpy(value, d, sizeof(double), hipMemcpyDeviceToHost);
        (void)hipFree(d);
        if (rc != 0) return fdw_fail(FDW_ECOMM, "ncclAllReduce failed: %s", rccl()->GetErrorString(rc));
        if (e != hipSuccess) return fdw_fail(FDW_EHIP, "allreduce: %s", hipGetErrorString(e));
        return FDW_OK;
    }
    if (c->shm) {
        ShmHeader* h = c->shm->hdr();
        const long gen = h->red_gen.load(std::memory_order_acquire);
        h->red[c->rank] = *value;
        if (h->red_count.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
            double acc = op_max ? h->red[0] : 0.0;
            for (int r = 0; r < c->world; r++) acc = op_max ? (h->red[r] > acc ? h->red[r] : acc) : acc + h->red[r];
            h->red_result = acc;
            h->red_count.store(0, std::memory_order_relaxed);
            h->red_gen.store(gen + 1, std::memory_order_release);
        } else if (!shm_wait([&] { return h->red_gen.load(std::memory_order_acquire) != gen; })) {
            return fdw_fail(FDW_ECOMM, "allreduce: rank %d waited %lds for the other ranks", c->rank, (long)kRendezvousTimeout.count());
        }
        *value = h->red_result;
        return FDW_OK;
    }
    LocalGroup& g = *c->grp;
    std::unique_lock<std::mutex> lk(g.m);
    const long gen = g.red_gen;
    g.red[c->rank] = *value;
    if (++g.red_count == g.world) {
        double acc = op_max ? g.red[0] : 0.0;
        for (double v : g.red) acc = op_max ? (v > acc ? v : acc) : acc + v;
        g.red_result = acc;
        g.red_count = 0;
        g.red_gen++;
        g.cv.notify_all();
    } else {
        if (!g.cv.wait_for(lk, kRendezvousTimeout, [&] { return g.red_gen != gen; })) {
            g.red_count--;
            return fdw_fail(FDW_ECOMM, "allreduce: rank %d waited %lds for the other ranks", c->rank, (long)kRendezvousTimeout.count());
        }
    }
    *value = g.red_result;
    return FDW_OK;
}

extern "C" int fdw_comm_barrier(fdw_comm* c)
{
    double v = 0.0;
    return fdw_comm_allreduce(c, &v, 0);
}

// One message to the own rank through the backend's send / receive path on a stream of its own: checks that librccl resolves,
// that the communicator works and that the transfer is ordered with the stream (a fill before, a read after).
extern "C" int fdw_comm_selftest(fdw_comm* c)
{
    if (!c) return fdw_fail(FDW_EINVAL, "comm is NULL");
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = 1 << 16;
    float *a = nullptr, *b = nullptr;
    HIP_TRY(hipMalloc((void**)&a, n * sizeof(float)));
    if (hipMalloc((void**)&b, n * sizeof(float)) != hipSuccess) {
        (void)hipFree(a);
        return fdw_fail(FDW_ENOMEM, "hipMalloc failed");
    }
    hipStream_t s = nullptr;
    int rc = FDW_OK;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; i++) h[i] = (float)(i % 977) + 0.5f * c->rank;
    hipError_t e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMemcpyAsync(a, h.data(), n * sizeof(float), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(b, 0, n * sizeof(float), s);
    if (e == hipSuccess) {
        if (c->nccl) {
            Rccl* r = rccl();
            ncclResult_t q = r->GroupStart();
            if (q == 0) q = r->Send(a, n, kNcclFloat, c->rank, c->nccl, s);
            if (q == 0) q = r->Recv(b, n, kNcclFloat, c->rank, c->nccl, s);
            if (q == 0) q = r->GroupEnd();
            if (q != 0) rc = fdw_fail(FDW_ECOMM, "self send/recv failed: %s", r->GetErrorString(q));
        } else {
            e = hipMemcpyAsync(b, a, n * sizeof(float), hipMemcpyDeviceToDevice, s);
        }
    }
    std::vector<float> back(n, -1.0f);
    if (e == hipSuccess && rc == FDW_OK) e = hipMemcpyAsync(back.data(), b, n * sizeof(float), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && rc == FDW_OK) e = hipStreamSynchronize(s);
    if (s) (void)hipStreamDestroy(s);
    (void)hipFree(a);
    (void)hipFree(b);
    if (rc != FDW_OK) return rc;
    if (e != hipSuccess) return fdw_fail(FDW_EHIP, "comm selftest: %s", hipGetErrorString(e));
    if (memcmp(back.data(), h.data(), n * sizeof(float)) != 0) return fdw_fail(FDW_ECOMM, "comm selftest: the received block differs from the one sent");
    return FDW_OK;
}
