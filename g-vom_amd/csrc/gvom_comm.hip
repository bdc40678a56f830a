// gvom_comm.hip -- the transport between the ranks of a sharded map (include/gvom_hip.h, "one map
// sharded over the GPUs of a node"): RCCL over xGMI, bound directly (no PyTorch).
//
//   * device data: grouped ncclSend / ncclRecv (the scan's quads and endpoints: a sparse all-to-all
//     whose sizes are only known after the trace) and an in-place ncclAllGather (the combine's
//     height-map rows), on the LIBRARY's stream -- no host synchronisation in between;
//   * host data: the ranks are the processes of ONE node, so the small per-scan vectors (counts,
//     in-grid flags) and the ncclUniqueId travel through a POSIX shared-memory segment
//     (/dev/shm/<name>): ~1 us, no GPU involved.  Double-buffered slots, sequence numbers, C11 atomics.
//
// librccl.so is loaded on first use (dlopen), so a single-GPU user of libgvom_hip.so does not
// depend on it.  The reference has no multi-GPU path; there is no interface to mirror.
#include "../../include/gvom_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <immintrin.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define VIS __attribute__((visibility("default")))
#define GVOM_COMM_MAX_RANKS 64
#define GVOM_COMM_MAX_VALUES 160          // int64 values per rank and exchange (>= 2 * ranks + 4)

namespace {

struct Rccl {
    void *lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load(std::string &err)
    {
        if (lib) return true;
        lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) { err = std::string("cannot load librccl.so: ") + dlerror(); return false; }
#define SYM(field, name)                                                                     \
        field = (decltype(field))dlsym(lib, name);                                            \
        if (!field) { err = std::string("librccl.so lacks ") + name; return false; }
        SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
        SYM(GroupStart, "ncclGroupStart") SYM(GroupEnd, "ncclGroupEnd") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
        SYM(AllGather, "ncclAllGather") SYM(GetErrorString, "ncclGetErrorString")
#undef SYM
        return true;
    }
};

struct Slot {                                  // one rank's mailbox for one exchange parity
    std::atomic<uint64_t> seq;
    int64_t values[GVOM_COMM_MAX_VALUES];
    char pad[64];
};

struct Segment {                               // the shared-memory rendezvous of one communicator
    std::atomic<uint32_t> magic;               // set last by rank 0
    uint32_t world;
    std::atomic<uint32_t> id_ready;
    std::atomic<uint32_t> attached;
    double created_s;                          // CLOCK_REALTIME at creation: a segment left behind by a crashed run is not joined
    int64_t creator_pid;                       // rank 0's process and its start time (/proc/<pid>/stat field 22): a joiner only
    uint64_t creator_start;                    // attaches to a segment whose creator is ALIVE -- a crashed job's is not
    ncclUniqueId id;
    Slot slots[2][GVOM_COMM_MAX_RANKS];
};

inline double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
inline double wall_s() { timespec t; clock_gettime(CLOCK_REALTIME, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

// start time of a process in clock ticks since boot (0: unknown / no such process); with the pid it names one
// process for the lifetime of the machine, whatever pid reuse does
uint64_t proc_start_time(int64_t pid)
{
    char path[64], buf[1024];
    snprintf(path, sizeof path, "/proc/%lld/stat", (long long)pid);
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char *p = strrchr(buf, ')');                 // the command name may contain spaces and parentheses
    if (!p) return 0;
    int field = 2;                                     // p points at the end of field 2
    for (++p; *p; ++p) {
        if (*p == ' ') { if (++field == 22) return strtoull(p + 1, nullptr, 10); }
    }
    return 0;
}
bool creator_alive(const Segment *sg)
{
    const int64_t pid = sg->creator_pid;
    if (pid <= 0) return false;
    if (kill((pid_t)pid, 0) != 0 && errno != EPERM) return false;
    const uint64_t st = proc_start_time(pid);
    return st == 0 || sg->creator_start == 0 || st == sg->creator_start;    // (no /proc: the pid test alone)
}

}  // namespace

struct gvom_comm {
    int rank = 0, world = 1, device = 0;
    Rccl rccl;
    ncclComm_t nccl = nullptr;
    Segment *seg = nullptr;
    std::string shm_name, err;
    uint64_t calls = 0;                        // host exchanges so far
    // rendezvous / host-exchange patience: the ranks of a job start seconds to minutes apart on a cold box
    // (first import of the interpreter's packages); GVOM_COMM_TIMEOUT_S overrides
    double timeout_s = 600.0;
};

namespace {

#define NCCLCHK(c, call)                                                                       \
    do {                                                                                       \
        ncclResult_t r_ = (call);                                                              \
        if (r_ != ncclSuccess) {                                                               \
            (c)->err = std::string(#call) + " failed: " + (c)->rccl.GetErrorString(r_);        \
            return GVOM_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)

}  // namespace

extern "C" {

// name: the same string on every rank of the communicator and unique to it on the node (e.g.
// "gvom_<master port>"); rank 0 creates /dev/shm/<name>, the others wait for it.  device: the HIP
// device this rank's handle lives on.  world == 1 is allowed (collectives degenerate to copies).
// device < 0: HOST-ONLY communicator -- the shared-memory rendezvous and gvom_comm_exchange_host /
// gvom_comm_barrier without RCCL or any HIP call (the CPU tests exercise the multi-process rendezvous
// with it); the device collectives return GVOM_ERR_INVALID on it.
VIS int gvom_comm_create(int rank, int world, int device, const char *name, gvom_comm_t **out)
{
    if (!out || !name || world < 1 || world > GVOM_COMM_MAX_RANKS || rank < 0 || rank >= world) return GVOM_ERR_INVALID;
    *out = nullptr;
    gvom_comm *c = new gvom_comm();
    c->rank = rank; c->world = world; c->device = device;
    c->shm_name = std::string("/") + name;
    if (const char *t = getenv("GVOM_COMM_TIMEOUT_S")) { const double v = atof(t); if (v > 0.0) c->timeout_s = v; }
    bool created = false;                                              // rank 0: the name exists and is ours
    auto fail = [&](const std::string &why, int code) {
        fprintf(stderr, "gvom_comm_create(rank %d of %d): %s\n", rank, world, why.c_str());
        if (c->seg) munmap(c->seg, sizeof(Segment));
        if (created) shm_unlink(c->shm_name.c_str());                  // never leave a segment behind for a later run to join
        delete c;
        return code;
    };
    const bool host_only = device < 0;
    if (!host_only) {
        if (!c->rccl.load(c->err)) return fail(c->err, GVOM_ERR_NO_DEVICE);
        if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed", GVOM_ERR_NO_DEVICE);
    }
    if (rank == 0) {
        shm_unlink(c->shm_name.c_str());                               // a stale segment of a crashed run
        int fd = shm_open(c->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        created = fd >= 0;
        if (fd < 0 || ftruncate(fd, sizeof(Segment)) != 0) { if (fd >= 0) close(fd); return fail("cannot create the shared-memory rendezvous", GVOM_ERR_HIP); }
        void *m = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (m == MAP_FAILED) return fail("mmap of the rendezvous failed", GVOM_ERR_HIP);
        c->seg = (Segment *)m;
        memset((void *)c->seg, 0, sizeof(Segment));
        c->seg->world = (uint32_t)world;
        c->seg->created_s = wall_s();
        c->seg->creator_pid = (int64_t)getpid();
        c->seg->creator_start = proc_start_time((int64_t)getpid());
        if (!host_only && c->rccl.GetUniqueId(&c->seg->id) != ncclSuccess) return fail("ncclGetUniqueId failed", GVOM_ERR_HIP);
        c->seg->id_ready.store(1, std::memory_order_release);
        c->seg->magic.store(0x47564f4du, std::memory_order_release);
    } else {
        // join the segment rank 0 has created for THIS run: complete (magic), fresh, same world size, created by a
        // process that is still alive (a job that crashed minutes ago leaves a complete, fresh-looking segment behind:
        // its ncclUniqueId would hang ncclCommInitRank), and still the one the name refers to once accepted
        const double deadline = now_s() + c->timeout_s;
        while (true) {
            int fd = shm_open(c->shm_name.c_str(), O_RDWR, 0600);
            struct stat sb;
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= sizeof(Segment)) {
                void *m = mmap(nullptr, sizeof(Segment), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                if (m != MAP_FAILED) {
                    Segment *sg = (Segment *)m;
                    const double t_end = now_s() + 0.05;
                    while (sg->magic.load(std::memory_order_acquire) != 0x47564f4du && now_s() < t_end) usleep(100);
                    const double age = wall_s() - sg->created_s;
                    if (sg->magic.load(std::memory_order_acquire) == 0x47564f4du && sg->id_ready.load(std::memory_order_acquire) == 1u &&
                        age > -5.0 && age < c->timeout_s + 60.0 && sg->attached.load(std::memory_order_acquire) < (uint32_t)world &&
                        creator_alive(sg)) {
                        // rank 0 of this run may have replaced the segment between our open and now
                        struct stat nb;
                        const int fd2 = shm_open(c->shm_name.c_str(), O_RDWR, 0600);
                        const bool same = fd2 >= 0 && fstat(fd2, &nb) == 0 && nb.st_ino == sb.st_ino && nb.st_dev == sb.st_dev;
                        // (name gone: the last rank to attach has unlinked it -- only possible for the live segment)
                        const bool gone = fd2 < 0 && errno == ENOENT;
                        if (fd2 >= 0) close(fd2);
                        if (same || gone) {
                            close(fd);
                            c->seg = sg;
                            break;
                        }
                    }
                    munmap(m, sizeof(Segment));
                }
            }
            if (fd >= 0) close(fd);
            if (now_s() > deadline) return fail("rank 0 never created the shared-memory rendezvous", GVOM_ERR_HIP);
            usleep(500);
        }
        if (c->seg->world != (uint32_t)world) return fail("world size differs from rank 0's", GVOM_ERR_INVALID);
    }
    if (!host_only) {
        ncclUniqueId id;
        memcpy(&id, &c->seg->id, sizeof id);
        ncclResult_t r = c->rccl.CommInitRank(&c->nccl, world, id, rank);
        if (r != ncclSuccess) return fail(std::string("ncclCommInitRank failed: ") + c->rccl.GetErrorString(r), GVOM_ERR_HIP);
    }
    // the name can go once everybody is attached (the mapping stays valid)
    if (c->seg->attached.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) shm_unlink(c->shm_name.c_str());
    *out = c;
    return GVOM_OK;
}

VIS void gvom_comm_destroy(gvom_comm_t *c)
{
    if (!c) return;
    if (c->nccl) { hipSetDevice(c->device); c->rccl.CommDestroy(c->nccl); }
    if (c->seg) munmap(c->seg, sizeof(Segment));
    if (c->rank == 0) shm_unlink(c->shm_name.c_str());                 // harmless if already gone
    delete c;
}

VIS const char *gvom_comm_last_error(gvom_comm_t *c) { return c ? c->err.c_str() : "null communicator"; }

// host-side all-gather of k int64 values per rank (k <= GVOM_COMM_MAX_VALUES): all[r*k + j] = rank
// r's mine[j].  Collective; ~1 us on one node.
VIS int gvom_comm_exchange_host(gvom_comm_t *c, const int64_t *mine, int k, int64_t *all)
{
    if (!c || !mine || !all || k < 0 || k > GVOM_COMM_MAX_VALUES) return GVOM_ERR_INVALID;
    const uint64_t call = ++c->calls;
    Slot *slots = c->seg->slots[call & 1];
    // (a rank can only be two calls ahead of the slowest one: it passes call n + 1 only after every
    // rank has published n + 1, i.e. has finished reading call n -- so two parities are enough)
    memcpy(slots[c->rank].values, mine, (size_t)k * 8);
    slots[c->rank].seq.store(call, std::memory_order_release);
    const double deadline = now_s() + c->timeout_s;
    for (int r = 0; r < c->world; ++r) {
        unsigned spins = 0;
        double slow_since = 0.0;
        while (slots[r].seq.load(std::memory_order_acquire) != call) {
            _mm_pause();
            if ((++spins & 0xfff) == 0) {
                // a peer that is milliseconds late is minutes late as likely as not (it died, or it is still importing):
                // stop burning the core
                const double t = now_s();
                if (slow_since == 0.0) slow_since = t;
                if (t > deadline) { c->err = "host exchange timed out (a rank is missing)"; return GVOM_ERR_HIP; }
                if (t - slow_since > 5e-3) usleep(t - slow_since > 1.0 ? 1000 : 50);
            }
        }
        memcpy(all + (size_t)r * k, slots[r].values, (size_t)k * 8);
    }
    return GVOM_OK;
}

VIS int gvom_comm_barrier(gvom_comm_t *c)
{
    int64_t z = 0, all[GVOM_COMM_MAX_RANKS];
    return gvom_comm_exchange_host(c, &z, 1, all);
}

// The scan's exchange: SEND regions of `h` (gvom_shard_scan_local) -> the owners' RECV regions.
// send_* / recv_*: [world] counts (quads: 4-byte id + 1 KiB each; endpoints: 8 bytes each); the
// caller has exchanged them (gvom_comm_exchange_host) and called gvom_shard_recv_reserve.  One
// grouped ncclSend / ncclRecv on the handle's stream; returns without synchronising.
VIS int gvom_comm_exchange_scan(gvom_comm_t *c, gvom_t *h, const int64_t *send_quads, const int64_t *send_eps,
                                const int64_t *recv_quads, const int64_t *recv_eps)
{
    if (!c || !h || !send_quads || !send_eps || !recv_quads || !recv_eps) return GVOM_ERR_INVALID;
    if (c->world == 1) return GVOM_OK;
    if (!c->nccl) { c->err = "host-only communicator: no device collectives"; return GVOM_ERR_INVALID; }
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return GVOM_ERR_HIP; }
    hipStream_t st = (hipStream_t)gvom_stream(h);
    NCCLCHK(c, c->rccl.GroupStart());
    // (inside the group every failure path must still close it: an open group stays with the thread)
    int rc = GVOM_OK;
    auto xfer = [&](bool send, int which, int p, size_t bytes) {
        if (rc != GVOM_OK || bytes == 0) return;
        void *ptr = nullptr;
        int64_t cap = 0;
        if (gvom_shard_buffer(h, which, p, &ptr, &cap) || (int64_t)bytes > cap) {
            c->err = "exchange region missing or smaller than the announced count";
            rc = GVOM_ERR_INVALID;
            return;
        }
        const ncclResult_t r = send ? c->rccl.Send(ptr, bytes, ncclUint8, p, c->nccl, st) : c->rccl.Recv(ptr, bytes, ncclUint8, p, c->nccl, st);
        if (r != ncclSuccess) { c->err = std::string(send ? "ncclSend" : "ncclRecv") + " failed: " + c->rccl.GetErrorString(r); rc = GVOM_ERR_HIP; }
    };
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        if (send_quads[p] < 0 || send_eps[p] < 0 || recv_quads[p] < 0 || recv_eps[p] < 0) { c->err = "negative count"; rc = GVOM_ERR_INVALID; break; }
        xfer(true, GVOM_XBUF_SEND_IDS, p, (size_t)send_quads[p] * 4);
        xfer(true, GVOM_XBUF_SEND_QUADS, p, (size_t)send_quads[p] * 1024);
        xfer(true, GVOM_XBUF_SEND_EPS, p, (size_t)send_eps[p] * 8);
        xfer(false, GVOM_XBUF_RECV_IDS, p, (size_t)recv_quads[p] * 4);
        xfer(false, GVOM_XBUF_RECV_QUADS, p, (size_t)recv_quads[p] * 1024);
        xfer(false, GVOM_XBUF_RECV_EPS, p, (size_t)recv_eps[p] * 8);
    }
    const ncclResult_t ge = c->rccl.GroupEnd();
    if (rc != GVOM_OK) return rc;
    if (ge != ncclSuccess) { c->err = std::string("ncclGroupEnd failed: ") + c->rccl.GetErrorString(ge); return GVOM_ERR_HIP; }
    return GVOM_OK;
}

// Statistics handles: the returns every other rank needs of this one (gvom_shard_stats_counts) -> their receive
// regions (gvom_shard_stats_reserve), bytes_per_return = 12 (float32 clouds) or 24.  One grouped send / recv.
VIS int gvom_comm_exchange_stats(gvom_comm_t *c, gvom_t *h, const int64_t *send_returns, const int64_t *recv_returns,
                                 int bytes_per_return)
{
    if (!c || !h || !send_returns || !recv_returns || (bytes_per_return != 12 && bytes_per_return != 24)) return GVOM_ERR_INVALID;
    if (c->world == 1) return GVOM_OK;
    if (!c->nccl) { c->err = "host-only communicator: no device collectives"; return GVOM_ERR_INVALID; }
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return GVOM_ERR_HIP; }
    hipStream_t st = (hipStream_t)gvom_stream(h);
    NCCLCHK(c, c->rccl.GroupStart());
    int rc = GVOM_OK;
    for (int p = 0; p < c->world && rc == GVOM_OK; ++p) {
        if (p == c->rank) continue;
        for (int dir = 0; dir < 2 && rc == GVOM_OK; ++dir) {
            const int64_t cnt = dir == 0 ? send_returns[p] : recv_returns[p];
            if (cnt < 0) { c->err = "negative count"; rc = GVOM_ERR_INVALID; break; }
            if (cnt == 0) continue;
            void *ptr = nullptr;
            int64_t cap = 0;
            const size_t bytes = (size_t)cnt * (size_t)bytes_per_return;
            if (gvom_shard_buffer(h, dir == 0 ? GVOM_XBUF_SEND_RETURNS : GVOM_XBUF_RECV_RETURNS, p, &ptr, &cap) || (int64_t)bytes > cap) {
                c->err = "statistics exchange region missing or smaller than the announced count";
                rc = GVOM_ERR_INVALID;
                break;
            }
            const ncclResult_t r = dir == 0 ? c->rccl.Send(ptr, bytes, ncclUint8, p, c->nccl, st) : c->rccl.Recv(ptr, bytes, ncclUint8, p, c->nccl, st);
            if (r != ncclSuccess) { c->err = std::string("ncclSend/Recv failed: ") + c->rccl.GetErrorString(r); rc = GVOM_ERR_HIP; }
        }
    }
    const ncclResult_t ge = c->rccl.GroupEnd();
    if (rc != GVOM_OK) return rc;
    if (ge != ncclSuccess) { c->err = std::string("ncclGroupEnd failed: ") + c->rccl.GetErrorString(ge); return GVOM_ERR_HIP; }
    return GVOM_OK;
}

// The combine's exchange: in-place all-gather of the handle's [height | inferred height | positive
// density] rows (GVOM_BUF_HEIGHT_MAPS; a rank's rows are one contiguous block) on the handle's stream.
VIS int gvom_comm_allgather_rows(gvom_comm_t *c, gvom_t *h)
{
    if (!c || !h) return GVOM_ERR_INVALID;
    if (!c->nccl) { c->err = "host-only communicator: no device collectives"; return GVOM_ERR_INVALID; }
    void *ptr = nullptr;
    int64_t bytes = 0, row = 0;
    int rc = gvom_device_buffer(h, GVOM_BUF_HEIGHT_MAPS, &ptr, &bytes, &row);
    if (rc) return rc;
    if (bytes % c->world) { c->err = "height-map rows do not divide among the ranks"; return GVOM_ERR_INVALID; }
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return GVOM_ERR_HIP; }
    const size_t share = (size_t)(bytes / c->world);
    NCCLCHK(c, c->rccl.AllGather((char *)ptr + share * c->rank, ptr, share, ncclUint8, c->nccl, (hipStream_t)gvom_stream(h)));
    return GVOM_OK;
}

VIS int gvom_comm_rank(gvom_comm_t *c) { return c ? c->rank : -1; }
VIS int gvom_comm_world(gvom_comm_t *c) { return c ? c->world : -1; }

}  // extern "C"
