// gvom_comm.hip -- the transport between the ranks of a sharded map (include/gvom_hip.h, "one map
// sharded over the GPUs of a node"): RCCL over xGMI, bound directly (no PyTorch).
//
//   * device data: grouped ncclSend / ncclRecv (the scan's quads and endpoints: a sparse all-to-all
//     whose sizes are only known after the trace) and an in-place ncclAllGather (the combine's
//     height-map rows), on the LIBRARY's stream -- no host synchronisation in between;
//   * device data, second transport (GVOM_TRANSPORT_PEER): PEER COPIES -- every rank exports its send regions
//     (hipIpcGetMemHandle; handles and per-destination offsets travel through the shared-memory segment), the
//     receiver maps them once (hipIpcOpenMemHandle, lazy peer access) and PULLS its bytes with hipMemcpyAsync on
//     its own handle's stream: over xGMI between the GPUs of a node, inside one GPU's memory when several ranks
//     share a device.  No RCCL involved: it is what a one-GPU box can run with several rank PROCESSES (RCCL
//     refuses two ranks on one device), and what GVOM_TRANSPORT_AUTO falls back to when RCCL cannot initialise.
//     Three rules keep the HSA runtime's inter-process memory honest (each found as a silently different map or a refused
//     call under tests/shard_procs.py's churn hook): an allocation is exported ONCE and a mapping
//     opened ONCE (both kept until the communicator goes), exported regions are whole multiples of 2 MiB, and a region a
//     peer may have mapped is never given back to the allocator while the process lives (gvom_capi.hip);
//   * device data, third transport (GVOM_TRANSPORT_LOOPBACK): RCCL LOOPBACK -- the ranks are THREADS of one process that share
//     one GPU (RCCL refuses two ranks of ONE communicator on one device), each with a 1-rank communicator of its own; what
//     the RCCL transport moves with ncclSend on the sender and ncclRecv on the receiver, the RECEIVER moves with
//     ncclSend(peer's send region, self) + ncclRecv(own receive region, self) in one group on its handle's stream, through the
//     same group / capacity-check / byte-count code (WireGroup below), and the combine ends in the in-place ncclAllGather.
//     It exists so that a one-GPU box executes the RCCL calls, byte counts and stream ordering of the product path;
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
#include <chrono>
#include <condition_variable>
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <signal.h>
#include <immintrin.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define VIS __attribute__((visibility("default")))
#if defined(GVOM_DIAG) || defined(GVOM_TEST_HOOKS)
#define GVOM_HOOKS 1                           // (as gvom_internal.h: the test hooks of include/gvom_hip_test.h)
#endif
#define GVOM_COMM_MAX_RANKS 64
#define GVOM_COMM_MAX_VALUES 208          // int64 values per rank and exchange (statistics handles exchange 3 * ranks + 3)
static_assert(GVOM_COMM_MAX_VALUES >= 3 * GVOM_COMM_MAX_RANKS + 4, "a communicator of GVOM_COMM_MAX_RANKS ranks must be able to exchange its counts");
extern "C" int gvom_shard_renew_region(gvom_t *h, int which);    // (gvom_capi.hip; same library)

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

struct PeerFlags {                             // one 64-byte line per rank
    std::atomic<uint64_t> pulled_scan;         // "I have pulled what the others held for me in scan exchange k" (they may repack)
    std::atomic<uint64_t> pulled_stats;        // the same for the statistics returns
    std::atomic<uint64_t> rows_ready;          // "my rows of combine k are complete" (the others may pull them)
    std::atomic<uint64_t> pulled_rows;         // "I have pulled everybody's rows of combine k" (they may fuse again)
    char pad[32];
};

#define GVOM_PEER_KINDS 5                   // exported regions: send ids, send quads, send endpoints, send returns, height-map rows
struct PeerExport {                            // one exported region of one rank (peer transport)
    hipIpcMemHandle_t handle;                  // of the ALLOCATION the region lies in
    uint64_t generation;                       // changes whenever `handle` does (0: nothing exported yet)
    uint64_t size;                             // of the allocation
    uint64_t offset[GVOM_COMM_MAX_RANKS];      // byte offset, inside the allocation, of the part meant for rank d
};

struct LoopRegion {                            // loopback transport: one rank's regions of one kind
    uint64_t ptr[GVOM_COMM_MAX_RANKS];         // device address of the part meant for rank d (0: nothing)
    uint64_t cap[GVOM_COMM_MAX_RANKS];         // its capacity in bytes
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
    // (the fields above keep their offsets: tests/test_comm_rendezvous.py writes stale segments byte by byte)
    std::atomic<uint32_t> rccl_failed;         // GVOM_TRANSPORT_AUTO: ranks whose ncclCommInitRank failed
    std::atomic<uint32_t> poison;              // rank + 1 of a rank whose device exchange failed: nobody waits for it again
    std::atomic<uint32_t> arrived;             // ranks that have mapped the segment (GVOM_TRANSPORT_AUTO waits for all before ncclCommInitRank)
    int64_t rank_pid[GVOM_COMM_MAX_RANKS];     // the ranks' processes (0 until attached), so that a wait for a rank that has
    uint64_t rank_start[GVOM_COMM_MAX_RANKS];  // died ends in an error at once instead of after the timeout
    Slot slots[2][GVOM_COMM_MAX_RANKS];
    PeerExport exports[GVOM_COMM_MAX_RANKS][GVOM_PEER_KINDS];   // written by their rank between two barriers only
    // peer transport, asynchronous form: exchange numbers written BY THE GPU of the rank they belong to (a one-thread kernel on
    // its stream, system-scope release) and read by the other ranks' hosts -- the segment is registered with HIP for that
    std::atomic<uint32_t> sync_only;           // some rank could not register the segment: everybody keeps the host-synchronised form
    PeerFlags flags[GVOM_COMM_MAX_RANKS];
    // recovery of refused imports: import_failed[s][kind] = the recovery round in which some rank could not open rank s's
    // export `kind` (written between two barriers, read after the second: every rank sees the same table)
    std::atomic<uint64_t> import_failed[GVOM_COMM_MAX_RANKS][GVOM_PEER_KINDS];
    // loopback transport (the ranks are threads of ONE process): loop[s][kind] = where rank s holds what it has of `kind` for
    // every destination, as plain device addresses (written by s between two barriers, read by the receivers behind the second)
    LoopRegion loop[GVOM_COMM_MAX_RANKS][GVOM_PEER_KINDS];
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
// a process that can still take part in an exchange: it exists, is the one recorded (start time) and is not a zombie
bool proc_running(int64_t pid, uint64_t start)
{
    if (pid <= 0) return true;                                         // unknown: assume so
    char path[64], buf[1024];
    snprintf(path, sizeof path, "/proc/%lld/stat", (long long)pid);
    FILE *f = fopen(path, "r");
    if (!f) return kill((pid_t)pid, 0) == 0 || errno == EPERM;         // (no /proc: the pid test alone)
    const size_t n = fread(buf, 1, sizeof buf - 1, f);
    fclose(f);
    buf[n] = 0;
    const char *p = strrchr(buf, ')');
    if (!p || !p[1] || !p[2]) return true;
    if (p[2] == 'Z' || p[2] == 'X' || p[2] == 'x') return false;
    const uint64_t st = proc_start_time(pid);
    return st == 0 || start == 0 || st == start;
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

// PROCESS-WIDE, for the life of the process (the three rules at the head of this file hold across communicators too: an
// allocation is exported once and a mapping opened once, whoever asks):
struct PeerImport {                            // a peer process's exported allocation, mapped into this process
    int64_t pid = 0; uint64_t start = 0;       // the exporting process
    uint64_t generation = 0;                   // its name for the allocation (gvom_region_generation: unique in that process)
    void *base = nullptr;
};
struct PeerOwn {                               // an allocation this process has exported
    void *base = nullptr;
    size_t size = 0;
    uint64_t generation = 0;
    hipIpcMemHandle_t handle;
};
static std::mutex g_ipc_mu;
static std::vector<PeerImport> g_imports;
static std::vector<PeerOwn> g_own;
// test hook (lib/libgvom_hip_test.so only; include/gvom_hip_test.h): GVOM_TEST_IPC_REFUSE="export:N" / "import:N" [",rank:R"] -- the
// N-th export (import) this process attempts is answered as the HSA runtime answers when it refuses (every repetition of it
// too), so that the recovery below runs.  The production library has no such switch: nothing is ever injected.
struct IpcFault { int export_n = 0, import_n = 0, rank = -1; std::atomic<int> exports{0}, imports{0}; bool parsed = false; };
static IpcFault g_fault;
static void parse_fault()
{
#ifdef GVOM_HOOKS
    if (g_fault.parsed) return;
    g_fault.parsed = true;
    const char *e = getenv("GVOM_TEST_IPC_REFUSE");
    if (!e) return;
    if (const char *p = strstr(e, "export:")) g_fault.export_n = atoi(p + 7);
    if (const char *p = strstr(e, "import:")) g_fault.import_n = atoi(p + 7);
    if (const char *p = strstr(e, "rank:")) g_fault.rank = atoi(p + 5);
#endif
}

struct gvom_comm {
    int rank = 0, world = 1, device = 0;
    int transport = GVOM_TRANSPORT_RCCL;       // the one in use (never AUTO)
    Rccl rccl;
    ncclComm_t nccl = nullptr;
    uint64_t recover_round = 0;                // rounds of the import recovery so far (the same on every rank)
    void *src_all[GVOM_COMM_MAX_RANKS][GVOM_PEER_KINDS] = {};       // scratch of one exchange: where each source's export is mapped
    bool need_all[GVOM_COMM_MAX_RANKS][GVOM_PEER_KINDS] = {};
    uint64_t peer_bytes = 0, peer_copies = 0;  // pulled so far (diagnostics)
    uint64_t wire_ops = 0, wire_bytes = 0, wire_groups = 0, wire_allgathers = 0;   // RCCL calls issued so far (ncclSend + ncclRecv, their bytes, groups, all-gathers)
    uint64_t export_seq = 0, peer_open_retries = 0, peer_renewed = 0;
    // asynchronous form of the peer transport (no host wait for the GPU inside an exchange)
    bool async = false, registered = false;
    Segment *seg_dev = nullptr;                // device view of the registered segment
    uint64_t scan_x = 0, stats_x = 0, rows_x = 0;   // exchanges so far (the same on every rank: they are collectives)
    Segment *seg = nullptr;
    std::string shm_name, err;
    uint64_t calls = 0;                        // host exchanges so far
    std::vector<int64_t> scan_table;           // gvom_comm_process_pointcloud: every rank's counts of the scan in hand
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

#define HIPCHK_C(c, call)                                                                      \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            (c)->err = std::string(#call) + " failed: " + hipGetErrorString(e_);               \
            return GVOM_ERR_HIP;                                                               \
        }                                                                                      \
    } while (0)

// ---- one grouped RCCL point-to-point exchange on a handle's stream -----------------------------------
// The RCCL transport (peers = the other ranks of the communicator) and the loopback transport (peer = this rank itself, the
// source a same-process peer's send region) issue their transfers through this one object: zero-byte transfers are skipped,
// regions of the handle are capacity-checked against the announced byte count, counts travel as size_t bytes of ncclUint8,
// and a group that has been opened is closed on every path (an open group stays with the thread).
struct WireGroup {
    gvom_comm *c;
    hipStream_t st;
    int rc = GVOM_OK;
    bool open = false;
    int begin()
    {
        const ncclResult_t r = c->rccl.GroupStart();
        if (r != ncclSuccess) { c->err = std::string("ncclGroupStart failed: ") + c->rccl.GetErrorString(r); return rc = GVOM_ERR_HIP; }
        open = true;
        return GVOM_OK;
    }
    void fail(int code, const char *why) { if (rc == GVOM_OK) { rc = code; c->err = why; } }
    // the part of region `which` of handle h that belongs to rank p, if it can hold `bytes`
    void *region(gvom_t *h, int which, int p, size_t bytes)
    {
        void *ptr = nullptr;
        int64_t cap = 0;
        if (gvom_shard_buffer(h, which, p, &ptr, &cap) || !ptr || (int64_t)bytes > cap) {
            fail(GVOM_ERR_INVALID, "exchange region missing or smaller than the announced count");
            return nullptr;
        }
        return ptr;
    }
    void wire(bool send, void *ptr, size_t bytes, int peer)
    {
        if (rc != GVOM_OK || bytes == 0 || !ptr) return;
        const ncclResult_t r = send ? c->rccl.Send(ptr, bytes, ncclUint8, peer, c->nccl, st) : c->rccl.Recv(ptr, bytes, ncclUint8, peer, c->nccl, st);
        if (r != ncclSuccess) { c->err = std::string(send ? "ncclSend" : "ncclRecv") + " failed: " + c->rccl.GetErrorString(r); rc = GVOM_ERR_HIP; return; }
        ++c->wire_ops; c->wire_bytes += bytes;
    }
    // a transfer out of / into this handle's own region
    void xfer(bool send, gvom_t *h, int which, int p, size_t bytes, int peer)
    {
        if (rc != GVOM_OK || bytes == 0) return;
        wire(send, region(h, which, p, bytes), bytes, peer);
    }
    int end()
    {
        if (!open) return rc;
        open = false;
        ++c->wire_groups;
        const ncclResult_t ge = c->rccl.GroupEnd();
        if (rc != GVOM_OK) return rc;
        if (ge != ncclSuccess) { c->err = std::string("ncclGroupEnd failed: ") + c->rccl.GetErrorString(ge); return rc = GVOM_ERR_HIP; }
        return GVOM_OK;
    }
};

// ---- loopback transport -----------------------------------------------------------------------------
struct LoopPull { int kind, send_which, recv_which; int64_t unit; const int64_t *recv_counts; const int64_t *send_counts; };

// where this rank holds what it has of `kind` for every other rank (between two barriers' worth of quiet, as peer_publish)
int loop_publish(gvom_comm *c, gvom_t *h, int kind, int send_which, const int64_t *counts)
{
    LoopRegion &e = c->seg->loop[c->rank][kind];
    for (int d = 0; d < c->world; ++d) {
        e.ptr[d] = 0; e.cap[d] = 0;
        if (d == c->rank || counts[d] <= 0) continue;
        void *ptr = nullptr;
        int64_t cap = 0;
        if (gvom_shard_buffer(h, send_which, d, &ptr, &cap) || !ptr) { c->err = "send region missing"; return GVOM_ERR_INVALID; }
        e.ptr[d] = (uint64_t)(uintptr_t)ptr; e.cap[d] = (uint64_t)cap;
    }
    return GVOM_OK;
}
int loop_same_process(gvom_comm *c)
{
    const int64_t me = (int64_t)getpid();
    for (int r = 0; r < c->world; ++r)
        if (c->seg->rank_pid[r] != me) { c->err = "loopback transport: rank " + std::to_string(r) + " is not a thread of this process"; return GVOM_ERR_INVALID; }
    return GVOM_OK;
}
// One exchange over RCCL loopback: the senders' regions are published and complete; barrier; every rank moves what the others
// hold for it -- ncclSend(their region, self) + ncclRecv(its own region, self), one group on ITS handle's stream, in front of
// the kernels that consume the data -- and waits for the group; barrier (the senders may repack).
int loop_pull(gvom_comm *c, gvom_t *h, hipStream_t st, const LoopPull *pulls, int n_pulls)
{
    HIPCHK_C(c, hipStreamSynchronize(st));                             // what the others read of me is complete
    int rc = gvom_comm_barrier(c);
    if (rc) return rc;
    rc = loop_same_process(c);
    WireGroup g{c, st};
    if (rc == GVOM_OK) g.begin();
    for (int s = 0; s < c->world && g.open && g.rc == GVOM_OK; ++s) {
        if (s == c->rank) continue;
        for (int k = 0; k < n_pulls && g.rc == GVOM_OK; ++k) {
            const int64_t cnt = pulls[k].recv_counts[s];
            if (cnt < 0) { g.fail(GVOM_ERR_INVALID, "negative count"); break; }
            if (cnt == 0) continue;
            const size_t bytes = (size_t)cnt * (size_t)pulls[k].unit;
            const LoopRegion &e = c->seg->loop[s][pulls[k].kind];
            if (!e.ptr[c->rank] || bytes > e.cap[c->rank]) { g.fail(GVOM_ERR_INVALID, "a peer announced more than its send region holds"); break; }
            g.wire(true, (void *)(uintptr_t)e.ptr[c->rank], bytes, 0);
            g.xfer(false, h, pulls[k].recv_which, s, bytes, 0);
        }
    }
    if (rc == GVOM_OK) rc = g.end();
    // (even a failed rank passes the second barrier: the others would wait for it for ever)
    const hipError_t se = hipStreamSynchronize(st);
    const int rb = gvom_comm_barrier(c);
    if (rc) return rc;
    if (se != hipSuccess) { c->err = std::string("hipStreamSynchronize failed: ") + hipGetErrorString(se); return GVOM_ERR_HIP; }
    return rb;
}

// ---- peer transport -------------------------------------------------------------------------------
__global__ void k_comm_flag(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);     // everything before it on the stream is complete and visible
}
int peer_flag(gvom_comm *c, hipStream_t st, std::atomic<uint64_t> *host_field, uint64_t v)
{
    unsigned long long *dev = (unsigned long long *)((char *)c->seg_dev + ((char *)host_field - (char *)c->seg));
    hipLaunchKernelGGL(k_comm_flag, dim3(1), dim3(1), 0, st, dev, (unsigned long long)v);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { c->err = std::string("flag kernel launch failed: ") + hipGetErrorString(e); return GVOM_ERR_HIP; }
    return GVOM_OK;
}
// the host waits until rank r's GPU has written exchange number `target` (normally it has, long ago)
int peer_wait(gvom_comm *c, int r, const std::atomic<uint64_t> &flag, uint64_t target, const char *what)
{
    if (flag.load(std::memory_order_acquire) >= target) return GVOM_OK;
    const double deadline = now_s() + c->timeout_s;
    double next_look = 0.0;
    unsigned spins = 0;
    while (flag.load(std::memory_order_acquire) < target) {
        _mm_pause();
        if ((++spins & 0x3ff) == 0) {
            const double t = now_s();
            const uint32_t bad = c->seg->poison.load(std::memory_order_acquire);
            if (bad) { c->err = "rank " + std::to_string(bad - 1) + " reported a failed device exchange"; return GVOM_ERR_HIP; }
            if (t > deadline) { c->err = std::string("timed out waiting for rank ") + std::to_string(r) + " (" + what + ")"; return GVOM_ERR_HIP; }
            if (t > next_look) {
                next_look = t + 0.05;
                if (!proc_running(c->seg->rank_pid[r], c->seg->rank_start[r])) { c->err = "rank " + std::to_string(r) + "'s process is gone"; return GVOM_ERR_HIP; }
            }
        }
    }
    return GVOM_OK;
}

// Publishes export `kind` of this rank: the allocation that holds what it has for the other ranks -- kinds 0..3: the send
// regions of `send_which[kind]` for every rank d with counts[d] > 0; kind 4: this rank's rows of the height maps -- as the
// allocation's handle + the regions' offsets inside it.  An allocation is exported once per process (g_own); one the HSA
// runtime REFUSES to export is replaced by a fresh one (gvom_shard_renew_region: contents move, the old one is parked) --
// up to three times before the refusal counts.  Call between two barriers' worth of quiet: readers look at the export only
// after the barrier that follows.
static const int region_of_kind[GVOM_PEER_KINDS] = {GVOM_XBUF_SEND_IDS, GVOM_XBUF_SEND_QUADS, GVOM_XBUF_SEND_EPS, GVOM_XBUF_SEND_RETURNS, -1};
int peer_publish(gvom_comm *c, gvom_t *h, int kind, const int64_t *counts, bool force_renew = false)
{
    PeerExport &e = c->seg->exports[c->rank][kind];
    parse_fault();
    for (int renewals = 0; ; ++renewals) {
        void *ptr[GVOM_COMM_MAX_RANKS] = {};
        void *any = nullptr;
        if (kind == 4) {
            void *rows = nullptr;
            int64_t bytes = 0, row = 0;
            const int rc = gvom_device_buffer(h, GVOM_BUF_HEIGHT_MAPS, &rows, &bytes, &row);
            if (rc) { c->err = "height-map rows missing"; return rc; }
            for (int d = 0; d < c->world; ++d) if (d != c->rank) ptr[d] = (char *)rows + (size_t)(bytes / c->world) * c->rank;
        } else {
            for (int d = 0; d < c->world; ++d) {
                int64_t cap = 0;
                if (d == c->rank || counts[d] <= 0) continue;
                if (gvom_shard_buffer(h, region_of_kind[kind], d, &ptr[d], &cap) || !ptr[d]) { c->err = "send region missing"; return GVOM_ERR_INVALID; }
            }
        }
        for (int d = 0; d < c->world; ++d) if (d != c->rank && ptr[d]) { any = ptr[d]; break; }
        if (!any) return GVOM_OK;                                      // nothing of this kind goes anywhere
        if (force_renew && renewals == 0) {                            // (a peer could not open the current allocation)
            const int rr = gvom_shard_renew_region(h, region_of_kind[kind]);
            if (rr) { c->err = "could not replace an exchange region a peer cannot open"; return GVOM_ERR_HIP; }
            ++c->peer_renewed;
            continue;
        }
        void *base = nullptr;
        size_t size = 0;
        HIPCHK_C(c, hipMemGetAddressRange((hipDeviceptr_t *)&base, &size, (hipDeviceptr_t)any));
        const uint64_t gen = gvom_region_generation(h, region_of_kind[kind]);
        PeerOwn own;
        bool have = false;
        {
            std::lock_guard<std::mutex> lk(g_ipc_mu);
            for (PeerOwn &k : g_own) if (k.base == base && k.size == size && k.generation == gen) { own = k; have = true; break; }
        }
        if (!have) {
            (void)gvom_set_tuning(h, "exported", 1);                // (its regions go to the pool instead of back to the allocator: gvom_capi.hip)
            hipError_t ge = hipSuccess;
            const bool injected = g_fault.export_n > 0 && (g_fault.rank < 0 || g_fault.rank == c->rank) && ++g_fault.exports == g_fault.export_n;
            for (int attempt = 0; attempt < 5; ++attempt) {            // (a refusal is asked again a few times before it counts)
                ge = injected ? hipErrorInvalidValue : hipIpcGetMemHandle(&own.handle, base);
                if (ge == hipSuccess) break;
                (void)hipGetLastError();
                ++c->peer_open_retries;
                usleep(1000 + 1000 * attempt);
            }
            if (ge != hipSuccess) {
                if (renewals >= 3 || gvom_shard_renew_region(h, region_of_kind[kind]) != GVOM_OK) {
                    c->err = std::string("hipIpcGetMemHandle failed: ") + hipGetErrorString(ge) + " (also for " + std::to_string(renewals) + " fresh allocation(s))";
                    return GVOM_ERR_HIP;
                }
                ++c->peer_renewed;
                continue;                                              // the region lies in a fresh allocation now: export that
            }
            own.base = base; own.size = size; own.generation = gen;
            std::lock_guard<std::mutex> lk(g_ipc_mu);
            g_own.push_back(own);
            ++c->export_seq;
        }
        if (e.generation != own.generation) {                          // (another allocation is current than at the last exchange)
            memcpy(&e.handle, &own.handle, sizeof own.handle);
            e.size = size;
            e.generation = own.generation;
        }
        for (int d = 0; d < c->world; ++d) {
            if (d == c->rank || !ptr[d]) { e.offset[d] = ~0ull; continue; }
            const size_t off = (size_t)((char *)ptr[d] - (char *)base);
            if ((char *)ptr[d] < (char *)base || off >= size) { c->err = "exchange regions of one kind lie in different allocations"; return GVOM_ERR_INVALID; }
            e.offset[d] = off;
        }
        return GVOM_OK;
    }
}

// Address, in this process, of what rank s exported as `kind` for this rank (mapped on first use of the allocation, for the
// life of the process).  refused (optional): set instead of failing when the HSA runtime refuses to open it.
int peer_source(gvom_comm *c, int s, int kind, void **src, bool *refused = nullptr)
{
    const PeerExport &e = c->seg->exports[s][kind];
    if (e.generation == 0 || e.offset[c->rank] == ~0ull) { c->err = "a peer announced data it has not exported"; return GVOM_ERR_INVALID; }
    const int64_t pid = c->seg->rank_pid[s];
    const uint64_t start = c->seg->rank_start[s];
    void *base = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_ipc_mu);
        for (PeerImport &k : g_imports) if (k.generation == e.generation && k.pid == pid && k.start == start) { base = k.base; break; }
    }
    if (!base) {
        parse_fault();
        hipIpcMemHandle_t hd;
        memcpy(&hd, &e.handle, sizeof hd);
        // (the exporter hands its allocation over through a helper thread of the HSA runtime that starts with its FIRST
        // export; a peer that asks in the same millisecond can be too early -- "invalid device pointer" -- so a refusal is
        // asked again a few times before it counts)
        hipError_t oe = hipSuccess;
        const bool injected = g_fault.import_n > 0 && (g_fault.rank < 0 || g_fault.rank == c->rank) && ++g_fault.imports == g_fault.import_n;
        for (int attempt = 0; attempt < 20; ++attempt) {
            oe = injected ? hipErrorInvalidDevicePointer : hipIpcOpenMemHandle(&base, hd, hipIpcMemLazyEnablePeerAccess);
            if (oe == hipSuccess) break;
            (void)hipGetLastError();
            ++c->peer_open_retries;
            if (injected) break;
            usleep(2000 + 1000 * attempt);
        }
        if (oe != hipSuccess) {
            c->err = "hipIpcOpenMemHandle failed (" + std::string(hipGetErrorString(oe)) + "): export " + std::to_string(kind) +
                     " of rank " + std::to_string(s) + ", " + std::to_string(e.size) + " bytes, generation " + std::to_string(e.generation);
            if (refused) { *refused = true; return GVOM_OK; }
            return GVOM_ERR_HIP;
        }
        PeerImport im;
        im.pid = pid; im.start = start; im.generation = e.generation; im.base = base;
        std::lock_guard<std::mutex> lk(g_ipc_mu);
        g_imports.push_back(im);
    }
    *src = (char *)base + e.offset[c->rank];
    return GVOM_OK;
}

struct PeerPull { int kind, recv_which; int64_t unit; const int64_t *recv_counts; const int64_t *send_counts; };

// After the barrier that made the exports visible: every rank opens what it needs (src[s][k] for need[s][k]).  An open the HSA
// runtime REFUSES is absorbed here: the rank says so in the segment, the owner of the allocation moves the region into a fresh
// one and exports that, and everybody tries again -- up to three rounds.  Whether a round is needed is decided COLLECTIVELY:
// every rank brings "an open of mine was refused" to one host exchange and every rank sees the OR (round 4 let each rank
// decide from the export generations it had seen: a rank that opened an OLD export for the first time -- three or more
// ranks, a source that had nothing for it in the earlier scans -- failed alone while the others went on to the next barrier;
// ADVICE r4).  One more ~1 us host barrier per exchange of the peer transport; both of its forms (host-synchronised and
// asynchronous) come through here.  republish(kind): the caller's way to export `kind` again from a fresh allocation.
template <typename Republish>
int peer_open_all(gvom_comm *c, const int *kinds, int nk, const bool (*need)[GVOM_PEER_KINDS], void *(*src)[GVOM_PEER_KINDS],
                  Republish republish)
{
    for (int round = 0; ; ++round) {
        const uint64_t tag = ++c->recover_round;
        bool mine_failed = false;
        std::string first_err;
        for (int s = 0; s < c->world; ++s) {
            if (s == c->rank) continue;
            for (int i = 0; i < nk; ++i) {
                const int k = kinds[i];
                if (!need[s][k]) continue;
                bool refused = false;
                const int rc = peer_source(c, s, k, &src[s][k], &refused);
                if (rc) return rc;
                if (refused) {
                    mine_failed = true;
                    if (first_err.empty()) first_err = c->err;
                    c->seg->import_failed[s][k].store(tag, std::memory_order_release);
                }
            }
        }
        // the refusals of this round are on the table, and every rank learns whether there were any
        int64_t mine = mine_failed ? 1 : 0, all[GVOM_COMM_MAX_RANKS];
        int rc = gvom_comm_exchange_host(c, &mine, 1, all);
        if (rc) return rc;
        bool any = false;
        for (int r = 0; r < c->world; ++r) any = any || all[r] != 0;
        if (!any) return GVOM_OK;
        if (round >= 3) { c->err = first_err.empty() ? "a peer could not open an exported region (three fresh allocations tried)" : first_err; return GVOM_ERR_HIP; }
        for (int i = 0; i < nk; ++i)
            if (c->seg->import_failed[c->rank][kinds[i]].load(std::memory_order_acquire) == tag && (rc = republish(kinds[i]))) break;
        const int rb = gvom_comm_barrier(c);                           // the fresh exports are visible
        if (rc) return rc;
        if (rb) return rb;
    }
}

// One exchange by peer copies: every rank has published its regions (peer_publish); barrier; every rank pulls what
// the others hold for it -- hipMemcpyAsync on ITS OWN handle's stream, the ordering ncclRecv on that stream gives --
// and waits for its copies; barrier (the senders may rewrite their regions).
// Asynchronous form (c->async; `done` = this rank's flag for the kind of exchange, `x` its number): the send regions are
// complete by construction (their counts were read by the hosts after the kernels that fill them), so one host barrier
// makes the exports visible, the pulls are enqueued, a one-thread kernel behind them writes `x` into `done`, and the call
// returns without waiting for the GPU; whoever is about to REWRITE its send regions waits for every peer's `done` first
// (gvom_comm_before_scan).
int peer_pull(gvom_comm *c, gvom_t *h, hipStream_t st, const PeerPull *pulls, int n_pulls, std::atomic<uint64_t> *done = nullptr, uint64_t x = 0)
{
    const bool async = c->async && done != nullptr;
    if (!async) HIPCHK_C(c, hipStreamSynchronize(st));                 // what I export is complete
    int rc = gvom_comm_barrier(c);
    if (rc) return rc;
    auto &src_all = c->src_all;
    auto &need_all = c->need_all;
    {
        int kinds[GVOM_PEER_KINDS];
        for (int s = 0; s < c->world; ++s)
            for (int k = 0; k < n_pulls; ++k) {
                need_all[s][pulls[k].kind] = s != c->rank && pulls[k].recv_counts[s] > 0;
                src_all[s][pulls[k].kind] = nullptr;
            }
        for (int k = 0; k < n_pulls; ++k) kinds[k] = pulls[k].kind;
        rc = peer_open_all(c, kinds, n_pulls, need_all, src_all, [&](int kind) {
            for (int k = 0; k < n_pulls; ++k) if (pulls[k].kind == kind) return peer_publish(c, h, kind, pulls[k].send_counts, true);
            return (int)GVOM_ERR_INVALID;
        });
    }
    for (int s = 0; s < c->world && rc == GVOM_OK; ++s) {
        if (s == c->rank) continue;
        for (int k = 0; k < n_pulls && rc == GVOM_OK; ++k) {
            const int64_t cnt = pulls[k].recv_counts[s];
            if (cnt < 0) { c->err = "negative count"; rc = GVOM_ERR_INVALID; break; }
            if (cnt == 0) continue;
            const size_t bytes = (size_t)cnt * (size_t)pulls[k].unit;
            void *dst = nullptr, *src = nullptr;
            int64_t cap = 0;
            if (gvom_shard_buffer(h, pulls[k].recv_which, s, &dst, &cap) || (int64_t)bytes > cap) {
                c->err = "exchange region missing or smaller than the announced count";
                rc = GVOM_ERR_INVALID;
                break;
            }
            src = src_all[s][pulls[k].kind];
            const hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st);
            if (e != hipSuccess) { c->err = std::string("hipMemcpyAsync (peer copy) failed: ") + hipGetErrorString(e); rc = GVOM_ERR_HIP; break; }
            c->peer_bytes += bytes; ++c->peer_copies;
        }
    }
    if (async) {
        if (rc) return rc;                                             // (the caller poisons the segment: nobody waits for this rank's flag)
        return peer_flag(c, st, done, x);
    }
    // (even a failed rank passes the second barrier: the others would wait for it for ever)
    const hipError_t se = hipStreamSynchronize(st);
    const int rb = gvom_comm_barrier(c);
    if (rc) return rc;
    if (se != hipSuccess) { c->err = std::string("hipStreamSynchronize failed: ") + hipGetErrorString(se); return GVOM_ERR_HIP; }
    return rb;
}

}  // namespace

extern "C" {

// name: the same string on every rank of the communicator and unique to it on the node (e.g.
// "gvom_<master port>"); rank 0 creates /dev/shm/<name>, the others wait for it.  device: the HIP
// device this rank's handle lives on.  world == 1 is allowed (collectives degenerate to copies).
// device < 0: HOST-ONLY communicator -- the shared-memory rendezvous and gvom_comm_exchange_host /
// gvom_comm_barrier without RCCL or any HIP call (the CPU tests exercise the multi-process rendezvous
// with it); the device collectives return GVOM_ERR_INVALID on it.
// transport: GVOM_TRANSPORT_RCCL (grouped ncclSend / ncclRecv + ncclAllGather), GVOM_TRANSPORT_PEER (peer copies through
// exported allocations: see the head of this file) or GVOM_TRANSPORT_AUTO (RCCL; if RCCL cannot be loaded or
// ncclCommInitRank fails on ANY rank, every rank uses peer copies -- gvom_comm_transport says which it became).
VIS int gvom_comm_create2(int rank, int world, int device, const char *name, int transport, gvom_comm_t **out)
{
    if (!out || !name || world < 1 || world > GVOM_COMM_MAX_RANKS || rank < 0 || rank >= world) return GVOM_ERR_INVALID;
    if (transport != GVOM_TRANSPORT_RCCL && transport != GVOM_TRANSPORT_PEER && transport != GVOM_TRANSPORT_AUTO && transport != GVOM_TRANSPORT_LOOPBACK) return GVOM_ERR_INVALID;
    *out = nullptr;
    gvom_comm *c = new gvom_comm();
    c->rank = rank; c->world = world; c->device = device;
    c->transport = transport == GVOM_TRANSPORT_AUTO ? GVOM_TRANSPORT_RCCL : transport;
    const bool want_rccl = transport != GVOM_TRANSPORT_PEER, may_fall_back = transport == GVOM_TRANSPORT_AUTO;
    const bool loopback = transport == GVOM_TRANSPORT_LOOPBACK;       // every rank a 1-rank communicator of its own
    bool rccl_ok = want_rccl;                                          // (this rank's view)
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
    if (!host_only && world > 1 && transport != GVOM_TRANSPORT_RCCL && !getenv("HSA_ENABLE_IPC_MODE_LEGACY")) {
        // (every measurement of this transport was taken with it: the host driver of the target pool only supports dmabuf IPC)
        static std::atomic<bool> said{false};
        if (!said.exchange(true))
            fprintf(stderr, "gvom_comm_create(rank %d of %d): HSA_ENABLE_IPC_MODE_LEGACY is not set; the peer-copy transport exports device memory "
                            "between processes (hipIpc*), which on dmabuf-only hosts fails without HSA_ENABLE_IPC_MODE_LEGACY=0\n", rank, world);
    }
    if (!host_only) {
        if (want_rccl && !c->rccl.load(c->err)) {
            if (!may_fall_back) return fail(c->err, GVOM_ERR_NO_DEVICE);
            fprintf(stderr, "gvom_comm_create(rank %d of %d): %s -- falling back to peer copies\n", rank, world, c->err.c_str());
            rccl_ok = false;
        }
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
        if (!host_only && rccl_ok && !loopback && c->rccl.GetUniqueId(&c->seg->id) != ncclSuccess) {
            if (!may_fall_back) return fail("ncclGetUniqueId failed", GVOM_ERR_HIP);
            rccl_ok = false;
        }
        // (AUTO: a rank 0 without RCCL says so BEFORE the others could try ncclCommInitRank on an id that was never made)
        if (!host_only && want_rccl && !rccl_ok) c->seg->rccl_failed.fetch_add(1, std::memory_order_acq_rel);
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
            if (now_s() > deadline)
                return fail("rank 0 never created the shared-memory rendezvous /dev/shm" + c->shm_name + " (do all ranks of the job compute the "
                            "same name?  gvom_sharded.rendezvous_name: MASTER_PORT and GVOM_JOB_NONCE)", GVOM_ERR_HIP);
            usleep(500);
        }
        if (c->seg->world != (uint32_t)world) return fail("world size differs from rank 0's", GVOM_ERR_INVALID);
    }
    if (!host_only && may_fall_back) {
        // (the ranks of a job start seconds to minutes apart: the watchdog below must time RCCL, not the slowest interpreter)
        c->seg->arrived.fetch_add(1, std::memory_order_acq_rel);
        const double deadline = now_s() + c->timeout_s;
        while (c->seg->arrived.load(std::memory_order_acquire) < (uint32_t)world) {
            if (now_s() > deadline) return fail("not every rank arrived at the rendezvous", GVOM_ERR_HIP);
            usleep(200);
        }
    }
    if (!host_only && want_rccl) {
        if (rank != 0 && !rccl_ok) c->seg->rccl_failed.fetch_add(1, std::memory_order_acq_rel);     // (rank 0 has counted itself)
        // nobody joins a communicator that some rank is already known to be missing from
        if (rccl_ok && !(may_fall_back && c->seg->rccl_failed.load(std::memory_order_acquire) != 0u)) {
            ncclUniqueId id;
            memcpy(&id, &c->seg->id, sizeof id);
            ncclResult_t r = ncclSuccess;
            std::string why;
            if (loopback) {
                // (the ranks are threads of one process on one device: their communicators are made one after the other)
                static std::mutex init_mu;
                std::lock_guard<std::mutex> lk(init_mu);
                r = c->rccl.GetUniqueId(&id);
                if (r == ncclSuccess) r = c->rccl.CommInitRank(&c->nccl, 1, id, 0);
            } else if (!may_fall_back) {
                r = c->rccl.CommInitRank(&c->nccl, world, id, rank);
            } else {
                // AUTO: ncclCommInitRank is a collective that can also HANG (a rank that failed early leaves the others
                // waiting inside it; a bootstrap interface that does not route).  It runs on a helper thread; this one
                // gives up -- for every rank, through the segment -- when it has not returned within
                // GVOM_RCCL_INIT_TIMEOUT_S (default 90 s, counted from the moment ALL ranks have arrived) or as soon
                // as some rank reports its failure.  A helper that never returns is left behind (detached).
                struct InitJob { std::mutex m; std::condition_variable cv; bool done = false; ncclComm_t comm = nullptr; ncclResult_t res = ncclSuccess; };
                auto job = std::make_shared<InitJob>();
                auto init = c->rccl.CommInitRank;
                std::thread([job, init, world, id, rank, device]() {
                    (void)hipSetDevice(device);
                    ncclComm_t cm = nullptr;
                    const ncclResult_t res = init(&cm, world, id, rank);
                    std::lock_guard<std::mutex> lk(job->m);
                    job->comm = cm; job->res = res; job->done = true;
                    job->cv.notify_all();
                }).detach();
                double patience = 90.0;
                if (const char *t = getenv("GVOM_RCCL_INIT_TIMEOUT_S")) { const double v = atof(t); if (v > 0.0) patience = v; }
                const double give_up = now_s() + patience;
                std::unique_lock<std::mutex> lk(job->m);
                bool abandoned = false;
                while (!job->done) {
                    job->cv.wait_for(lk, std::chrono::milliseconds(50));
                    if (job->done) break;
                    if (c->seg->rccl_failed.load(std::memory_order_acquire) != 0u) { abandoned = true; why = "another rank could not initialise RCCL"; break; }
                    if (now_s() > give_up) { abandoned = true; why = "ncclCommInitRank did not return within " + std::to_string((int)patience) + " s"; break; }
                }
                if (abandoned) r = ncclInternalError;
                else { r = job->res; c->nccl = job->comm; }
            }
            if (r != ncclSuccess) {
                c->nccl = nullptr;
                if (why.empty()) why = std::string("ncclCommInitRank failed: ") + c->rccl.GetErrorString(r);
                if (!may_fall_back) return fail(why, GVOM_ERR_HIP);
                fprintf(stderr, "gvom_comm_create(rank %d of %d): %s -- falling back to peer copies\n", rank, world, why.c_str());
                c->seg->rccl_failed.fetch_add(1, std::memory_order_acq_rel);
            }
        }
    }
    c->seg->rank_start[rank] = proc_start_time((int64_t)getpid());
    c->seg->rank_pid[rank] = (int64_t)getpid();
    // the name can go once everybody is attached (the mapping stays valid)
    if (c->seg->attached.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) shm_unlink(c->shm_name.c_str());
    if (!host_only && may_fall_back) {
        // every rank has tried by the time this barrier is passed: one failure anywhere -> peer copies everywhere
        if (gvom_comm_barrier(c) != GVOM_OK) return fail("rendezvous barrier: " + c->err, GVOM_ERR_HIP);
        if (c->seg->rccl_failed.load(std::memory_order_acquire) != 0u) {
            if (c->nccl) { c->rccl.CommDestroy(c->nccl); c->nccl = nullptr; }
            c->transport = GVOM_TRANSPORT_PEER;
        }
    }
    if (!host_only && c->transport == GVOM_TRANSPORT_PEER && world > 1) {
        // GVOM_PEER_ASYNC=1 asks for the asynchronous form.  It needs the GPU to write into the segment: register it; one rank
        // that cannot (or was not asked to) keeps everybody on the host-synchronised form.  (Not the default: with several
        // ranks on ONE GPU -- the only place it could be measured -- it is the slower form, the processes' kernels then
        // interleave on the device: m256 x 2 ranks 677 us per step against 421, c4 x 4 ranks 2959 against 2862.)
        const char *want = getenv("GVOM_PEER_ASYNC");
        bool ok = want && atoi(want) != 0;
        if (ok) {
            ok = hipHostRegister(c->seg, sizeof(Segment), hipHostRegisterMapped | hipHostRegisterPortable) == hipSuccess;
            if (ok) {
                c->registered = true;
                ok = hipHostGetDevicePointer((void **)&c->seg_dev, c->seg, 0) == hipSuccess && c->seg_dev != nullptr;
            }
            (void)hipGetLastError();
        }
        if (!ok) c->seg->sync_only.fetch_add(1, std::memory_order_acq_rel);
        if (gvom_comm_barrier(c) != GVOM_OK) return fail("rendezvous barrier: " + c->err, GVOM_ERR_HIP);
        c->async = c->seg->sync_only.load(std::memory_order_acquire) == 0u;
    }
    *out = c;
    return GVOM_OK;
}

VIS int gvom_comm_create(int rank, int world, int device, const char *name, gvom_comm_t **out)
{
    return gvom_comm_create2(rank, world, device, name, GVOM_TRANSPORT_RCCL, out);
}

VIS int gvom_comm_transport(gvom_comm_t *c) { return c ? c->transport : -1; }

VIS void gvom_comm_destroy(gvom_comm_t *c)
{
    if (!c) return;
    if (c->nccl) { hipSetDevice(c->device); c->rccl.CommDestroy(c->nccl); }
    // (what the peer transport has mapped of other processes' memory stays mapped for the life of the process: g_imports)
    if (c->registered) { hipSetDevice(c->device); (void)hipDeviceSynchronize(); (void)hipHostUnregister(c->seg); }
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
        double slow_since = 0.0, next_look = 0.0;
        while (slots[r].seq.load(std::memory_order_acquire) != call) {
            _mm_pause();
            if ((++spins & 0xfff) == 0) {
                // a peer that is milliseconds late is minutes late as likely as not (it died, or it is still importing):
                // stop burning the core
                const double t = now_s();
                if (slow_since == 0.0) slow_since = t;
                if (t > deadline) { c->err = "host exchange timed out (a rank is missing)"; return GVOM_ERR_HIP; }
                if (t - slow_since > 5e-3) {
                    // late by milliseconds: look whether the rank can still come
                    const uint32_t bad = c->seg->poison.load(std::memory_order_acquire);
                    if (bad) { c->err = "rank " + std::to_string(bad - 1) + " reported a failed device exchange"; return GVOM_ERR_HIP; }
                    if (t > next_look) {
                        next_look = t + 0.05;
                        if (slots[r].seq.load(std::memory_order_acquire) != call &&
                            !proc_running(c->seg->rank_pid[r], c->seg->rank_start[r])) {
                            c->err = "rank " + std::to_string(r) + "'s process is gone";
                            return GVOM_ERR_HIP;
                        }
                    }
                    usleep(t - slow_since > 1.0 ? 1000 : 50);
                }
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
static int exchange_scan_impl(gvom_comm_t *c, gvom_t *h, const int64_t *send_quads, const int64_t *send_eps,
                                const int64_t *recv_quads, const int64_t *recv_eps)
{
    if (!c || !h || !send_quads || !send_eps || !recv_quads || !recv_eps) return GVOM_ERR_INVALID;
    if (c->world == 1) return GVOM_OK;
    if (c->device < 0) { c->err = "host-only communicator: no device collectives"; return GVOM_ERR_INVALID; }
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return GVOM_ERR_HIP; }
    hipStream_t st = (hipStream_t)gvom_stream(h);
    if (c->transport == GVOM_TRANSPORT_PEER) {
        for (int k = 0; k < 3; ++k) {
            const int rc = peer_publish(c, h, k, k < 2 ? send_quads : send_eps);
            if (rc) return rc;
        }
        const PeerPull pulls[3] = {{0, GVOM_XBUF_RECV_IDS, 4, recv_quads, send_quads}, {1, GVOM_XBUF_RECV_QUADS, 1024, recv_quads, send_quads},
                                   {2, GVOM_XBUF_RECV_EPS, 8, recv_eps, send_eps}};
        ++c->scan_x;
        return peer_pull(c, h, st, pulls, 3, &c->seg->flags[c->rank].pulled_scan, c->scan_x);
    }
    if (!c->nccl) { c->err = "communicator without RCCL"; return GVOM_ERR_INVALID; }
    if (c->transport == GVOM_TRANSPORT_LOOPBACK) {
        int rc = GVOM_OK;
        for (int k = 0; k < 3 && rc == GVOM_OK; ++k) rc = loop_publish(c, h, k, region_of_kind[k], k < 2 ? send_quads : send_eps);
        if (rc) return rc;
        const LoopPull pulls[3] = {{0, GVOM_XBUF_SEND_IDS, GVOM_XBUF_RECV_IDS, 4, recv_quads, send_quads},
                                   {1, GVOM_XBUF_SEND_QUADS, GVOM_XBUF_RECV_QUADS, 1024, recv_quads, send_quads},
                                   {2, GVOM_XBUF_SEND_EPS, GVOM_XBUF_RECV_EPS, 8, recv_eps, send_eps}};
        return loop_pull(c, h, st, pulls, 3);
    }
    WireGroup g{c, st};
    if (g.begin()) return g.rc;
    for (int p = 0; p < c->world && g.rc == GVOM_OK; ++p) {
        if (p == c->rank) continue;
        if (send_quads[p] < 0 || send_eps[p] < 0 || recv_quads[p] < 0 || recv_eps[p] < 0) { g.fail(GVOM_ERR_INVALID, "negative count"); break; }
        g.xfer(true, h, GVOM_XBUF_SEND_IDS, p, (size_t)send_quads[p] * 4, p);
        g.xfer(true, h, GVOM_XBUF_SEND_QUADS, p, (size_t)send_quads[p] * 1024, p);
        g.xfer(true, h, GVOM_XBUF_SEND_EPS, p, (size_t)send_eps[p] * 8, p);
        g.xfer(false, h, GVOM_XBUF_RECV_IDS, p, (size_t)recv_quads[p] * 4, p);
        g.xfer(false, h, GVOM_XBUF_RECV_QUADS, p, (size_t)recv_quads[p] * 1024, p);
        g.xfer(false, h, GVOM_XBUF_RECV_EPS, p, (size_t)recv_eps[p] * 8, p);
    }
    return g.end();
}

// Statistics handles: the returns every other rank needs of this one (gvom_shard_stats_counts) -> their receive
// regions (gvom_shard_stats_reserve), bytes_per_return = 12 (float32 clouds) or 24.  One grouped send / recv.
static int exchange_stats_impl(gvom_comm_t *c, gvom_t *h, const int64_t *send_returns, const int64_t *recv_returns,
                                 int bytes_per_return)
{
    if (!c || !h || !send_returns || !recv_returns || (bytes_per_return != 12 && bytes_per_return != 24)) return GVOM_ERR_INVALID;
    if (c->world == 1) return GVOM_OK;
    if (c->device < 0) { c->err = "host-only communicator: no device collectives"; return GVOM_ERR_INVALID; }
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return GVOM_ERR_HIP; }
    hipStream_t st = (hipStream_t)gvom_stream(h);
    if (c->transport == GVOM_TRANSPORT_PEER) {
        const int rcp = peer_publish(c, h, 3, send_returns);
        if (rcp) return rcp;
        const PeerPull pull = {3, GVOM_XBUF_RECV_RETURNS, bytes_per_return, recv_returns, send_returns};
        ++c->stats_x;
        return peer_pull(c, h, st, &pull, 1, &c->seg->flags[c->rank].pulled_stats, c->stats_x);
    }
    if (!c->nccl) { c->err = "communicator without RCCL"; return GVOM_ERR_INVALID; }
    if (c->transport == GVOM_TRANSPORT_LOOPBACK) {
        const int rc = loop_publish(c, h, 3, GVOM_XBUF_SEND_RETURNS, send_returns);
        if (rc) return rc;
        const LoopPull pull = {3, GVOM_XBUF_SEND_RETURNS, GVOM_XBUF_RECV_RETURNS, bytes_per_return, recv_returns, send_returns};
        return loop_pull(c, h, st, &pull, 1);
    }
    WireGroup g{c, st};
    if (g.begin()) return g.rc;
    for (int p = 0; p < c->world && g.rc == GVOM_OK; ++p) {
        if (p == c->rank) continue;
        if (send_returns[p] < 0 || recv_returns[p] < 0) { g.fail(GVOM_ERR_INVALID, "negative count"); break; }
        g.xfer(true, h, GVOM_XBUF_SEND_RETURNS, p, (size_t)send_returns[p] * (size_t)bytes_per_return, p);
        g.xfer(false, h, GVOM_XBUF_RECV_RETURNS, p, (size_t)recv_returns[p] * (size_t)bytes_per_return, p);
    }
    return g.end();
}

// The combine's exchange: in-place all-gather of the handle's [height | inferred height | positive
// density] rows (GVOM_BUF_HEIGHT_MAPS; a rank's rows are one contiguous block) on the handle's stream.
static int allgather_rows_impl(gvom_comm_t *c, gvom_t *h)
{
    if (!c || !h) return GVOM_ERR_INVALID;
    if (c->device < 0) { c->err = "host-only communicator: no device collectives"; return GVOM_ERR_INVALID; }
    void *ptr = nullptr;
    int64_t bytes = 0, row = 0;
    int rc = gvom_device_buffer(h, GVOM_BUF_HEIGHT_MAPS, &ptr, &bytes, &row);
    if (rc) return rc;
    if (bytes % c->world) { c->err = "height-map rows do not divide among the ranks"; return GVOM_ERR_INVALID; }
    if (hipSetDevice(c->device) != hipSuccess) { c->err = "hipSetDevice failed"; return GVOM_ERR_HIP; }
    const size_t share = (size_t)(bytes / c->world);
    if (c->transport == GVOM_TRANSPORT_PEER) {
        if (c->world == 1) return GVOM_OK;
        hipStream_t st = (hipStream_t)gvom_stream(h);
        // every rank offers ITS rows (the same part whoever asks) and pulls the others' into the same place of its own buffer
        if ((rc = peer_publish(c, h, 4, nullptr))) return rc;
        if (c->async) {
            // my rows are complete when my stream gets here; the others' when their flag says so (their hosts wait for it, the GPUs
            // for nothing); "I have pulled" goes behind my copies, and whoever fuses again waits for it (gvom_comm_before_combine)
            const uint64_t x = ++c->rows_x;
            if ((rc = peer_flag(c, st, &c->seg->flags[c->rank].rows_ready, x))) return rc;
            if ((rc = gvom_comm_barrier(c))) return rc;                // (exports visible)
            {
                const int kind4 = 4;
                for (int s = 0; s < c->world; ++s) { c->need_all[s][4] = s != c->rank; c->src_all[s][4] = nullptr; }
                if ((rc = peer_open_all(c, &kind4, 1, c->need_all, c->src_all, [&](int) { return peer_publish(c, h, 4, nullptr, true); }))) return rc;
                // (this rank's own rows may lie in a fresh allocation now: a renewal waits for the stream, so rows_ready below
                // would be stale -- publish it again behind the move)
                void *p2 = nullptr;
                if (gvom_device_buffer(h, GVOM_BUF_HEIGHT_MAPS, &p2, &bytes, &row)) { c->err = "height-map rows missing"; return GVOM_ERR_INVALID; }
                if (p2 != ptr) { ptr = p2; if ((rc = peer_flag(c, st, &c->seg->flags[c->rank].rows_ready, x))) return rc; }
            }
            for (int s = 0; s < c->world; ++s) {
                if (s == c->rank) continue;
                if ((rc = peer_wait(c, s, c->seg->flags[s].rows_ready, x, "its rows of the combine"))) return rc;
                void *src = c->src_all[s][4];
                HIPCHK_C(c, hipMemcpyAsync((char *)ptr + share * s, src, share, hipMemcpyDefault, st));
                c->peer_bytes += share; ++c->peer_copies;
            }
            return peer_flag(c, st, &c->seg->flags[c->rank].pulled_rows, x);
        }
        HIPCHK_C(c, hipStreamSynchronize(st));                         // my rows are complete
        if ((rc = gvom_comm_barrier(c))) return rc;
        {
            const int kind4 = 4;
            for (int s = 0; s < c->world; ++s) { c->need_all[s][4] = s != c->rank; c->src_all[s][4] = nullptr; }
            rc = peer_open_all(c, &kind4, 1, c->need_all, c->src_all, [&](int) { return peer_publish(c, h, 4, nullptr, true); });
            // (this rank's own rows may lie in a fresh allocation now)
            if (rc == GVOM_OK && (rc = gvom_device_buffer(h, GVOM_BUF_HEIGHT_MAPS, &ptr, &bytes, &row))) c->err = "height-map rows missing";
        }
        for (int s = 0; s < c->world && rc == GVOM_OK; ++s) {
            if (s == c->rank) continue;
            const hipError_t e = hipMemcpyAsync((char *)ptr + share * s, c->src_all[s][4], share, hipMemcpyDefault, st);
            if (e != hipSuccess) { c->err = std::string("hipMemcpyAsync (peer copy) failed: ") + hipGetErrorString(e); rc = GVOM_ERR_HIP; }
            c->peer_bytes += share; ++c->peer_copies;
        }
        const hipError_t se = hipStreamSynchronize(st);                // nobody's next fusion rewrites its rows before everyone has pulled
        const int rb = gvom_comm_barrier(c);
        if (rc) return rc;
        if (se != hipSuccess) { c->err = std::string("hipStreamSynchronize failed: ") + hipGetErrorString(se); return GVOM_ERR_HIP; }
        return rb;
    }
    if (c->transport == GVOM_TRANSPORT_LOOPBACK) {
        // every rank offers ITS rows and moves the others' into the same place of its own buffer with ncclSend / ncclRecv to
        // itself (one group on its handle's stream); then the in-place all-gather of its 1-rank communicator over its own
        // share: the product's call with the product's arguments, complete before k_map2d by stream order
        if (!c->nccl) { c->err = "communicator without RCCL"; return GVOM_ERR_INVALID; }
        hipStream_t st = (hipStream_t)gvom_stream(h);
        LoopRegion &e = c->seg->loop[c->rank][4];
        for (int d = 0; d < c->world; ++d) { e.ptr[d] = (uint64_t)(uintptr_t)((char *)ptr + share * c->rank); e.cap[d] = share; }
        HIPCHK_C(c, hipStreamSynchronize(st));                         // my rows are complete
        if ((rc = gvom_comm_barrier(c))) return rc;
        rc = loop_same_process(c);
        WireGroup g{c, st};
        if (rc == GVOM_OK) g.begin();
        for (int s = 0; s < c->world && g.open && g.rc == GVOM_OK; ++s) {
            if (s == c->rank) continue;
            const LoopRegion &o = c->seg->loop[s][4];
            if (!o.ptr[c->rank] || o.cap[c->rank] != share) { g.fail(GVOM_ERR_INVALID, "the ranks' height-map rows differ in size"); break; }
            g.wire(true, (void *)(uintptr_t)o.ptr[c->rank], share, 0);
            g.wire(false, (char *)ptr + share * s, share, 0);
        }
        if (rc == GVOM_OK) rc = g.end();
        if (rc == GVOM_OK) {
            const ncclResult_t r = c->rccl.AllGather((char *)ptr + share * c->rank, (char *)ptr + share * c->rank, share, ncclUint8, c->nccl, st);
            if (r != ncclSuccess) { c->err = std::string("ncclAllGather failed: ") + c->rccl.GetErrorString(r); rc = GVOM_ERR_HIP; }
            else ++c->wire_allgathers;
        }
        const hipError_t se = hipStreamSynchronize(st);                // nobody's next fusion rewrites its rows before everyone has pulled
        const int rb = gvom_comm_barrier(c);
        if (rc) return rc;
        if (se != hipSuccess) { c->err = std::string("hipStreamSynchronize failed: ") + hipGetErrorString(se); return GVOM_ERR_HIP; }
        return rb;
    }
    if (c->world == 1) return GVOM_OK;                     // (one rank: its rows are all the rows)
    if (!c->nccl) { c->err = "communicator without RCCL"; return GVOM_ERR_INVALID; }
    NCCLCHK(c, c->rccl.AllGather((char *)ptr + share * c->rank, ptr, share, ncclUint8, c->nccl, (hipStream_t)gvom_stream(h)));
    ++c->wire_allgathers;
    return GVOM_OK;
}

VIS int gvom_comm_exchange_scan(gvom_comm_t *c, gvom_t *h, const int64_t *send_quads, const int64_t *send_eps,
                                const int64_t *recv_quads, const int64_t *recv_eps)
{
    const int rc = exchange_scan_impl(c, h, send_quads, send_eps, recv_quads, recv_eps);
    // a rank whose device exchange failed will not come to the next rendezvous: the others must not wait for it
    if (rc != GVOM_OK && c && c->seg && c->world > 1) c->seg->poison.store((uint32_t)c->rank + 1u, std::memory_order_release);
    return rc;
}

VIS int gvom_comm_exchange_stats(gvom_comm_t *c, gvom_t *h, const int64_t *send_returns, const int64_t *recv_returns,
                                 int bytes_per_return)
{
    const int rc = exchange_stats_impl(c, h, send_returns, recv_returns, bytes_per_return);
    // a rank whose device exchange failed will not come to the next rendezvous: the others must not wait for it
    if (rc != GVOM_OK && c && c->seg && c->world > 1) c->seg->poison.store((uint32_t)c->rank + 1u, std::memory_order_release);
    return rc;
}

VIS int gvom_comm_allgather_rows(gvom_comm_t *c, gvom_t *h)
{
    const int rc = allgather_rows_impl(c, h);
    // a rank whose device exchange failed will not come to the next rendezvous: the others must not wait for it
    if (rc != GVOM_OK && c && c->seg && c->world > 1) c->seg->poison.store((uint32_t)c->rank + 1u, std::memory_order_release);
    return rc;
}

// Asynchronous peer transport: before a rank REWRITES what its peers pull from -- its send regions (the next scan's pack) or its
// rows (the next fusion) -- every peer must have finished pulling the previous exchange.  Their GPUs say so in the segment;
// normally they did long ago and these calls return at once.  No-ops on the other transports.
VIS int gvom_comm_before_scan(gvom_comm_t *c)
{
    if (!c) return GVOM_ERR_INVALID;
    if (!c->async || c->world == 1) return GVOM_OK;
    int rc = GVOM_OK;
    for (int p = 0; p < c->world && rc == GVOM_OK; ++p) {
        if (p == c->rank) continue;
        rc = peer_wait(c, p, c->seg->flags[p].pulled_scan, c->scan_x, "pulling the previous scan");
        if (rc == GVOM_OK) rc = peer_wait(c, p, c->seg->flags[p].pulled_stats, c->stats_x, "pulling the previous scan's returns");
    }
    return rc;
}
VIS int gvom_comm_peer_async(gvom_comm_t *c) { return c && c->async ? 1 : 0; }
VIS int gvom_comm_before_combine(gvom_comm_t *c)
{
    if (!c) return GVOM_ERR_INVALID;
    if (!c->async || c->world == 1) return GVOM_OK;
    int rc = GVOM_OK;
    for (int p = 0; p < c->world && rc == GVOM_OK; ++p)
        if (p != c->rank) rc = peer_wait(c, p, c->seg->flags[p].pulled_rows, c->rows_x, "pulling the previous combine's rows");
    return rc;
}

// ONE sharded scan, natively: what ShardedGvom.process_pointcloud does call by call from Python (before_scan, scan_local, the
// host exchange of the counts, recv_reserve, the device exchange, scan_merge), for handles without per-voxel statistics.
// Every rank calls it with ITS share of the cloud.  out = {the scan was accepted (some rank saw a return in the grid,
// gvom.py:147-150), returns of all ranks, bytes this rank sent, bytes it received}.
VIS int gvom_comm_process_pointcloud(gvom_comm_t *c, gvom_t *h, const void *xyz, int on_device, int64_t n, int64_t row_stride_bytes,
                                     int dtype, const double ego[3], const double *transform_4x4, int64_t out[4])
{
    if (!c || !h || !out) return GVOM_ERR_INVALID;
    const int W = c->world, me = c->rank;
    if (W > GVOM_COMM_MAX_RANKS) return GVOM_ERR_INVALID;
    int rc = gvom_comm_before_scan(c);
    if (rc) return rc;
    int64_t mine[2 * GVOM_COMM_MAX_RANKS + 2], sq[GVOM_COMM_MAX_RANKS], se[GVOM_COMM_MAX_RANKS];
    int any = 0;
    if ((rc = gvom_shard_scan_local(h, xyz, on_device, n, row_stride_bytes, dtype, ego, transform_4x4, sq, se, &any))) {
        c->err = std::string("gvom_shard_scan_local failed: ") + gvom_last_error(h);
        if (c->seg && W > 1) c->seg->poison.store((uint32_t)me + 1u, std::memory_order_release);   // the others must not wait for this rank
        return rc;
    }
    for (int d = 0; d < W; ++d) { mine[d] = sq[d]; mine[W + d] = se[d]; }
    mine[2 * W] = any; mine[2 * W + 1] = n;
    const int k = 2 * W + 2;
    std::vector<int64_t> &table = c->scan_table;
    table.resize((size_t)k * W);
    if ((rc = gvom_comm_exchange_host(c, mine, k, table.data()))) return rc;
    int64_t rq[GVOM_COMM_MAX_RANKS], re[GVOM_COMM_MAX_RANKS], accept = 0, total_n = 0, sent = 0, got = 0;
    for (int s_ = 0; s_ < W; ++s_) {
        const int64_t *row = table.data() + (size_t)s_ * k;
        rq[s_] = s_ != me ? row[me] : 0;
        re[s_] = s_ != me ? row[W + me] : 0;
        accept |= row[2 * W] != 0;
        total_n += row[2 * W + 1];
        got += 1028 * rq[s_] + 8 * re[s_];
    }
    sq[me] = 0; se[me] = 0;
    for (int d = 0; d < W; ++d) sent += 1028 * sq[d] + 8 * se[d];
    if ((rc = gvom_shard_recv_reserve(h, re))) { c->err = std::string("gvom_shard_recv_reserve failed: ") + gvom_last_error(h); return rc; }
    if ((rc = gvom_comm_exchange_scan(c, h, sq, se, rq, re))) return rc;
    if ((rc = gvom_shard_scan_merge(h, rq, re, accept ? 1 : 0))) { c->err = std::string("gvom_shard_scan_merge failed: ") + gvom_last_error(h); return rc; }
    out[0] = accept; out[1] = total_n; out[2] = sent; out[3] = got;
    return GVOM_OK;
}

// ONE sharded combine, natively (before_combine, the slab's fusion, the all-gather of the height rows, the 2-D maps into the
// pinned buffer): returns GVOM_EMPTY_BUFFER as gvom_combine_fuse does.
VIS int gvom_comm_combine_maps_into(gvom_comm_t *c, gvom_t *h, double origin_world[3], void *pinned_out)
{
    if (!c || !h || !pinned_out) return GVOM_ERR_INVALID;
    int rc = gvom_comm_before_combine(c);
    if (rc) return rc;
    rc = gvom_combine_fuse(h, nullptr);
    if (rc == GVOM_EMPTY_BUFFER) return rc;
    if (rc) { c->err = std::string("gvom_combine_fuse failed: ") + gvom_last_error(h); return rc; }
    if ((rc = gvom_comm_allgather_rows(c, h))) return rc;
    if ((rc = gvom_combine_map2d_into(h, origin_world, pinned_out))) c->err = std::string("gvom_combine_map2d_into failed: ") + gvom_last_error(h);
    return rc;
}

// peer transport bookkeeping: {bytes pulled, copies, exports made, refused hipIpcOpenMemHandle calls that were repeated}
VIS int gvom_comm_peer_stats(gvom_comm_t *c, int64_t out[4])
{
    if (!c || !out) return GVOM_ERR_INVALID;
    out[0] = (int64_t)c->peer_bytes; out[1] = (int64_t)c->peer_copies; out[2] = (int64_t)c->export_seq; out[3] = (int64_t)c->peer_open_retries;
    return GVOM_OK;
}

// RCCL calls this rank has issued: {ncclSend + ncclRecv calls, their bytes, groups closed, ncclAllGather calls}
VIS int gvom_comm_wire_stats(gvom_comm_t *c, int64_t out[4])
{
    if (!c || !out) return GVOM_ERR_INVALID;
    out[0] = (int64_t)c->wire_ops; out[1] = (int64_t)c->wire_bytes; out[2] = (int64_t)c->wire_groups; out[3] = (int64_t)c->wire_allgathers;
    return GVOM_OK;
}

// A rank that cannot go on (its caller failed outside the library) says so: the others' next wait ends with an error naming it
// instead of lasting GVOM_COMM_TIMEOUT_S.  The communicator stays broken.
VIS int gvom_comm_abort(gvom_comm_t *c)
{
    if (!c || !c->seg) return GVOM_ERR_INVALID;
    c->seg->poison.store((uint32_t)c->rank + 1u, std::memory_order_release);
    return GVOM_OK;
}

// regions that were moved into a fresh allocation because the HSA runtime refused to export the old one or a peer to open it
VIS int64_t gvom_comm_peer_renewed(gvom_comm_t *c) { return c ? (int64_t)c->peer_renewed : -1; }

// What the communicator itself says about the job: out = {ranks RCCL counts in its communicator (ncclCommCount; -1 without
// RCCL), this rank's number there (ncclCommUserRank), the HIP device, the transport in use}; busid: the device's PCI bus id.
// For the bench line: N ranks on N distinct bus ids = N GPUs.
VIS int gvom_comm_info(gvom_comm_t *c, int64_t out[4], char *busid, size_t busid_len)
{
    if (!c || !out) return GVOM_ERR_INVALID;
    out[0] = out[1] = -1; out[2] = c->device; out[3] = c->transport;
    if (c->nccl) {
        typedef ncclResult_t (*count_fn)(const ncclComm_t, int *);
        count_fn cnt = (count_fn)dlsym(c->rccl.lib, "ncclCommCount"), usr = (count_fn)dlsym(c->rccl.lib, "ncclCommUserRank");
        int v = -1;
        if (cnt && cnt(c->nccl, &v) == ncclSuccess) out[0] = v;
        v = -1;
        if (usr && usr(c->nccl, &v) == ncclSuccess) out[1] = v;
    }
    if (busid && busid_len) {
        busid[0] = 0;
        if (c->device >= 0 && hipDeviceGetPCIBusId(busid, (int)busid_len, c->device) != hipSuccess) { (void)hipGetLastError(); busid[0] = 0; }
    }
    return GVOM_OK;
}

VIS int gvom_comm_rank(gvom_comm_t *c) { return c ? c->rank : -1; }
VIS int gvom_comm_world(gvom_comm_t *c) { return c ? c->world : -1; }

}  // extern "C"
