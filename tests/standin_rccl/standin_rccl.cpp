// standin_rccl.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A stand-in for librccl that lets several processes SHARING ONE GPU run the engine's native
// exchange path (cholmod_hip_rccl_attach: ncclCommInitRank, ncclCommSplit, stream-ordered
// collectives) -- the real RCCL refuses two ranks on one device, and the test box has one
// device.  The engine binds it through CHOLMOD_HIP_RCCL_LIBRARY (engine.hip: rccl_api).
//
// Semantics kept:
//  * communicators, ncclCommSplit by colour / key (NCCL_SPLIT_NOCOLOR -> NULL);
//  * ASYNCHRONY (round 4): a collective call returns at once.  It records an event on the stream
//    it is given, hands the operation to a helper thread of the process and enqueues a small
//    kernel on that stream that waits for the helper's completion flag (host-coherent memory).
//    The helper waits for the event ON THE DEVICE (hipStreamWaitEvent on its own stream), stages
//    the data through a POSIX shared-memory segment, meets the peers there, writes the result back
//    and raises the flag.  So the collective is ordered with the work before and after it on ITS
//    stream and with nothing else: kernels on other streams run beside it, and an engine that
//    forgot an event between its two streams reads stale data here as it would over RCCL.
//    The operations of one process are executed in the order of the calls (one helper), which is
//    what NCCL promises per communicator and the engine provides across communicators (every rank
//    walks the same global launch list).  STANDIN_RCCL_SYNC=1 restores the blocking behaviour of
//    round 3 (the host waits inside every call).
//  * summation order: a reduce-scatter sums segment d starting with member d + 1 and ending with
//    member d, as a ring does -- the copies of a value that travels in several segments differ in
//    the last bits between the members, so the engine may not rely on bit-identical sums
//    (STANDIN_RCCL_RANK_ORDER=1: every segment in rank order).  All-reduce sums in rank order and
//    hands every member the same bits, as RCCL's ring does (reduce-scatter + all-gather).
// Not kept: bandwidth, concurrency of two communicators of one process.  fp64 sum / plain byte moves only.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace {

constexpr int MAXR = 16 ;               // ranks per world
constexpr int MAXCOMM = 4096 ;          // communicators per world (ids are never reused)
constexpr size_t CHUNK = (size_t) 1 << 20 ;     // doubles per rank slot (8 MB)

struct Barrier { std::atomic<int> count ; std::atomic<int> gen ; } ;
struct Shm {
    std::atomic<int> magic ;
    std::atomic<int> next_comm ;        // id allocator
    std::atomic<int> attached ;
    Barrier bar [MAXCOMM] ;
    int color [MAXCOMM][MAXR], key [MAXCOMM][MAXR] ;   // ncclCommSplit scratch of the parent communicator
    int base [MAXCOMM] ;
    std::atomic<long long> calls ;      // collectives executed (all ranks), for the tests
    alignas (64) double slot [MAXR][CHUNK] ;
} ;

struct Comm {
    Shm *shm ;
    int id ;                    // barrier / scratch index
    int nranks, rank ;          // in this communicator
    int world_rank [MAXR] ;     // slot of every member
    bool is_world ;
    char name [64] ;
    hipStream_t side ;          // the process's staging stream (helper thread)
} ;

// ---- the helper thread of the process -----------------------------------------------------
// One per process, started with the first communicator.  A collective call pushes a closure
// (the blocking implementation of round 3, run on the helper's stream) and returns.
__global__ void k_standin_wait (const volatile unsigned *done, unsigned id)
{
    // (ids only grow and the helper completes them in order: >= with wrap-around)
    while ((int) (__hip_atomic_load (done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - id) < 0) __builtin_amdgcn_s_sleep (64) ;
}

struct Helper {
    std::mutex mu ;
    std::condition_variable cv, idle ;
    std::deque<std::function<ncclResult_t ()>> q ;
    std::deque<std::pair<hipEvent_t, unsigned>> meta ;      // (event to wait for, id) of each queued closure
    unsigned next_id = 1, done_host = 0 ;
    volatile unsigned *done = nullptr ;                     // host-coherent, read by k_standin_wait
    hipStream_t side = nullptr ;
    int device = 0 ;
    std::atomic<int> failed {0} ;                           // sticky: a staged operation failed
    bool sync = false ;
    std::thread th ;
    void run ()
    {
        (void) hipSetDevice (device) ;
        for ( ; ; )
        {
            std::function<ncclResult_t ()> fn ;
            std::pair<hipEvent_t, unsigned> m ;
            {
                std::unique_lock<std::mutex> lk (mu) ;
                cv.wait (lk, [this] { return !q.empty () ; }) ;
                fn = std::move (q.front ()) ; m = meta.front () ;
            }
            ncclResult_t r = ncclSuccess ;
            if (hipStreamWaitEvent (side, m.first, 0) != hipSuccess) r = ncclUnhandledCudaError ;
            if (r == ncclSuccess) r = fn () ;
            if (r == ncclSuccess && hipStreamSynchronize (side) != hipSuccess) r = ncclUnhandledCudaError ;
            if (r != ncclSuccess)
            {
                fprintf (stderr, "standin_rccl: a queued collective failed (%d)\n", (int) r) ;
                failed.store ((int) r) ;
            }
            (void) hipEventDestroy (m.first) ;
            __atomic_store_n ((unsigned *) done, m.second, __ATOMIC_RELEASE) ;      // the waiting kernel goes on (also after a failure: no hang)
            {
                std::lock_guard<std::mutex> lk (mu) ;
                q.pop_front () ; meta.pop_front () ;
                done_host = m.second ;
            }
            idle.notify_all () ;
        }
    }
} ;
Helper *g_helper = nullptr ;
std::mutex g_helper_mu ;

Helper *helper ()
{
    std::lock_guard<std::mutex> lk (g_helper_mu) ;
    if (g_helper) return g_helper ;
    Helper *h = new Helper ;        // (never destroyed: its thread lives as long as the process)
    if (hipGetDevice (&h->device) != hipSuccess) { delete h ; return nullptr ; }
    if (hipStreamCreateWithFlags (&h->side, hipStreamNonBlocking) != hipSuccess) { delete h ; return nullptr ; }
    void *p = nullptr ;
    if (hipHostMalloc (&p, 64, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) { delete h ; return nullptr ; }
    h->done = (volatile unsigned *) p ;
    *h->done = 0 ;
    const char *e = getenv ("STANDIN_RCCL_SYNC") ;
    h->sync = e && atoi (e) != 0 ;
    h->th = std::thread ([h] { h->run () ; }) ;
    h->th.detach () ;
    g_helper = h ;
    return h ;
}

// everything queued so far has run (communicator management, blocking mode)
void drain (Helper *h)
{
    std::unique_lock<std::mutex> lk (h->mu) ;
    h->idle.wait (lk, [h] { return h->q.empty () ; }) ;
}

// The asynchronous frame of every collective: `fn` is its blocking implementation on the helper's stream.
ncclResult_t enqueue (Comm *c, hipStream_t stream, std::function<ncclResult_t ()> fn)
{
    Helper *h = helper () ;
    if (!h) return ncclUnhandledCudaError ;
    if (h->failed.load ()) return (ncclResult_t) h->failed.load () ;
    hipEvent_t ev ;
    if (hipEventCreateWithFlags (&ev, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError ;
    if (hipEventRecord (ev, stream) != hipSuccess) { (void) hipEventDestroy (ev) ; return ncclUnhandledCudaError ; }
    unsigned id ;
    {
        std::lock_guard<std::mutex> lk (h->mu) ;
        id = h->next_id++ ;
        h->q.push_back (std::move (fn)) ;
        h->meta.push_back ({ev, id}) ;
    }
    h->cv.notify_one () ;
    c->shm->calls.fetch_add (1) ;
    hipLaunchKernelGGL (k_standin_wait, dim3 (1), dim3 (1), 0, stream, h->done, id) ;
    if (hipGetLastError () != hipSuccess) return ncclUnhandledCudaError ;
    if (h->sync)
    {
        if (hipStreamSynchronize (stream) != hipSuccess) return ncclUnhandledCudaError ;
        if (h->failed.load ()) return (ncclResult_t) h->failed.load () ;
    }
    return ncclSuccess ;
}

void spin_barrier (Comm *c)
{
    Barrier &b = c->shm->bar [c->id] ;
    int g = b.gen.load (std::memory_order_acquire) ;
    if (b.count.fetch_add (1, std::memory_order_acq_rel) == c->nranks - 1)
    {
        b.count.store (0, std::memory_order_relaxed) ;
        b.gen.store (g + 1, std::memory_order_release) ;
        return ;
    }
    long spins = 0 ;
    while (b.gen.load (std::memory_order_acquire) == g)
    {
        if (++spins > 2000) { struct timespec ts = {0, 50000} ; nanosleep (&ts, nullptr) ; }
        if (spins > 2000 + 20L * 60 * 1000 * 20)    // ~20 minutes: a peer died
        {
            fprintf (stderr, "standin_rccl: barrier of communicator %d timed out (rank %d of %d)\n", c->id, c->rank, c->nranks) ;
            abort () ;
        }
    }
}

#define HCHK(x) do { hipError_t e_ = (x) ; if (e_ != hipSuccess) { \
    fprintf (stderr, "standin_rccl: %s: %s\n", #x, hipGetErrorString (e_)) ; return ncclUnhandledCudaError ; } } while (0)

size_t type_bytes (ncclDataType_t t)
{
    switch (t)
    {
        case ncclInt8: case ncclUint8: return 1 ;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4 ;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8 ;
        default: return 0 ;
    }
}

// One staged exchange of at most CHUNK doubles per member: everybody publishes `mine`
// (n_pub elements), then `consume` runs with all slots visible, then a closing barrier.
template <typename F>
ncclResult_t staged (Comm *c, const void *dev_src, size_t bytes_pub, F consume)
{
    double *mine = c->shm->slot [c->world_rank [c->rank]] ;
    if (bytes_pub)
    {
        HCHK (hipMemcpyAsync (mine, dev_src, bytes_pub, hipMemcpyDeviceToHost, c->side)) ;
        HCHK (hipStreamSynchronize (c->side)) ;
    }
    spin_barrier (c) ;
    ncclResult_t r = consume () ;
    spin_barrier (c) ;
    return r ;
}

} // namespace

extern "C" {

const char *ncclGetErrorString (ncclResult_t r)
{
    return r == ncclSuccess ? "success" : r == ncclUnhandledCudaError ? "hip error (stand-in)" :
           r == ncclInvalidArgument ? "invalid argument (stand-in)" : "error (stand-in)" ;
}

ncclResult_t ncclGetUniqueId (ncclUniqueId *id)
{
    if (!id) return ncclInvalidArgument ;
    memset (id, 0, sizeof (*id)) ;
    unsigned long long r = 0 ;
    int fd = open ("/dev/urandom", O_RDONLY) ;
    if (fd >= 0) { if (read (fd, &r, sizeof (r)) != (ssize_t) sizeof (r)) r = (unsigned long long) getpid () * 2654435761u ; close (fd) ; }
    snprintf (id->internal, sizeof (id->internal), "/standin_rccl_%d_%016llx", (int) getpid (), r) ;
    int sfd = shm_open (id->internal, O_CREAT | O_EXCL | O_RDWR, 0600) ;
    if (sfd < 0) { perror ("standin_rccl: shm_open") ; return ncclSystemError ; }
    if (ftruncate (sfd, (off_t) sizeof (Shm)) != 0) { perror ("standin_rccl: ftruncate") ; close (sfd) ; return ncclSystemError ; }
    void *p = mmap (nullptr, sizeof (Shm), PROT_READ | PROT_WRITE, MAP_SHARED, sfd, 0) ;
    close (sfd) ;
    if (p == MAP_FAILED) { perror ("standin_rccl: mmap") ; return ncclSystemError ; }
    Shm *s = (Shm *) p ;        // (fresh segment: zero pages)
    s->next_comm.store (1) ;    // id 0 = the world
    s->magic.store (0x5ca1ab1e, std::memory_order_release) ;
    munmap (p, sizeof (Shm)) ;
    return ncclSuccess ;
}

ncclResult_t ncclCommInitRank (ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return ncclInvalidArgument ;
    int sfd = shm_open (id.internal, O_RDWR, 0600) ;
    if (sfd < 0) { perror ("standin_rccl: shm_open (attach)") ; return ncclSystemError ; }
    void *p = mmap (nullptr, sizeof (Shm), PROT_READ | PROT_WRITE, MAP_SHARED, sfd, 0) ;
    close (sfd) ;
    if (p == MAP_FAILED) return ncclSystemError ;
    Shm *s = (Shm *) p ;
    if (s->magic.load (std::memory_order_acquire) != 0x5ca1ab1e) return ncclInvalidArgument ;
    Comm *c = new Comm ;
    c->shm = s ; c->id = 0 ; c->nranks = nranks ; c->rank = rank ; c->is_world = true ;
    for (int q = 0 ; q < nranks ; q++) c->world_rank [q] = q ;
    snprintf (c->name, sizeof (c->name), "%s", id.internal) ;
    Helper *h = helper () ;
    if (!h) { delete c ; return ncclUnhandledCudaError ; }
    c->side = h->side ;
    // the staging slots as pinned memory of this process: the copies between the device and the segment are then plain DMA
    // (a pageable shared mapping moves at a few GB/s through the runtime's bounce buffers; eight ranks at Poisson 100^3
    // stage ~8 GB per factorization).  Best effort: without it the copies still work.
    if (hipHostRegister ((void *) s->slot, sizeof (s->slot), hipHostRegisterDefault) != hipSuccess) (void) hipGetLastError () ;
    spin_barrier (c) ;
    if (rank == 0) shm_unlink (id.internal) ;       // everybody holds a mapping now
    *comm = (ncclComm_t) c ;
    return ncclSuccess ;
}

ncclResult_t ncclCommSplit (ncclComm_t comm, int color, int key, ncclComm_t *newcomm, ncclConfig_t *)
{
    Comm *c = (Comm *) comm ;
    if (!c || !newcomm) return ncclInvalidArgument ;
    if (Helper *h = helper ()) drain (h) ;         // (communicator management blocks the host, as ncclCommSplit does)
    Shm *s = c->shm ;
    s->color [c->id][c->rank] = color ;
    s->key [c->id][c->rank] = key ;
    if (c->rank == 0) s->base [c->id] = s->next_comm.fetch_add (c->nranks) ;
    spin_barrier (c) ;
    *newcomm = nullptr ;
    ncclResult_t res = ncclSuccess ;
    if (color != NCCL_SPLIT_NOCOLOR)
    {
        std::vector<std::pair<std::pair<int, int>, int>> mem ;     // ((key, parent rank), parent rank)
        int lowest = -1 ;
        for (int q = 0 ; q < c->nranks ; q++)
            if (s->color [c->id][q] == color) { mem.push_back ({{s->key [c->id][q], q}, q}) ; if (lowest < 0) lowest = q ; }
        std::sort (mem.begin (), mem.end ()) ;
        Comm *n = new Comm ;
        n->shm = s ; n->id = s->base [c->id] + lowest ; n->nranks = (int) mem.size () ; n->rank = -1 ; n->is_world = false ;
        n->name [0] = 0 ;
        for (int q = 0 ; q < n->nranks ; q++)
        {
            n->world_rank [q] = c->world_rank [mem [q].second] ;
            if (mem [q].second == c->rank) n->rank = q ;
        }
        if (n->id >= MAXCOMM || n->rank < 0) { delete n ; res = ncclInternalError ; }
        else { n->side = c->side ; *newcomm = (ncclComm_t) n ; }
    }
    spin_barrier (c) ;          // the scratch of the parent may be reused
    return res ;
}

ncclResult_t ncclCommDestroy (ncclComm_t comm)
{
    Comm *c = (Comm *) comm ;
    if (!c) return ncclSuccess ;
    if (Helper *h = helper ()) drain (h) ;         // nothing of this communicator is in flight any more
    if (c->is_world) { (void) hipHostUnregister ((void *) c->shm->slot) ; (void) hipGetLastError () ; munmap (c->shm, sizeof (Shm)) ; }
    delete c ;
    return ncclSuccess ;
}

ncclResult_t ncclCommCount (const ncclComm_t comm, int *count) { if (!comm || !count) return ncclInvalidArgument ; *count = ((Comm *) comm)->nranks ; return ncclSuccess ; }
ncclResult_t ncclCommUserRank (const ncclComm_t comm, int *rank) { if (!comm || !rank) return ncclInvalidArgument ; *rank = ((Comm *) comm)->rank ; return ncclSuccess ; }

// test hook: collectives this world has executed so far (summed over the ranks)
long long standin_rccl_calls (ncclComm_t comm) { return comm ? ((Comm *) comm)->shm->calls.load () : -1 ; }

static ncclResult_t do_allreduce (Comm *c, const void *sendbuff, void *recvbuff, size_t count)
{
    std::vector<double> acc ;
    for (size_t o = 0 ; o < count || (count == 0 && o == 0) ; o += CHUNK)
    {
        size_t n = std::min (CHUNK, count - o) ;
        acc.assign (n, 0.0) ;
        ncclResult_t r = staged (c, (const double *) sendbuff + o, n * sizeof (double), [&] () -> ncclResult_t
        {
            for (int q = 0 ; q < c->nranks ; q++)
            {
                const double *s = c->shm->slot [c->world_rank [q]] ;
                for (size_t e = 0 ; e < n ; e++) acc [e] += s [e] ;
            }
            return ncclSuccess ;
        }) ;
        if (r != ncclSuccess) return r ;
        if (n)
        {
            HCHK (hipMemcpyAsync ((double *) recvbuff + o, acc.data (), n * sizeof (double), hipMemcpyHostToDevice, c->side)) ;
            HCHK (hipStreamSynchronize (c->side)) ;
        }
        if (count == 0) break ;
    }
    return ncclSuccess ;
}

ncclResult_t ncclAllReduce (const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op,
    ncclComm_t comm, hipStream_t stream)
{
    Comm *c = (Comm *) comm ;
    if (!c || dt != ncclDouble || op != ncclSum) return ncclInvalidArgument ;
    return enqueue (c, stream, [=] () { return do_allreduce (c, sendbuff, recvbuff, count) ; }) ;
}

static ncclResult_t do_reduce (Comm *c, const void *sendbuff, void *recvbuff, size_t count, int root)
{
    std::vector<double> acc ;
    for (size_t o = 0 ; o < count ; o += CHUNK)
    {
        size_t n = std::min (CHUNK, count - o) ;
        ncclResult_t r = staged (c, (const double *) sendbuff + o, n * sizeof (double), [&] () -> ncclResult_t
        {
            if (c->rank != root) return ncclSuccess ;
            acc.assign (n, 0.0) ;
            for (int q = 0 ; q < c->nranks ; q++)
            {
                const double *s = c->shm->slot [c->world_rank [q]] ;
                for (size_t e = 0 ; e < n ; e++) acc [e] += s [e] ;
            }
            return ncclSuccess ;
        }) ;
        if (r != ncclSuccess) return r ;
        if (c->rank == root)
        {
            HCHK (hipMemcpyAsync ((double *) recvbuff + o, acc.data (), n * sizeof (double), hipMemcpyHostToDevice, c->side)) ;
            HCHK (hipStreamSynchronize (c->side)) ;
        }
    }
    return ncclSuccess ;
}

ncclResult_t ncclReduce (const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, ncclRedOp_t op, int root,
    ncclComm_t comm, hipStream_t stream)
{
    Comm *c = (Comm *) comm ;
    if (!c || dt != ncclDouble || op != ncclSum || root < 0 || root >= c->nranks) return ncclInvalidArgument ;
    return enqueue (c, stream, [=] () { return do_reduce (c, sendbuff, recvbuff, count, root) ; }) ;
}

static ncclResult_t do_broadcast (Comm *c, const void *sendbuff, void *recvbuff, size_t bytes, int root)
{
    size_t cb = CHUNK * sizeof (double) ;
    for (size_t o = 0 ; o < bytes ; o += cb)
    {
        size_t n = std::min (cb, bytes - o) ;
        bool isroot = c->rank == root ;
        ncclResult_t r = staged (c, (const char *) sendbuff + o, isroot ? n : 0, [&] () -> ncclResult_t
        {
            if (isroot && recvbuff == sendbuff) return ncclSuccess ;
            HCHK (hipMemcpyAsync ((char *) recvbuff + o, c->shm->slot [c->world_rank [root]], n, hipMemcpyHostToDevice, c->side)) ;
            HCHK (hipStreamSynchronize (c->side)) ;
            return ncclSuccess ;
        }) ;
        if (r != ncclSuccess) return r ;
    }
    return ncclSuccess ;
}

ncclResult_t ncclBroadcast (const void *sendbuff, void *recvbuff, size_t count, ncclDataType_t dt, int root,
    ncclComm_t comm, hipStream_t stream)
{
    Comm *c = (Comm *) comm ;
    size_t tb = type_bytes (dt) ;
    if (!c || !tb || root < 0 || root >= c->nranks) return ncclInvalidArgument ;
    return enqueue (c, stream, [=] () { return do_broadcast (c, sendbuff, recvbuff, count * tb, root) ; }) ;
}

// recvbuff (recvcount) = sum over the members of their sendbuff [rank * recvcount ...)
static ncclResult_t do_reduce_scatter (Comm *c, const void *sendbuff, void *recvbuff, size_t recvcount)
{
    static const bool rank_order = [] () { const char *e = getenv ("STANDIN_RCCL_RANK_ORDER") ; return e && atoi (e) != 0 ; } () ;
    // the members' slots carry the piece of one destination at a time
    std::vector<double> acc ;
    for (int dest = 0 ; dest < c->nranks ; dest++)
        for (size_t o = 0 ; o < recvcount ; o += CHUNK)
        {
            size_t n = std::min (CHUNK, recvcount - o) ;
            ncclResult_t r = staged (c, (const double *) sendbuff + (size_t) dest * recvcount + o, n * sizeof (double), [&] () -> ncclResult_t
            {
                if (c->rank != dest) return ncclSuccess ;
                // ring order: segment d is summed starting with member d + 1 and ending with member d
                const int q0 = rank_order ? 0 : (dest + 1) % c->nranks ;
                const double *s0 = c->shm->slot [c->world_rank [q0]] ;
                acc.assign (s0, s0 + n) ;
                for (int t = 1 ; t < c->nranks ; t++)
                {
                    const double *s = c->shm->slot [c->world_rank [(q0 + t) % c->nranks]] ;
                    for (size_t e = 0 ; e < n ; e++) acc [e] += s [e] ;
                }
                return ncclSuccess ;
            }) ;
            if (r != ncclSuccess) return r ;
            if (c->rank == dest)
            {
                HCHK (hipMemcpyAsync ((double *) recvbuff + o, acc.data (), n * sizeof (double), hipMemcpyHostToDevice, c->side)) ;
                HCHK (hipStreamSynchronize (c->side)) ;
            }
        }
    return ncclSuccess ;
}

ncclResult_t ncclReduceScatter (const void *sendbuff, void *recvbuff, size_t recvcount, ncclDataType_t dt, ncclRedOp_t op,
    ncclComm_t comm, hipStream_t stream)
{
    Comm *c = (Comm *) comm ;
    if (!c || dt != ncclDouble || op != ncclSum) return ncclInvalidArgument ;
    return enqueue (c, stream, [=] () { return do_reduce_scatter (c, sendbuff, recvbuff, recvcount) ; }) ;
}

// recvbuff [q * sendcount ...) = member q's sendbuff
static ncclResult_t do_all_gather (Comm *c, const void *sendbuff, void *recvbuff, size_t bytes)
{
    size_t cb = CHUNK * sizeof (double) ;
    for (size_t o = 0 ; o < bytes ; o += cb)
    {
        size_t n = std::min (cb, bytes - o) ;
        ncclResult_t r = staged (c, (const char *) sendbuff + o, n, [&] () -> ncclResult_t
        {
            for (int q = 0 ; q < c->nranks ; q++)
            {
                char *dst = (char *) recvbuff + (size_t) q * bytes + o ;
                if (q == c->rank && dst == (const char *) sendbuff + o) continue ;      // in place
                HCHK (hipMemcpyAsync (dst, c->shm->slot [c->world_rank [q]], n, hipMemcpyHostToDevice, c->side)) ;
            }
            HCHK (hipStreamSynchronize (c->side)) ;
            return ncclSuccess ;
        }) ;
        if (r != ncclSuccess) return r ;
    }
    return ncclSuccess ;
}

ncclResult_t ncclAllGather (const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t dt,
    ncclComm_t comm, hipStream_t stream)
{
    Comm *c = (Comm *) comm ;
    size_t tb = type_bytes (dt) ;
    if (!c || !tb) return ncclInvalidArgument ;
    return enqueue (c, stream, [=] () { return do_all_gather (c, sendbuff, recvbuff, sendcount * tb) ; }) ;
}

ncclResult_t ncclGroupStart (void) { return ncclSuccess ; }
ncclResult_t ncclGroupEnd (void) { return ncclSuccess ; }

} // extern "C"
