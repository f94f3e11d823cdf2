// uwip_copier: the host <-> HBM copy engine of a rank (GpuMat::upload / download around the timed region,
// modules/histretch/src/histretch.cpp:165-216, for a stream of frame batches).
//
// Why it exists (measured, DESIGN.md section 5 "host-buffer mode"): with one upload and one download HIP stream per
// sub-batch (8 copy streams + 4 compute streams) the runtime multiplexes the 12 streams onto its 4 hardware queues;
// the barrier packet behind every hipStreamWaitEvent / hipEventRecord that involves a DMA copy then sits in a queue it
// shares with an unrelated compute stream and holds that stream's kernels for the length of the copy (7-30 ms).
// Here no such packet exists: a lane thread waits for a request's dependency ON THE HOST, hands the copy to the DMA
// engine when it can run, waits for it on the host, and publishes the ticket; the pipes wait for tickets on the host.
// Copies of one direction run one at a time at the full link rate (57 GB/s).  A lane serves the FIRST QUEUED request whose
// dependency has completed (hipEventQuery), so a download whose stream is still busy does not hold back the ready
// downloads of other pipes; requests that depend on the same stream are still served in their order (an event recorded
// later on a stream cannot complete before an earlier one).  When nothing is ready the lane sleeps on the head request's
// event.  Completion is published per ticket.
#include "uwip_internal.hpp"
#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <set>
#include <thread>

namespace {

struct Request {
    void *dst;
    const void *src;
    size_t bytes;
    hipEvent_t after;       // nullptr: no dependency
    uint64_t seq;
};

struct Lane {
    hipMemcpyKind kind;
    hipStream_t stream = nullptr;
    hipEvent_t done_ev = nullptr;          // blocking-sync: the lane thread sleeps while its copy runs
    std::thread th;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<Request> q;
    std::vector<hipEvent_t> free_events;
    uint64_t next_seq = 1, done_seq = 0;   // every seq <= done_seq has completed ...
    std::set<uint64_t> done_ahead;         // ... and so have these (served ahead of an unready predecessor)
    bool is_done(uint64_t seq) const { return seq <= done_seq || done_ahead.count(seq) != 0; }
    bool stop = false;
};

}  // namespace

struct uwip_copier {
    int device = 0;
    Lane lane[2];           // 0 = upload, 1 = download
    std::mutex err_mu;
    std::string err;
    std::atomic<bool> failed{false};

    void fail(const char *what, hipError_t e)
    {
        std::lock_guard<std::mutex> g(err_mu);
        if (!failed.exchange(true)) { err = what; err += ": "; err += hipGetErrorString(e); }
    }
    void run(Lane &L)
    {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) fail("hipSetDevice (copy lane)", e);
        for (;;) {
            Request r;
            bool ready = false;
            {
                // The first queued request that can run now; none: the head, and sleep on its dependency.  A request never
                // overtakes an earlier one when one of the two WRITES bytes the other touches (an upload with after = NULL into
                // a buffer an earlier upload with a dependency still targets; a download into a host buffer an earlier one
                // still fills): those stay FIFO; two reads of one buffer do not order each other.  The dependency events are queried on a snapshot, outside the lane's mutex (submitters only append).
                std::vector<Request> snap;
                {
                    std::unique_lock<std::mutex> lk(L.mu);
                    L.cv_work.wait(lk, [&] { return L.stop || !L.q.empty(); });
                    if (L.q.empty()) return;                    // stop requested and nothing left
                    snap.assign(L.q.begin(), L.q.end());
                }
                auto overlaps = [](const void *a, size_t an, const void *b, size_t bn) {
                    const char *x = (const char *)a, *y = (const char *)b;
                    return x < y + bn && y < x + an;
                };
                uint64_t pick_seq = snap[0].seq;
                for (size_t i = 0; i < snap.size(); ++i) {
                    bool blocked = false;
                    for (size_t j = 0; j < i && !blocked; ++j)
                        blocked = overlaps(snap[i].dst, snap[i].bytes, snap[j].dst, snap[j].bytes) ||
                                  overlaps(snap[i].dst, snap[i].bytes, snap[j].src, snap[j].bytes) || overlaps(snap[i].src, snap[i].bytes, snap[j].dst, snap[j].bytes);
                    if (blocked) continue;
                    if (!snap[i].after || hipEventQuery(snap[i].after) == hipSuccess) { pick_seq = snap[i].seq; ready = true; break; }
                }
                std::lock_guard<std::mutex> lk(L.mu);
                size_t pick = 0;
                while (pick < L.q.size() && L.q[pick].seq != pick_seq) ++pick;      // this lane's thread is the only remover
                r = L.q[pick];
                L.q.erase(L.q.begin() + (std::ptrdiff_t)pick);
            }
            if (r.after && !ready) {
                e = uwip_event_wait(r.after, 100);
                if (e != hipSuccess) fail("hipEventSynchronize (copy dependency)", e);
            }
            if (!failed.load()) {
                e = hipMemcpyAsync(r.dst, r.src, r.bytes, L.kind, L.stream);
                // sleep-poll until the DMA engine is done (hipStreamSynchronize spins for the 7 ms of a 400 MB copy); 50 us
                // steps: the next copy of this direction starts when this one is seen complete
                if (e == hipSuccess) e = hipEventRecord(L.done_ev, L.stream);
                if (e == hipSuccess) e = uwip_event_wait(L.done_ev, 50);
                if (e != hipSuccess) fail(L.kind == hipMemcpyHostToDevice ? "upload" : "download", e);
            }
            {
                std::lock_guard<std::mutex> lk(L.mu);
                if (r.after) L.free_events.push_back(r.after);
                L.done_ahead.insert(r.seq);
                while (!L.done_ahead.empty() && *L.done_ahead.begin() == L.done_seq + 1) {
                    L.done_seq++;
                    L.done_ahead.erase(L.done_ahead.begin());
                }
            }
            L.cv_done.notify_all();
        }
    }
};

static int submit(uwip_copier *c, int which, uwip_ctx *after, void *dst, const void *src, size_t bytes, uint64_t *ticket)
{
    if (!c || !ticket) return UWIP_ERR_INVALID;
    *ticket = 0;
    if (bytes == 0) return UWIP_OK;
    if (!dst || !src) return UWIP_ERR_INVALID;
    if (c->failed.load()) return UWIP_ERR_HIP;
    Lane &L = c->lane[which];
    hipEvent_t ev = nullptr;
    if (after) {
        if (after->device != c->device) return UWIP_ERR_INVALID;
        if (int rc = uwip_enter(after)) return rc;
        {
            std::lock_guard<std::mutex> lk(L.mu);
            if (!L.free_events.empty()) { ev = L.free_events.back(); L.free_events.pop_back(); }
        }
        // the lane thread sleep-polls the dependency (uwip_event_wait) instead of spinning in hipEventSynchronize
        if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return UWIP_ERR_HIP;
        hipError_t e = hipEventRecord(ev, after->stream);
        if (e != hipSuccess) { (void)hipEventDestroy(ev); return after->fail(UWIP_ERR_HIP, "hipEventRecord (copier)", hipGetErrorString(e)); }
    }
    {
        std::lock_guard<std::mutex> lk(L.mu);
        const uint64_t seq = L.next_seq++;
        L.q.push_back(Request{dst, src, bytes, ev, seq});
        *ticket = (seq << 1) | (uint64_t)which;
    }
    L.cv_work.notify_one();
    return UWIP_OK;
}

UWIP_API int uwip_copier_create(int device, uwip_copier **out)
{
    if (!out) return UWIP_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return UWIP_ERR_HIP;
    if (device < 0 || device >= n) return UWIP_ERR_INVALID;
    int caller_dev = 0;
    (void)hipGetDevice(&caller_dev);
    if (hipSetDevice(device) != hipSuccess) return UWIP_ERR_HIP;
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{caller_dev};   // the caller's current device is left as found
    uwip_copier *c = new (std::nothrow) uwip_copier();
    if (!c) return UWIP_ERR_NOMEM;
    c->device = device;
    c->lane[0].kind = hipMemcpyHostToDevice;
    c->lane[1].kind = hipMemcpyDeviceToHost;
    for (auto &L : c->lane)
        if (hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&L.done_ev, hipEventDisableTiming) != hipSuccess) {
            for (auto &M : c->lane) { if (M.stream) (void)hipStreamDestroy(M.stream); if (M.done_ev) (void)hipEventDestroy(M.done_ev); }
            delete c;
            return UWIP_ERR_HIP;
        }
    for (auto &L : c->lane) L.th = std::thread([c, &L] { c->run(L); });
    *out = c;
    return UWIP_OK;
}

UWIP_API int uwip_copier_destroy(uwip_copier *c)
{
    if (!c) return UWIP_OK;
    for (auto &L : c->lane) {
        { std::lock_guard<std::mutex> lk(L.mu); L.stop = true; }
        L.cv_work.notify_all();
    }
    for (auto &L : c->lane) if (L.th.joinable()) L.th.join();       // each lane finishes its queue first
    int caller_dev = 0;
    (void)hipGetDevice(&caller_dev);
    (void)hipSetDevice(c->device);
    struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{caller_dev};
    for (auto &L : c->lane) {
        for (auto ev : L.free_events) (void)hipEventDestroy(ev);
        (void)hipStreamDestroy(L.stream);
        if (L.done_ev) (void)hipEventDestroy(L.done_ev);
    }
    delete c;
    return UWIP_OK;
}

UWIP_API int uwip_copier_upload(uwip_copier *c, uwip_ctx *after, void *d_dst, const void *h_src, size_t bytes, uint64_t *ticket)
{
    return submit(c, 0, after, d_dst, h_src, bytes, ticket);
}

UWIP_API int uwip_copier_download(uwip_copier *c, uwip_ctx *after, void *h_dst, const void *d_src, size_t bytes, uint64_t *ticket)
{
    return submit(c, 1, after, h_dst, d_src, bytes, ticket);
}

UWIP_API int uwip_copier_wait(uwip_copier *c, uint64_t ticket)
{
    if (!c) return UWIP_ERR_INVALID;
    if (ticket == 0) return UWIP_OK;
    Lane &L = c->lane[ticket & 1];
    const uint64_t seq = ticket >> 1;
    std::unique_lock<std::mutex> lk(L.mu);
    if (seq >= L.next_seq) return UWIP_ERR_INVALID;
    L.cv_done.wait(lk, [&] { return L.is_done(seq); });
    return c->failed.load() ? UWIP_ERR_HIP : UWIP_OK;
}

UWIP_API int uwip_copier_query(uwip_copier *c, uint64_t ticket, int *done)
{
    if (!c || !done) return UWIP_ERR_INVALID;
    *done = 1;
    if (ticket == 0) return UWIP_OK;
    Lane &L = c->lane[ticket & 1];
    std::lock_guard<std::mutex> lk(L.mu);
    if ((ticket >> 1) >= L.next_seq) return UWIP_ERR_INVALID;
    *done = L.is_done(ticket >> 1) ? 1 : 0;
    return c->failed.load() ? UWIP_ERR_HIP : UWIP_OK;
}

UWIP_API const char *uwip_copier_last_error(const uwip_copier *c)
{
    if (!c) return "null copier";
    std::lock_guard<std::mutex> g(const_cast<uwip_copier *>(c)->err_mu);
    return c->err.c_str();
}
