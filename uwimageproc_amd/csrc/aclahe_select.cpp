// ACLAHE parameter selection, host C++.
//
// The reference does this stage in Python with scipy
// (modules/aclahe/python/ACLAHE.py:66-129, functions.py:49-93):
//   curve_fit(f, u, Y, p0=(7,0.4,0.9,5))        -> MINPACK lmdif (leastsq defaults)
//   splrep(x22, f(x22)), splev(.., der=1|2)     -> FITPACK interpolating cubic
//   curvature |x'y'' - y'x''| / (x'^2+y'^2)^1.5 -> arg-max index
// and TODO:13 of the reference says the C++ port stalled for want of a curve
// fitting library.  This file restates the two published algorithms it needs:
// MINPACK-1 lmdif (More, Garbow, Hillstrom 1980: fdjac2, qrfac, lmpar, qrsolv,
// enorm) with scipy.optimize.leastsq's defaults (ftol = xtol = 1.49012e-8,
// gtol = 0, maxfev = 200*(n+1), factor = 100, automatic scaling), and the
// not-a-knot cubic interpolant that FITPACK's curfit returns for s = 0.
// Pinned by tests/golden/aclahe_knee.npz (indices produced by the reference's
// own functions.py under scipy 1.15.3).
#include "uwip_internal.hpp"
#include "lm_core.hpp"
#include <cmath>
#include <cstring>
#include <sched.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

namespace {

using namespace uwip_lm;
using HV = HostLanes::vec;

const SplineElim &spline_elim()
{
    static const SplineElim E = [] { SplineElim e; make_spline_elim(e); return e; }();
    return E;
}

// DerivadaX (functions.py:67-80): the fit of the clip-limit axis and the derivatives of its spline.  The axis does not
// depend on the image: for the standard grid 0.5 .. 24.5 it is computed once per process (and uploaded once per context
// for the device form).
struct XFit {
    int info;
    HV d1, d2;
};

XFit fit_x(const float *xs)
{
    XFit r{};
    HV xd{};
    for (int i = 0; i < MS; ++i) xd.v[i] = (double)xs[i];
    double px[4] = {7, 0.4, 0.9, 5};
    r.info = lmdif<HostLanes>(xd, MS, px);
    HV x22;
    model_at_knots<HostLanes>(px, x22);
    spline_derivs<HostLanes>(spline_elim(), x22, r.d1, r.d2);
    return r;
}

const XFit &standard_xfit()
{
    static const XFit cached = [] {
        float xs[MS];
        for (int i = 0; i < MS; ++i) xs[i] = 0.5f * (float)(i + 1);
        return fit_x(xs);
    }();
    return cached;
}

// DerivadaY + DerivadaX + Curvatura (functions.py:49-93); -1 when curve_fit would raise
int knee_index(const float *xs, const float *ys)
{
    HV yd{};
    bool standard = true;
    for (int i = 0; i < MS; ++i) {
        yd.v[i] = (double)ys[i];
        standard = standard && (xs[i] == 0.5f * (float)(i + 1));
    }
    XFit local;
    const XFit *xf = &local;
    if (standard) xf = &standard_xfit();
    else local = fit_x(xs);
    return knee_from_curve<HostLanes>(spline_elim(), yd, xf->d1, xf->d2, xf->info);
}

}  // namespace

// aclahe_device.hip: the constants the device form reads (fitted / eliminated by the host form's own code)
void uwip_aclahe_knee_consts(uwip_lm::SplineElim *E, double *xd1, double *xd2, int *x_info)
{
    *E = spline_elim();
    const XFit &xf = standard_xfit();
    for (int i = 0; i < 64; ++i) {
        xd1[i] = i < HostLanes::W ? xf.d1.v[i] : 0.0;
        xd2[i] = i < HostLanes::W ? xf.d2.v[i] : 0.0;
    }
    *x_info = xf.info;
}

namespace {

// float32 -> float16 (round to nearest even) -> float, for the BS comparison of ACLAHE.py:102
float through_half(float f)
{
    _Float16 h = (_Float16)f;
    return (float)h;
}

// ---- persistent host worker pool -----------------------------------------------------------------------------------------
// One per process (= per rank), created on first use and shared by all contexts: min(16, CPUs in the process's affinity
// mask - 1) workers (UWIP_HOST_THREADS overrides), so eight ranks that each pinned themselves to their GPU's NUMA node
// share the host without oversubscribing it, and a step creates no threads.  parallel_for splits [0, n) into chunks that
// the workers AND the calling thread pull from a shared counter; several callers (one per sub-batch stream) may be inside
// at once.
class HostPool {
public:
    static HostPool &get()
    {
        static HostPool pool;
        return pool;
    }
    void parallel_for(int n, const std::function<void(int, int)> &fn)
    {
        if (n <= 0) return;
        const int workers = (int)th_.size();
        if (n < 4 || workers == 0) { fn(0, n); return; }
        auto job = std::make_shared<Job>();
        job->fn = &fn;
        job->n = n;
        job->grain = std::max(1, n / (4 * (workers + 1)));
        job->chunks = (n + job->grain - 1) / job->grain;
        job->left.store(job->chunks);
        const int helpers = std::min(workers, job->chunks - 1);
        {
            std::lock_guard<std::mutex> lk(mu_);
            for (int i = 0; i < helpers; ++i) q_.push_back(job);
        }
        if (helpers == 1) cv_.notify_one(); else cv_.notify_all();
        run(*job);                                           // the caller works too
        std::unique_lock<std::mutex> lk(job->mu);
        job->cv.wait(lk, [&] { return job->left.load() == 0; });
    }

private:
    struct Job {
        const std::function<void(int, int)> *fn = nullptr;
        int n = 0, grain = 1, chunks = 0;
        std::atomic<int> next{0}, left{0};
        std::mutex mu;
        std::condition_variable cv;
    };
    static void run(Job &j)
    {
        for (;;) {
            const int c = j.next.fetch_add(1);
            if (c >= j.chunks) return;
            const int a = c * j.grain, b = std::min(j.n, a + j.grain);
            (*j.fn)(a, b);
            if (j.left.fetch_sub(1) == 1) {
                std::lock_guard<std::mutex> lk(j.mu);
                j.cv.notify_all();
            }
        }
    }
    HostPool()
    {
        int want = -1;
        if (const char *e = std::getenv("UWIP_HOST_THREADS")) want = std::atoi(e);
        budget_ = cpu_budget(&ranks_);
        if (want < 0) {
            // the caller works too, so budget - 1 helpers; never more than 16 (a sub-batch has 64 frames: the selection
            // is ~0.3 ms each, more threads only add wake-ups)
            want = std::min(16, std::max(0, (int)std::floor(budget_ + 1e-9) - 1));
        }
        for (int i = 0; i < want; ++i) th_.emplace_back([this] { loop(); });
    }
    // CPUs this rank may really use: min(affinity mask, cgroup CPU quota) / ranks on the node.  The affinity mask alone
    // overstates it: the GPU boxes show 256 CPUs in the mask under a cgroup quota of 16, and eight ranks of a node share it.
    static double cpu_budget(int *ranks_out)
    {
        double cpus = 0;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = CPU_COUNT(&set);
        if (cpus <= 0) cpus = (double)std::thread::hardware_concurrency();
        const char *root = std::getenv("UWIP_CGROUP_ROOT");           // tests point this at a fake tree
        const std::string base = root && *root ? root : "/sys/fs/cgroup";
        double quota = 0;
        if (FILE *f = std::fopen((base + "/cpu.max").c_str(), "r")) {          // cgroup v2: "<quota|max> <period>"
            char q[64] = {0};
            double per = 0;
            if (std::fscanf(f, "%63s %lf", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) quota = std::atof(q) / per;
            std::fclose(f);
        } else {
            double q = 0, per = 0;
            if (FILE *a = std::fopen((base + "/cpu/cpu.cfs_quota_us").c_str(), "r")) { if (std::fscanf(a, "%lf", &q) != 1) q = 0; std::fclose(a); }
            if (FILE *b = std::fopen((base + "/cpu/cpu.cfs_period_us").c_str(), "r")) { if (std::fscanf(b, "%lf", &per) != 1) per = 0; std::fclose(b); }
            if (q > 0 && per > 0) quota = q / per;
        }
        if (quota > 0 && quota < cpus) cpus = quota;
        int ranks = 1;
        for (const char *name : {"UWIP_RANKS_ON_NODE", "LOCAL_WORLD_SIZE"})
            if (const char *e = std::getenv(name)) { const int v = std::atoi(e); if (v > 0) { ranks = v; break; } }
        if (ranks_out) *ranks_out = ranks;
        return std::max(1.0, cpus / ranks);
    }
public:
    int workers() const { return (int)th_.size(); }
    double budget() const { return budget_; }
    int ranks() const { return ranks_; }
private:
    double budget_ = 1.0;
    int ranks_ = 1;
    ~HostPool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void loop()
    {
        for (;;) {
            std::shared_ptr<Job> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                job = q_.front();
                q_.pop_front();
            }
            run(*job);
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Job>> q_;
    bool stop_ = false;
};

}  // namespace

UWIP_API int uwip_host_pool_info(int *workers, double *cpu_budget, int *ranks_on_node)
{
    HostPool &p = HostPool::get();
    if (workers) *workers = p.workers();
    if (cpu_budget) *cpu_budget = p.budget();
    if (ranks_on_node) *ranks_on_node = p.ranks();
    return UWIP_OK;
}

UWIP_API int uwip_aclahe_knee(const float *h_xs49, const float *h_ys49, int32_t *index)
{
    if (!h_xs49 || !h_ys49 || !index) return UWIP_ERR_INVALID;
    *index = knee_index(h_xs49, h_ys49);
    return UWIP_OK;
}

// h_entropy [frames][5][51] -> per frame BS (block size), CL (clip limit, an index as the reference
// uses it).  h_knee (optional) [frames][5].  h_need_eval (optional) [frames]: 1 when 2*CL > 50, i.e.
// the BS choice needs entropies at a clip limit outside the swept grid (the caller supplies them
// through h_extra [frames][5], used when h_extra_valid[frame] != 0).
int uwip_aclahe_select_internal(const float *h_entropy, int frames, int32_t *h_bs, int32_t *h_cl, int32_t *h_knee,
                                int32_t *h_need_eval, const float *h_extra, const int32_t *h_extra_valid);

UWIP_API int uwip_aclahe_select(const float *h_entropy, int frames, int32_t *h_bs, int32_t *h_cl, int32_t *h_knee)
{
    return uwip_aclahe_select_internal(h_entropy, frames, h_bs, h_cl, h_knee, nullptr, nullptr, nullptr);
}

int uwip_aclahe_select_internal(const float *h_entropy, int frames, int32_t *h_bs, int32_t *h_cl, int32_t *h_knee,
                                int32_t *h_need_eval, const float *h_extra, const int32_t *h_extra_valid)
{
    static const int BlockSize[5] = {2, 4, 8, 16, 32};
    if (frames < 0 || (frames > 0 && (!h_entropy || !h_bs || !h_cl))) return UWIP_ERR_INVALID;
    const std::function<void(int, int)> work = [&](int f0, int f1) {
        for (int f = f0; f < f1; ++f) {
            const float *tab = h_entropy + (size_t)f * 5 * 51;
            float xs[49];
            for (int i = 0; i < 49; ++i) xs[i] = 0.5f * (float)(i + 1);
            int d = -1;
            for (int g = 0; g < 5; ++g) {
                const int k = knee_index(xs, tab + (size_t)g * 51 + 1);
                if (h_knee) h_knee[(size_t)f * 5 + g] = k;
                if (k > d) d = k;
            }
            if (d < 0) d = 0;
            const bool outside = 2 * d > 50;
            const bool have_extra = outside && h_extra && h_extra_valid && h_extra_valid[f];
            if (h_need_eval) h_need_eval[f] = (outside && !have_extra) ? 1 : 0;
            int w = 0;
            float best = 0.f;
            for (int g = 0; g < 5; ++g) {
                const float e = have_extra ? h_extra[(size_t)f * 5 + g] : tab[(size_t)g * 51 + (outside ? 50 : 2 * d)];
                const float h = through_half(e);
                if (g == 0 || h >= best) { best = h; w = g; }      // last maximum wins (ACLAHE.py:118-124)
            }
            h_bs[f] = BlockSize[w];
            h_cl[f] = d;
        }
    };
    // ~0.3 ms of Levenberg-Marquardt per frame: spread over the process's persistent host pool (no thread is created
    // per call; the pool is sized from the CPUs this rank may run on, see HostPool)
    HostPool::get().parallel_for(frames, work);
    return UWIP_OK;
}
