// Host search service: whole autophase searches (reference processing/phasing.py:257-287) run by a few NATIVE threads
// of this library -- differential evolution (xm_solver_de: scipy's generations bit for bit) followed by the test
// scipy's L-BFGS-B polish starts with (xm_solver_fg + the projected gradient; xmris_amd/autophase_solver.py,
// polish="exact") -- with the result left in the same record a search kernel fills (xm_search_result, `seq` last).
//
// Why: the streaming executor used to run every search on a Python thread (create the objective, call the
// generations, the gradient test, wrap the result: ~60-100 us of interpreter per search), and those threads share
// the interpreter lock with the thread that queues the kernels -- on the small configurations (a dataset every
// ~0.25 ms) the lock, not the arithmetic, paced the device.  A submitted search touches no Python object.
// Host code only; IEEE semantics like xm_solver.cpp (it only calls the solver's entry points).
#include "../../include/xmris_hip.h"

#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct Job {
  // the service's OWN copies of the slice and the axis: a search that was started a second time (or submitted late by
  // a test hook) may still be running when its caller has moved on and freed the buffers it was submitted from --
  // only the result record must outlive it (the executor keeps abandoned records alive)
  std::vector<double> slice_copy, coords_copy;
  const double* slice;
  const double* coords;
  xm_search_result* out;
  unsigned long long seq;
  double tol;
  int n, method, target_idx, index_width, p0_only, maxiter, threads;
  unsigned seed;
};

struct Service {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Job> queue;
  std::vector<std::thread> workers;
  int idle = 0;
  int running = 0;      // searches being run right now
  int max_workers = 4;  // searches running at once (xm_hostsearch_set_workers); more submissions wait in the queue
  bool quit = false;

  void run_job(const Job& j) {
    const auto t0 = std::chrono::steady_clock::now();
    xm_search_result r;
    std::memset(&r, 0, sizeof(r));
    int k = j.target_idx;
    if (k < 0) {  // first arg-max of |slice| (phasing.py:229 on the winning row), numpy's hypot
      double best = -1.0;
      k = 0;
      for (int i = 0; i < j.n; ++i) {
        const double a = std::hypot(j.slice[2 * i], j.slice[2 * i + 1]);
        if (a > best) {
          best = a;
          k = i;
        }
      }
    }
    r.target_idx = k;
    void* h = xm_solver_create(j.slice, j.coords, j.n, j.coords[k], j.method, k, j.index_width);
    if (!h) {
      r.status = -1;
    } else {
      if (j.threads > 0) xm_solver_set_threads(h, j.threads);
      double x[2] = {0., 0.}, fun = 0.;
      int nfev = 0, nit = 0;
      r.status = xm_solver_de(h, j.p0_only, j.seed, j.tol, j.maxiter, x, &fun, &nfev, &nit);
      const auto t1 = std::chrono::steady_clock::now();
      const int nx = j.p0_only ? 1 : 2;
      const double lb[2] = {-180.0, -4000.0}, ub[2] = {180.0, 4000.0};
      double xc[2], f0 = 0., g[2] = {0., 0.};
      for (int i = 0; i < nx; ++i) xc[i] = x[i] < lb[i] ? lb[i] : (x[i] > ub[i] ? ub[i] : x[i]);
      double pgn = INFINITY;
      if (xm_solver_fg(h, xc, nx, lb, ub, &f0, g) == 0) {
        pgn = 0.;
        for (int i = 0; i < nx; ++i) {  // L-BFGS-B's projgr, both bounds set
          const double pg = g[i] < 0. ? std::fmax(xc[i] - ub[i], g[i]) : std::fmin(xc[i] - lb[i], g[i]);
          pgn = std::fmax(pgn, std::fabs(pg));
        }
      }
      xm_solver_destroy(h);
      r.x[0] = x[0];
      r.x[1] = j.p0_only ? 0. : x[1];
      r.fun = fun;
      r.pg_norm = pgn;
      r.nfev = nfev;
      r.nit = nit;
      r.needs_polish = pgn <= 0.5e-5 ? 0 : 1;
      r.t_us[0] = std::chrono::duration<double, std::micro>(t1 - t0).count();
    }
    r.t_us[5] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    r.seq = 0;
    std::memcpy((void*)j.out, &r, sizeof(r));
    __atomic_store_n(&j.out->seq, (uint64_t)j.seq, __ATOMIC_RELEASE);
  }

  void worker() {
    for (;;) {
      Job j;
      {
        std::unique_lock<std::mutex> lk(mu);
        ++idle;
        // (the cap holds for the workers that exist, too: a hedged second start raises it for one submission and the
        // thread it brought stays)
        cv.wait(lk, [&] { return quit || (!queue.empty() && running < max_workers); });
        --idle;
        if (quit) return;
        j = std::move(queue.front());
        queue.pop_front();
        ++running;
      }
      j.slice = j.slice_copy.data();
      j.coords = j.coords_copy.data();
      run_job(j);
      {
        std::lock_guard<std::mutex> lk(mu);
        --running;
      }
      cv.notify_one();  // (a job may have been waiting for this slot)
    }
  }
};

Service* g_service = nullptr;  // created on first use, intentionally leaked (its threads may outlive static destruction)
std::mutex g_service_mu;

}  // namespace

extern "C" {

int xm_hostsearch_submit(const void* slice, int n, const double* coords, int method, int target_idx, int index_width,
                         int p0_only, unsigned seed, double tol, int maxiter, int threads, uint64_t seq,
                         xm_search_result* out) {
  if (!slice || !coords || !out || n < 2 || method < 0 || method > 2 || target_idx >= n || maxiter < 1) return XM_ERR_INVALID_ARG;
  Service* s;
  {
    std::lock_guard<std::mutex> lk(g_service_mu);
    if (!g_service) g_service = new Service();
    s = g_service;
  }
  Job j;
  j.slice_copy.assign((const double*)slice, (const double*)slice + 2 * (size_t)n);
  j.coords_copy.assign(coords, coords + n);
  j.slice = nullptr;  // (set from the copies by the worker: the vectors move with the job)
  j.coords = nullptr;
  j.out = out;
  j.seq = seq;
  j.tol = tol;
  j.n = n;
  j.method = method;
  j.target_idx = target_idx;
  j.index_width = index_width < 1 ? 1 : index_width;
  j.p0_only = p0_only ? 1 : 0;
  j.maxiter = maxiter;
  j.threads = threads;
  j.seed = seed;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    s->queue.push_back(std::move(j));
    // one more worker whenever none is idle to take this job, up to the cap: later submissions wait their turn (the
    // executor submits further ahead than it wants searches to run side by side)
    if (s->idle < (int)s->queue.size() && (int)s->workers.size() < s->max_workers) s->workers.emplace_back([s] { s->worker(); });
  }
  s->cv.notify_one();
  return XM_OK;
}

// searches the service runs side by side (1 ... 8; default 4); returns the value set.
int xm_hostsearch_set_workers(int n) {
  Service* s;
  {
    std::lock_guard<std::mutex> lk(g_service_mu);
    if (!g_service) g_service = new Service();
    s = g_service;
  }
  int v;
  {
    std::lock_guard<std::mutex> lk(s->mu);
    v = s->max_workers = n < 1 ? 1 : (n > 8 ? 8 : n);
  }
  s->cv.notify_all();  // (a raised cap may let queued jobs start)
  return v;
}

}  // extern "C"
