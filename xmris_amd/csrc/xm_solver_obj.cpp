// Objectives of the autophase search (reference processing/phasing.py:100-157) as vectorised C++:
// one pass over the arg-max spectrum per evaluation, split over a small pool of spinning host threads
// (an evaluation lasts ~2 us, far below what a fork/join runtime costs), libmvec sin/cos/log
// (compiled with -O3 -ffast-math -mavx2 -mfma).  Host code only.
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// One 256-sample chunk of the ACME objective.  Built twice (function multi-versioning): AVX-512 for
// CPUs that have it (8-wide libmvec sin/cos/log), AVX2 otherwise; the dispatcher picks at load time.
__attribute__((target_clones("avx512f", "default"))) void acme_chunk_kernel(
    const double* __restrict__ re, const double* __restrict__ im, const double* __restrict__ u, int nn, int c0,
    double p0r, double p1r, double* __restrict__ pc) {
  const int c1 = std::min(nn, c0 + 256);
  double d[257], cs[257], sn[257];
  const int m = std::min(nn, c1 + 1) - c0;  // one extra sample for the forward difference
  // separate loops so that gcc uses the libmvec vector cos / sin (a fused sincos call stays scalar)
#pragma omp simd
  for (int k = 0; k < m; ++k) cs[k] = std::cos(p0r + p1r * u[c0 + k]);
#pragma omp simd
  for (int k = 0; k < m; ++k) sn[k] = std::sin(p0r + p1r * u[c0 + k]);
#pragma omp simd
  for (int k = 0; k < m; ++k) d[k] = re[c0 + k] * cs[k] - im[c0 + k] * sn[k];
  double a_ds = 0, a_dl = 0, a_as = 0, a_as2 = 0, a_mx = -DBL_MAX;
#pragma omp simd reduction(+ : a_ds, a_dl, a_as, a_as2) reduction(max : a_mx)
  for (int k = 0; k < c1 - c0; ++k) {
    const double v = d[k];
    const double as_ = v - std::fabs(v);
    a_as += as_;
    a_as2 += (0.5 * as_) * (0.5 * as_);
    a_mx = std::max(a_mx, v);
    if (c0 + k + 1 < nn) {
      const double ds = std::fabs((d[k + 1] - v) * 0.5);
      a_ds += ds;
      a_dl += ds > 0 ? ds * std::log(ds) : 0.0;
    }
  }
  pc[0] = a_ds;
  pc[1] = a_dl;
  pc[2] = a_as;
  pc[3] = a_as2;
  pc[4] = a_mx;
}

struct Solver {
  int n = 0, method = 0, target_idx = 0, index_width = 1;
  double pivot = 0, x_range = 0;
  std::vector<double> re, im, u;  // u[k] = (c[k] - pivot) / (max c - min c)   (phasing.py:69)
  long nfev = 0;
  int threads = 1;  // OpenMP team for one objective evaluation (1 = serial)

  // Re(slice[k] * e^{i phi_k}),  phi_k = rad(p0) + rad(p1) * u[k]        (phasing.py:62-73)
  inline double phased_real(int k, double p0r, double p1r) const {
    const double ph = x_range == 0 ? p0r : p0r + p1r * u[k];
    return re[k] * std::cos(ph) - im[k] * std::sin(ph);
  }

  // partial sums of one 256-sample chunk: {sum ds, sum ds*ln ds, sum a, sum (a/2)^2, max d}
  void acme_chunk(int c0, double p0r, double p1r, double* pc) const {
    acme_chunk_kernel(re.data(), im.data(), u.data(), n, c0, p0r, p1r, pc);
  }

  double acme(double p0r, double p1r) const;  // phasing.py:100-122, one pass (defined after Pool)

  double peak_minima(double p0r, double p1r) const {  // phasing.py:125-139
    const int start = std::max(0, target_idx - index_width), end = std::min(n, target_idx + index_width);
    const double dt = phased_real(target_idx, p0r, p1r);
    double mina = dt, minb = dt;
    if (start < target_idx) {
      mina = DBL_MAX;
      for (int k = start; k < target_idx; ++k) mina = std::min(mina, phased_real(k, p0r, p1r));
    }
    if (end > target_idx) {
      minb = DBL_MAX;
      for (int k = target_idx; k < end; ++k) minb = std::min(minb, phased_real(k, p0r, p1r));
    }
    return std::fabs(mina - minb);
  }

  double positivity(double p0r, double p1r) const {  // phasing.py:142-157
    const int start = std::max(0, target_idx - index_width), end = std::min(n, target_idx + index_width);
    double pos = 0, neg = 0;
    for (int k = start; k < end; ++k) {
      const double v = phased_real(k, p0r, p1r);
      if (v > 0) pos += v;
      if (v < 0) neg += std::fabs(v);
    }
    return neg * 5.0 - pos;
  }

  double score(const double* x, int nx) {
    ++nfev;
    const double kRad = M_PI / 180.0;  // np.radians
    const double p0r = x[0] * kRad, p1r = (nx > 1 ? x[1] : 0.0) * kRad;
    if (method == 0) {
      if (x_range == 0) {  // scalar phase: same formula with u == 0
        Solver tmp = *this;
        std::fill(tmp.u.begin(), tmp.u.end(), 0.0);
        return tmp.acme(p0r, 0.0);
      }
      return acme(p0r, p1r);
    }
    return method == 1 ? peak_minima(p0r, p1r) : positivity(p0r, p1r);
  }
};



// A few persistent worker threads that SPIN while a search is running (activated by xm_solver_de via
// Pool::Scope) and sleep on a condition variable otherwise.  One evaluation = publish (p0, p1), bump a
// generation counter, every thread does its chunks, the caller combines the per-chunk partial sums
// serially in chunk order (so the value is independent of the thread count).
struct Pool {
  static constexpr int kMaxWorkers = 31, kMaxLocal = 32;
  // One slot per team member: its acknowledgement AND its chunk partial sums share the same cache
  // lines, so the caller pays one coherence miss per worker (prefetched together), not one per flag
  // plus one per result line, and nobody does a contended read-modify-write.
  struct alignas(128) Slot {
    std::atomic<uint64_t> ack{0};
    double sums[5 * kMaxLocal];
  };
  std::vector<std::thread> th;
  alignas(128) std::atomic<uint64_t> gen{0};
  alignas(128) std::atomic<int> state{0};  // 0 parked, 1 spinning, 2 exit
  std::mutex mu;
  std::condition_variable cv;
  Slot slot[kMaxWorkers + 1];
  // current job
  const Solver* s = nullptr;
  double p0r = 0, p1r = 0;
  int nchunk = 0, team = 1;

  static Pool& get() {
    static Pool* p = new Pool();  // intentionally leaked: workers may outlive static destruction
    return *p;
  }
  void run_share(int id) {  // chunks id, id + team, ... -> slot[id].sums[local]
    int local = 0;
    for (int c = id; c < nchunk; c += team, ++local) s->acme_chunk(c * 256, p0r, p1r, slot[id].sums + 5 * local);
  }
  void worker(int id) {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return state.load() != 0; });
        if (state.load() == 2) return;
      }
      while (state.load(std::memory_order_acquire) == 1) {
        const uint64_t g = gen.load(std::memory_order_acquire);
        if (g != seen) {
          seen = g;
          if (id < team) {
            run_share(id);
            slot[id].ack.store(g, std::memory_order_release);
          }
        } else {
          _mm_pause();
        }
      }
    }
  }
  void ensure(int workers) {
    workers = std::min(workers, kMaxWorkers);
    while ((int)th.size() < workers) {
      const int id = (int)th.size() + 1;  // id 0 is the calling thread
      th.emplace_back([this, id] { worker(id); });
    }
  }
  // (No CPU pinning: on the shared MI355X hosts the scheduler finds idle cores better than a static
  // same-L3 placement did -- measured 2.6 us vs 4.7 us per evaluation with 16 threads.)
  void activate() {
    {
      std::lock_guard<std::mutex> lk(mu);
      state.store(1);
    }
    cv.notify_all();
  }
  void park() { state.store(0, std::memory_order_release); }
  bool active() const { return state.load(std::memory_order_acquire) == 1; }
  bool fits(int chunks, int t) const { return t - 1 <= (int)th.size() && (chunks + t - 1) / t <= kMaxLocal; }
  // partial sums of chunk c after eval(): slot[c % team].sums[5 * (c / team)]
  const double* chunk_sums(int c) const { return slot[c % team].sums + 5 * (c / team); }
  void eval(const Solver* sv, double a, double b, int chunks, int t) {
    s = sv;
    p0r = a;
    p1r = b;
    nchunk = chunks;
    team = t;
    const uint64_t g = gen.fetch_add(1, std::memory_order_release) + 1;
    run_share(0);
    for (int w = 1; w < team; ++w) __builtin_prefetch(&slot[w], 0, 3);
    for (int w = 1; w < team; ++w)
      while (slot[w].ack.load(std::memory_order_acquire) != g) _mm_pause();
  }
};

double Solver::acme(double p0r, double p1r) const {
  const int nn = n, nchunk = (nn + 255) / 256;
  Pool& pool = Pool::get();
  const bool par = threads > 1 && nn >= 2048 && pool.active() && pool.fits(nchunk, threads);
  double local[5];
  if (par) pool.eval(this, p0r, p1r, nchunk, threads);
  // per-chunk partial sums are combined serially in chunk order: the value does not depend on the
  // team size (the optimiser's path must be repeatable)
  double s_ds = 0, s_dslog = 0, s_as = 0, s_as2 = 0, dmax = -DBL_MAX;
  for (int c = 0; c < nchunk; ++c) {
    const double* pc;
    if (par) {
      pc = pool.chunk_sums(c);
    } else {
      acme_chunk(c * 256, p0r, p1r, local);
      pc = local;
    }
    s_ds += pc[0];
    s_dslog += pc[1];
    s_as += pc[2];
    s_as2 += pc[3];
    dmax = std::max(dmax, pc[4]);
  }
  // H = -sum p ln p with p = ds / S, zeros skipped  ==  ln S - (sum ds ln ds) / S
  const double h = std::log(s_ds) - s_dslog / s_ds;
  const double pfun = s_as < 0 ? s_as2 : 0.0;
  return (h + 1000.0 * pfun) / (double)n / dmax;
}

}  // namespace

extern "C" {

void* xm_solver_create(const double* slice_re_im, const double* coords, int n, double pivot, int method,
                       int target_idx, int index_width) {
  if (!slice_re_im || !coords || n < 2 || method < 0 || method > 2 || target_idx < 0 || target_idx >= n)
    return nullptr;
  Solver* s = new Solver();
  s->n = n;
  s->method = method;
  s->target_idx = target_idx;
  s->index_width = index_width < 1 ? 1 : index_width;
  s->pivot = pivot;
  s->re.resize(n);
  s->im.resize(n);
  s->u.resize(n);
  double cmin = coords[0], cmax = coords[0];
  for (int k = 0; k < n; ++k) {
    cmin = std::min(cmin, coords[k]);
    cmax = std::max(cmax, coords[k]);
  }
  s->x_range = cmax - cmin;
  const int hw = (int)std::thread::hardware_concurrency();
  s->threads = std::max(1, std::min(16, hw / 2));
  for (int k = 0; k < n; ++k) {
    s->re[k] = slice_re_im[2 * k];
    s->im[k] = slice_re_im[2 * k + 1];
    s->u[k] = s->x_range == 0 ? 0.0 : (coords[k] - pivot) / s->x_range;
  }
  return s;
}

void xm_solver_destroy(void* h) { delete (Solver*)h; }

double xm_solver_score(void* h, const double* x, int nx) { return ((Solver*)h)->score(x, nx); }

long xm_solver_nfev(void* h) { return ((Solver*)h)->nfev; }

// Team size of one objective evaluation: threads <= 0 picks min(16, hardware threads / 2), 1 = serial.
// The team only engages between xm_solver_pool_begin() and xm_solver_pool_end() (workers spin then);
// outside that window every evaluation is serial, which is what isolated calls from Python want.
int xm_solver_set_threads(void* h, int threads) {
  if (threads <= 0) {
    const int hw = (int)std::thread::hardware_concurrency();
    threads = std::max(1, std::min(16, hw / 2));
  }
  ((Solver*)h)->threads = std::min(threads, Pool::kMaxWorkers + 1);
  return threads;
}

int xm_solver_get_threads(void* h) { return ((Solver*)h)->threads; }

// One search at a time owns the pool (two Python threads may call xm_solver_de concurrently).
static std::mutex g_search_mu;

void xm_solver_pool_begin(int threads) {
  g_search_mu.lock();
  Pool& p = Pool::get();
  p.ensure(threads - 1);
  p.activate();
}

void xm_solver_pool_end(void) {
  Pool::get().park();
  g_search_mu.unlock();
}

}  // extern "C"
