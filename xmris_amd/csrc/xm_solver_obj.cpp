// Objectives of the autophase search (reference processing/phasing.py:100-157) as vectorised C++:
// one pass over the arg-max spectrum per evaluation in 256-sample chunks (AVX-512 / AVX2 clones; the phase
// factor by a rotation recurrence re-anchored every 16 samples on the uniform frequency axis; a gather-free
// log), evaluations handed to a small pool of spinning host threads in work units of 2048 samples (an
// evaluation lasts ~6 us on one core, far below what a fork/join runtime costs).  Compiled with
// -O3 -ffast-math -mavx2 -mfma; the summation tree is fixed, so values do not depend on the team.  Host code only.
#include <immintrin.h>
#include <x86intrin.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// ln x for finite x > 0 (fdlibm's reduction x = 2^k (1 + f), sqrt(1/2) <= 1 + f < sqrt(2), and its degree-14
// polynomial): branch-free and table-free, so the loop around it vectorises without gathers.  Absolute
// error ~1e-16 on the reduced range; 0 maps to a finite value (the caller multiplies by x).
static inline double fast_log(double x) {
  int64_t ix;
  std::memcpy(&ix, &x, 8);
  const int64_t k = (ix - 0x3fe6a09e667f3bcdLL) >> 52;
  const int64_t im = ix - (k << 52);
  double m;
  std::memcpy(&m, &im, 8);
  const int64_t kb = k + 0x4338000000000000LL;  // double(k) = bits(k + 2^52 + 2^51) - (2^52 + 2^51)
  double kd;
  std::memcpy(&kd, &kb, 8);
  kd -= 6755399441055744.0;
  const double f = m - 1.0;
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
  const double t2 = z * (6.666666666666735130e-01 +
                         w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
  return kd * 6.93147180559945286227e-01 + (f - s * (f - (t2 + t1)));
}

// One 256-sample chunk of the ACME objective.  Built twice (function multi-versioning): AVX-512 for
// CPUs that have it (8-wide libmvec sin/cos), AVX2 otherwise; the dispatcher picks at load time.
// `uniform` says the coordinate is uniformly spaced (u[k] = u[0] + k du to rounding, the fftfreq axis the
// path produces): e^{i phi} is then evaluated exactly only at every 16th sample and rotated by the 16
// precomputed e^{i j p1 du} inside each block (one complex multiply instead of a sin and a cos per sample;
// the rounding error stays at a few ulp because every block restarts from an exact anchor).
__attribute__((target_clones("avx512f", "default"))) void acme_chunk_kernel(
    const double* __restrict__ re, const double* __restrict__ im, const double* __restrict__ u, int nn, int c0,
    double p0r, double p1r, double du, bool uniform, double* __restrict__ pc) {
  const int c1 = std::min(nn, c0 + 256);
  alignas(64) double d[272];
  const int m = std::min(nn, c1 + 1) - c0;  // one extra sample for the forward difference
  if (uniform) {
    alignas(64) double ang[32], cc[32], ss[32];
    for (int b = 0; b < 16; ++b) ang[b] = p0r + p1r * u[std::min(c0 + 16 * b, nn - 1)];  // block anchors
    for (int j = 0; j < 16; ++j) ang[16 + j] = p1r * du * (double)j;                      // in-block rotations
#pragma omp simd
    for (int i = 0; i < 32; ++i) cc[i] = std::cos(ang[i]);
#pragma omp simd
    for (int i = 0; i < 32; ++i) ss[i] = std::sin(ang[i]);
    const double* rc = cc + 16;
    const double* rs = ss + 16;
    if (c0 + 256 <= nn) {
      for (int b = 0; b < 16; ++b) {
        const double ca = cc[b], sa = ss[b];
#pragma omp simd
        for (int j = 0; j < 16; ++j) {
          const int k = c0 + 16 * b + j;
          d[16 * b + j] = re[k] * (ca * rc[j] - sa * rs[j]) - im[k] * (sa * rc[j] + ca * rs[j]);
        }
      }
      if (m > 256) {  // sample 256 = anchor 15 advanced by 16 steps (rotation 15, then rotation 1)
        const double c16 = rc[15] * rc[1] - rs[15] * rs[1], s16 = rs[15] * rc[1] + rc[15] * rs[1];
        d[256] = re[c0 + 256] * (cc[15] * c16 - ss[15] * s16) - im[c0 + 256] * (ss[15] * c16 + cc[15] * s16);
      }
    } else {
      for (int k = 0; k < m; ++k) {
        const int b = k >> 4, j = k & 15;
        d[k] = re[c0 + k] * (cc[b] * rc[j] - ss[b] * rs[j]) - im[c0 + k] * (ss[b] * rc[j] + cc[b] * rs[j]);
      }
    }
  } else {
    alignas(64) double cs[272], sn[272];
    // separate loops so that gcc uses the libmvec vector cos / sin (a fused sincos call stays scalar)
#pragma omp simd
    for (int k = 0; k < m; ++k) cs[k] = std::cos(p0r + p1r * u[c0 + k]);
#pragma omp simd
    for (int k = 0; k < m; ++k) sn[k] = std::sin(p0r + p1r * u[c0 + k]);
#pragma omp simd
    for (int k = 0; k < m; ++k) d[k] = re[c0 + k] * cs[k] - im[c0 + k] * sn[k];
  }
  double a_ds = 0, a_dl = 0, a_as = 0, a_as2 = 0, a_mx = -DBL_MAX;
  const int nk = c1 - c0, nd = std::min(nk, nn - 1 - c0);  // the very last sample has no forward difference
#pragma omp simd reduction(+ : a_as, a_as2) reduction(max : a_mx)
  for (int k = 0; k < nk; ++k) {
    const double v = d[k];
    const double as_ = v - std::fabs(v);
    a_as += as_;
    a_as2 += (0.5 * as_) * (0.5 * as_);
    a_mx = std::max(a_mx, v);
  }
#pragma omp simd reduction(+ : a_ds, a_dl)
  for (int k = 0; k < nd; ++k) {
    const double ds = std::fabs((d[k + 1] - d[k]) * 0.5);
    a_ds += ds;
    a_dl += ds * fast_log(ds);  // ds == 0 contributes 0 (fast_log(0) is finite), as the reference's zeros -> 1 rule
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
  double du = 0;         // spacing of u when `uniform`
  bool uniform = false;  // u[k] == u[0] + k du to a few ulp (checked in xm_solver_create)
  std::vector<double> re, im, u;  // u[k] = (c[k] - pivot) / (max c - min c)   (phasing.py:69)
  long nfev = 0;
  int threads = 1;  // team size for objective evaluations (1 = serial)
  int batch = 4;    // most trials xm_solver_de evaluates speculatively in one hand-off (measured optimum, 16 threads)

  // Re(slice[k] * e^{i phi_k}),  phi_k = rad(p0) + rad(p1) * u[k]        (phasing.py:62-73)
  inline double phased_real(int k, double p0r, double p1r) const {
    const double ph = x_range == 0 ? p0r : p0r + p1r * u[k];
    return re[k] * std::cos(ph) - im[k] * std::sin(ph);
  }

  // partial sums of one 256-sample chunk: {sum ds, sum ds*ln ds, sum a, sum (a/2)^2, max d}
  void acme_chunk(int c0, double p0r, double p1r, double* pc) const {
    acme_chunk_kernel(re.data(), im.data(), u.data(), n, c0, p0r, p1r, du, uniform, pc);
  }
  // A "part" = kPartChunks consecutive chunks (2048 samples), the unit of work handed to a team member; its
  // chunk sums are combined in chunk order.  The summation tree (chunks -> parts -> total) depends only on n.
  static constexpr int kPartChunks = 8;
  int nparts() const { return ((n + 255) / 256 + kPartChunks - 1) / kPartChunks; }
  void acme_part(int p, double p0r, double p1r, double* pp) const {
    const int nchunk = (n + 255) / 256, lo = p * kPartChunks, hi = std::min(nchunk, lo + kPartChunks);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = -DBL_MAX, pc[5];
    for (int c = lo; c < hi; ++c) {
      acme_chunk(c * 256, p0r, p1r, pc);
      a0 += pc[0];
      a1 += pc[1];
      a2 += pc[2];
      a3 += pc[3];
      a4 = std::max(a4, pc[4]);
    }
    pp[0] = a0;
    pp[1] = a1;
    pp[2] = a2;
    pp[3] = a3;
    pp[4] = a4;
  }

  double acme(double p0r, double p1r) const;  // phasing.py:100-122, one pass (defined after Pool)
  void acme_batch(const double* p01r, int count, double* out) const;
  // combine the per-part partial sums serially in part order: the value does not depend on the team
  // size or on batching (the optimiser's path must be repeatable)
  template <class F>
  double acme_combine(int nchunk, F&& chunk_sums) const {
    double s_ds = 0, s_dslog = 0, s_as = 0, s_as2 = 0, dmax = -DBL_MAX;
    for (int c = 0; c < nchunk; ++c) {
      const double* pc = chunk_sums(c);
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

  // `count` parameter vectors (nx values each, stride 2) -> out[count]; ACME evaluations share one
  // hand-off to the worker team
  void score_batch(const double* xs, int nx, int count, double* out) {
    const double kRad = M_PI / 180.0;
    if (method != 0 || x_range == 0 || count == 1) {
      for (int e = 0; e < count; ++e) out[e] = score(xs + 2 * e, nx);
      return;
    }
    nfev += count;
    std::vector<double> pr(2 * count);
    for (int e = 0; e < count; ++e) {
      pr[2 * e] = xs[2 * e] * kRad;
      pr[2 * e + 1] = (nx > 1 ? xs[2 * e + 1] : 0.0) * kRad;
    }
    acme_batch(pr.data(), count, out);
  }

  double score(const double* x, int nx) {
    ++nfev;
    const double kRad = M_PI / 180.0;  // np.radians
    const double p0r = x[0] * kRad, p1r = (nx > 1 ? x[1] : 0.0) * kRad;
    if (method == 0) return acme(p0r, x_range == 0 ? 0.0 : p1r);  // zero range: scalar phase (u is all zero)
    return method == 1 ? peak_minima(p0r, p1r) : positivity(p0r, p1r);
  }
};

// A few persistent worker threads that SPIN while a search is running (between xm_solver_pool_begin / _end,
// i.e. inside xm_solver_de) and sleep on a condition variable otherwise.  One job = a batch of evaluations:
// publish their (p0, p1), bump a generation counter, every member takes its share of the work units
// (evaluation x 2048-sample part), the caller combines the per-part sums serially in part order (so the
// value is independent of the team size and of the batching).
//
// Stragglers.  A batch is ~10 us of work per member and a search runs ~150 of them, so a member that is not there --
// still asleep when the search starts (a futex wake-up on a busy 256-CPU host was seen to take 3-8 ms, with no
// involuntary context switch and no cgroup throttling on record) or descheduled in mid-search -- used to stall the
// whole team for as long as it stayed away: isolated searches of 4-8 ms instead of 1.4, one in every few runs of 20
// datasets, each a 3-9 ms hole between two main passes.  The caller therefore waits for a missing member only for a
// grace period and then computes that member's share ITSELF into a backup buffer (the work units are deterministic:
// whoever computes them gets the same five sums), a few microseconds instead of milliseconds.  What this needs:
//   * jobs are immutable once published: a ring of descriptors (with their own copy of the parameters); a member
//     copies the descriptor of the generation it saw and discards the copy when the caller has lapped the ring
//     meanwhile (it may be torn);
//   * a late member writes only its own slot and acknowledges only the generation it computed, so the caller never
//     reads a slot that is being written;
//   * the solver outlives every member that may touch it: a member raises `busy` before and checks the pool's phase
//     after (Dekker), park() flips the phase first and then waits for the busy flags -- a member that wakes up after
//     the search is over never starts on its job.
struct Pool {
  static constexpr int kMaxWorkers = 31, kMaxLocal = 128, kRing = 32, kMaxJobEvals = 128;
  static constexpr uint64_t kGraceCycles = 80000, kSuspectGraceCycles = 6000;  // TSC ticks: ~30 us, ~2.5 us
  // One slot per team member: its acknowledgement AND its chunk partial sums share the same cache
  // lines, so the caller pays one coherence miss per worker (prefetched together), not one per flag
  // plus one per result line, and nobody does a contended read-modify-write.
  struct alignas(128) Slot {
    std::atomic<uint64_t> ack{0};
    std::atomic<int> busy{0};
    double sums[5 * kMaxLocal];
  };
  struct Job {
    const Solver* s = nullptr;
    int nchunk = 0, neval = 1, team = 1;
    uint64_t phase = 0;  // the pool phase (= the search) that published it: a member that wakes into the NEXT search
                         // on this pool before that search's first job still finds this descriptor in the ring
                         // (its `seen` lags `gen`), and its solver may be gone -- it must not run it (advisor, round 3)
    double params[2 * kMaxJobEvals];  // (p0, p1) in radians per evaluation
  };
  std::vector<std::thread> th;
  alignas(128) std::atomic<uint64_t> gen{0};
  alignas(128) std::atomic<uint64_t> phase{0};  // even: parked, odd: a search is running (one value per search)
  std::atomic<int> quit{0};
  std::atomic<int> spin_team{0};           // members (caller included) of the search that activated the pool: only
                                           // workers with id < spin_team wake up and spin -- a pool that once served a
                                           // 16-thread search must not spin 15 workers beside a 2-thread one (round 2
                                           // did: ~25 cores busy under a 16-CPU quota with four searches in flight)
  std::mutex mu;
  std::condition_variable cv;
  Slot slot[kMaxWorkers + 1];
  Job ring[kRing];
  // caller's side of the current job
  const Job* cur = nullptr;
  bool backed_up[kMaxWorkers + 1] = {};
  bool suspect[kMaxWorkers + 1] = {};  // missed a batch of this search: short grace until it is on time again
  std::vector<double> backup = std::vector<double>((size_t)(kMaxWorkers + 1) * 5 * kMaxLocal);
  std::atomic<long> backups{0};  // shares the caller computed for a missing member (diagnostics)

  // test hook (XM_SOLVER_TEST_STALL="<member>,<microseconds>"): that member sleeps before every 5th job it takes
  int stall_id = -1, stall_us = 0;
  bool no_backup = false;  // tuning switch XM_SOLVER_NO_BACKUP: wait for every member however long it takes (round 2)
  Pool() {
    if (const char* e = std::getenv("XM_SOLVER_TEST_STALL")) std::sscanf(e, "%d,%d", &stall_id, &stall_us);
    no_backup = std::getenv("XM_SOLVER_NO_BACKUP") != nullptr;
  }

  bool in_use = false;  // guarded by g_pools_mu
  // task j = evaluation j / nchunk, part j % nchunk (nchunk = parts per evaluation); member id takes tasks
  // id, id + team, ... -> sums[local]
  static void run_share(const Job& jb, int id, double* sums) {
    int local = 0;
    const int ntask = jb.nchunk * jb.neval;
    for (int j = id; j < ntask; j += jb.team, ++local) {
      const int e = j / jb.nchunk, c = j - e * jb.nchunk;
      jb.s->acme_part(c, jb.params[2 * e], jb.params[2 * e + 1], sums + 5 * local);
    }
  }
  void worker(int id) {
    uint64_t seen = 0;
    Job jb;
    for (;;) {
      uint64_t ph;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return quit.load() || ((phase.load() & 1) && id < spin_team.load()); });
        if (quit.load()) return;
        ph = phase.load();
      }
      while (phase.load(std::memory_order_acquire) == ph) {
        const uint64_t g = gen.load(std::memory_order_acquire);
        if (g == seen) {
          _mm_pause();
          continue;
        }
        seen = g;
        const Job& src = ring[g % kRing];
        if (id >= src.team) continue;  // (a torn read here only costs a pass of the loop)
        jb = src;
        std::atomic_thread_fence(std::memory_order_acquire);
        if (gen.load(std::memory_order_acquire) - g >= (uint64_t)(kRing - 1)) continue;  // lapped: the copy may be torn
        if (id >= jb.team || jb.phase != ph) continue;  // (a job of an earlier search on this pool: stale)
        slot[id].busy.store(1, std::memory_order_seq_cst);
        if (phase.load(std::memory_order_seq_cst) == ph) {  // the search (and its solver) is still there
          if (id == stall_id && g % 5 == 0) std::this_thread::sleep_for(std::chrono::microseconds(stall_us));
          run_share(jb, id, slot[id].sums);
          slot[id].ack.store(g, std::memory_order_release);
        }
        slot[id].busy.store(0, std::memory_order_release);
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
  void activate(int team_size) {
    {
      std::lock_guard<std::mutex> lk(mu);
      spin_team.store(team_size);
      for (int w = 0; w <= kMaxWorkers; ++w) suspect[w] = false;
      phase.fetch_add(1, std::memory_order_seq_cst);  // -> odd
    }
    cv.notify_all();
  }
  void park() {
    phase.fetch_add(1, std::memory_order_seq_cst);  // -> even: no member starts on a job from here on
    for (size_t w = 1; w <= th.size(); ++w)
      while (slot[w].busy.load(std::memory_order_seq_cst)) _mm_pause();  // a member in mid-share still reads the solver
  }
  bool active() const { return (phase.load(std::memory_order_acquire) & 1) != 0; }
  bool fits(int tasks, int t) const {
    return t - 1 <= (int)th.size() && t <= spin_team.load() && (tasks + t - 1) / t <= kMaxLocal;
  }
  // partial sums of task j after eval(): member j % team, its local task j / team
  const double* task_sums(int j) const {
    const int w = j % cur->team;
    return (backed_up[w] ? backup.data() + (size_t)w * 5 * kMaxLocal : slot[w].sums) + 5 * (j / cur->team);
  }
  void eval(const Solver* sv, const double* p01r, int count, int chunks, int t) {
    const uint64_t g = gen.load(std::memory_order_relaxed) + 1;
    Job& jb = ring[g % kRing];
    jb.s = sv;
    jb.nchunk = chunks;
    jb.neval = count;
    jb.team = t;
    jb.phase = phase.load(std::memory_order_relaxed);  // (odd: this search; only this thread changes it while it runs)
    std::memcpy(jb.params, p01r, sizeof(double) * 2 * (size_t)count);
    cur = &jb;
    gen.store(g, std::memory_order_release);
    run_share(jb, 0, slot[0].sums);
    backed_up[0] = false;
    for (int w = 1; w < t; ++w) __builtin_prefetch(&slot[w], 0, 3);
    const uint64_t t0 = __rdtsc();
    for (int w = 1; w < t; ++w) {
      backed_up[w] = false;
      const uint64_t grace = suspect[w] ? kSuspectGraceCycles : kGraceCycles;
      while (slot[w].ack.load(std::memory_order_acquire) != g) {
        if (!no_backup && __rdtsc() - t0 > grace) {  // not there: its share is a few microseconds of the caller's time
          run_share(jb, w, backup.data() + (size_t)w * 5 * kMaxLocal);
          backed_up[w] = suspect[w] = true;
          backups.fetch_add(1, std::memory_order_relaxed);
          break;
        }
        _mm_pause();
      }
      if (!backed_up[w]) suspect[w] = false;  // there (again)
    }
  }
};

// A few pools so that several searches (independent datasets, one Python thread each) can run at the same time, each
// with its own team.  A search owns its pool from xm_solver_pool_begin to xm_solver_pool_end ON ITS OWN THREAD
// (t_pool): evaluations issued by any other thread (a polish, score calls from Python) never touch a team.
constexpr int kPools = 6;  // up to four searches in flight + one that was hedged and still runs + its second start
Pool* g_pools[kPools] = {};  // created on first use, intentionally leaked: workers may outlive static destruction
std::mutex g_pools_mu;
thread_local Pool* t_pool = nullptr;

double Solver::acme(double p0r, double p1r) const {
  double out;
  const double p[2] = {p0r, p1r};
  acme_batch(p, 1, &out);
  return out;
}

void Solver::acme_batch(const double* p01r, int count, double* out) const {
  const int nn = n, nchunk = nparts();  // work units per evaluation
  Pool* const poolp = t_pool;  // only the thread that owns a running search has one
  const bool par = threads > 1 && nn >= 2048 && poolp && poolp->active() && poolp->fits(nchunk, threads);
  if (!par) {
    for (int e = 0; e < count; ++e) {
      double local[5];
      out[e] = acme_combine(nchunk, [&](int c) {
        acme_part(c, p01r[2 * e], p01r[2 * e + 1], local);
        return (const double*)local;
      });
    }
    return;
  }
  Pool& pool = *poolp;
  // as many evaluations per hand-off as the per-member result slots hold
  const int per = std::max(1, std::min(Pool::kMaxJobEvals, (Pool::kMaxLocal * threads) / nchunk));
  for (int e0 = 0; e0 < count; e0 += per) {
    const int g = std::min(per, count - e0);
    pool.eval(this, p01r + 2 * e0, g, nchunk, threads);
    for (int e = 0; e < g; ++e)
      out[e0 + e] = acme_combine(nchunk, [&](int c) { return pool.task_sums(e * nchunk + c); });
  }
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
  if (const char* e = std::getenv("XM_SOLVER_BATCH")) s->batch = std::max(1, std::atoi(e));
  for (int k = 0; k < n; ++k) {
    s->re[k] = slice_re_im[2 * k];
    s->im[k] = slice_re_im[2 * k + 1];
    s->u[k] = s->x_range == 0 ? 0.0 : (coords[k] - pivot) / s->x_range;
  }
  s->du = (s->u[n - 1] - s->u[0]) / (double)(n - 1);
  double dev = 0;
  for (int k = 0; k < n; ++k) dev = std::max(dev, std::fabs(s->u[k] - (s->u[0] + (double)k * s->du)));
  s->uniform = dev <= 4e-15 && !std::getenv("XM_SOLVER_NO_RECURRENCE");
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

// `count` parameter vectors, 2 doubles apart (p0[, p1] in degrees) -> out[count]
void xm_solver_score_batch(void* h, const double* xs, int nx, int count, double* out) {
  ((Solver*)h)->score_batch(xs, nx, count, out);
}

int xm_solver_get_batch(void* h) { return ((Solver*)h)->batch; }

// e^{i phi} over a coordinate axis, phi = rad(p0) + rad(p1) * (c - pivot) / (max c - min c)  (scalar phase when the
// range is zero; reference processing/phasing.py:56-73), computed in fp64 and rounded once to the storage
// precision: `out` = n interleaved (re, im) pairs of float32 (as_float != 0) or float64.  Host memory (the caller's
// pinned staging buffer).  The cos and sin loops are separate so that the vector math library is used.
int xm_phase_table(const double* coords, int n, double p0_deg, double p1_deg, double pivot, void* out, int as_float) {
  if (!coords || !out || n < 1) return -1;
  double cmin = coords[0], cmax = coords[0];
  for (int k = 1; k < n; ++k) {
    cmin = std::min(cmin, coords[k]);
    cmax = std::max(cmax, coords[k]);
  }
  const double kRad = M_PI / 180.0, range = cmax - cmin;
  const double p0r = p0_deg * kRad, p1r = p1_deg * kRad;
  std::vector<double> ang(n), cs(n), sn(n);
  if (range == 0) {
    std::fill(ang.begin(), ang.end(), p0r);
  } else {
    for (int k = 0; k < n; ++k) ang[k] = p0r + p1r * ((coords[k] - pivot) / range);
  }
#pragma omp simd
  for (int k = 0; k < n; ++k) cs[k] = std::cos(ang[k]);
#pragma omp simd
  for (int k = 0; k < n; ++k) sn[k] = std::sin(ang[k]);
  if (as_float) {
    float* o = (float*)out;
    for (int k = 0; k < n; ++k) {
      o[2 * k] = (float)cs[k];
      o[2 * k + 1] = (float)sn[k];
    }
  } else {
    double* o = (double*)out;
    for (int k = 0; k < n; ++k) {
      o[2 * k] = cs[k];
      o[2 * k + 1] = sn[k];
    }
  }
  return 0;
}

// A search takes a free pool for its duration; with every pool taken it evaluates serially.
void xm_solver_pool_begin(int threads) {
  Pool* p = nullptr;
  if (threads <= 1) {  // a serial search needs no team: any number of them may run side by side
    t_pool = nullptr;
    return;
  }
  {
    std::lock_guard<std::mutex> lk(g_pools_mu);
    for (int i = 0; i < kPools && !p; ++i) {
      if (!g_pools[i]) g_pools[i] = new Pool();
      if (!g_pools[i]->in_use) {
        p = g_pools[i];
        p->in_use = true;
      }
    }
  }
  t_pool = p;
  if (!p) return;
  p->ensure(threads - 1);
  p->activate(threads);
}

// shares of a batch that a search's own thread computed because the member they belong to was not there in time
// (asleep, descheduled), over all pools since the library was loaded: diagnostics
long xm_solver_pool_backups(void) {
  long n = 0;
  std::lock_guard<std::mutex> lk(g_pools_mu);
  for (int i = 0; i < kPools; ++i)
    if (g_pools[i]) n += g_pools[i]->backups.load(std::memory_order_relaxed);
  return n;
}

void xm_solver_pool_end(void) {
  Pool* p = t_pool;
  if (!p) return;
  t_pool = nullptr;
  p->park();
  std::lock_guard<std::mutex> lk(g_pools_mu);
  p->in_use = false;
}

}  // extern "C"
