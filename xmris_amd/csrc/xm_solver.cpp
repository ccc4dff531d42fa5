// Host side of autophase: the O(1)-per-dataset (p0, p1) search on the ONE arg-max spectrum
// (reference processing/phasing.py:100-157 objectives, :276-284 differential evolution).
//
// The reference calls scipy.optimize.differential_evolution(best1bin, tol=0.01, seed=42) with Python
// objectives (~90 ms for an 8192-point slice).  This file restates
//   * the three objectives as vectorised C++ (xm_solver_obj.cpp: one pass over the slice, shared by a small
//     team of spinning host threads), and
//   * scipy 1.15.3's DifferentialEvolutionSolver for exactly the configuration the reference uses
//     (latin-hypercube init, best1bin, dithered mutation U[0.5,1), recombination 0.7, immediate
//     updating, std/mean convergence) driven by numpy's legacy RandomState(seed) MT19937 stream, so the
//     search visits the same trial vectors as scipy does given the same objective values.
// The final L-BFGS-B polish stays in scipy (xmris_amd/autophase_solver.py) with this objective.
// This file holds the optimiser bookkeeping and must keep IEEE semantics (no fast-math, no FMA
// contraction: -O2 -ffp-contract=off) so that trial vectors equal scipy's bit for bit; the objectives
// live in xm_solver_obj.cpp (fast-math).  Host code only.

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

// ---- numpy legacy RandomState (MT19937) -------------------------------------------------------
struct MT19937 {
  uint32_t mt[624];
  int pos;
  void seed(uint32_t s) {  // numpy/random/src/mt19937/mt19937.c  mt19937_seed
    mt[0] = s;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    pos = 624;
  }
  void gen() {
    const uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX = 0x9908b0dfu;
    int i;
    uint32_t y;
    for (i = 0; i < 624 - 397; ++i) {
      y = (mt[i] & UPPER) | (mt[i + 1] & LOWER);
      mt[i] = mt[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX);
    }
    for (; i < 623; ++i) {
      y = (mt[i] & UPPER) | (mt[i + 1] & LOWER);
      mt[i] = mt[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX);
    }
    y = (mt[623] & UPPER) | (mt[0] & LOWER);
    mt[623] = mt[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & MATRIX);
    pos = 0;
  }
  uint32_t next32() {
    if (pos == 624) gen();
    uint32_t y = mt[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  double next_double() {  // random_double: 53 bits from two draws
    const int32_t a = next32() >> 5, b = next32() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  uint32_t interval(uint32_t max) {  // legacy random_interval: masked rejection
    if (max == 0) return 0;
    uint32_t mask = max;
    mask |= mask >> 1;
    mask |= mask >> 2;
    mask |= mask >> 4;
    mask |= mask >> 8;
    mask |= mask >> 16;
    uint32_t v;
    while ((v = (next32() & mask)) > max) {
    }
    return v;
  }
  void shuffle(int* a, int n) {  // RandomState._shuffle_raw
    for (int i = n - 1; i >= 1; --i) {
      const int j = (int)interval((uint32_t)i);
      std::swap(a[i], a[j]);
    }
  }
};

// numpy's pairwise summation for n <= 128 (the population has 15 or 30 members)
double np_sum_small(const double* a, int n) {
  if (n < 8) {
    double r = 0.;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  }
  double r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i;
  for (i = 8; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] += a[i + j];
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += a[i];
  return res;
}

}  // namespace

extern "C" {

double xm_solver_score(void* h, const double* x, int nx);
void xm_solver_score_batch(void* h, const double* xs, int nx, int count, double* out);
int xm_solver_get_threads(void* h);
int xm_solver_get_batch(void* h);
void xm_solver_pool_begin(int threads);
void xm_solver_pool_end(void);

// scipy 1.15.3 DifferentialEvolutionSolver.solve() without the polish, for
// strategy="best1bin", popsize=15, mutation=(0.5, 1), recombination=0.7, init="latinhypercube",
// updating="immediate", atol=0, maxiter=1000; bounds p0 in [-180, 180], p1 in [-4000, 4000].
// Returns 0 when converged, 1 when maxiter was reached.
int xm_solver_de(void* h, int p0_only, unsigned seed, double tol, int maxiter, double* x_out, double* fun_out,
                 int* nfev_out, int* nit_out) {
  const int N = p0_only ? 1 : 2;
  // evaluations come back to back from here on: let the worker pool spin for the duration
  struct PoolScope {
    explicit PoolScope(int t) { xm_solver_pool_begin(t); }
    ~PoolScope() { xm_solver_pool_end(); }
  } scope(xm_solver_get_threads(h));
  const double lo[2] = {-180.0, -4000.0}, hi[2] = {180.0, 4000.0};
  double arg1[2], arg2[2];
  for (int j = 0; j < N; ++j) {
    arg1[j] = 0.5 * (lo[j] + hi[j]);
    arg2[j] = std::fabs(lo[j] - hi[j]);
  }
  const int M = std::max(5, 15 * N);
  MT19937 rng;
  rng.seed(seed);
  auto scale = [&](const double* t, double* p) {
    for (int j = 0; j < N; ++j) p[j] = arg1[j] + (t[j] - 0.5) * arg2[j];
  };

  // init_population_lhs
  std::vector<double> samples(M * N), pop(M * N), en(M);
  const double seg = 1.0 / M;
  for (int i = 0; i < M; ++i)
    for (int j = 0; j < N; ++j) {
      // np.linspace(0., 1., M, endpoint=False)[i] = i * (1/M)   (step = 1.0/M, y = arange * step)
      samples[i * N + j] = seg * rng.next_double() + (double)i * (1.0 / M);
    }
  std::vector<int> order(M);
  for (int j = 0; j < N; ++j) {
    for (int i = 0; i < M; ++i) order[i] = i;
    rng.shuffle(order.data(), M);  // rng.permutation(range(M))
    for (int i = 0; i < M; ++i) pop[i * N + j] = samples[order[i] * N + j];
  }
  std::vector<int> ridx(M);
  for (int i = 0; i < M; ++i) ridx[i] = i;
  int nfev = 0;
  double par[2];
  {  // initial energies: M independent evaluations, one batch
    std::vector<double> ps(2 * M, 0.0);
    for (int i = 0; i < M; ++i) scale(&pop[i * N], &ps[2 * i]);
    xm_solver_score_batch(h, ps.data(), N, M, en.data());
    nfev += M;
  }
  auto promote = [&]() {  // _promote_lowest_energy: first arg-min to slot 0; returns the slot it came from
    int l = 0;
    for (int i = 1; i < M; ++i)
      if (en[i] < en[l]) l = i;
    std::swap(en[0], en[l]);
    for (int j = 0; j < N; ++j) std::swap(pop[j], pop[l * N + j]);
    return l;
  };
  promote();

  // scipy walks the candidates one by one ("immediate" updating): trial c is built from the population as
  // it stands after candidates < c were decided.  One evaluation is far shorter than a thread hand-off, so
  // the next `batch` trials are built SPECULATIVELY from the current population, evaluated together
  // (xm_solver_score_batch: one hand-off for the lot), and committed in order.  A trial reads only
  // pop[c], pop[0] (the best), pop[r0], pop[r1] and the random stream, so it is exactly the trial scipy would
  // build unless one of those four members changed earlier in the same batch; at the first such trial the
  // rest of the batch is thrown away, the generator and the index array are rewound to their state before
  // that trial, and speculation restarts there.  The sequence of accepted trials, energies and random
  // draws is therefore identical to the sequential algorithm's.
  struct Spec {
    double trial[2];
    int r0, r1;
    MT19937 rng_before;
    int ridx_before[30];
  };
  // (a serial search gains nothing from evaluating ahead: every discarded trial is wasted time; two threads look one
  // trial ahead.  The committed sequence is the same for every batch size.)
  const int team = xm_solver_get_threads(h);
  const int batch_max = std::max(1, std::min(M, team <= 1 ? 1 : (team == 2 ? 2 : xm_solver_get_batch(h))));
  std::vector<Spec> spec(batch_max);
  std::vector<double> ps(2 * batch_max, 0.0), es(batch_max);
  std::vector<char> mod(M);
  int batch = batch_max;

  int nit = 0, status = 1;
  std::vector<double> dev(M);
  for (nit = 1; nit <= maxiter; ++nit) {
    const double scl = 0.5 + (1.0 - 0.5) * rng.next_double();  // dither: rng.uniform(0.5, 1)
    for (int c = 0; c < M;) {
      const int B = std::min(batch, M - c);
      for (int b = 0; b < B; ++b) {
        Spec& sp = spec[b];
        const int cand = c + b;
        sp.rng_before = rng;
        std::memcpy(sp.ridx_before, ridx.data(), sizeof(int) * M);
        // _mutate
        int fill_point = 0;
        if (N > 1) fill_point = (int)rng.interval((uint32_t)(N - 1));  // rng_integers(rng, N) -> randint(0, N)
        rng.shuffle(ridx.data(), M);  // _select_samples(candidate, 5)
        int smp[5], ns = 0;
        for (int i = 0; i < 6 && ns < 5; ++i)
          if (ridx[i] != cand) smp[ns++] = ridx[i];
        sp.r0 = smp[0];
        sp.r1 = smp[1];
        double bprime[2];
        for (int j = 0; j < N; ++j) {
          sp.trial[j] = pop[cand * N + j];
          bprime[j] = pop[j] + scl * (pop[sp.r0 * N + j] - pop[sp.r1 * N + j]);  // _best1
        }
        bool cross[2];
        for (int j = 0; j < N; ++j) cross[j] = rng.next_double() < 0.7;
        cross[fill_point] = true;
        for (int j = 0; j < N; ++j)
          if (cross[j]) sp.trial[j] = bprime[j];
        // _ensure_constraint
        for (int j = 0; j < N; ++j)
          if (sp.trial[j] > 1 || sp.trial[j] < 0) sp.trial[j] = rng.next_double();
        scale(sp.trial, &ps[2 * b]);
      }
      xm_solver_score_batch(h, ps.data(), N, B, es.data());
      std::fill(mod.begin(), mod.end(), 0);
      int b = 0;
      for (; b < B; ++b) {
        const Spec& sp = spec[b];
        const int cand = c + b;
        if (b > 0 && (mod[cand] | mod[0] | mod[sp.r0] | mod[sp.r1])) break;  // stale: rebuild from here
        ++nfev;
        if (es[b] <= en[cand]) {
          for (int j = 0; j < N; ++j) pop[cand * N + j] = sp.trial[j];
          en[cand] = es[b];
          mod[cand] = 1;
          if (es[b] <= en[0]) {
            mod[promote()] = 1;
            mod[0] = 1;
          }
        }
      }
      if (b < B) {  // rewind the random stream to just before the first stale trial
        rng = spec[b].rng_before;
        std::memcpy(ridx.data(), spec[b].ridx_before, sizeof(int) * M);
        batch = std::max(std::min(4, batch_max), std::min(batch, 2 * b));
      } else {
        batch = std::min(batch_max, batch * 2);
      }
      c += b;
    }
    // converged(): std(energies) <= atol + tol * |mean(energies)|
    bool any_inf = false;
    for (int i = 0; i < M; ++i) any_inf |= !(std::fabs(en[i]) <= 1.79769313486231570815e308);
    if (!any_inf) {
      const double mean = np_sum_small(en.data(), M) / M;
      for (int i = 0; i < M; ++i) dev[i] = (en[i] - mean) * (en[i] - mean);
      const double sd = std::sqrt(np_sum_small(dev.data(), M) / M);
      if (sd <= tol * std::fabs(mean)) {
        status = 0;
        break;
      }
    }
  }
  if (nit > maxiter) nit = maxiter;
  scale(&pop[0], par);
  for (int j = 0; j < N; ++j) x_out[j] = par[j];
  if (N == 1) x_out[1] = 0.0;
  *fun_out = en[0];
  *nfev_out = nfev;
  *nit_out = nit;
  return status;
}

// f and its forward-difference gradient at x (n = 1 or 2 parameters), as scipy's L-BFGS-B front end requests them:
// approx_derivative(method="2-point", abs_step=1e-8, bounds=(lb, ub)) -- absolute step, relative fallback where it
// vanishes in x's precision, flipped where x + h leaves the bounds, one-sided towards the wider side where neither
// direction fits -- with the step taken as the exactly representable (x + h) - x.  The n + 1 points go to the objective
// in one batch.  This translation unit keeps IEEE semantics (no fast-math, no contraction): the values equal the numpy
// statement of the same arithmetic (`autophase_solver.polish_lbfgsb`) bit for bit.
int xm_solver_fg(void* h, const double* x, int n, const double* lb, const double* ub, double* f_out, double* g_out) {
  if (!h || !x || !lb || !ub || !f_out || !g_out || n < 1 || n > 2) return -1;
  double pts[3][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}}, dx[2] = {0.0, 0.0}, vals[3];
  for (int r = 0; r <= n; ++r)
    for (int i = 0; i < n; ++i) pts[r][i] = x[i];
  for (int i = 0; i < n; ++i) {
    const double xc = x[i];
    double step = 1e-8;
    volatile double moved = xc + step;  // (x + h) - x in double, never folded
    if (moved - xc == 0.0) step = 1.4901161193847656e-08 * (xc >= 0.0 ? 1.0 : -1.0) * std::max(1.0, std::fabs(xc));
    const double lower = xc - lb[i], upper = ub[i] - xc;
    const double xt = xc + step;
    const bool violated = xt < lb[i] || xt > ub[i];
    const bool fitting = std::fabs(step) <= std::max(lower, upper);
    if (violated && fitting) step = -step;
    if (!fitting) step = upper >= lower ? upper : -lower;
    volatile double p = xc + step;
    pts[1 + i][i] = p;
    dx[i] = p - xc;
  }
  xm_solver_score_batch(h, &pts[0][0], n, n + 1, vals);
  *f_out = vals[0];
  for (int i = 0; i < n; ++i) g_out[i] = (vals[1 + i] - vals[0]) / dx[i];
  return 0;
}

}  // extern "C"
