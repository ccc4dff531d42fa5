// Objectives of the autophase search (reference processing/phasing.py:100-157) as vectorised C++:
// one pass over the arg-max spectrum per evaluation, OpenMP across host cores, libmvec sin/cos/log
// (compiled with -O3 -ffast-math -mavx2 -mfma -fopenmp).  Host code only.
#include <omp.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <vector>

namespace {

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

  double acme(double p0r, double p1r) const {  // phasing.py:100-122, one pass
    // H = -sum p ln p with p = ds / S, zeros skipped  ==  ln S - (sum ds ln ds) / S
    // per-chunk partial sums, combined serially in chunk order afterwards: the value does not depend
    // on the number of threads or on OpenMP's reduction order (the optimiser's path must be repeatable)
    const int nn = n, nchunk = (nn + 255) / 256;
    std::vector<double> part(5 * (size_t)nchunk);
#pragma omp parallel for schedule(static) num_threads(threads) if (nn >= 2048 && threads > 1)
    for (int c0 = 0; c0 < nn; c0 += 256) {
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
      double* pc = &part[5 * (size_t)(c0 / 256)];
      pc[0] = a_ds;
      pc[1] = a_dl;
      pc[2] = a_as;
      pc[3] = a_as2;
      pc[4] = a_mx;
    }
    double s_ds = 0, s_dslog = 0, s_as = 0, s_as2 = 0, dmax = -DBL_MAX;
    for (int c = 0; c < nchunk; ++c) {
      s_ds += part[5 * c];
      s_dslog += part[5 * c + 1];
      s_as += part[5 * c + 2];
      s_as2 += part[5 * c + 3];
      dmax = std::max(dmax, part[5 * c + 4]);
    }
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

// threads <= 0: pick min(16, available cores).  Back-to-back evaluations (the DE generations) profit
// from a team; isolated calls from Python (the L-BFGS-B polish) are faster serial because a sleeping
// OpenMP team takes longer to wake than one evaluation lasts.
int xm_solver_set_threads(void* h, int threads) {
  if (threads <= 0) threads = std::min(16, omp_get_max_threads());
  ((Solver*)h)->threads = threads < 1 ? 1 : threads;
  return ((Solver*)h)->threads;
}

}  // extern "C"
