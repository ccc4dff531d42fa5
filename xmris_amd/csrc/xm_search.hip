// Device-resident autophase search (reference processing/phasing.py:100-122 ACME objective, :276-284
// scipy.optimize.differential_evolution(best1bin, tol=0.01, seed=42) and the first step of its L-BFGS-B polish).
//
// The (p0, p1) search is O(1) work per dataset, but it sits between a dataset's guess stage and its main pass, and on
// the host it costs 3 core-milliseconds per dataset (8192 bins): a node that streams small datasets -- or eight ranks
// of it -- runs out of cores long before the GPUs run out of bandwidth, and a shared host's scheduler decides the
// tail.  Here ONE workgroup per search runs scipy's algorithm on a CU of its own beside the streaming kernels:
//
//   wave 0        the optimiser: numpy's legacy RandomState (MT19937: state and tempered outputs in LDS, regenerated
//                 two generations ahead in parallel, consumed through a 64-word register window by v_readlane; the
//                 population-index shuffle takes its accepted draws from per-step ballots over that window), scipy 1.15.3's
//                 DifferentialEvolutionSolver for the reference's configuration (latin hypercube, best1bin, dither
//                 U[0.5, 1), CR 0.7, immediate updating, std/mean convergence) with the population held one member
//                 per LANE -- the same statements in the same order as xm_solver.cpp, IEEE arithmetic without
//                 contraction, so the trial vectors are scipy's bit for bit given equal comparisons of the energies;
//   waves 1..7    the objective: every thread keeps P + 1 consecutive bins of the arg-max spectrum (complex128) in
//                 registers for the whole search; e^{i phi_k} = A[t / 32] B[t % 32] C[j] from three small tables (one
//                 sincos per table entry and evaluation, 32 + 14 + P + 1 entries, instead of one per bin), one pass,
//                 five sums (the same five as xm_solver_obj.cpp), wave reductions through DPP, 7 partial sums to wave 0.
//   While the workers evaluate trial t, wave 0 already draws the random part of trial t + 1 (fill point, the
//   population-index shuffle, the crossover mask -- none of it depends on the population), so the serial part between
//   two evaluations is: combine, accept / reject, build the next trial, its phase tables.
//
// The search ends with the test scipy's polish starts with (f and its forward-difference gradient at the best member,
// approx_derivative's steps; projected gradient against pgtol): passed -- the usual case -- means scipy returns that
// member, and so does this kernel; failed is reported (`needs_polish`) and the caller polishes on the reference's
// own route (xmris_amd/autophase_solver.py, polish="exact").  The result record is written to device-accessible
// memory (pinned host memory), its sequence word last.  gfx950 only.
#include "xm_host.h"

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstring>
#include <map>
#include <type_traits>
#include <mutex>
#include <vector>

namespace {

constexpr int kThreads = 512;                 // 8 waves, up to 256 VGPRs each: the fp64 objective (sincos, log, divisions) needs room;
                                              // nine waves (512 workers) would cap them at 168 and spill hundreds
constexpr int kWaves = kThreads / XM_WAVE;
constexpr int kWorkers = kThreads - XM_WAVE;  // waves 1..7
constexpr int kTabA = kWorkers / 32;          // 14
constexpr int kMaxP = 37;

struct SearchArgs {
  const double* slice;          // n complex128 (re, im)
  const unsigned* mt0;          // MT19937 state right after seeding (624 words)
  xm_search_result* out;        // device-accessible result record
  const double* xs;             // evaluation mode: n_eval parameter pairs (degrees)
  double* fs;                   // ... their scores
  double c0, cstep, x_range;    // uniform coordinate axis: c[k] = c0 + k cstep; x_range = max c - min c
  double tol;
  unsigned long long seq;
  int n, n_eval, p0_only, maxiter, target_idx;
};

struct alignas(16) SearchLds {
  unsigned mt[624];
  unsigned y[1248];          // tempered outputs of TWO generations of the state (the stream never waits at a seam)
  double tab[2 * 96];        // A[14] | B[32] | C[P + 1], (cos, sin) pairs
  double part[kWaves][6];    // per worker wave: sum ds, sum ds ln ds, sum a, sum (a/2)^2, max d
  double prm[4];             // p0r, p1r of the evaluation in flight
  double lhs[30][2];         // latin hypercube samples
  double en[32];             // energies (convergence test)
  double dv[32];             // ... their squared deviations
  double pts[3][2], dx[2], vals[4];  // gradient test: the points, the exact steps, the scores
  double amax_v[kWaves];
  int amax_i[kWaves];
  int stop;
};

#define SDEV __device__ __forceinline__

SDEV double rdlane_d(double v, int lane) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}

// numpy's pairwise summation for n <= 128 (xm_solver.cpp np_sum_small): the population has 15 or 30 members
SDEV double np_sum_small(const double* a, int n) {
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

// ln x for finite x > 0: fdlibm's reduction and polynomial, as xm_solver_obj.cpp::fast_log (no special cases: the
// caller multiplies by x and skips x == 0)
SDEV double log_pos(double x) {
  long long ix = __double_as_longlong(x);
  const long long k = (ix - 0x3fe6a09e667f3bcdLL) >> 52;
  const double m = __longlong_as_double(ix - (k << 52));
  const double kd = (double)(int)k;
  const double f = m - 1.0;
  // s = f / (2 + f): v_rcp_f64 + two Newton steps + one correction of the quotient (<= 1 ulp; an IEEE division is
  // twice the instructions, and this quotient only feeds a polynomial)
  const double y = 2.0 + f;
  double r = __builtin_amdgcn_rcp(y);
  r = fma(fma(-y, r, 1.0), r, r);
  r = fma(fma(-y, r, 1.0), r, r);
  double s = f * r;
  s = fma(fma(-y, s, f), r, s);
  const double z = s * s;
  const double w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                            6.666666666666735130e-01);
  return fma(kd, 6.93147180559945286227e-01, f - s * (f - (t2 + t1)));
}

// wave-wide sum / maximum of a double through DPP (no LDS round trips); the result is in lane 63
template <bool MAX>
SDEV double wave_reduce_d(double v) {
  const double ident = MAX ? -DBL_MAX : 0.0;
  const long long ib = __double_as_longlong(ident);
  const int ilo = (int)(ib & 0xffffffffll), ihi = (int)(ib >> 32);
  auto step = [&](auto ctrl, auto rows) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(ilo, (int)(b & 0xffffffffll), decltype(ctrl)::value, decltype(rows)::value, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(ihi, (int)(b >> 32), decltype(ctrl)::value, decltype(rows)::value, 0xf, false);
    const double o = __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
    v = MAX ? fmax(v, o) : v + o;
  };
  using I = std::integral_constant<int, 0>;
  (void)sizeof(I);
  step(std::integral_constant<int, 0xB1>{}, std::integral_constant<int, 0xf>{});   // quad_perm:[1,0,3,2]
  step(std::integral_constant<int, 0x4E>{}, std::integral_constant<int, 0xf>{});   // quad_perm:[2,3,0,1]
  step(std::integral_constant<int, 0x141>{}, std::integral_constant<int, 0xf>{});  // row_half_mirror
  step(std::integral_constant<int, 0x140>{}, std::integral_constant<int, 0xf>{});  // row_mirror
  step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15 -> rows 1, 3
  step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31 -> rows 2, 3
  return v;
}

// The optimiser's random stream (wave 0; every value here is wave-uniform unless it is "one per lane").
struct Rng {
  SearchLds* L;
  unsigned yw;    // one per lane: word (pos - widx + lane) of the stream
  int pos, widx;  // pos: index of the next word in y[0 .. 1247] (two generations, circular); widx: its place in the window

#define XM_LDS_ORDER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
  // numpy mt19937_gen in three dependency-free phases, then the tempering of all 624 words into half `half` of y
  SDEV void produce(int half, int lane) {
    unsigned* mt = L->mt;
    const unsigned UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX = 0x9908b0dfu;
    // Lanes exchange words through the LDS inside one wave: every phase reads ALL of its inputs, then writes -- the
    // waits between are also compiler barriers (per-thread alias analysis would let a store pass another lane's load).
    unsigned v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // words 0 .. 226: mt[i + 397] is an old word
      const int i = lane + XM_WAVE * q;
      if (i < 227) {
        const unsigned yv = (mt[i] & UPPER) | (mt[i + 1] & LOWER);
        v[q] = mt[i + 397] ^ (yv >> 1) ^ ((0u - (yv & 1u)) & MATRIX);
      }
    }
    XM_LDS_ORDER();
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (lane + XM_WAVE * q < 227) mt[lane + XM_WAVE * q] = v[q];
    XM_LDS_ORDER();
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // words 227 .. 453: mt[i - 227] is a new word of the first phase
      const int i = 227 + lane + XM_WAVE * q;
      if (i < 454) {
        const unsigned yv = (mt[i] & UPPER) | (mt[i + 1] & LOWER);
        v[q] = mt[i - 227] ^ (yv >> 1) ^ ((0u - (yv & 1u)) & MATRIX);
      }
    }
    XM_LDS_ORDER();
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (227 + lane + XM_WAVE * q < 454) mt[227 + lane + XM_WAVE * q] = v[q];
    XM_LDS_ORDER();
#pragma unroll
    for (int q = 0; q < 3; ++q) {  // words 454 .. 622: mt[i - 227] is a new word of the second phase
      const int i = 454 + lane + XM_WAVE * q;
      if (i < 623) {
        const unsigned yv = (mt[i] & UPPER) | (mt[i + 1] & LOWER);
        v[q] = mt[i - 227] ^ (yv >> 1) ^ ((0u - (yv & 1u)) & MATRIX);
      }
    }
    XM_LDS_ORDER();
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (454 + lane + XM_WAVE * q < 623) mt[454 + lane + XM_WAVE * q] = v[q];
    XM_LDS_ORDER();
    if (lane == 0) {  // word 623 reads the NEW word 0
      const unsigned yv = (mt[623] & UPPER) | (mt[0] & LOWER);
      mt[623] = mt[396] ^ (yv >> 1) ^ ((0u - (yv & 1u)) & MATRIX);
    }
    XM_LDS_ORDER();
    unsigned* y = L->y + 624 * half;
    for (int i = lane; i < 624; i += XM_WAVE) {
      unsigned w = mt[i];
      w ^= (w >> 11);
      w ^= (w << 7) & 0x9d2c5680u;
      w ^= (w << 15) & 0xefc60000u;
      w ^= (w >> 18);
      y[i] = w;
    }
    XM_LDS_ORDER();
  }
  SDEV void start(int lane) {  // (the seeded state has produced nothing yet: numpy generates on the first draw)
    produce(0, lane);
    produce(1, lane);
    pos = 0;
    widx = XM_WAVE;
  }
  SDEV void refill(int lane) {  // the next 64 words of the stream, one per lane
    int i = pos + lane;
    i -= i >= 1248 ? 1248 : 0;
    yw = L->y[i];
    widx = 0;
  }
  // `k` (<= 64) words were consumed.  Leaving a half frees it for the generation after the one ahead: the words of a
  // loaded window that are still unread all lie beyond the seam.
  SDEV void advance(int k, int lane) {
    const int np = pos + k;
    if (pos < 624 && np >= 624) {
      produce(0, lane);
      pos = np;
    } else if (np >= 1248) {
      produce(1, lane);
      pos = np - 1248;
    } else {
      pos = np;
    }
  }
  SDEV unsigned next32(int lane) {
    if (widx == XM_WAVE) refill(lane);
    const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)yw, widx);
    ++widx;
    advance(1, lane);
    return w;
  }
  SDEV double next_double(int lane) {  // random_double: 53 bits from two draws
    const int a = (int)(next32(lane) >> 5), b = (int)(next32(lane) >> 6);
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
  }
  SDEV unsigned interval(unsigned max, int lane) {  // legacy random_interval: masked rejection
    if (max == 0) return 0;
    const unsigned mask = 0xffffffffu >> __builtin_clz(max);
    unsigned v;
    while ((v = (next32(lane) & mask)) > max) {
    }
    return v;
  }
  // RandomState._shuffle_raw (for i = n-1 .. 1: j = interval(i); swap(a[i], a[j])) on an array held one element per
  // lane.  The accepted draw of step i is the first word at or behind the stream position whose masked value is <= i:
  // a ballot over the 64-word window answers that for every position at once, so a step is a shift, a count of
  // trailing zeros and one v_readlane instead of a rejection loop; the swaps are tracked as "where has this lane's
  // element gone" (two compares and two selects per step, no cross-lane traffic) and applied by ONE ds_permute.
  // Fully unrolled and free of branches, so that the ballots, the scalar chain (position -> next position), the
  // v_readlanes and the tracking overlap in the wave's one instruction stream.  A window that runs out (more than
  // 64 draws for <= 29 steps: rare) leaves everything untouched and the plain loop does the shuffle.
  template <int NN>
  SDEV int shuffle_n(int arr, int lane) {
    refill(lane);
    int p = lane, cur = 0;
    bool ok = true;
#pragma unroll
    for (int i = NN - 1; i >= 1; --i) {
      const unsigned mask = 0xffffffffu >> __builtin_clz((unsigned)i);
      const unsigned long long acc = __ballot((yw & mask) <= (unsigned)i);
      const int c = cur < XM_WAVE - 1 ? cur : XM_WAVE - 1;
      const unsigned long long m = acc >> c;
      ok = ok && cur < XM_WAVE && m != 0ull;
      const int q0 = c + (int)__builtin_ctzll(m | 0x8000000000000000ull);
      const int q = q0 < XM_WAVE - 1 ? q0 : XM_WAVE - 1;
      const int j = (int)((unsigned)__builtin_amdgcn_readlane((int)yw, q) & mask);
      cur = q + 1;
      p = p == i ? j : (p == j ? i : p);
    }
    if (ok) {
      widx = cur;
      advance(cur, lane);
      return __builtin_amdgcn_ds_permute(p << 2, arr);  // lane l's element goes to lane p
    }
    for (int i = NN - 1; i >= 1; --i) {  // (the window is loaded, nothing was consumed)
      const int j = (int)interval((unsigned)i, lane);
      const int ai = __builtin_amdgcn_readlane(arr, i), aj = __builtin_amdgcn_readlane(arr, j);
      arr = lane == i ? aj : (lane == j ? ai : arr);
    }
    return arr;
  }
  SDEV int shuffle(int arr, int n, int lane) { return n == 30 ? shuffle_n<30>(arr, lane) : shuffle_n<15>(arr, lane); }
};

struct Drawn {  // the random part of one trial (does not depend on the population)
  int fill, r0, r1;
  bool cross[2];
};

template <int P, bool FULL>
__global__ __launch_bounds__(kThreads) void k_search(SearchArgs A) {
  __shared__ SearchLds L;
  const int t = (int)threadIdx.x, lane = t & (XM_WAVE - 1), wave = t / XM_WAVE;
  const int n = A.n;
  const bool worker = wave > 0;
  const int tw = t - XM_WAVE;  // worker index
  const int k0 = tw * P;       // first bin of a worker

  // ---- the spectrum: P + 1 bins per worker in registers (the last one is the next worker's first) ----------------
  double re[P + 1], im[P + 1];
  if (worker) {
#pragma unroll
    for (int j = 0; j <= P; ++j) {
      const int k = k0 + j;
      const bool ok = k < n;
      const double2 v = ok ? reinterpret_cast<const double2*>(A.slice)[k] : make_double2(0., 0.);
      re[j] = v.x;
      im[j] = v.y;
    }
  } else {
    for (int i = lane; i < 624; i += XM_WAVE) L.mt[i] = A.mt0[i];
    if (lane == 0) L.stop = 0;
  }
  // ---- first arg-max of |slice| (phasing.py:229 on the winning row; target_idx >= 0: given) ------------------------
  int kwin = A.target_idx;
  if (kwin < 0) {
    double bv = -1.;
    int bi = 0x7fffffff;
    if (worker) {
#pragma unroll
      for (int j = 0; j < P; ++j) {
        const double m2 = re[j] * re[j] + im[j] * im[j];
        if (k0 + j < n && m2 > bv) {
          bv = m2;
          bi = k0 + j;
        }
      }
      for (int m = 1; m < XM_WAVE; m <<= 1) {
        const double ov = __shfl_xor(bv, m);
        const int oi = __shfl_xor(bi, m);
        if (ov > bv || (ov == bv && oi < bi)) {
          bv = ov;
          bi = oi;
        }
      }
      if (lane == 0) {
        L.amax_v[wave] = bv;
        L.amax_i[wave] = bi;
      }
    }
    __syncthreads();
    bv = -1.;
    bi = 0x7fffffff;
    for (int w = 1; w < kWaves; ++w) {
      const double ov = L.amax_v[w];
      const int oi = L.amax_i[w];
      if (ov > bv || (ov == bv && oi < bi)) {
        bv = ov;
        bi = oi;
      }
    }
    kwin = bi;
  }
  __syncthreads();
  // u[k] = (c[k] - pivot) / x_range = u0 + k du   (phasing.py:69 on a uniform axis)
  const double pivot = A.c0 + A.cstep * (double)kwin;
  const double u0 = (A.c0 - pivot) / A.x_range, du = A.cstep / A.x_range;
  const double kRad = 3.14159265358979323846 / 180.0;  // np.radians

  // One evaluation, workers' side: the phase tables of (p0r, p1r), then this thread's share of the five sums.
  auto tables = [&]() {
    if (worker && tw < kTabA + 32 + P + 1) {
      const double p0r = L.prm[0], p1r = L.prm[1];
      double ang;
      if (tw < kTabA)
        ang = p0r + p1r * (u0 + du * (double)(32 * P * tw));
      else if (tw < kTabA + 32)
        ang = p1r * (du * (double)(P * (tw - kTabA)));
      else
        ang = p1r * (du * (double)(tw - kTabA - 32));
      double sn, cs;
      sincos(ang, &sn, &cs);
      L.tab[2 * tw] = cs;
      L.tab[2 * tw + 1] = sn;
    }
  };
  auto evaluate = [&]() {
    if (!worker) return;
    const double* ta = L.tab + 2 * (tw >> 5);
    const double* tb = L.tab + 2 * (kTabA + (tw & 31));
    const double* tc = L.tab + 2 * (kTabA + 32);
    const double ar = ta[0], ai = ta[1], br = tb[0], bi = tb[1];
    const double xr = fma(ar, br, -(ai * bi)), xi = fma(ar, bi, ai * br);
    auto phased = [&](int j) {  // Re(slice[k] e^{i phi_k})   (phasing.py:73, 104)
      const double cr = tc[2 * j], ci = tc[2 * j + 1];
      const double tr = fma(xr, cr, -(xi * ci)), ti = fma(xr, ci, xi * cr);
      return fma(re[j], tr, -(im[j] * ti));
    };
    double s_ds = 0., s_dl = 0., s_as = 0., s_as2 = 0., mx = -DBL_MAX;
    double dj = phased(0);
#pragma unroll
    for (int j = 0; j < P; ++j) {
      const double dn = phased(j + 1);
      const int k = k0 + j;
      // FULL (n == 448 P): all bins exist, and only the very last one has no forward difference -- no per-bin predicates
      const bool bin = FULL || k < n, diff = FULL ? (j + 1 < P || k + 1 < n) : k + 1 < n;
      const double as_ = dj - fabs(dj);
      s_as += bin ? as_ : 0.;
      s_as2 = fma(0.5 * as_, bin ? 0.5 * as_ : 0., s_as2);
      mx = bin ? fmax(mx, dj) : mx;
      const double ds = diff ? fabs((dn - dj) * 0.5) : 0.;
      s_ds += ds;
      // zeros -> p = 1 -> contribute 0 (phasing.py:110): ds ln(max(ds, tiny)) is exactly 0 there, without a branch
      s_dl = fma(ds, log_pos(fmax(ds, 2.2250738585072014e-308)), s_dl);
      dj = dn;
      // (four bins at a time: let loose, the scheduler interleaves all P logarithms and spills hundreds of registers)
      if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    s_ds = wave_reduce_d<false>(s_ds);
    s_dl = wave_reduce_d<false>(s_dl);
    s_as = wave_reduce_d<false>(s_as);
    s_as2 = wave_reduce_d<false>(s_as2);
    mx = wave_reduce_d<true>(mx);
    if (lane == XM_WAVE - 1) {
      double* p = L.part[wave];
      p[0] = s_ds;
      p[1] = s_dl;
      p[2] = s_as;
      p[3] = s_as2;
      p[4] = mx;
    }
  };
  // ... wave 0's side: the 7 partial sums -> the score (xm_solver_obj.cpp::acme_combine's formula).  Lane q < 5 adds
  // up quantity q in wave order.
  auto combine = [&]() -> double {
    double acc = lane == 4 ? -DBL_MAX : 0.;
    if (lane < 5) {
#pragma unroll
      for (int w = 1; w < kWaves; ++w) {
        const double v = L.part[w][lane];
        acc = lane == 4 ? fmax(acc, v) : acc + v;
      }
    }
    const double s_ds = rdlane_d(acc, 0), s_dl = rdlane_d(acc, 1), s_as = rdlane_d(acc, 2), s_as2 = rdlane_d(acc, 3),
                 mx = rdlane_d(acc, 4);
    const double h = log(s_ds) - s_dl / s_ds;  // H = -sum p ln p,  p = ds / S
    const double pfun = s_as < 0. ? s_as2 : 0.;
    return (h + 1000.0 * pfun) / (double)n / mx;
  };

  // ---- wave 0: the optimiser's state (xm_solver.cpp::xm_solver_de, statement for statement) -----------------------
  const int N = A.p0_only ? 1 : 2;
  const int M = 15 * N;  // max(5, popsize * N)
  const double lo0 = -180.0, hi0 = 180.0, lo1 = -4000.0, hi1 = 4000.0;
  const double arg1_0 = 0.5 * (lo0 + hi0), arg2_0 = fabs(lo0 - hi0), arg1_1 = 0.5 * (lo1 + hi1), arg2_1 = fabs(lo1 - hi1);
  Rng rng;
  rng.L = &L;
  rng.yw = 0u;
  rng.pos = 0;
  rng.widx = XM_WAVE;
  double px0 = 0., px1 = 0., en = DBL_MAX;  // one population member per lane (lanes >= M: unused)
  int ridx = lane;                          // _random_population_index
  int nfev = 0, nit = 1, status = 1;
  double scl = 0., scl_next = 0., tr0 = 0., tr1 = 0.;
  double x0 = 0., x1 = 0., fun = 0.;
  Drawn dr;
  dr.fill = dr.r0 = dr.r1 = 0;
  dr.cross[0] = dr.cross[1] = false;

  auto promote = [&]() {  // _promote_lowest_energy: first arg-min to slot 0
    double m = lane < M ? en : DBL_MAX;
    for (int s = 1; s < XM_WAVE; s <<= 1) m = fmin(m, __shfl_xor(m, s));
    const unsigned long long hit = __ballot(lane < M && en == m);
    const int l = hit ? (int)__builtin_ctzll(hit) : 0;
    if (l != 0) {
      const double a0 = rdlane_d(px0, 0), a1 = rdlane_d(px1, 0), ae = rdlane_d(en, 0);
      const double b0 = rdlane_d(px0, l), b1 = rdlane_d(px1, l), be = rdlane_d(en, l);
      if (lane == 0) {
        px0 = b0;
        px1 = b1;
        en = be;
      } else if (lane == l) {
        px0 = a0;
        px1 = a1;
        en = ae;
      }
    }
  };
  auto draw = [&](int cand) {  // _mutate's random part: fill point, _select_samples, crossover mask -> dr
    dr.fill = N > 1 ? (int)rng.interval((unsigned)(N - 1), lane) : 0;
    ridx = rng.shuffle(ridx, M, lane);
    const int a0 = __builtin_amdgcn_readlane(ridx, 0), a1 = __builtin_amdgcn_readlane(ridx, 1),
              a2 = __builtin_amdgcn_readlane(ridx, 2);
    if (a0 == cand) {
      dr.r0 = a1;
      dr.r1 = a2;
    } else if (a1 == cand) {
      dr.r0 = a0;
      dr.r1 = a2;
    } else {
      dr.r0 = a0;
      dr.r1 = a1;
    }
    dr.cross[0] = rng.next_double(lane) < 0.7;
    dr.cross[1] = N > 1 ? rng.next_double(lane) < 0.7 : false;
    if (dr.fill == 0)
      dr.cross[0] = true;
    else
      dr.cross[1] = true;
  };

  // where a search's time goes (wave 0's view, ticks of the 100 MHz wall clock): naming the point, the tables, the
  // overlapped draw, waiting for the workers' sums, taking the score
  unsigned long long tk[5] = {0, 0, 0, 0, 0}, tk0 = wall_clock64(), tk_start = tk0;
  auto lap = [&](int which) {
    const unsigned long long now = wall_clock64();
    tk[which] += now - tk0;
    tk0 = now;
  };
  enum { PH_LIST = 0, PH_INIT = 1, PH_TRIAL = 2, PH_GRAD = 3 };
  int phase = A.n_eval > 0 ? PH_LIST : PH_INIT, idx = 0;
  if (!worker && phase == PH_INIT) {  // init_population_lhs
    XM_LDS_ORDER();  // (the seeded state, copied above)
    rng.start(lane);
    const double seg = 1.0 / (double)M;
    for (int i = 0; i < M; ++i)
      for (int j = 0; j < N; ++j) {
        const double v = seg * rng.next_double(lane) + (double)i * (1.0 / (double)M);
        if (lane == 0) L.lhs[i][j] = v;
      }
    XM_LDS_ORDER();
    for (int j = 0; j < N; ++j) {
      const int order = rng.shuffle(lane, M, lane);  // rng.permutation(range(M))
      const double v = lane < M ? L.lhs[order][j] : 0.;
      if (j == 0)
        px0 = v;
      else
        px1 = v;
    }
  }

  // ---- ONE loop, one evaluation per turn: wave 0 names the point (or ends the search), the workers' tables and sums,
  // wave 0 takes the score.  While the workers sum, wave 0 draws the random part of the NEXT trial.
  for (;;) {
    if (!worker) {
      double p0d = 0., p1d = 0.;  // the point, in degrees
      bool fin = false;
      if (phase == PH_LIST) {
        fin = idx >= A.n_eval;
        if (!fin) {
          p0d = A.xs[2 * idx];
          p1d = A.p0_only ? 0. : A.xs[2 * idx + 1];
        }
      } else if (phase == PH_INIT) {  // initial energies
        p0d = arg1_0 + (rdlane_d(px0, idx) - 0.5) * arg2_0;
        p1d = N > 1 ? arg1_1 + (rdlane_d(px1, idx) - 0.5) * arg2_1 : 0.;
      } else if (phase == PH_TRIAL) {
        // _mutate (best1bin) + _ensure_constraint on the population as it stands now
        const int c = idx;
        const double b0 = rdlane_d(px0, 0) + scl * (rdlane_d(px0, dr.r0) - rdlane_d(px0, dr.r1));
        const double b1 = rdlane_d(px1, 0) + scl * (rdlane_d(px1, dr.r0) - rdlane_d(px1, dr.r1));
        tr0 = dr.cross[0] ? b0 : rdlane_d(px0, c);
        tr1 = N > 1 ? (dr.cross[1] ? b1 : rdlane_d(px1, c)) : 0.;
        if (tr0 > 1 || tr0 < 0) tr0 = rng.next_double(lane);
        if (N > 1 && (tr1 > 1 || tr1 < 0)) tr1 = rng.next_double(lane);
        p0d = arg1_0 + (tr0 - 0.5) * arg2_0;
        p1d = N > 1 ? arg1_1 + (tr1 - 0.5) * arg2_1 : 0.;
      } else {  // PH_GRAD: x, x + h0 e0[, x + h1 e1]
        fin = idx > N;
        if (!fin) {
          p0d = L.pts[idx][0];
          p1d = L.pts[idx][1];
        }
      }
      if (lane == 0) {
        L.stop = fin ? 1 : 0;
        L.prm[0] = p0d * kRad;
        L.prm[1] = p1d * kRad;
      }
      lap(0);
    }
    __syncthreads();
    if (L.stop) break;
    tables();
    __syncthreads();
    if (!worker) lap(1);
    if (worker) {
      evaluate();
    } else if (phase == PH_TRIAL) {  // meanwhile: the next trial's random part (a new generation's dither comes first)
      if (idx + 1 < M) {
        draw(idx + 1);
      } else {
        scl_next = 0.5 + (1.0 - 0.5) * rng.next_double(lane);  // dither: rng.uniform(0.5, 1)
        draw(0);
      }
    }
    if (!worker) lap(2);
    __syncthreads();
    if (!worker) {
      lap(3);
      const double f = combine();
      if (phase == PH_LIST) {
        if (lane == 0) A.fs[idx] = f;
        ++idx;
      } else if (phase == PH_INIT) {
        if (lane == idx) en = f;
        ++nfev;
        if (++idx == M) {
          promote();
          scl = 0.5 + (1.0 - 0.5) * rng.next_double(lane);
          draw(0);
          phase = PH_TRIAL;
          idx = 0;
        }
      } else if (phase == PH_TRIAL) {
        const int c = idx;
        ++nfev;
        if (f <= rdlane_d(en, c)) {
          if (lane == c) {
            px0 = tr0;
            px1 = tr1;
            en = f;
          }
          if (f <= rdlane_d(en, 0)) promote();
        }
        if (++idx == M) {  // end of a generation: converged()?  std(energies) <= atol + tol * |mean(energies)|
          if (lane < M) L.en[lane] = en;
          XM_LDS_ORDER();
          bool any_inf = false;
          for (int i = 0; i < M; ++i) any_inf |= !(fabs(L.en[i]) <= 1.79769313486231570815e308);
          bool stop = false;
          if (!any_inf) {
            const double mean = np_sum_small(L.en, M) / (double)M;
            if (lane < M) L.dv[lane] = (en - mean) * (en - mean);
            XM_LDS_ORDER();
            const double sd = sqrt(np_sum_small(L.dv, M) / (double)M);
            stop = sd <= A.tol * fabs(mean);
          }
          if (stop) status = 0;
          if (stop || nit >= A.maxiter) {
            // the test scipy's polish starts with: f and the forward-difference gradient at the best member
            // (xm_solver.cpp::xm_solver_fg: approx_derivative "2-point", abs_step 1e-8, bounds-aware steps)
            x0 = arg1_0 + (rdlane_d(px0, 0) - 0.5) * arg2_0;
            x1 = N > 1 ? arg1_1 + (rdlane_d(px1, 0) - 0.5) * arg2_1 : 0.;
            fun = rdlane_d(en, 0);
            if (lane == 0) {
              for (int r = 0; r < 3; ++r) {
                L.pts[r][0] = x0;
                L.pts[r][1] = x1;
              }
              for (int i = 0; i < N; ++i) {
                const double xc = i == 0 ? x0 : x1, lb = i == 0 ? lo0 : lo1, ub = i == 0 ? hi0 : hi1;
                double step = 1e-8;
                if ((xc + step) - xc == 0.0) step = 1.4901161193847656e-08 * (xc >= 0.0 ? 1.0 : -1.0) * fmax(1.0, fabs(xc));
                const double lower = xc - lb, upper = ub - xc;
                const double xt = xc + step;
                const bool violated = xt < lb || xt > ub;
                const bool fitting = fabs(step) <= fmax(lower, upper);
                if (violated && fitting) step = -step;
                if (!fitting) step = upper >= lower ? upper : -lower;
                const double pnt = xc + step;
                L.pts[1 + i][i] = pnt;
                L.dx[i] = pnt - xc;
              }
            }
            XM_LDS_ORDER();
            phase = PH_GRAD;
            idx = 0;
          } else {
            ++nit;
            scl = scl_next;  // drawn, with candidate 0's random part, while the generation's last trial was summed
            idx = 0;
          }
        }
      } else {  // PH_GRAD
        if (lane == 0) L.vals[idx] = f;
        ++idx;
      }
      lap(4);
    }
  }

  if (t == 0) {
    xm_search_result* o = A.out;
    if (A.n_eval > 0) {
      if (o) o->target_idx = kwin;
    } else {
      double pgn = 0.;
      for (int i = 0; i < N; ++i) {
        const double xi = i == 0 ? x0 : x1, lb = i == 0 ? lo0 : lo1, ub = i == 0 ? hi0 : hi1;
        const double g = (L.vals[1 + i] - L.vals[0]) / L.dx[i];
        const double pg = g < 0. ? fmax(xi - ub, g) : fmin(xi - lb, g);  // L-BFGS-B's projgr, both bounds set
        pgn = fmax(pgn, fabs(pg));
      }
      o->x[0] = x0;
      o->x[1] = x1;
      o->fun = fun;
      o->pg_norm = pgn;
      o->nfev = nfev;
      o->nit = nit;
      o->status = status;
      o->needs_polish = pgn <= 0.5e-5 ? 0 : 1;
      o->target_idx = kwin;
      o->pad_ = 0;
      for (int i = 0; i < 5; ++i) o->t_us[i] = 0.01 * (double)tk[i];
      o->t_us[5] = 0.01 * (double)(wall_clock64() - tk_start);
      o->t_us[6] = o->t_us[7] = 0.;
    }
    if (o) {
      __threadfence_system();
      __hip_atomic_store((unsigned long long*)&o->seq, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// numpy/random/src/mt19937/mt19937.c mt19937_seed
void mt_seed(unsigned s, unsigned* mt) {
  mt[0] = s;
  for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (unsigned)i;
}

std::mutex g_seed_mu;
std::map<std::pair<int, unsigned>, unsigned*> g_seed_tables;  // (device, seed) -> 624 words in device memory

int seed_table(unsigned seed, const unsigned** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_seed_mu);
  auto it = g_seed_tables.find({dev, seed});
  if (it == g_seed_tables.end()) {
    unsigned h[624];
    mt_seed(seed, h);
    unsigned* d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(h)));
    HIP_TRY(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
    it = g_seed_tables.emplace(std::make_pair(dev, seed), d).first;
  }
  *out = it->second;
  return XM_OK;
}

int points_per_worker(int n) {
  const int need = (n + kWorkers - 1) / kWorkers;
  for (int p : {1, 2, 3, 5, 10, 19, kMaxP})
    if (p >= need) return p;
  return 0;
}

template <int PP>
void launch_p(const SearchArgs& A, hipStream_t st) {
  if (A.n == kWorkers * PP)
    hipLaunchKernelGGL((k_search<PP, true>), dim3(1), dim3(kThreads), 0, st, A);
  else
    hipLaunchKernelGGL((k_search<PP, false>), dim3(1), dim3(kThreads), 0, st, A);
}

int launch(const SearchArgs& A, hipStream_t st) {
  switch (points_per_worker(A.n)) {
    case 1: launch_p<1>(A, st); break;
    case 2: launch_p<2>(A, st); break;
    case 3: launch_p<3>(A, st); break;
    case 5: launch_p<5>(A, st); break;
    case 10: launch_p<10>(A, st); break;
    case 19: launch_p<19>(A, st); break;
    case 37: launch_p<37>(A, st); break;
    default:
      return xm_fail(XM_ERR_UNSUPPORTED_N, "device search: at most " + std::to_string(kWorkers * kMaxP) + " bins");
  }
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

}  // namespace

extern "C" {

int xm_search_supported(int n, int method, double x_range) {
  return n >= 2 && method == 0 && points_per_worker(n) > 0 && x_range > 0.0 ? 1 : 0;
}

int xm_search_launch(const void* slice, int n, double c0, double cstep, double x_range, int method, int p0_only,
                     unsigned seed, double tol, int maxiter, uint64_t seq, xm_search_result* out, void* stream) {
  if (!slice || !out || n < 2 || maxiter < 1 || !(x_range > 0.0)) return xm_fail(XM_ERR_INVALID_ARG, "xm_search_launch: bad argument");
  if (method != 0) return xm_fail(XM_ERR_INVALID_ARG, "xm_search_launch: only the ACME objective (method 0) runs on the device");
  SearchArgs A;
  std::memset(&A, 0, sizeof(A));
  int rc = seed_table(seed, &A.mt0);
  if (rc) return rc;
  A.slice = (const double*)slice;
  A.out = out;
  A.c0 = c0;
  A.cstep = cstep;
  A.x_range = x_range;
  A.tol = tol;
  A.seq = seq;
  A.n = n;
  A.n_eval = 0;
  A.p0_only = p0_only ? 1 : 0;
  A.maxiter = maxiter;
  A.target_idx = -1;
  return launch(A, (hipStream_t)stream);
}

int xm_search_eval(const void* slice, int n, double c0, double cstep, double x_range, int target_idx, int p0_only,
                   const double* xs, int count, double* fs, void* stream) {
  if (!slice || !xs || !fs || n < 2 || count < 1 || !(x_range > 0.0) || target_idx >= n)
    return xm_fail(XM_ERR_INVALID_ARG, "xm_search_eval: bad argument");
  SearchArgs A;
  std::memset(&A, 0, sizeof(A));
  int rc = seed_table(42u, &A.mt0);
  if (rc) return rc;
  A.slice = (const double*)slice;
  A.xs = xs;
  A.fs = fs;
  A.c0 = c0;
  A.cstep = cstep;
  A.x_range = x_range;
  A.n = n;
  A.n_eval = count;
  A.p0_only = p0_only ? 1 : 0;
  A.maxiter = 1;
  A.target_idx = target_idx;
  return launch(A, (hipStream_t)stream);
}

}  // extern "C"
