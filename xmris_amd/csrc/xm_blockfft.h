// Workgroup-level Stockham (auto-sort, decimation-in-time) FFT for gfx950.
//
// One spectrum of length N is held by NT threads, P = N/NT points per thread, in the
// "strided layout"  v[q] = x[t + NT*q].  A stage of radix R does P/R in-register radix-R
// butterflies per thread, multiplies by the stage twiddles, and (except for the last stage)
// exchanges through LDS; the last stage's outputs land back in the strided layout, so a
// K-stage plan needs K-1 exchanges.  Stage s (sub-transform length so far Ns) of butterfly
// b = t + NT*u:
//     in : x[b + (N/R)*r]                     (= register u + (P/R)*r)
//     tw : W_{Ns*R}^{r*(b mod Ns)}            (table [(r-1)*Ns + (b mod Ns)], fp64-computed)
//     out: y[(b/Ns)*Ns*R + (b mod Ns) + r*Ns]
// LDS rows are padded by one element every 16 so that the stage-0 scatter (lane stride R
// elements) and the strided gather are both (nearly) bank-conflict free for ds_write_b64 /
// ds_read_b64 (bank = (addr/4) mod 32 resp. 64, serviced in 16/32-lane groups).
#pragma once
#include "xm_dft.h"

#include <type_traits>

// LDS rows are padded by one element every 128 bytes: 16 x 8 B (c64), 8 x 16 B (c128 or a two-lane
// c64 pair), 4 x 32 B (two-lane c128).
constexpr int xm_pad_shift(int elem_bytes) { return elem_bytes <= 8 ? 4 : (elem_bytes == 16 ? 3 : 2); }
template <int SH>
XM_DEV int xm_pad(int i) {
  return i + (i >> SH);
}

template <int N_, int NT_, int... Rs>
struct FftPlan {
  static constexpr int N = N_;
  static constexpr int NT = NT_;
  static constexpr int P = N_ / NT_;
  static constexpr int K = sizeof...(Rs);
  static_assert(N_ % NT_ == 0, "NT must divide N");
  static constexpr int radix(int i) {
    constexpr int r[] = {Rs...};
    return r[i];
  }
  static constexpr int ns(int i) {  // product of radices before stage i
    int p = 1;
    for (int j = 0; j < i; ++j) p *= radix(j);
    return p;
  }
  static constexpr int tw_offset(int i) {  // table offset (elements) of stage i (i >= 1)
    int o = 0;
    for (int j = 1; j < i; ++j) o += (radix(j) - 1) * ns(j);
    return o;
  }
  static constexpr int tw_size() { return tw_offset(K); }
  // distinguishes the plans of one length in the table cache: threads, stage count and the first two radices
  static constexpr int signature() { return NT | (K << 12) | (radix(0) << 16) | ((K > 1 ? radix(1) : 0) << 24); }
  static constexpr bool valid() {
    int p = 1;
    for (int j = 0; j < K; ++j) {
      if (P % radix(j) != 0) return false;
      p *= radix(j);
    }
    return p == N;
  }
  static_assert(valid(), "radices must multiply to N and each divide P");
  // exchange-buffer elements for an element of `es` bytes (one pad element per 128 bytes)
  static constexpr int lds_elems(int es) { return K > 1 ? (N + (N >> xm_pad_shift(es))) : 0; }
  // all radices (and NT) powers of two: index sums below are bitwise ORs, so the padded LDS address of
  // (lane part + compile-time part) splits into a per-lane base plus an immediate offset
  static constexpr bool pow2() {
    for (int j = 0; j < K; ++j)
      if (radix(j) & (radix(j) - 1)) return false;
    return (NT & (NT - 1)) == 0;
  }
  // the same for every stage that SCATTERS (all but the last, whose outputs stay in registers): a plan that keeps its
  // odd factor for the last stage (4.4.4.4.20) addresses the LDS like a power-of-two plan
  static constexpr bool scatter_pow2() {
    for (int j = 0; j + 1 < K; ++j)
      if (radix(j) & (radix(j) - 1)) return false;
    return (NT & (NT - 1)) == 0;
  }
};


// The same plan with the plane-by-plane exchange forced (BlockFFT::SPLIT): half the exchange buffer, i.e. room for a
// second workgroup on the CU, for twice the barriers
template <class PL>
struct SplitPlan : PL {
  static constexpr bool kForceSplit = true;
};
template <class PL, class = void>
struct xm_force_split : std::false_type {};
template <class PL>
struct xm_force_split<PL, typename std::enable_if<PL::kForceSplit>::type> : std::true_type {};

// Twiddle sources.  get<S, U, R_IDX>(k): twiddle of stage S, butterfly u, input r (1-based), table
// column k = (b mod Ns).  All template arguments are compile-time so register tables stay in VGPRs.
// Twiddles are scalar complex (Cx<S>) even when the data is two-lane (Cx<V>, S = ScalarOf<V>).
template <class S, class PL>
struct GlobalTw {  // every stage from the (L2-resident) global table
  static constexpr bool kHasR0 = false;
  const Cx<S>* __restrict__ tw;
  template <int ST, int U, int R1>
  XM_DEV Cx<S> get(int k) const {
    return tw[PL::tw_offset(ST) + (R1 - 1) * PL::ns(ST) + k];
  }
};

// R0: the last stage also multiplies its input r = 0 (a per-thread unit factor folded into the last stage's
// twiddles, e.g. the per-thread part of a linear output phase: see HotTw::fold)
template <class S, class PL, bool R0 = false>
struct HotTw {  // middle stages from an LDS copy of the table (or the global table when that copy would be
                // larger than 8 KiB), last stage from per-thread registers
  static constexpr bool kHasR0 = R0;
  static constexpr int K = PL::K;
  static constexpr int RL = PL::radix(K - 1);
  static constexpr int NREG = (K > 1) ? (PL::P / RL) * (RL - 1) : 1;
  static constexpr int mid_size() { return K > 1 ? PL::tw_offset(K - 1) : 0; }
  static constexpr bool mid_in_lds() { return mid_size() * (int)sizeof(Cx<S>) <= 8192; }
  static constexpr int mid_lds_size() { return mid_in_lds() ? mid_size() : 0; }
  const Cx<S>* mid;  // mid_size() entries: LDS copy or the global table itself
  Cx<S> reg[NREG];
  Cx<S> r0[R0 ? (K > 1 ? PL::P / RL : 1) : 1];  // R0: twiddle of input r = 0 of each last-stage butterfly
  // multiply every last-stage twiddle (and the implicit 1 of input r = 0) by the unit factor f: the transform's
  // outputs of this thread all come out multiplied by f, for one extra complex multiply per butterfly
  XM_DEV void fold(Cx<S> f) {
    static_assert(R0, "fold needs the r = 0 twiddle slot");
#pragma unroll
    for (int i = 0; i < NREG; ++i) reg[i] = reg[i] * f;
#pragma unroll
    for (int u = 0; u < (K > 1 ? PL::P / RL : 1); ++u) r0[u] = f;
  }
  // thread t loads its last-stage twiddles: butterfly b = t + NT*u, input r -> table[(r-1)*Ns + b]
  XM_DEV void load(const Cx<S>* __restrict__ tw, int t) {
    if constexpr (K > 1) {
      constexpr int Ns = PL::ns(K - 1);
#pragma unroll
      for (int u = 0; u < PL::P / RL; ++u)
#pragma unroll
        for (int r = 1; r < RL; ++r)
          reg[u * (RL - 1) + (r - 1)] = tw[PL::tw_offset(K - 1) + (r - 1) * Ns + t + PL::NT * u];
    }
  }
  template <int ST, int U, int R1>
  XM_DEV Cx<S> get(int k) const {
    if constexpr (ST == K - 1)
      return reg[U * (RL - 1) + (R1 - 1)];
    else
      return mid[PL::tw_offset(ST) + (R1 - 1) * PL::ns(ST) + k];
  }
};

// V = element real type: a scalar (one spectrum per NT threads) or a two-lane vector (two transforms
// carried side by side in the packed-math lanes).
template <class V, class PL, int SH_OVERRIDE = -1>
struct BlockFFT {
  using S = typename ScalarOf<V>::type;
  static constexpr int N = PL::N, NT = PL::NT, P = PL::P, K = PL::K;
  // pad shift: one pad element per 2^SH elements.  Default: per 128 bytes.  For 16-byte elements and a
  // power-of-two first radix R0 >= 8 the stage-0 scatter (lane stride R0 elements) wants exactly one pad per
  // R0 elements: the lane stride becomes R0*16+16 bytes = 4 dwords (mod 32 banks), i.e. the 8 lanes of a
  // ds_write_b128 group land on 8 distinct 4-bank slots (R0 = 16 with a pad every 8 elements was a 2-way
  // conflict: 37 % of the LDS cycles of the 16-point plan).  A kernel that must squeeze two workgroups
  // into the LDS may override.
  static constexpr int r0_shift() {
    const int r0 = PL::radix(0);
    if (sizeof(Cx<V>) >= 16 && r0 >= 8 && (r0 & (r0 - 1)) == 0) {
      int s = 0;
      while ((1 << s) < r0) ++s;
      return s;
    }
    return xm_pad_shift((int)sizeof(Cx<V>));
  }
  // SPLIT: the padded exchange buffer of whole complex elements would not fit the 160 KiB LDS (complex128 16384:
  // 256 KiB), so the exchange goes through ONE plane of real numbers twice -- real parts, then imaginary parts (the
  // two are independent registers): half the buffer for twice the barriers.
  static constexpr bool SPLIT =
      SH_OVERRIDE < 0 && K > 1 &&
      (xm_force_split<PL>::value || (long long)(N + (N >> r0_shift())) * (long long)sizeof(Cx<V>) > 160LL * 1024);
  static constexpr int SH = SPLIT ? xm_pad_shift((int)sizeof(V)) : (SH_OVERRIDE >= 0 ? SH_OVERRIDE : r0_shift());
  // in units of Cx<V> (SPLIT: a plane of N padded V's)
  static constexpr int lds_elems() { return K > 1 ? (SPLIT ? (N + (N >> SH) + 1) / 2 : (N + (N >> SH))) : 0; }

  template <int ST, int U, int R1, int R, class TW>
  XM_DEV static void twiddle_row(Cx<V>* a, const TW& tw, int k) {
    if constexpr (R1 < R) {
      a[R1] = a[R1] * tw.template get<ST, U, R1>(k);
      twiddle_row<ST, U, R1 + 1, R>(a, tw, k);
    }
  }

  template <class TW, class = void>
  struct has_r0 : std::false_type {};
  template <class TW>
  struct has_r0<TW, typename std::enable_if<TW::kHasR0>::type> : std::true_type {};

  // `t0`: the butterfly (column) index this thread takes in stage 0 -- v[q] = x[t0 + NT*q] on entry.  Stage 0
  // has no twiddles and scatters its outputs by address, so any bijection thread -> column works there; every
  // later stage (and the result) uses the strided layout of `t`.
  template <int ST, int U, class TW>
  XM_DEV static void butterflies(Cx<V> (&v)[P], Cx<V>* lds, const TW& tw, int t, int t0) {
    constexpr int R = PL::radix(ST);
    constexpr int Ns = PL::ns(ST);
    constexpr int PR = P / R;
    constexpr bool last = (ST == K - 1);
    if constexpr (U < PR) {
      Cx<V> a[R];
#pragma unroll
      for (int r = 0; r < R; ++r) a[r] = v[U + PR * r];
      const int b = (ST == 0 ? t0 : t) + NT * U;
      if constexpr (ST > 0) {
        constexpr bool full = (Ns * R == N);  // last stage: b < Ns always
        const int k = full ? b : (b % Ns);
        if constexpr (last && has_r0<TW>::value) a[0] = a[0] * tw.r0[U];
        twiddle_row<ST, U, 1, R>(a, tw, k);
      }
      Dft<V, R>::run(a);
      if constexpr (last || SPLIT) {  // SPLIT: exchange_plane scatters the outputs, one component at a time
#pragma unroll
        for (int r = 0; r < R; ++r) v[U + PR * r] = a[r];
      } else {
        const int o = (b / Ns) * (Ns * R) + (b % Ns);
        if constexpr (PL::scatter_pow2()) {  // o has zero bits where r*Ns lives: pad(o + c) = pad(o) + pad(c)
          Cx<V>* wp = lds + xm_pad<SH>(o);
#pragma unroll
          for (int r = 0; r < R; ++r) wp[r * Ns + ((r * Ns) >> SH)] = a[r];
        } else {
#pragma unroll
          for (int r = 0; r < R; ++r) lds[xm_pad<SH>(o + r * Ns)] = a[r];
        }
      }
      butterflies<ST, U + 1>(v, lds, tw, t, t0);
    }
  }

  // SPLIT exchange of component C (0 = re, 1 = im) after stage ST: scatter from the butterfly layout left in v,
  // gather back in the strided layout; leaves the plane free for the next component / stage
  template <int ST, int C>
  XM_DEV static void exchange_plane(Cx<V> (&v)[P], V* plane, int t, int t0) {
    constexpr int R = PL::radix(ST);
    constexpr int Ns = PL::ns(ST);
    constexpr int PR = P / R;
#pragma unroll
    for (int u = 0; u < PR; ++u) {
      const int b = (ST == 0 ? t0 : t) + NT * u;
      const int o = (b / Ns) * (Ns * R) + (b % Ns);
      if constexpr (PL::scatter_pow2()) {  // per-lane base + immediate offsets, as in butterflies()
        V* wp = plane + xm_pad<SH>(o);
#pragma unroll
        for (int r = 0; r < R; ++r) wp[r * Ns + ((r * Ns) >> SH)] = C ? v[u + PR * r].im : v[u + PR * r].re;
      } else {
#pragma unroll
        for (int r = 0; r < R; ++r) plane[xm_pad<SH>(o + r * Ns)] = C ? v[u + PR * r].im : v[u + PR * r].re;
      }
    }
    __syncthreads();
    const V* rp = plane + (PL::scatter_pow2() ? xm_pad<SH>(t) : 0);
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const V x = PL::scatter_pow2() ? rp[NT * q + ((NT * q) >> SH)] : plane[xm_pad<SH>(t + NT * q)];
      if constexpr (C)
        v[q].im = x;
      else
        v[q].re = x;
    }
    __syncthreads();
  }

  struct NoHook {
    XM_DEV void operator()() const {}
  };

  // `hook` runs once per transform, in front of the first barrier of the LAST exchange (plans with K >= 2): whatever
  // it writes to LDS is visible to every thread when run() returns, at no extra barrier
  template <int ST, class TW, class HOOK>
  XM_DEV static void stage(Cx<V> (&v)[P], Cx<V>* lds, const TW& tw, int t, int t0, const HOOK& hook) {
    constexpr bool last = (ST == K - 1);
    butterflies<ST, 0>(v, lds, tw, t, t0);
    if constexpr (!last && SPLIT) {
      if constexpr (ST == K - 2) hook();
      V* plane = reinterpret_cast<V*>(lds);
      exchange_plane<ST, 0>(v, plane, t, t0);
      exchange_plane<ST, 1>(v, plane, t, t0);
      stage<ST + 1>(v, lds, tw, t, t0, hook);
    } else if constexpr (!last) {
      if constexpr (ST == K - 2) hook();
      __syncthreads();
      if constexpr (PL::scatter_pow2()) {
        const Cx<V>* rp = lds + xm_pad<SH>(t);
#pragma unroll
        for (int q = 0; q < P; ++q) v[q] = rp[NT * q + ((NT * q) >> SH)];
      } else {
#pragma unroll
        for (int q = 0; q < P; ++q) v[q] = lds[xm_pad<SH>(t + NT * q)];
      }
      __syncthreads();
      stage<ST + 1>(v, lds, tw, t, t0, hook);
    }
  }

  // forward, unnormalised.  `lds` = this spectrum's padded exchange buffer (lds_elems()).
  // All threads of the workgroup must call it (it contains workgroup barriers).
  template <class TW>
  XM_DEV static void run(Cx<V> (&v)[P], Cx<V>* lds, const TW& tw, int t) {
    stage<0>(v, lds, tw, t, t, NoHook{});
  }
  // stage-0 columns remapped: v[q] = x[t0 + NT*q] on entry (see butterflies)
  template <class TW>
  XM_DEV static void run_cols(Cx<V> (&v)[P], Cx<V>* lds, const TW& tw, int t, int t0) {
    stage<0>(v, lds, tw, t, t0, NoHook{});
  }
  template <class TW, class HOOK>
  XM_DEV static void run_cols(Cx<V> (&v)[P], Cx<V>* lds, const TW& tw, int t, int t0, const HOOK& hook) {
    stage<0>(v, lds, tw, t, t0, hook);
  }
  XM_DEV static void run(Cx<V> (&v)[P], Cx<V>* lds, const Cx<S>* __restrict__ tw, int t) {
    GlobalTw<S, PL> g{tw};
    stage<0>(v, lds, g, t, t, NoHook{});
  }
};
