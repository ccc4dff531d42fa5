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

template <int... Rs>
struct RadixList {};

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
  static constexpr bool valid() {
    int p = 1;
    for (int j = 0; j < K; ++j) {
      if (P % radix(j) != 0) return false;
      p *= radix(j);
    }
    return p == N;
  }
  static_assert(valid(), "radices must multiply to N and each divide P");
  static constexpr int lds_elems() { return K > 1 ? (N + (N >> 4)) : 0; }
};

XM_DEV int xm_pad(int i) { return i + (i >> 4); }

template <class T, class PL>
struct BlockFFT {
  static constexpr int N = PL::N, NT = PL::NT, P = PL::P, K = PL::K;

  template <int S>
  XM_DEV static void stage(Cx<T> (&v)[P], Cx<T>* lds, const Cx<T>* __restrict__ tw, int t) {
    constexpr int R = PL::radix(S);
    constexpr int Ns = PL::ns(S);
    constexpr int PR = P / R;
    constexpr bool last = (S == K - 1);
    const Cx<T>* __restrict__ tws = tw + PL::tw_offset(S);
#pragma unroll
    for (int u = 0; u < PR; ++u) {
      Cx<T> a[R];
#pragma unroll
      for (int r = 0; r < R; ++r) a[r] = v[u + PR * r];
      const int b = t + NT * u;
      if constexpr (S > 0) {
        constexpr bool full = (Ns * R == N);  // last stage: b < Ns always
        const int k = full ? b : (b % Ns);
#pragma unroll
        for (int r = 1; r < R; ++r) a[r] = a[r] * tws[(r - 1) * Ns + k];
      }
      Dft<T, R>::run(a);
      if constexpr (last) {
#pragma unroll
        for (int r = 0; r < R; ++r) v[u + PR * r] = a[r];
      } else {
        const int o = (b / Ns) * (Ns * R) + (b % Ns);
#pragma unroll
        for (int r = 0; r < R; ++r) lds[xm_pad(o + r * Ns)] = a[r];
      }
    }
    if constexpr (!last) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < P; ++q) v[q] = lds[xm_pad(t + NT * q)];
      __syncthreads();
      stage<S + 1>(v, lds, tw, t);
    }
  }

  // forward, unnormalised.  `lds` = this spectrum's padded exchange buffer (PL::lds_elems()).
  // All threads of the workgroup must call it (it contains workgroup barriers).
  XM_DEV static void run(Cx<T> (&v)[P], Cx<T>* lds, const Cx<T>* __restrict__ tw, int t) {
    stage<0>(v, lds, tw, t);
  }
};
