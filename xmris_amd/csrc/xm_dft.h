// In-register small DFTs (forward, e^{-2*pi*i*nk/R}), natural order in and out.
// Every index is a compile-time constant after unrolling so arrays stay in VGPRs
// (runtime-indexed register arrays go to scratch on gfx950: cdna guide rule 20).
#pragma once
#include "xm_common.h"

template <class T, int R>
struct Dft;

template <class T>
struct Dft<T, 1> {
  XM_DEV static void run(Cx<T>*) {}
};

template <class T>
struct Dft<T, 2> {
  XM_DEV static void run(Cx<T>* a) {
    Cx<T> t = a[0] - a[1];
    a[0] = a[0] + a[1];
    a[1] = t;
  }
};

template <class T>
struct Dft<T, 3> {
  XM_DEV static void run(Cx<T>* a) {
    using S = typename ScalarOf<T>::type;
    constexpr S c = S(-0.5);
    constexpr S s = S(0.86602540378443864676);  // sin(2pi/3)
    Cx<T> s12 = a[1] + a[2];
    Cx<T> d12 = a[1] - a[2];
    Cx<T> m = mk<T>(a[0].re + c * s12.re, a[0].im + c * s12.im);
    Cx<T> r = mk<T>(s * d12.im, -s * d12.re);  // -i*s*(a1-a2)
    a[0] = a[0] + s12;
    a[1] = m + r;
    a[2] = m - r;
  }
};

template <class T>
struct Dft<T, 4> {
  XM_DEV static void run(Cx<T>* a) {
    Cx<T> s02 = a[0] + a[2], d02 = a[0] - a[2];
    Cx<T> s13 = a[1] + a[3], d13 = mul_mi(a[1] - a[3]);
    a[0] = s02 + s13;
    a[1] = d02 + d13;
    a[2] = s02 - s13;
    a[3] = d02 - d13;
  }
};

template <class T>
struct Dft<T, 5> {
  XM_DEV static void run(Cx<T>* a) {
    using S = typename ScalarOf<T>::type;
    constexpr S c1 = S(0.30901699437494742410);   // cos(2pi/5)
    constexpr S c2 = S(-0.80901699437494742410);  // cos(4pi/5)
    constexpr S s1 = S(0.95105651629515357212);   // sin(2pi/5)
    constexpr S s2 = S(0.58778525229247312917);   // sin(4pi/5)
    Cx<T> s14 = a[1] + a[4], d14 = a[1] - a[4];
    Cx<T> s23 = a[2] + a[3], d23 = a[2] - a[3];
    Cx<T> m1 = mk<T>(a[0].re + c1 * s14.re + c2 * s23.re, a[0].im + c1 * s14.im + c2 * s23.im);
    Cx<T> m2 = mk<T>(a[0].re + c2 * s14.re + c1 * s23.re, a[0].im + c2 * s14.im + c1 * s23.im);
    // -i * (s1*d14 + s2*d23) and -i * (s2*d14 - s1*d23)
    Cx<T> r1 = mk<T>(s1 * d14.im + s2 * d23.im, -(s1 * d14.re + s2 * d23.re));
    Cx<T> r2 = mk<T>(s2 * d14.im - s1 * d23.im, -(s2 * d14.re - s1 * d23.re));
    a[0] = a[0] + s14 + s23;
    a[1] = m1 + r1;
    a[4] = m1 - r1;
    a[2] = m2 + r2;
    a[3] = m2 - r2;
  }
};

// Composite radix R = R1 * R2 (Cooley-Tukey in registers):
//   X[k1 + R1*k2] = sum_{n2} W_R^{n2*k1} ( sum_{n1} x[n1*R2 + n2] W_R1^{n1*k1} ) W_R2^{n2*k2}
template <int R>
struct DftSplit {
  static constexpr int R1 = (R % 4 == 0) ? 4 : (R % 2 == 0) ? 2 : (R % 3 == 0) ? 3 : (R % 5 == 0) ? 5 : R;
  static constexpr int R2 = R / R1;
  static_assert(R1 != R || R <= 5, "unsupported prime radix");
};

template <class T, int R, int N2>
struct DftTwRow {  // y[k1] *= W_R^{N2*k1} for k1 = 0..R1-1
  template <int K1>
  XM_DEV static void apply(Cx<T>* y) {
    if constexpr (K1 < DftSplit<R>::R1) {
      y[K1] = mul_w<N2 * K1, R, T>(y[K1]);
      apply<K1 + 1>(y);
    }
  }
};

template <class T, int R>
struct Dft {
  static constexpr int R1 = DftSplit<R>::R1;
  static constexpr int R2 = DftSplit<R>::R2;

  template <int N2>
  XM_DEV static void cols(const Cx<T>* a, Cx<T> (*y)[R2]) {  // step 1 + twiddle, column n2
    if constexpr (N2 < R2) {
      Cx<T> c[R1];
#pragma unroll
      for (int n1 = 0; n1 < R1; ++n1) c[n1] = a[n1 * R2 + N2];
      Dft<T, R1>::run(c);
      DftTwRow<T, R, N2>::template apply<0>(c);
#pragma unroll
      for (int k1 = 0; k1 < R1; ++k1) y[k1][N2] = c[k1];
      cols<N2 + 1>(a, y);
    }
  }

  XM_DEV static void run(Cx<T>* a) {
    Cx<T> y[R1][R2];
    cols<0>(a, y);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) {
      Dft<T, R2>::run(y[k1]);
#pragma unroll
      for (int k2 = 0; k2 < R2; ++k2) a[k1 + R1 * k2] = y[k1][k2];
    }
  }
};
