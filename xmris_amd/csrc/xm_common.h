// xmris_amd -- common device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Wave = 64 lanes, LDS = 160 KiB/CU.  No CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#define XM_WAVE 64
#define XM_DEV __device__ __forceinline__

// Two-lane vector of a real type.  Arithmetic on it maps to the packed-f32 VALU instructions
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) on gfx950, which do two lanes for the issue cost of
// one: the hot kernel carries (even-bin FFT, odd-bin FFT) in the two lanes.
typedef float xm_f2 __attribute__((ext_vector_type(2)));
typedef double xm_d2 __attribute__((ext_vector_type(2)));

template <class T>
struct ScalarOf {
  using type = T;
};
template <>
struct ScalarOf<xm_f2> {
  using type = float;
};
template <>
struct ScalarOf<xm_d2> {
  using type = double;
};
template <class S>
struct PairOf;
template <>
struct PairOf<float> {
  using type = xm_f2;
};
template <>
struct PairOf<double> {
  using type = xm_d2;
};

// Complex value.  T is a real scalar (storage precision) or a two-lane vector of it.
// alignas(2*sizeof(T)) so that a c64 moves as one dwordx2 and a c128 / pair of c64 as one dwordx4.
template <class T>
struct alignas(2 * sizeof(T)) Cx {
  T re, im;
};

template <class T>
XM_DEV Cx<T> mk(T a, T b) {
  Cx<T> r;
  r.re = a;
  r.im = b;
  return r;
}
template <class T>
XM_DEV Cx<T> operator+(Cx<T> a, Cx<T> b) {
  return mk<T>(a.re + b.re, a.im + b.im);
}
template <class T>
XM_DEV Cx<T> operator-(Cx<T> a, Cx<T> b) {
  return mk<T>(a.re - b.re, a.im - b.im);
}
template <class T>
XM_DEV Cx<T> operator*(Cx<T> a, Cx<T> b) {
  return mk<T>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
template <class T>
XM_DEV Cx<T> operator*(Cx<T> a, T s) {
  return mk<T>(a.re * s, a.im * s);
}
// two-lane complex times a scalar complex / scalar real (the scalar is broadcast to both lanes)
template <class V, class S, class = typename std::enable_if<!std::is_same<V, S>::value &&
                                                            std::is_same<typename ScalarOf<V>::type, S>::value>::type>
XM_DEV Cx<V> operator*(Cx<V> a, Cx<S> b) {
  return mk<V>(a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re);
}
template <class V, class S, class = typename std::enable_if<!std::is_same<V, S>::value &&
                                                            std::is_same<typename ScalarOf<V>::type, S>::value>::type>
XM_DEV Cx<V> operator*(Cx<V> a, S s) {
  return mk<V>(a.re * s, a.im * s);
}
template <class T>
XM_DEV Cx<T> conj(Cx<T> a) {
  return mk<T>(a.re, -a.im);
}
// a * (-i)  and  a * (+i)
template <class T>
XM_DEV Cx<T> mul_mi(Cx<T> a) {
  return mk<T>(a.im, -a.re);
}
template <class T>
XM_DEV Cx<T> mul_pi(Cx<T> a) {
  return mk<T>(-a.im, a.re);
}

// ---- compile-time cos/sin of 2*pi*k/R (exact to double rounding) ---------------------------
namespace xmct {
constexpr double kPi = 3.14159265358979323846264338327950288;
constexpr double tcos(double x) {  // |x| <= pi/4
  double x2 = x * x, term = 1.0, sum = 1.0;
  for (int i = 1; i <= 12; ++i) {
    term *= -x2 / double((2 * i - 1) * (2 * i));
    sum += term;
  }
  return sum;
}
constexpr double tsin(double x) {  // |x| <= pi/4
  double x2 = x * x, term = x, sum = x;
  for (int i = 1; i <= 12; ++i) {
    term *= -x2 / double((2 * i) * (2 * i + 1));
    sum += term;
  }
  return sum;
}
// cos(2*pi*k/R), sin(2*pi*k/R) with octant reduction on the INTEGER fraction (no rounding
// in the argument reduction).
constexpr double cos2pi(long k, long R) {
  k %= R;
  if (k < 0) k += R;
  // use symmetry: angle = 2*pi*k/R in [0, 2pi)
  if (2 * k > R) return cos2pi(R - k, R);            // cos(2pi - a) = cos a
  if (4 * k > R) return -cos2pi(R - 2 * k, 2 * R) ;  // a in (pi/2, pi]: cos a = -cos(pi - a); pi - a = 2pi (R-2k)/(2R)
  if (8 * k > R) {                                   // a in (pi/4, pi/2]: cos a = sin(pi/2 - a); pi/2 - a = 2pi (R-4k)/(4R)
    return tsin(2.0 * kPi * double(R - 4 * k) / double(4 * R));
  }
  return tcos(2.0 * kPi * double(k) / double(R));
}
constexpr double sin2pi(long k, long R) {
  k %= R;
  if (k < 0) k += R;
  if (2 * k > R) return -sin2pi(R - k, R);
  if (4 * k > R) return sin2pi(R - 2 * k, 2 * R);
  if (8 * k > R) return tcos(2.0 * kPi * double(R - 4 * k) / double(4 * R));
  return tsin(2.0 * kPi * double(k) / double(R));
}
}  // namespace xmct

// a * W_R^K  with  W_R = exp(-2*pi*i/R); trivial rotations cost no multiplies.
template <int K_, int R, class T>
XM_DEV Cx<T> mul_w(Cx<T> a) {
  using S = typename ScalarOf<T>::type;
  constexpr int K = ((K_ % R) + R) % R;
  if constexpr (K == 0) {
    return a;
  } else if constexpr (4 * K == R) {
    return mul_mi(a);
  } else if constexpr (2 * K == R) {
    return mk<T>(-a.re, -a.im);
  } else if constexpr (4 * K == 3 * R) {
    return mul_pi(a);
  } else if constexpr (8 * K == R) {  // (1 - i)/sqrt2
    constexpr S c = S(0.70710678118654752440);
    return mk<T>((a.re + a.im) * c, (a.im - a.re) * c);
  } else if constexpr (8 * K == 3 * R) {  // (-1 - i)/sqrt2
    constexpr S c = S(0.70710678118654752440);
    return mk<T>((a.im - a.re) * c, -(a.re + a.im) * c);
  } else if constexpr (8 * K == 5 * R) {  // (-1 + i)/sqrt2
    constexpr S c = S(0.70710678118654752440);
    return mk<T>(-(a.re + a.im) * c, (a.re - a.im) * c);
  } else if constexpr (8 * K == 7 * R) {  // (1 + i)/sqrt2
    constexpr S c = S(0.70710678118654752440);
    return mk<T>((a.re - a.im) * c, (a.re + a.im) * c);
  } else {
    constexpr S c = S(xmct::cos2pi(K, R));
    constexpr S s = S(-xmct::sin2pi(K, R));  // W = cos - i sin
    return mk<T>(a.re * c - a.im * s, a.re * s + a.im * c);
  }
}
