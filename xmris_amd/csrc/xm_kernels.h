// Device kernels of the xmris spectral hot path for gfx950.
//   k_pipe      : generic fused  zero-fill + window + FFT(+shift) + |X|^2 arg-max + phase
//   k_pipe_zf2  : same, specialised for n_out >= 2*(pad_left+n_in) ("end" zero-fill to >= 2x):
//                 the upper half of the input is zero, so X[2m] = FFT_H(z)[m] and
//                 X[2m+1] = FFT_H(z * W_N^k)[m] -- two half-length FFTs, first radix-2 level free,
//                 and every thread stores adjacent (even, odd) outputs as one 16-byte word.
//   k_bluestein : arbitrary length through a power-of-two chirp-z convolution, same fused I/O.
//   elementwise : zero_fill, apodize, phase, roll, per-row arg-max, final arg-max.
#pragma once
#include "xm_blockfft.h"

template <class T>
struct PipeArgs {
  const Cx<T>* in;
  Cx<T>* out;
  const T* window;        // n_out reals, indexed by padded position (before SHIFT_IN roll)
  const Cx<T>* phase;     // n_out complex, indexed by output position (after SHIFT_OUT roll)
  const Cx<T>* tw;        // stage twiddles of the plan
  const Cx<T>* aux;       // zf2: W_N^k (k < H);  bluestein: chirp a[k] = e^{-i pi k^2/n}
  const Cx<T>* aux2;      // bluestein: FFT_M(b) / M
  T* absmax2;
  int32_t* argidx;
  long long in_stride;
  long long n_batch;
  int n;                  // transform length (n_out)
  int n_in, pad_left, in_shift, out_shift;
  int inverse;
  T scale;
};

// ---- (value, index) arg-max helpers: larger value wins, ties -> smaller index -------------------
template <class T>
XM_DEV void amax_take(T& bv, int& bi, T v, int i) {
  if (v > bv || (v == bv && i < bi)) {
    bv = v;
    bi = i;
  }
}

template <class T>
XM_DEV T shfl_xor_t(T v, int m) {
  return __shfl_xor(v, m, XM_WAVE);
}

// Reduce (bv, bi) over the NT threads that hold one spectrum and let its thread t == 0 write the
// result.  NT is a power of two; `red` is LDS scratch for the NT > 64 case (>= blockDim/64 entries
// of each kind); the function contains workgroup barriers when NT > 64.
template <class T, int NT>
XM_DEV void amax_reduce_store(T bv, int bi, int t, bool live, long long s, T* absmax2, int32_t* argidx,
                              T* red_v, int* red_i) {
  constexpr int W = NT < XM_WAVE ? NT : XM_WAVE;
#pragma unroll
  for (int m = W / 2; m >= 1; m >>= 1) {
    T ov = shfl_xor_t(bv, m);
    int oi = __shfl_xor(bi, m, XM_WAVE);
    amax_take(bv, bi, ov, oi);
  }
  if constexpr (NT <= XM_WAVE) {
    if (t == 0 && live) {
      absmax2[s] = bv;
      argidx[s] = bi;
    }
  } else {
    constexpr int NW = NT / XM_WAVE;  // waves per spectrum
    const int wave = threadIdx.x / XM_WAVE;
    __syncthreads();  // the exchange buffer is free again
    if ((threadIdx.x & (XM_WAVE - 1)) == 0) {
      red_v[wave] = bv;
      red_i[wave] = bi;
    }
    __syncthreads();
    if (t == 0 && live) {
      const int w0 = wave;  // first wave of this spectrum
#pragma unroll
      for (int w = 1; w < NW; ++w) amax_take(bv, bi, red_v[w0 + w], red_i[w0 + w]);
      absmax2[s] = bv;
      argidx[s] = bi;
    }
  }
}

// =================================================================================================
// Generic fused kernel.  blockDim = NT * SPB, one spectrum per NT threads.
// =================================================================================================
template <class T, class PL, int SPB>
__global__ __launch_bounds__(PL::NT* SPB) void k_pipe(PipeArgs<T> A) {
  constexpr int N = PL::N, NT = PL::NT, P = PL::P;
  extern __shared__ __attribute__((aligned(16))) char xm_smem[];
  const int t = threadIdx.x % NT;
  const int ls = threadIdx.x / NT;
  const long long s = (long long)blockIdx.x * SPB + ls;
  const bool live = s < A.n_batch;
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(xm_smem) + (size_t)ls * PL::lds_elems();

  Cx<T> v[P];
  const Cx<T>* __restrict__ row = A.in + (live ? s : 0) * A.in_stride;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int pos = t + NT * q;
    int j = pos - A.in_shift;
    if (j < 0) j += N;
    const int src = j - A.pad_left;
    Cx<T> x = mk<T>(T(0), T(0));
    if (live && src >= 0 && src < A.n_in) {
      x = row[src];
      if (A.window) x = x * A.window[j];
      if (A.inverse) x = conj(x);
    }
    v[q] = x;
  }

  BlockFFT<T, PL>::run(v, lds, A.tw, t);

  T bv = T(-1);
  int bi = 0x7fffffff;
  Cx<T>* __restrict__ orow = A.out ? A.out + (live ? s : 0) * (long long)N : nullptr;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int m = t + NT * q;
    int k = m + A.out_shift;
    if (k >= N) k -= N;
    Cx<T> x = v[q] * A.scale;
    if (A.inverse) x = conj(x);
    if (A.absmax2) amax_take(bv, bi, x.re * x.re + x.im * x.im, k);
    if (orow) {
      if (A.phase) x = x * A.phase[k];
      if (live) orow[k] = x;
    }
  }
  if (A.absmax2) {
    T* red_v = reinterpret_cast<T*>(xm_smem);
    int* red_i = reinterpret_cast<int*>(red_v + (NT * SPB) / XM_WAVE + 1);
    amax_reduce_store<T, NT>(bv, bi, t, live, s, A.absmax2, A.argidx, red_v, red_i);
  }
}

// =================================================================================================
// Zero-fill-by->=2 fused kernel: N = 2*H, PL is the plan of the HALF length H.
// =================================================================================================
template <class T>
struct alignas(4 * sizeof(T)) CxPair {
  Cx<T> a, b;
};

template <class T, class PL, int SPB>
__global__ __launch_bounds__(PL::NT* SPB) void k_pipe_zf2(PipeArgs<T> A) {
  constexpr int H = PL::N, N = 2 * PL::N, NT = PL::NT, P = PL::P;
  extern __shared__ __attribute__((aligned(16))) char xm_smem[];
  const int t = threadIdx.x % NT;
  const int ls = threadIdx.x / NT;
  const long long s = (long long)blockIdx.x * SPB + ls;
  const bool live = s < A.n_batch;
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(xm_smem) + (size_t)ls * PL::lds_elems();

  Cx<T> ve[P], vo[P];
  const Cx<T>* __restrict__ row = A.in + (live ? s : 0) * A.in_stride;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int j = t + NT * q;  // < H
    const int src = j - A.pad_left;
    Cx<T> x = mk<T>(T(0), T(0));
    if (live && src >= 0 && src < A.n_in) {
      x = row[src];
      if (A.window) x = x * A.window[j];
    }
    ve[q] = x;
    vo[q] = x * A.aux[j];  // W_N^j
  }

  BlockFFT<T, PL>::run(ve, lds, A.tw, t);
  BlockFFT<T, PL>::run(vo, lds, A.tw, t);

  T bv = T(-1);
  int bi = 0x7fffffff;
  Cx<T>* __restrict__ orow = A.out ? A.out + (live ? s : 0) * (long long)N : nullptr;
  const bool paired = (A.out_shift & 1) == 0;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int m = t + NT * q;
    int k0 = 2 * m + A.out_shift;
    if (k0 >= N) k0 -= N;
    int k1 = k0 + 1;
    if (k1 >= N) k1 -= N;
    Cx<T> xe = ve[q] * A.scale;
    Cx<T> xo = vo[q] * A.scale;
    if (A.absmax2) {
      amax_take(bv, bi, xe.re * xe.re + xe.im * xe.im, k0);
      amax_take(bv, bi, xo.re * xo.re + xo.im * xo.im, k1);
    }
    if (orow) {
      if (paired) {  // k0 even, k1 = k0 + 1: one 16-byte (c64) word per thread, fully coalesced
        if (A.phase) {
          const CxPair<T> ph = *reinterpret_cast<const CxPair<T>*>(A.phase + k0);
          xe = xe * ph.a;
          xo = xo * ph.b;
        }
        CxPair<T> o;
        o.a = xe;
        o.b = xo;
        if (live) *reinterpret_cast<CxPair<T>*>(orow + k0) = o;
      } else {
        if (A.phase) {
          xe = xe * A.phase[k0];
          xo = xo * A.phase[k1];
        }
        if (live) {
          orow[k0] = xe;
          orow[k1] = xo;
        }
      }
    }
  }
  if (A.absmax2) {
    T* red_v = reinterpret_cast<T*>(xm_smem);
    int* red_i = reinterpret_cast<int*>(red_v + (NT * SPB) / XM_WAVE + 1);
    amax_reduce_store<T, NT>(bv, bi, t, live, s, A.absmax2, A.argidx, red_v, red_i);
  }
}

// =================================================================================================
// Bluestein (chirp-z) kernel for lengths without a direct plan.  PL = power-of-two plan of
// length M >= 2n-1.   X[m] = a[m] * sum_k (z[k] a[k]) b[m-k],  a[k] = e^{-i pi k^2/n}, b = conj(a).
// aux = a (n entries), aux2 = FFT_M(b wrapped) / M (M entries), both fp64-computed on the host.
// =================================================================================================
template <class T, class PL, int SPB>
__global__ __launch_bounds__(PL::NT* SPB) void k_bluestein(PipeArgs<T> A) {
  constexpr int M = PL::N, NT = PL::NT, P = PL::P;
  extern __shared__ __attribute__((aligned(16))) char xm_smem[];
  const int t = threadIdx.x % NT;
  const int ls = threadIdx.x / NT;
  const long long s = (long long)blockIdx.x * SPB + ls;
  const bool live = s < A.n_batch;
  const int n = A.n;
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(xm_smem) + (size_t)ls * PL::lds_elems();

  Cx<T> v[P];
  const Cx<T>* __restrict__ row = A.in + (live ? s : 0) * A.in_stride;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int pos = t + NT * q;
    Cx<T> x = mk<T>(T(0), T(0));
    if (pos < n) {
      int j = pos - A.in_shift;
      if (j < 0) j += n;
      const int src = j - A.pad_left;
      if (live && src >= 0 && src < A.n_in) {
        x = row[src];
        if (A.window) x = x * A.window[j];
        if (A.inverse) x = conj(x);
        x = x * A.aux[pos];
      }
    }
    v[q] = x;
  }
  BlockFFT<T, PL>::run(v, lds, A.tw, t);
#pragma unroll
  for (int q = 0; q < P; ++q) v[q] = conj(v[q] * A.aux2[t + NT * q]);  // conj -> inverse via forward
  if constexpr (PL::K > 1) __syncthreads();
  BlockFFT<T, PL>::run(v, lds, A.tw, t);

  T bv = T(-1);
  int bi = 0x7fffffff;
  Cx<T>* __restrict__ orow = A.out ? A.out + (live ? s : 0) * (long long)n : nullptr;
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const int m = t + NT * q;
    if (m < n) {
      int k = m + A.out_shift;
      if (k >= n) k -= n;
      Cx<T> x = (conj(v[q]) * A.aux[m]) * A.scale;
      if (A.inverse) x = conj(x);
      if (A.absmax2) amax_take(bv, bi, x.re * x.re + x.im * x.im, k);
      if (orow) {
        if (A.phase) x = x * A.phase[k];
        if (live) orow[k] = x;
      }
    }
  }
  if (A.absmax2) {
    T* red_v = reinterpret_cast<T*>(xm_smem);
    int* red_i = reinterpret_cast<int*>(red_v + (NT * SPB) / XM_WAVE + 1);
    amax_reduce_store<T, NT>(bv, bi, t, live, s, A.absmax2, A.argidx, red_v, red_i);
  }
}

// =================================================================================================
// Element-wise staged kernels (one complex sample per thread per step, grid-stride).
// =================================================================================================
template <class T>
__global__ void k_zero_fill(const Cx<T>* __restrict__ in, Cx<T>* __restrict__ out, long long n_batch,
                            int n_in, int n_out, int pad_left) {
  const long long total = n_batch * (long long)n_out;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / n_out;
    const int j = (int)(i - b * n_out);
    const int src = j - pad_left;
    Cx<T> x = mk<T>(T(0), T(0));
    if (src >= 0 && src < n_in) x = in[b * n_in + src];
    out[i] = x;
  }
}

template <class T>
__global__ void k_apodize(const Cx<T>* in, Cx<T>* out, const T* __restrict__ w, long long n_batch, int n) {
  const long long total = n_batch * (long long)n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % n);
    out[i] = in[i] * w[j];
  }
}

template <class T>
__global__ void k_phase(const Cx<T>* in, Cx<T>* out, const Cx<T>* __restrict__ ph, long long n_batch, int n) {
  const long long total = n_batch * (long long)n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % n);
    out[i] = in[i] * ph[j];
  }
}

template <class T>
__global__ void k_roll(const Cx<T>* __restrict__ in, Cx<T>* __restrict__ out, long long n_batch, int n,
                       int shift) {
  const long long total = n_batch * (long long)n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long b = i / n;
    int j = (int)(i - b * n) + shift;
    if (j >= n) j -= n;
    out[b * n + j] = in[i];
  }
}

// one 256-thread workgroup per spectrum
template <class T>
__global__ __launch_bounds__(256) void k_absmax_rows(const Cx<T>* __restrict__ in, long long n_batch, int n,
                                                      T* absmax2, int32_t* argidx) {
  __shared__ T red_v[4];
  __shared__ int red_i[4];
  for (long long b = blockIdx.x; b < n_batch; b += gridDim.x) {
    const Cx<T>* row = in + b * n;
    T bv = T(-1);
    int bi = 0x7fffffff;
    for (int j = threadIdx.x; j < n; j += 256) {
      const Cx<T> x = row[j];
      amax_take(bv, bi, x.re * x.re + x.im * x.im, j);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      T ov = shfl_xor_t(bv, m);
      int oi = __shfl_xor(bi, m, XM_WAVE);
      amax_take(bv, bi, ov, oi);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
      red_v[threadIdx.x >> 6] = bv;
      red_i[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) amax_take(bv, bi, red_v[w], red_i[w]);
      absmax2[b] = bv;
      argidx[b] = bi;
    }
  }
}

// single workgroup: global (max, first flat index) over the per-spectrum pairs
template <class T>
__global__ __launch_bounds__(1024) void k_argmax_final(const T* __restrict__ absmax2,
                                                        const int32_t* __restrict__ argidx, long long n_batch,
                                                        int n, T* out_max2, long long* out_flat) {
  __shared__ T red_v[16];
  __shared__ long long red_i[16];
  T bv = T(-1);
  long long bi = 0x7fffffffffffffffLL;
  for (long long b = threadIdx.x; b < n_batch; b += 1024) {
    const T v = absmax2[b];
    const long long f = b * (long long)n + argidx[b];
    if (v > bv || (v == bv && f < bi)) {
      bv = v;
      bi = f;
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    T ov = shfl_xor_t(bv, m);
    long long oi = __shfl_xor(bi, m, XM_WAVE);
    if (ov > bv || (ov == bv && oi < bi)) {
      bv = ov;
      bi = oi;
    }
  }
  if ((threadIdx.x & 63) == 0) {
    red_v[threadIdx.x >> 6] = bv;
    red_i[threadIdx.x >> 6] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) {
      if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) {
        bv = red_v[w];
        bi = red_i[w];
      }
    }
    out_max2[0] = bv;
    out_flat[0] = bi;
  }
}
