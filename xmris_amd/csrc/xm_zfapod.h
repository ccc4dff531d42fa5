// k_zf_apod -- zero fill + apodisation in ONE pass over the FIDs (reference processing/fid.py:251 `da.pad(...)`
// followed by fid.py:136-139 `da * exp(-pi lb t)`): out[b, j] = in[b, j - pad_left] * window[j] inside the acquired
// samples, 0 elsewhere.  The chain `apodize_*(zero_fill(fid))` stopped before the FFT used to be two launches with a
// full-size intermediate (32 + 64 + 64 + 64 KiB per 4096 -> 8192 complex64 row instead of 32 + 64).
//
// A pure streaming kernel, shaped like what `tools/stream_lab.hip` measured as this chip's best for a 1 : 2 read : write
// row pattern: persistent 256-thread workgroups, 16-byte loads and 16-byte NONTEMPORAL stores, a row per ticket from a
// device-scope counter (static splits lose ~10 % to the workgroups' unequal speeds), the next ticket claimed while the
// current row's loads are in flight.  The window is staged through the LDS once per workgroup -- coalesced 16-byte
// reads -- and, where a thread's columns fit (n_out <= 64 x 256 elements... 32 weights per thread), kept in registers
// for the whole launch.  TI / TO: storage precision of the input / output (complex64 rows times a float64 window give
// complex128, numpy's promotion, fid.py:136-139).  gfx950 only.
#pragma once
#include "xm_common.h"

typedef float xm_f4 __attribute__((ext_vector_type(4)));

template <class TI, class TO>
struct ZfApodArgs {
  const Cx<TI>* in;
  Cx<TO>* out;
  const TO* window;  // n_out weights of the OUTPUT precision
  unsigned* queue;   // {head, done}, both 0 at launch; the last workgroup out leaves them 0
  long long in_stride, n_batch;
  int n_in, n_out, pad_left;
};

template <class TI, int EPL>
XM_DEV void load_vec(const Cx<TI>* p, Cx<TI> (&v)[EPL]) {  // 16 bytes, nontemporal: the FIDs are read once
  if constexpr (sizeof(Cx<TI>) == 8) {
    const xm_f4 q = __builtin_nontemporal_load(reinterpret_cast<const xm_f4*>(p));
    v[0] = mk<TI>(q.x, q.y);
    v[1] = mk<TI>(q.z, q.w);
  } else {
    const xm_d2 q = __builtin_nontemporal_load(reinterpret_cast<const xm_d2*>(p));
    v[0] = mk<TI>(q.x, q.y);
  }
}
template <class TO, int EPL>
XM_DEV void store_vec(Cx<TO>* p, const Cx<TO> (&y)[EPL]) {  // 16-byte nontemporal stores
  if constexpr (sizeof(Cx<TO>) == 8) {
    static_assert(EPL == 2, "two complex64 per 16 bytes");
    xm_f4 q;
    q.x = (float)y[0].re;
    q.y = (float)y[0].im;
    q.z = (float)y[1].re;
    q.w = (float)y[1].im;
    __builtin_nontemporal_store(q, reinterpret_cast<xm_f4*>(p));
  } else {
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      xm_d2 q;
      q.x = (double)y[e].re;
      q.y = (double)y[e].im;
      __builtin_nontemporal_store(q, reinterpret_cast<xm_d2*>(p + e));
    }
  }
}

// VEC: every row of the input and the output is 16-byte aligned and pad_left / n_in are even for 8-byte elements, so a
// lane moves 16 bytes of input per step.  Otherwise: one element per lane and step (any geometry).
template <class TI, class TO, bool VEC>
__global__ __launch_bounds__(256) void k_zf_apod(ZfApodArgs<TI, TO> A) {
  constexpr int NT = 256;
  constexpr int EPL = VEC ? (int)(16 / sizeof(Cx<TI>)) : 1;  // input elements per lane and step (2 x c64, 1 x c128)
  extern __shared__ __attribute__((aligned(16))) char xm_smem[];
  TO* wl = reinterpret_cast<TO*>(xm_smem);  // the window over the ACQUIRED samples only (n_in weights from pad_left on:
                                            // the padding needs none) -- 8192 doubles would allow two workgroups per CU
  __shared__ long long ticket[2];
  const int t = (int)threadIdx.x;
  for (int j = t; j < A.n_in && A.pad_left + j < A.n_out; j += NT) wl[j] = A.window[A.pad_left + j];
  long long row = blockIdx.x;  // the first round is static, then tickets
  if (t == 0) ticket[0] = (long long)gridDim.x + (long long)atomicAdd(A.queue, 1u);
  __syncthreads();
  const int per_step = NT * EPL;
  for (int it = 0; row < A.n_batch; ++it) {
    // the row after this one was claimed an iteration ago; the one after that is claimed now, while this row's
    // loads are in flight (two slots: this iteration's readers and its writer never meet; one barrier per row)
    const long long nxt = ticket[it & 1];
    if (t == 0) ticket[(it + 1) & 1] = (long long)gridDim.x + (long long)atomicAdd(A.queue, 1u);
    const Cx<TI>* __restrict__ irow = A.in + row * A.in_stride;
    Cx<TO>* __restrict__ orow = A.out + row * (long long)A.n_out;
    auto one_step = [&](int j0) {
      const int k = j0 - A.pad_left;  // input index of the first element
      Cx<TO> y[EPL];
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        const int ke = k + e;
        const bool in_range = ke >= 0 && ke < A.n_in && j0 + e < A.n_out;
        const Cx<TI> v = in_range ? irow[ke] : mk<TI>(TI(0), TI(0));
        const TO w = in_range ? wl[ke] : TO(0);
        y[e] = in_range ? mk<TO>((TO)v.re * w, (TO)v.im * w) : mk<TO>(TO(0), TO(0));  // the padding is +0
      }
      if constexpr (VEC) {
        store_vec(orow + j0, y);
      } else {
        if (j0 < A.n_out) orow[j0] = y[0];
      }
    };
    int j0 = t * EPL;
    if constexpr (VEC) {
      // one step at a time: the steps that lie inside the acquired samples (a 16-byte load, the products, the store)
      // or inside the padding (a store) take no per-element tests.  (Four steps per iteration with four loads in
      // flight were SLOWER -- 5.0 instead of 5.4 TB/s, complex64 -> complex128 2.8 instead of 4.3: eight to ten
      // workgroups per CU already overlap one another's loads and stores, the unrolled body halves that.)
      constexpr int U = 1;
      for (; j0 + (U - 1) * per_step + EPL <= A.n_out; j0 += U * per_step) {
        const int k = j0 - A.pad_left;
        if (k >= 0 && k + (U - 1) * per_step + EPL <= A.n_in) {
          Cx<TI> v[U][EPL];
#pragma unroll
          for (int u = 0; u < U; ++u) load_vec(irow + k + u * per_step, v[u]);
#pragma unroll
          for (int u = 0; u < U; ++u) {
            Cx<TO> y[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
              const TO w = wl[k + u * per_step + e];
              y[e] = mk<TO>((TO)v[u][e].re * w, (TO)v[u][e].im * w);  // complex times real: two products, as numpy's
            }
            store_vec(orow + j0 + u * per_step, y);
          }
        } else if (k >= A.n_in || k + (U - 1) * per_step + EPL <= 0) {
          Cx<TO> z[EPL];
#pragma unroll
          for (int e = 0; e < EPL; ++e) z[e] = mk<TO>(TO(0), TO(0));
#pragma unroll
          for (int u = 0; u < U; ++u) store_vec(orow + j0 + u * per_step, z);
        } else {
#pragma unroll
          for (int u = 0; u < U; ++u) one_step(j0 + u * per_step);
        }
      }
    }
    for (; j0 < A.n_out; j0 += per_step) one_step(j0);
    __syncthreads();
    row = nxt;
  }
  if (t == 0) {  // the last workgroup out leaves the counters zero for the next launch
    __threadfence();
    if (atomicAdd(A.queue + 1, 1u) == gridDim.x - 1) {
      A.queue[0] = 0u;
      A.queue[1] = 0u;
    }
  }
}
