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

#include <type_traits>

// result record of a global arg-max key (xmris_hip.h: xm_argmax_result)
struct XmKeyResult {
  float max2;
  float pad_;
  long long flat;
};

#define XM_KEY_SLOTS 64
#define XM_KEY_C128_WORD 8192  // 64-bit word of a key buffer where the complex128 kernels keep their per-wave (value, row) slots
#define XM_KEY_GBEST_WORD 2048  // 32-bit word of a key buffer behind the 64 partial keys: xm_guess_refine's 64 partial bounds
#define XM_KEY_STRIDE 16  // 64-bit words between partial keys (128 bytes)

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
  int amax_value_only;    // skip the first-index scan (argidx written as 0)
  T scale;
  // Linear output phase e^{i (a + b k)} (k = output index after the roll) in factorised form, for kernels whose
  // output index is (wave-uniform base_q) + 2t + (0 | 1): ramp_c[2q], ramp_c[2q+1] = e^{i (a + b base_q)} (re, im),
  // ramp_e = e^{i b} (fp64 on the host, rounded once), ramp_db = b; the per-thread factor e^{i b 2t} is computed
  // in fp64 by the kernel once per launch (kernels instantiated with ZF2_RAMP; the table pointer is unused then).
  unsigned* queue;        // persistent kernels with dynamic row hand-out: {head, done} counters, both 0 at launch
  int queue_chunk;        // ... rows per ticket (>= 1)
  // XM_AMAX_GLOBAL_KEY (complex64, value-only maxima): instead of per-row outputs every wave keeps the best
  // (max |X|^2, row) of the rows it transforms and merges it into this ONE 64-bit key with a single atomic max when
  // it leaves: key = float bits << 32 | (0xffffffff - row), i.e. larger value first, then the lower row.  The key is
  // kept in XM_KEY_SLOTS partial keys on cache lines of their own (slot = workgroup mod XM_KEY_SLOTS; thousands of
  // waves leave a kernel within microseconds and one address takes ~90 atomics per microsecond), all zero at launch;
  // k_key_take merges and clears them.
  unsigned long long* gkey;
  XmKeyResult* key_result;  // optional (kernels with a row queue): the last workgroup decodes + clears the key itself
  double ramp_db;
  T ramp_e[2];
  T ramp_c[32];
  // Guess stage of the speculative schedule (xm_guess_rows / xm_guess_refine; kernels in xm_zf2p.h, always float):
  float* est;                         // per-row coarse estimate of max |X|^2: written by the guess pass, read by the refine pass
  unsigned long long* gkey_in;        // refine pass: the guess pass's key -- its maximum sets the candidate threshold
  float band2;                        // ... rows with est >= band2 * max(est) are transformed exactly
  unsigned* gbest;                    // ... float bits of the best exact max |X|^2 so far: 64 partial bounds, 128 B apart (0 at launch, left 0)
  float* take_max2;                   // refine pass, last workgroup out: max |X|^2 of the winning candidate,
  long long* take_flat;               // ... its row * n, and
  Cx<double>* take_row;               // ... its n_in samples as complex128 (the search's input is recomputed in fp64)
};

// ---- (value, index) arg-max helpers: larger value wins, ties -> smaller index; a NaN outranks every number
// (np.argmax returns the first NaN, phasing.py:229), NaN against NaN -> smaller index ------------------------
template <class T>
XM_DEV void amax_take(T& bv, int& bi, T v, int i) {  // branch-free
  const bool vn = v != v, bn = bv != bv;
  const bool take = (v > bv) | ((v == bv) & (i < bi)) | (vn & !bn) | (vn & bn & (i < bi));
  bv = take ? v : bv;
  bi = take ? i : bi;
}
// fmax drops NaNs: a thread-local maximum that is still at its start value -1 saw nothing but NaNs
template <class T>
XM_DEV T amax_nan_if_unset(T bv) {
  return bv < T(0) ? T(__builtin_nan("")) : bv;
}

template <class T>
XM_DEV T shfl_xor_t(T v, int m) {
  return __shfl_xor(v, m, XM_WAVE);
}

// DPP controls (gfx9): within a 16-lane row, then across rows
#define XM_DPP_QUAD_XOR1 0xB1     // quad_perm:[1,0,3,2]
#define XM_DPP_QUAD_XOR2 0x4E     // quad_perm:[2,3,0,1]
#define XM_DPP_HALF_MIRROR 0x141  // row_half_mirror
#define XM_DPP_ROW_MIRROR 0x140   // row_mirror
#define XM_DPP_BCAST15 0x142      // row_bcast:15
#define XM_DPP_BCAST31 0x143      // row_bcast:31

// Wave-wide (64 lanes) reduction of a 32-bit unsigned key with v_max_u32 / v_min_u32 through DPP: six
// VALU instructions and no LDS round trips (a __shfl_xor butterfly costs six ds_bpermute latencies).
// Non-negative floats order like their bit patterns, so |X|^2 is reduced as an unsigned key.
template <bool MAX>
XM_DEV unsigned wave_reduce_u32(unsigned v) {
  auto op = [](unsigned a, unsigned b) { return MAX ? (a > b ? a : b) : (a < b ? a : b); };
  const unsigned ident = MAX ? 0u : 0xffffffffu;
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)ident, (int)v, XM_DPP_QUAD_XOR1, 0xf, 0xf, false));
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)ident, (int)v, XM_DPP_QUAD_XOR2, 0xf, 0xf, false));
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)ident, (int)v, XM_DPP_HALF_MIRROR, 0xf, 0xf, false));
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)ident, (int)v, XM_DPP_ROW_MIRROR, 0xf, 0xf, false));
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)ident, (int)v, XM_DPP_BCAST15, 0xa, 0xf, false));
  v = op(v, (unsigned)__builtin_amdgcn_update_dpp((int)ident, (int)v, XM_DPP_BCAST31, 0xc, 0xf, false));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// wave-wide maximum of a 64-bit unsigned key (the bit pattern of a non-negative double, or of a NaN, which outranks
// every number): the high words first, then the low words of the lanes that hold the largest high word
XM_DEV unsigned long long wave_reduce_u64_max(unsigned long long v) {
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned mh = wave_reduce_u32<true>(hi);
  const unsigned ml = wave_reduce_u32<true>(hi == mh ? lo : 0u);
  return ((unsigned long long)mh << 32) | ml;
}

// (max value, first index) over the 64 lanes of a wave; result uniform across the wave
XM_DEV void wave_amax(float& bv, int& bi) {
  // bv >= 0 (a squared magnitude), -1 for "nothing", or a (positive quiet) NaN, whose bit pattern outranks every number
  const unsigned key = bv != bv ? 0x7fc00000u : __float_as_uint(bv);
  const unsigned kmax = wave_reduce_u32<true>(bv < 0.f ? 0u : key);
  const unsigned cand = (!(bv < 0.f) && key == kmax) ? (unsigned)bi : 0xffffffffu;
  bi = (int)wave_reduce_u32<false>(cand);
  bv = __uint_as_float(kmax);
}
XM_DEV void wave_amax(double& bv, int& bi) {  // not on the hot path: shuffle butterfly
#pragma unroll
  for (int m = XM_WAVE / 2; m >= 1; m >>= 1) {
    const double ov = shfl_xor_t(bv, m);
    const int oi = __shfl_xor(bi, m, XM_WAVE);
    amax_take(bv, bi, ov, oi);
  }
}

// Reduce (bv, bi) over the NT threads that hold one spectrum and let one thread write the result.
// NT is a power of two.  NT <= 64: shuffle butterfly inside the NT-lane group.  NT > 64: DPP wave
// reduction, one LDS slot per wave, ONE workgroup barrier, then the spectrum's first wave combines the
// NT/64 slots.  `red_v` / `red_i` must be LDS that nothing else writes until the next barrier.
template <class T, int NT>
XM_DEV void amax_reduce_store(T bv, int bi, int t, bool live, long long s, T* absmax2, int32_t* argidx,
                              T* red_v, int* red_i) {
  if constexpr (NT <= XM_WAVE) {
#pragma unroll
    for (int m = NT / 2; m >= 1; m >>= 1) {
      T ov = shfl_xor_t(bv, m);
      int oi = __shfl_xor(bi, m, XM_WAVE);
      amax_take(bv, bi, ov, oi);
    }
    if (t == 0 && live) {
      absmax2[s] = bv;
      argidx[s] = bi;
    }
  } else {
    constexpr int NW = NT / XM_WAVE;  // waves per spectrum
    const int wave = threadIdx.x / XM_WAVE, lane = threadIdx.x & (XM_WAVE - 1);
    wave_amax(bv, bi);
    if (lane == 0) {
      red_v[wave] = bv;
      red_i[wave] = bi;
    }
    __syncthreads();
    if (t < XM_WAVE) {  // first wave of this spectrum (t == lane there)
      const int w0 = wave;
      T v = T(-1);
      int i = 0x7fffffff;
      if (lane < NW) {
        v = red_v[w0 + lane];
        i = red_i[w0 + lane];
      }
#pragma unroll
      for (int m = (NW > 1 ? NW / 2 : 0); m >= 1; m >>= 1) {
        T ov = shfl_xor_t(v, m);
        int oi = __shfl_xor(i, m, XM_WAVE);
        amax_take(v, i, ov, oi);
      }
      if (lane == 0 && live) {
        absmax2[s] = v;
        argidx[s] = i;
      }
    }
  }
}


// =================================================================================================
// Chunked dynamic hand-out of work items (rows) to the workgroups of a persistent kernel -- k_zf2p's row queue
// (xm_zf2p.h) as a helper, used by the first-generation kernel k_zf2 (complex128).  One ticket = `ch` consecutive
// items; the first round is static (chunk = blockIdx.x); at the first item of every chunk thread 0 claims the chunk
// AFTER next, the ticket returns while the transform runs and is published through LDS in front of one of the
// transform's own barriers (BlockFFT's hook).  queue == nullptr: plain static stride.
// Per iteration:  next_item() -> prefetch;  ticket = claim(t);  FFT(hook: publish(t, ticket));  collect();  ...;  advance().
// After the loop: finish(t).
// =================================================================================================
struct WorkQueue {
  unsigned* q;
  unsigned* slot;  // one LDS word
  long long n, ch, c_cur, c_nxt, c_nn, item;
  unsigned off;
  XM_DEV void init(unsigned* queue, int chunk, unsigned* lds_slot, long long n_items, unsigned t) {
    q = queue;
    slot = lds_slot;
    n = n_items;
    ch = chunk > 0 ? chunk : 1;
    c_cur = blockIdx.x;
    c_nxt = c_cur + gridDim.x;
    c_nn = c_nxt + gridDim.x;
    off = 0;
    item = c_cur * ch;
    if (q) {
      if (t == 0) *slot = atomicAdd(q, 1u) + gridDim.x;
      __syncthreads();
      c_nxt = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)*slot);
      __syncthreads();
    }
  }
  XM_DEV bool chunk_end() const { return (off + 1 == (unsigned)ch) || (item + 1 >= n); }
  XM_DEV long long next_item() const { return chunk_end() ? c_nxt * ch : item + 1; }  // may be >= n: nothing left
  XM_DEV unsigned claim(unsigned t) const {
    unsigned ticket = 0;
    if (q && off == 0u && t == 0u) ticket = atomicAdd(q, 1u) + gridDim.x;
    return ticket;
  }
  XM_DEV void publish(unsigned t, unsigned ticket) const {
    if (q && off == 0u && t == 0u) *slot = ticket;
  }
  XM_DEV void collect() {  // after the transform's barriers
    if (q && off == 0u) c_nn = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)*slot);
  }
  XM_DEV void advance() {
    const long long nx = next_item();
    if (chunk_end()) {
      c_cur = c_nxt;
      c_nxt = c_nn;
      if (!q) c_nn = c_nxt + gridDim.x;
      off = 0;
    } else {
      ++off;
    }
    item = nx;
  }
  XM_DEV void finish(unsigned t) const {  // the last workgroup out leaves the counters at zero for the next launch
    if (q && t == 0u) {
      const unsigned d = atomicAdd(q + 1, 1u);
      if (d == gridDim.x - 1u) {
        __hip_atomic_store(q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
};

// =================================================================================================
// Generic fused kernel.  blockDim = NT * SPB, one spectrum per NT threads.
// =================================================================================================
// waves per SIMD to leave room for: a plan that asks for the plane-by-plane exchange does so to fit TWO workgroups
template <class PL, int SPB>
constexpr int pipe_waves() {
  return xm_force_split<PL>::value ? (2 * PL::NT * SPB / 256 > 8 ? 8 : 2 * PL::NT * SPB / 256) : 1;
}

template <class T, class PL, int SPB>
__global__ __launch_bounds__(PL::NT* SPB, (pipe_waves<PL, SPB>())) void k_pipe(PipeArgs<T> A) {
  constexpr int N = PL::N, NT = PL::NT, P = PL::P;
  extern __shared__ __attribute__((aligned(16))) char xm_smem[];
  const int t = threadIdx.x % NT;
  const int ls = threadIdx.x / NT;
  const long long s = (long long)blockIdx.x * SPB + ls;
  const bool live = s < A.n_batch;
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(xm_smem) + (size_t)ls * BlockFFT<T, PL>::lds_elems();

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

// ---- raw buffer access: SGPR descriptor (base, size) + 32-bit lane offset + scalar offset.  Reads
// beyond the descriptor's size return 0 and such writes are dropped, in hardware, so the zero fill
// needs neither branches nor 64-bit per-lane address arithmetic.
typedef unsigned xm_u2 __attribute__((ext_vector_type(2)));
typedef unsigned xm_u4 __attribute__((ext_vector_type(4)));

XM_DEV __amdgpu_buffer_rsrc_t xm_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

XM_DEV Cx<double> buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, Cx<double>*) {
  const xm_u4 u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  Cx<double> c;
  __builtin_memcpy(&c, &u, 16);
  return c;
}
XM_DEV CxPair<float> buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, CxPair<float>*) {
  const xm_u4 u = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  CxPair<float> c;
  __builtin_memcpy(&c, &u, 16);
  return c;
}
XM_DEV CxPair<double> buf_load(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, CxPair<double>*) {
  CxPair<double> c;
  c.a = buf_load(r, voff, soff, (Cx<double>*)nullptr);
  c.b = buf_load(r, voff + 16u, soff, (Cx<double>*)nullptr);
  return c;
}
// VMEM store-data hazard.  A buffer store of more than 64 bits reads its data VGPRs over several cycles; a VALU
// instruction issued right behind it that overwrites one of them corrupts what is stored.  LLVM's hazard
// recogniser inserts the wait state only for MUBUF stores WITHOUT an SGPR soffset (the documented SI-era rule),
// but gfx950 shows the hazard with an SGPR soffset too: in the write-only mode of k_zf2 -- nothing between one
// store and the next output's arithmetic -- dword 1 of lanes 12..15 of every 16-lane group carried the NEXT
// butterfly's value.  Hence these wide stores never use an SGPR soffset: the wave-uniform part of the address
// goes into the buffer descriptor's base (scalar ALU) and `soffset` is the literal 0, which keeps the
// compiler's own hazard handling in charge.
XM_DEV void buf_store(__amdgpu_buffer_rsrc_t r, unsigned voff, CxPair<float> v) {
  xm_u4 u;
  __builtin_memcpy(&u, &v, 16);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, 0, 0);
}
XM_DEV void buf_store(__amdgpu_buffer_rsrc_t r, unsigned voff, CxPair<double> v) {
  xm_u4 u;
  __builtin_memcpy(&u, &v.a, 16);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, voff, 0, 0);
  __builtin_memcpy(&u, &v.b, 16);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, voff + 16u, 0, 0);
}

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N)
template <int I, int N_, class F>
XM_DEV void static_for(F&& f) {
  if constexpr (I < N_) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N_>(f);
  }
}

// Persistent kernel: gridDim.x workgroups of NT threads loop over the spectra (stride gridDim.x).
//   * everything that does not depend on the spectrum is loaded ONCE per workgroup: the window
//     samples w[j], the per-thread rotation W_N^t, the last-stage twiddles (registers) and the
//     middle-stage twiddles (LDS copy) -- no table traffic inside the loop except the phase table;
//   * the next spectrum's FID samples are prefetched into registers before the current FFTs start,
//     so the HBM read of spectrum i+1 overlaps the butterflies / LDS exchanges of spectrum i;
//   * W_N^{t + NT*q} = W_N^t * W_{2P}^q : one per-thread complex rotation plus compile-time
//     constants, i.e. the odd-bin half is a half-sample-shifted copy of the even-bin butterfly;
//   * output index of (E[m], O[m]), m = t + NT*q, after the fftshift roll: (2*NT*q + shift) mod N
//     + 2t with shift a multiple of 2*NT -- a scalar base per q plus one per-lane offset, so all
//     addressing is SGPR base + 32-bit lane offset, and each lane stores one 16-byte (c64) word.
// MODE bits: 1 = write the spectrum, 2 = multiply by the phase table, 4 = per-spectrum arg-max.
// ZF2_VALUE_ONLY: the arg-max part is compiled WITHOUT the first-index scan (complex128 write + arg-max modes: the
// scan's live state pushed them over 256 VGPRs); the runtime flag PipeArgs::amax_value_only alone only skips it.
// ZF2_RAMP (with ZF2_WRITE, instead of ZF2_PHASE): the output phase is a linear ramp given in factorised form
// (PipeArgs::ramp_*), see xm_zf2p.h.  ZF2_PAIR (complex128): both half transforms ride through ONE pass of the
// block FFT as a two-lane element (32-byte exchange elements, half the barriers) instead of one after the other.
enum { ZF2_WRITE = 1, ZF2_PHASE = 2, ZF2_AMAX = 4, ZF2_RAMP = 8, ZF2_VALUE_ONLY = 16, ZF2_PAIR = 32,
       ZF2_GKEY = 64,    // k_zf2d: the maxima go into the launch's arg-max key (PipeArgs::gkey) instead of per-row arrays
       ZF2_DMA = 128 };  // k_zf2d: the next row is prefetched into the idle exchange buffer (global_load_lds, no registers)
template <class T, int MODE>
constexpr bool zf2_paired() {
  return sizeof(T) == 4 || (MODE & ZF2_PAIR) != 0;
}

// waves per SIMD the register allocator must leave room for: two resident workgroups per CU
template <class T, class PL>
constexpr int zf2_waves() {
  int w = PL::NT / 128;  // 2 workgroups of NT threads over 4 SIMDs
  if (sizeof(T) == 8) w /= 2;  // complex128 (8 points per thread): ~170 VGPRs, one workgroup per CU
  return w < 1 ? 1 : (w > 4 ? 4 : w);
}

template <class T, class PL, int MODE>
__global__ __launch_bounds__(PL::NT, (zf2_waves<T, PL>())) void k_zf2(PipeArgs<T> A) {
  constexpr unsigned N = 2 * PL::N, NT = PL::NT;
  constexpr int P = PL::P;
  constexpr bool WRITE = (MODE & ZF2_WRITE) != 0, PHASE = (MODE & ZF2_PHASE) != 0, AMAX = (MODE & ZF2_AMAX) != 0;
  constexpr bool RAMP = (MODE & ZF2_RAMP) != 0;
  static_assert(!(PHASE && RAMP) && (!RAMP || WRITE), "a ramp replaces the phase table of a writing mode");
  // float: the two half-FFTs ride in the two lanes of the packed-f32 VALU (lane x: even bins, lane y:
  // odd bins).  double: no packed f64 math exists, so the halves run one after the other through a
  // 16-byte-element exchange buffer (padded every 16 elements so that two workgroups fit the LDS).
  constexpr bool PACKED = sizeof(T) == 4;
  using V = typename std::conditional<PACKED, typename PairOf<T>::type, T>::type;
  using FFT = BlockFFT<V, PL>;
  using HT = HotTw<T, PL, RAMP>;
  extern __shared__ __attribute__((aligned(16))) char xm_smem[];
  Cx<V>* lds = reinterpret_cast<Cx<V>*>(xm_smem);
  Cx<T>* mid = reinterpret_cast<Cx<T>*>(lds + FFT::lds_elems());
  T* red_v = reinterpret_cast<T*>(mid + HT::mid_size());  // own region: the loop reuses `lds` at once
  int* red_i = reinterpret_cast<int*>(red_v + NT / XM_WAVE + 1);
  unsigned* wq_slot = reinterpret_cast<unsigned*>(red_i + NT / XM_WAVE + 1);
  const unsigned t = threadIdx.x;

  HT tw;
  tw.mid = mid;
  tw.load(A.tw, (int)t);
  for (unsigned i = t; i < (unsigned)HT::mid_size(); i += NT) mid[i] = A.tw[i];
  Cx<T> rot = A.aux[t];  // W_N^t
  if constexpr (RAMP) {  // e^{i b 2t} into the last-stage twiddles, the odd bins' e^{i b} into their rotation (xm_zf2p.h)
    double sn, cs;
    sincos(A.ramp_db * (double)(2u * t), &sn, &cs);
    tw.fold(mk<T>((T)cs, (T)sn));
    rot = rot * mk<T>(A.ramp_e[0], A.ramp_e[1]);
  }
  const unsigned n_in = (unsigned)A.n_in;
  const unsigned toff = t - (unsigned)A.pad_left;  // wraps for t < pad_left -> fails the range test
  T w[P];  // window sample * FFT scale; 0 outside the acquired samples (the zero fill)
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const bool ok = (toff + NT * q) < n_in;
    w[q] = ok ? (A.window ? A.window[t + NT * q] * A.scale : A.scale) : T(0);
  }
  // The pre-pass (arg-max only, packed) has ~50 VGPRs to spare and is instruction bound: it folds the window
  // into the odd-bin rotation once per launch, R_q = w_q * W_N^{t + NT q}, so that one sample costs two
  // packed multiplies + two FMAs instead of ~11 scalar operations (window, rotation, W_2P^q, lane packing).
  constexpr bool FOLD = PACKED && MODE == ZF2_AMAX;
  constexpr bool SEQ = !PACKED && MODE == ZF2_AMAX;
  constexpr bool CVO = (MODE & ZF2_VALUE_ONLY) != 0;
  Cx<T> wr[FOLD ? P : 1];
  if constexpr (FOLD) {
    static_for<0, P>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const Cx<T> r = mul_w<q, 2 * P, T>(rot);
      wr[q] = mk<T>(r.re * w[q], r.im * w[q]);
    });
  }
  __syncthreads();

  // FID sample j = t + NT*q of a row lives at row[j - pad_left].  Positions outside the acquired
  // samples read a clamped (valid) address and are zeroed by their window weight w = 0, so the loads
  // need no branches: uniform row pointer + 32-bit lane offset.  (ROCm 7.2's
  // __builtin_amdgcn_raw_buffer_load_b64 lowers to a ONE-dword load, so the 8-byte FID loads do not
  // use the buffer path; the 16-byte phase loads and spectrum stores below do.)
  constexpr unsigned CB = sizeof(Cx<T>);
  const unsigned last_in = n_in - 1u;
  Cx<T> xr[P];
  WorkQueue wq;
  wq.init(A.queue, A.queue_chunk, wq_slot, A.n_batch, t);
  if (wq.item < A.n_batch) {
    const Cx<T>* __restrict__ row = A.in + wq.item * A.in_stride;
#pragma unroll
    for (int q = 0; q < P; ++q) xr[q] = row[min(toff + NT * q, last_in)];
  }

  for (; wq.item < A.n_batch; wq.advance()) {
    const long long s = wq.item;
    // Opaque copies: every LDS / global address below is a cheap function of (t, shift, n_in).  Without
    // this the compiler hoists all of them out of the persistent loop (they are loop-invariant) and
    // pays for it with >100 live VGPRs/SGPRs; recomputing them per spectrum costs a few ALU ops.
    unsigned tt = t, sh = (unsigned)A.out_shift, nin = n_in, pl = (unsigned)A.pad_left;
    asm volatile("" : "+v"(tt));
    asm volatile("" : "+s"(sh));
    asm volatile("" : "+s"(nin));
    asm volatile("" : "+s"(pl));
    const unsigned toff2 = tt - pl;
    auto prefetch = [&]() {  // the next FID, while the current one is transformed
      const long long s2 = wq.next_item();
      if (s2 < A.n_batch) {
        const Cx<T>* __restrict__ row = A.in + s2 * A.in_stride;
        if (nin == NT * P && pl == 0u) {
          // exactly half full (the 2x zero fill): every slot holds a sample, no clamp -- scalar row base +
          // compile-time slot offset + one 32-bit lane offset, no per-load vector address arithmetic
          const unsigned lane_off = tt * CB;  // 32-bit byte offset: lets the load use SGPR base + VGPR offset
#pragma unroll
          for (int q = 0; q < P; ++q) {
            const char* rq = reinterpret_cast<const char*>(row + NT * q);
            xr[q] = *reinterpret_cast<const Cx<T>*>(rq + lane_off);
          }
        } else {
#pragma unroll
          for (int q = 0; q < P; ++q) xr[q] = row[min(toff2 + NT * q, nin - 1u)];
        }
      }
    };
    if constexpr (SEQ) {
      // complex128 arg-max pre-pass: the halves run one after the other anyway, so each is reduced to its
      // (max, first index) as soon as it is transformed and only ONE half-spectrum is ever live in registers
      // (both halves + the prefetched FID did not fit 256 VGPRs); the odd half is formed from the samples
      // after the even half is done, the prefetch overlaps the second transform.
      const unsigned t2 = 2u * tt;
      Cx<T> h[P];
      T bv = T(-1);
      int bi = 0x7fffffff;
      auto reduce_half = [&](unsigned odd) {
        T hv = T(-1);
#pragma unroll
        for (int q = 0; q < P; ++q) hv = fmax(hv, h[q].re * h[q].re + h[q].im * h[q].im);
        hv = amax_nan_if_unset(hv);
        const bool all_nan = hv != hv;
        int hi = 0;
        if (!A.amax_value_only) {  // wave-uniform
          hi = 0x7fffffff;
#pragma unroll
          for (int q = 0; q < P; ++q) {
            const int k0 = (int)(((2u * NT * q + sh) & (N - 1u)) + t2 + odd);
            hi = min(hi, (((h[q].re * h[q].re + h[q].im * h[q].im) == hv) | all_nan) ? k0 : 0x7fffffff);
          }
        }
        amax_take(bv, bi, hv, hi);
      };
#pragma unroll
      for (int q = 0; q < P; ++q) h[q] = xr[q] * w[q];
      FFT::run(h, lds, tw, (int)tt);
      reduce_half(0u);
      static_for<0, P>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        h[q] = mul_w<q, 2 * P, T>((xr[q] * w[q]) * rot);
      });
      prefetch();
      const unsigned ticket = wq.claim(tt);
      FFT::run_cols(h, lds, tw, (int)tt, (int)tt, [&]() { wq.publish(tt, ticket); });
      wq.collect();
      reduce_half(1u);
      if (A.amax_value_only) bi = 0;
      amax_reduce_store<T, (int)NT>(bv, bi, (int)tt, true, s, A.absmax2, A.argidx, red_v, red_i);
      continue;
    }
    Cx<V> v[P];                   // packed: both halves; unpacked: the even-bin half
    Cx<T> vo[PACKED ? 1 : P];     // unpacked: the odd-bin half
    static_for<0, P>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      if constexpr (FOLD) {
        // lane x: x * w (even bins), lane y: x * R_q (odd bins)
        const V a = V{w[q], wr[q].re};
        V re = V{xr[q].re, xr[q].re} * a;
        V im = V{xr[q].im, xr[q].im} * a;
        re.y = __builtin_fmaf(-xr[q].im, wr[q].im, re.y);
        im.y = __builtin_fmaf(xr[q].re, wr[q].im, im.y);
        v[q].re = re;
        v[q].im = im;
      } else {
        const Cx<T> e = xr[q] * w[q];
        const Cx<T> o = mul_w<q, 2 * P, T>(e * rot);
        if constexpr (PACKED) {
          v[q].re = V{e.re, o.re};
          v[q].im = V{e.im, o.im};
        } else {
          v[q] = e;
          vo[q] = o;
        }
      }
    });
    prefetch();
    const unsigned ticket = wq.claim(tt);
    FFT::run_cols(v, lds, tw, (int)tt, (int)tt, [&]() { wq.publish(tt, ticket); });
    wq.collect();
    if constexpr (!PACKED) FFT::run(vo, lds, tw, (int)tt);
    // (even, odd) outputs of butterfly q as scalars
    auto even = [&](int q) -> Cx<T> {
      if constexpr (PACKED) return mk<T>(v[q].re.x, v[q].im.x); else return v[q];
    };
    auto odd = [&](int q) -> Cx<T> {
      if constexpr (PACKED) return mk<T>(v[q].re.y, v[q].im.y); else return vo[q];
    };

    const unsigned t2 = 2u * tt;
    if constexpr (AMAX) {  // thread-local max |X|^2 first (packed), then the first index holding it
      auto mag2 = [&](int q, T& me, T& mo) {
        if constexpr (PACKED) {
          const V m2 = v[q].re * v[q].re + v[q].im * v[q].im;
          me = m2.x;
          mo = m2.y;
        } else {
          me = v[q].re * v[q].re + v[q].im * v[q].im;
          mo = vo[q].re * vo[q].re + vo[q].im * vo[q].im;
        }
      };
      T bv = T(-1);
#pragma unroll
      for (int q = 0; q < P; ++q) {
        T me, mo;
        mag2(q, me, mo);
        bv = fmax(bv, fmax(me, mo));
      }
      bv = amax_nan_if_unset(bv);
      const bool all_nan = bv != bv;  // then every index of this thread qualifies
      int bi = 0;
      if (!CVO && !A.amax_value_only) {  // wave-uniform (CVO: compiled out)
        bi = 0x7fffffff;
#pragma unroll
        for (int q = 0; q < P; ++q) {
          const int k0 = (int)(((2u * NT * q + sh) & (N - 1u)) + t2);
          T me, mo;
          mag2(q, me, mo);
          bi = min(bi, ((me == bv) | all_nan) ? k0 : 0x7fffffff);
          bi = min(bi, ((mo == bv) | all_nan) ? k0 + 1 : 0x7fffffff);
        }
      }
      if constexpr (sizeof(T) == 4 && NT > XM_WAVE) {
        if (A.amax_value_only) {
          // value only: one atomic max per wave into the row's slot (zeroed by the launcher; squared magnitudes
          // order like their bit patterns) -- no LDS slot, no workgroup barrier
          const unsigned key = wave_reduce_u32<true>(__float_as_uint(bv));
          if ((tt & (XM_WAVE - 1)) == 0u) atomicMax(reinterpret_cast<unsigned*>(A.absmax2 + s), key);
          if (tt == 0u) A.argidx[s] = 0;
        } else {
          amax_reduce_store<T, (int)NT>(bv, bi, (int)tt, true, s, A.absmax2, A.argidx, red_v, red_i);
        }
      } else {
        amax_reduce_store<T, (int)NT>(bv, bi, (int)tt, true, s, A.absmax2, A.argidx, red_v, red_i);
      }
    }
    if constexpr (WRITE) {
      Cx<T>* __restrict__ orow = A.out + s * (long long)N;
      const __amdgpu_buffer_rsrc_t rph = xm_rsrc(A.phase, PHASE ? N * CB : 0u);
      // RAMP: the wave-uniform factors e^{i(a + b base_q)}, re-read from the kernel-argument segment every spectrum
      typedef const T __attribute__((address_space(4))) * kptr_t;
      kptr_t rc = (kptr_t)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() +
                           __builtin_offsetof(PipeArgs<T>, ramp_c));
      asm volatile("" : "+s"(rc));
      static_for<0, P>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const unsigned base = (2u * NT * q + sh) & (N - 1u);  // wave-uniform
        // this butterfly's 2*NT contiguous outputs: descriptor base = row + base (see buf_store)
        const __amdgpu_buffer_rsrc_t rout = xm_rsrc(orow + base, 2u * NT * CB);
        Cx<T> xe = even(q);
        Cx<T> xo = odd(q);
        if constexpr (RAMP) {
          const Cx<T> c = mk<T>(rc[2 * q], rc[2 * q + 1]);
          xe = xe * c;
          xo = xo * c;
        } else if constexpr (PHASE) {
          const CxPair<T> ph = buf_load(rph, t2 * CB, base * CB, (CxPair<T>*)nullptr);
          xe = xe * ph.a;
          xo = xo * ph.b;
        }
        CxPair<T> o;
        o.a = xe;
        o.b = xo;
        buf_store(rout, t2 * CB, o);
      });
    }
  }
  wq.finish(t);
}

// =================================================================================================
// Generic persistent kernel: TWO spectra ride in the two packed-f32 lanes (lane x: row 2g, lane y: row
// 2g+1), same structure as k_zf2 (persistent workgroups, window / last-stage twiddles in registers,
// middle-stage twiddles in LDS, next pair prefetched).  Any direct plan with NT >= 64 whose 16-byte
// exchange buffer fits the LDS; rolls, inverse, window, phase and arg-max as in k_pipe.
// =================================================================================================
// waves per SIMD the register allocator must leave room for: as many workgroups per CU as the LDS holds
// (at most two)
template <class PL>
constexpr int fft2_waves() {
  const long lds = (long)BlockFFT<xm_f2, PL>::lds_elems() * (long)sizeof(Cx<xm_f2>);
  const int wg = 160 * 1024 / lds >= 2 ? 2 : 1;
  int w = wg * PL::NT / 256;
  // 512 threads x 8 points x two packed spectra need ~150 VGPRs: one 512-thread workgroup per CU without spills
  // beats two with ~40 spilled registers (4096-point FFT seam 4.3 -> 4.8 TB/s, fused main pass 3.6 -> 5.1 TB/s)
  if (PL::NT == 512 && w > 2) w = 2;
  return w < 1 ? 1 : (w > 4 ? 4 : w);
}

template <class PL, int MODE>
__global__ __launch_bounds__(PL::NT, (fft2_waves<PL>())) void k_fft2(PipeArgs<float> A) {
  using T = float;
  using V = xm_f2;
  constexpr int N = PL::N, NT = PL::NT, P = PL::P;
  constexpr bool WRITE = (MODE & ZF2_WRITE) != 0, PHASE = (MODE & ZF2_PHASE) != 0, AMAX = (MODE & ZF2_AMAX) != 0;
  // ZF2_RAMP: a linear output phase e^{i (a + b k)} in factorised form, as in k_zf2p: output k of thread t, slot q is
  // base_q + t with base_q = (NT q + out_shift) mod N a multiple of NT (the launcher checks out_shift % NT == 0), so
  // e^{i b t} is folded into the last stage's register twiddles once per launch and the wave-uniform e^{i (a + b base_q)}
  // comes from the kernel arguments -- no table, no per-output load (forward transforms only)
  constexpr bool RAMP = (MODE & ZF2_RAMP) != 0;
  static_assert(!(PHASE && RAMP) && (!RAMP || (WRITE && P <= 16)), "a ramp replaces the table of a writing mode; 16 slots");
  // ZF2_GKEY (round 4): the launch's global arg-max goes into the arg-max key (PipeArgs::gkey, as in k_zf2p) instead
  // of per-row arrays -- every wave keeps the best (max |X|^2 bits, row) of the row PARTS it transforms, merges it with
  // one atomic when it leaves, the last workgroup out decodes the key into the caller's record: no index scan, no
  // per-row reduction through the LDS, no barrier, and no reduction launches behind the kernel.
  constexpr bool GKEY = (MODE & ZF2_GKEY) != 0;
  static_assert(!GKEY || AMAX, "the key holds maxima");
  using FFT = BlockFFT<V, PL>;
  using HT = HotTw<T, PL, RAMP>;
  extern __shared__ __attribute__((aligned(16))) char xm_smem[];
  Cx<V>* lds = reinterpret_cast<Cx<V>*>(xm_smem);
  Cx<T>* mid = reinterpret_cast<Cx<T>*>(lds + FFT::lds_elems());
  T* red_v = reinterpret_cast<T*>(mid + HT::mid_lds_size());
  int* red_i = reinterpret_cast<int*>(red_v + 2 * (NT / XM_WAVE) + 2);
  const int t = threadIdx.x;

  HT tw;
  tw.load(A.tw, t);
  if constexpr (HT::mid_in_lds()) {
    tw.mid = mid;
    for (int i = t; i < HT::mid_size(); i += NT) mid[i] = A.tw[i];
  } else {
    tw.mid = A.tw;
  }
  if constexpr (RAMP) {
    double sn, cs;
    sincos(A.ramp_db * (double)t, &sn, &cs);
    tw.fold(mk<T>((T)cs, (T)sn));
  }
  // FFT input position pos = t + NT*q holds padded sample j = (pos - in_shift) mod N = input sample j - pad_left
  T w[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    int j = t + NT * q - A.in_shift;
    if (j < 0) j += N;
    const bool ok = (unsigned)(j - A.pad_left) < (unsigned)A.n_in;
    w[q] = ok ? (A.window ? A.window[j] * A.scale : A.scale) : T(0);
  }
  __syncthreads();

  const long long npairs = (A.n_batch + 1) / 2;
  const unsigned last_in = (unsigned)A.n_in - 1u;
  Cx<T> x0[P], x1[P];
  auto fetch = [&](long long g, int tt, int shift, int padl, unsigned lastv) {
    const long long s0 = 2 * g, s1 = (2 * g + 1 < A.n_batch) ? 2 * g + 1 : 2 * g;  // odd tail: duplicate row
    const Cx<T>* __restrict__ r0 = A.in + s0 * A.in_stride;
    const Cx<T>* __restrict__ r1 = A.in + s1 * A.in_stride;
#pragma unroll
    for (int q = 0; q < P; ++q) {
      int j = tt + NT * q - shift;
      if (j < 0) j += N;
      const unsigned src = min((unsigned)(j - padl), lastv);  // clamped; zero-filled positions have w = 0
      x0[q] = r0[src];
      x1[q] = r1[src];
    }
  };
  long long g = blockIdx.x;
  if (g < npairs) fetch(g, t, A.in_shift, A.pad_left, last_in);
  unsigned best_key = 0u, best_row = 0u;  // GKEY: this wave's best so far (wave-uniform)
  bool have = false;

  for (; g < npairs; g += gridDim.x) {
    int tt = t, osh = A.out_shift, ish = A.in_shift, padl = A.pad_left;
    unsigned lastv = last_in;
    asm volatile("" : "+v"(tt));  // keep per-lane address arithmetic inside the loop (see k_zf2)
    asm volatile("" : "+s"(osh));
    asm volatile("" : "+s"(ish));
    asm volatile("" : "+s"(padl));
    asm volatile("" : "+s"(lastv));
    Cx<V> v[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
      v[q].re = V{x0[q].re, x1[q].re} * w[q];
      v[q].im = V{x0[q].im, x1[q].im} * w[q];
      if (A.inverse) v[q].im = -v[q].im;
    }
    if (g + gridDim.x < npairs) fetch(g + gridDim.x, tt, ish, padl, lastv);

    FFT::run(v, lds, tw, tt);

    const long long s0 = 2 * g, s1 = 2 * g + 1;
    const bool has1 = s1 < A.n_batch;
    if constexpr (AMAX) {
      T b0 = T(-1), b1 = T(-1);
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const V m2 = v[q].re * v[q].re + v[q].im * v[q].im;
        b0 = fmax(b0, m2.x);
        b1 = fmax(b1, m2.y);
      }
      b0 = amax_nan_if_unset(b0);
      b1 = amax_nan_if_unset(b1);
      if constexpr (GKEY) {
        // rows come in ascending order: strict > keeps the lowest row among equal values; a NaN (bit pattern above every
        // number) outranks everything, as np.argmax has it
        const unsigned k0 = wave_reduce_u32<true>(b0 != b0 ? 0x7fc00000u : __float_as_uint(b0));
        const unsigned k1 = wave_reduce_u32<true>(b1 != b1 ? 0x7fc00000u : __float_as_uint(b1));
        if (!have || k0 > best_key) {
          best_key = k0;
          best_row = (unsigned)s0;
          have = true;
        }
        if (has1 && k1 > best_key) {
          best_key = k1;
          best_row = (unsigned)s1;
        }
      } else {
      const bool nan0 = b0 != b0, nan1 = b1 != b1;
      int i0 = 0x7fffffff, i1 = 0x7fffffff;
#pragma unroll
      for (int q = 0; q < P; ++q) {
        int k = tt + NT * q + osh;
        if (k >= N) k -= N;
        const V m2 = v[q].re * v[q].re + v[q].im * v[q].im;
        i0 = min(i0, ((m2.x == b0) | nan0) ? k : 0x7fffffff);
        i1 = min(i1, ((m2.y == b1) | nan1) ? k : 0x7fffffff);
      }
      amax_reduce_store<T, NT>(b0, i0, tt, true, s0, A.absmax2, A.argidx, red_v, red_i);
      amax_reduce_store<T, NT>(b1, i1, tt, has1, has1 ? s1 : s0, A.absmax2, A.argidx, red_v + NT / XM_WAVE + 1,
                               red_i + NT / XM_WAVE + 1);
      }
    }
    if constexpr (WRITE) {
      Cx<T>* __restrict__ o0 = A.out + s0 * (long long)N;
      Cx<T>* __restrict__ o1 = A.out + (has1 ? s1 : s0) * (long long)N;
#pragma unroll
      for (int q = 0; q < P; ++q) {
        int k = tt + NT * q + osh;
        if (k >= N) k -= N;
        Cx<V> y = v[q];
        if constexpr (RAMP) {  // (wave-uniform factor of slot q: scalar loads from the kernel-argument segment)
          y = y * mk<T>(A.ramp_c[2 * q], A.ramp_c[2 * q + 1]);
        } else {
          if (A.inverse) y.im = -y.im;
          if constexpr (PHASE) y = y * A.phase[k];
        }
        o0[k] = mk<T>(y.re.x, y.im.x);
        if (has1) o1[k] = mk<T>(y.re.y, y.im.y);
      }
    }
  }
  if constexpr (GKEY) {
    // merge, count out, and let the last workgroup decode + clear the key (the scheme of k_zf2p's epilogue; the
    // launcher hands this mode a {head, done} counter pair)
    const unsigned lane = (unsigned)t & (XM_WAVE - 1u);
    if (have && lane == 0u)
      atomicMax(A.gkey + (blockIdx.x % XM_KEY_SLOTS) * XM_KEY_STRIDE,
                ((unsigned long long)best_key << 32) | (unsigned long long)(0xffffffffu - best_row));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned last = 0;
    if (t == 0) {
      const unsigned d = atomicAdd(A.queue + 1, 1u);
      last = d == gridDim.x - 1u;
      if (last) __hip_atomic_store(A.queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (A.key_result && t < XM_WAVE) {  // first wave: one partial key per lane
      last = (unsigned)__builtin_amdgcn_readfirstlane((int)last);
      if (last) {
        unsigned long long k = __hip_atomic_load(A.gkey + t * XM_KEY_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.gkey + t * XM_KEY_STRIDE, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int m = XM_WAVE / 2; m >= 1; m >>= 1) {
          const unsigned hi = (unsigned)__shfl_xor((int)(k >> 32), m, XM_WAVE), lo = (unsigned)__shfl_xor((int)(unsigned)k, m, XM_WAVE);
          const unsigned long long o = ((unsigned long long)hi << 32) | lo;
          k = o > k ? o : k;
        }
        if (t == 0) {
          const unsigned row = k ? 0xffffffffu - (unsigned)(k & 0xffffffffu) : 0u;  // nothing published: row 0
          A.key_result->max2 = __uint_as_float((unsigned)(k >> 32));
          A.key_result->flat = (long long)row * (long long)N;
        }
      }
    }
  }
}

// =================================================================================================
// Scalar persistent kernel (complex128: no packed f64 math exists, so one spectrum per workgroup pass):
// the structure of k_fft2 -- persistent workgroups, window and last-stage twiddles in registers,
// middle-stage twiddles in LDS, the next spectrum prefetched while the current one is transformed.
// =================================================================================================
template <class T, class PL>
constexpr int fft1_waves() {
  const long lds = (long)BlockFFT<T, PL>::lds_elems() * (long)sizeof(Cx<T>);
  const int wg = 160 * 1024 / lds >= 2 ? 2 : 1;
  int w = wg * PL::NT / 256;
  return w < 1 ? 1 : (w > 4 ? 4 : w);
}

template <class T, class PL, int MODE>
__global__ __launch_bounds__(PL::NT, (fft1_waves<T, PL>())) void k_fft1(PipeArgs<T> A) {
  constexpr int N = PL::N, NT = PL::NT, P = PL::P;
  constexpr bool WRITE = (MODE & ZF2_WRITE) != 0, PHASE = (MODE & ZF2_PHASE) != 0, AMAX = (MODE & ZF2_AMAX) != 0;
  using FFT = BlockFFT<T, PL>;
  using HT = HotTw<T, PL>;
  extern __shared__ __attribute__((aligned(16))) char xm_smem[];
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(xm_smem);
  Cx<T>* mid = lds + FFT::lds_elems();
  T* red_v = reinterpret_cast<T*>(mid + HT::mid_lds_size());
  int* red_i = reinterpret_cast<int*>(red_v + NT / XM_WAVE + 1);
  const int t = threadIdx.x;

  HT tw;
  tw.load(A.tw, t);
  if constexpr (HT::mid_in_lds()) {
    tw.mid = mid;
    for (int i = t; i < HT::mid_size(); i += NT) mid[i] = A.tw[i];
  } else {
    tw.mid = A.tw;
  }
  T w[P];
#pragma unroll
  for (int q = 0; q < P; ++q) {
    int j = t + NT * q - A.in_shift;
    if (j < 0) j += N;
    const bool ok = (unsigned)(j - A.pad_left) < (unsigned)A.n_in;
    w[q] = ok ? (A.window ? A.window[j] * A.scale : A.scale) : T(0);
  }
  __syncthreads();

  const unsigned last_in = (unsigned)A.n_in - 1u;
  Cx<T> xr[P];
  auto fetch = [&](long long s, int tt, int shift, int padl, unsigned lastv) {
    const Cx<T>* __restrict__ row = A.in + s * A.in_stride;
#pragma unroll
    for (int q = 0; q < P; ++q) {
      int j = tt + NT * q - shift;
      if (j < 0) j += N;
      xr[q] = row[min((unsigned)(j - padl), lastv)];  // clamped; zero-filled positions have w = 0
    }
  };
  long long s = blockIdx.x;
  if (s < A.n_batch) fetch(s, t, A.in_shift, A.pad_left, last_in);

  for (; s < A.n_batch; s += gridDim.x) {
    int tt = t, osh = A.out_shift, ish = A.in_shift, padl = A.pad_left;
    unsigned lastv = last_in;
    asm volatile("" : "+v"(tt));  // keep per-lane address arithmetic inside the loop (see k_zf2)
    asm volatile("" : "+s"(osh));
    asm volatile("" : "+s"(ish));
    asm volatile("" : "+s"(padl));
    asm volatile("" : "+s"(lastv));
    Cx<T> v[P];
#pragma unroll
    for (int q = 0; q < P; ++q) {
      v[q] = xr[q] * w[q];
      if (A.inverse) v[q].im = -v[q].im;
    }
    if (s + gridDim.x < A.n_batch) fetch(s + gridDim.x, tt, ish, padl, lastv);

    FFT::run(v, lds, tw, tt);

    if constexpr (AMAX) {
      T bv = T(-1);
#pragma unroll
      for (int q = 0; q < P; ++q) bv = fmax(bv, v[q].re * v[q].re + v[q].im * v[q].im);
      bv = amax_nan_if_unset(bv);
      const bool all_nan = bv != bv;
      int bi = 0x7fffffff;
#pragma unroll
      for (int q = 0; q < P; ++q) {
        int k = tt + NT * q + osh;
        if (k >= N) k -= N;
        bi = min(bi, (((v[q].re * v[q].re + v[q].im * v[q].im) == bv) | all_nan) ? k : 0x7fffffff);
      }
      amax_reduce_store<T, NT>(bv, bi, tt, true, s, A.absmax2, A.argidx, red_v, red_i);
    }
    if constexpr (WRITE) {
      Cx<T>* __restrict__ orow = A.out + s * (long long)N;
#pragma unroll
      for (int q = 0; q < P; ++q) {
        int k = tt + NT * q + osh;
        if (k >= N) k -= N;
        Cx<T> y = v[q];
        if (A.inverse) y.im = -y.im;
        if constexpr (PHASE) y = y * A.phase[k];
        orow[k] = y;
      }
    }
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
  // 1024-thread workgroups have 128 VGPRs per thread: the scheduler must not hoist the table loads of all P positions
  // in front of the arithmetic (complex128 M = 16384: 34 spilled registers without the fences, 13 with them)
  constexpr bool FENCE = (NT * SPB >= 1024) && (P * (int)sizeof(Cx<T>) >= 256);
  extern __shared__ __attribute__((aligned(16))) char xm_smem[];
  const int t = threadIdx.x % NT;
  const int ls = threadIdx.x / NT;
  const long long s = (long long)blockIdx.x * SPB + ls;
  const bool live = s < A.n_batch;
  const int n = A.n;
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(xm_smem) + (size_t)ls * BlockFFT<T, PL>::lds_elems();

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
    if constexpr (FENCE) {
      if (q % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  BlockFFT<T, PL>::run(v, lds, A.tw, t);
#pragma unroll
  for (int q = 0; q < P; ++q) {
    v[q] = conj(v[q] * A.aux2[t + NT * q]);  // conj -> inverse via forward
    if constexpr (FENCE) {
      if (q % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if constexpr (PL::K > 1) __syncthreads();
  // the second transform reads its twiddles again: with the same pointer the compiler keeps the first transform's
  // twiddle registers alive for it (complex128 M = 16384: 117 spilled VGPRs, stored in one FFT and reloaded in the other)
  const Cx<T>* tw2 = A.tw;
  asm volatile("" : "+s"(tw2));
  BlockFFT<T, PL>::run(v, lds, tw2, t);

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
    if constexpr (FENCE) {
      if (q % 4 == 3) __builtin_amdgcn_sched_barrier(0);
    }
  }
  if (A.absmax2) {
    T* red_v = reinterpret_cast<T*>(xm_smem);
    int* red_i = reinterpret_cast<int*>(red_v + (NT * SPB) / XM_WAVE + 1);
    amax_reduce_store<T, NT>(bv, bi, t, live, s, A.absmax2, A.argidx, red_v, red_i);
  }
}

// =================================================================================================
// Persistent Bluestein kernel: the chirp-z of k_bluestein with the structure of k_fft2 / k_fft1.
// V = xm_f2: two spectra ride in the packed-f32 lanes (rows 2g, 2g+1); V = double: one spectrum per pass.
// Per-thread invariants live in registers for the whole launch: the chirp a[pos] (0 beyond n), the
// window weight of each position (0 where the padded input is structurally zero), the chirp spectrum
// aux2[pos] and the last-stage twiddles; the next rows are prefetched while the two FFTs of the current
// ones run.
// =================================================================================================
template <class V, class PL>
constexpr int blue_waves() {
  // ~165 VGPRs of per-thread state: two waves per SIMD at most (four would spill ~50 registers, measured 25 % slower)
  const long lds = (long)BlockFFT<V, PL>::lds_elems() * (long)sizeof(Cx<V>);
  const int wg = 160 * 1024 / lds >= 2 ? 2 : 1;
  int w = wg * PL::NT / 256;
  return w < 1 ? 1 : (w > 2 ? 2 : w);
}

template <class V, class PL, int MODE>
__global__ __launch_bounds__(PL::NT, (blue_waves<V, PL>())) void k_blue(PipeArgs<typename ScalarOf<V>::type> A) {
  using T = typename ScalarOf<V>::type;
  constexpr bool PACKED = sizeof(V) != sizeof(T);
  constexpr int NS = PACKED ? 2 : 1;  // spectra per pass
  constexpr int NT = PL::NT, P = PL::P;
  // M >= 2n - 1 is the next power of two, so n <= M/2: positions t + NT*q with q >= P/2 never hold input
  // samples and never produce outputs -- only the lower half of the per-thread slots carries chirps,
  // window weights, prefetched samples and results
  constexpr int PH = P / 2;
  constexpr bool WRITE = (MODE & ZF2_WRITE) != 0, PHASE = (MODE & ZF2_PHASE) != 0, AMAX = (MODE & ZF2_AMAX) != 0;
  using FFT = BlockFFT<V, PL>;
  using HT = HotTw<T, PL>;
  extern __shared__ __attribute__((aligned(16))) char xm_smem[];
  Cx<V>* lds = reinterpret_cast<Cx<V>*>(xm_smem);
  Cx<T>* mid = reinterpret_cast<Cx<T>*>(lds + FFT::lds_elems());
  T* red_v = reinterpret_cast<T*>(mid + HT::mid_lds_size());
  int* red_i = reinterpret_cast<int*>(red_v + NS * (NT / XM_WAVE + 1));
  const int t = threadIdx.x;
  const int n = A.n;

  HT tw;
  tw.load(A.tw, t);
  if constexpr (HT::mid_in_lds()) {
    tw.mid = mid;
    for (int i = t; i < HT::mid_size(); i += NT) mid[i] = A.tw[i];
  } else {
    tw.mid = A.tw;
  }
  // position pos = t + NT*q of the length-M convolution input holds padded sample j = (pos - in_shift) mod n
  // = input sample j - pad_left, for pos < n
  Cx<T> ca[PH], cb[P];
  T wg[PH];
#pragma unroll
  for (int q = 0; q < P; ++q) cb[q] = A.aux2[t + NT * q];
#pragma unroll
  for (int q = 0; q < PH; ++q) {
    const int pos = t + NT * q;
    ca[q] = pos < n ? A.aux[pos] : mk<T>(T(0), T(0));
    int j = pos - A.in_shift;
    if (j < 0) j += n;
    const bool ok = pos < n && (unsigned)(j - A.pad_left) < (unsigned)A.n_in;
    wg[q] = ok ? (A.window ? A.window[j] : T(1)) : T(0);
  }
  __syncthreads();

  const long long ngroups = (A.n_batch + NS - 1) / NS;
  const unsigned last_in = (unsigned)A.n_in - 1u;
  Cx<T> xr[NS][PH];
  auto fetch = [&](long long g, int tt, int shift, int padl, unsigned lastv, int nn) {
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      long long s = NS * g + u;
      if (s >= A.n_batch) s = NS * g;  // odd tail: duplicate row
      const Cx<T>* __restrict__ row = A.in + s * A.in_stride;
#pragma unroll
      for (int q = 0; q < PH; ++q) {
        int j = tt + NT * q - shift;
        if (j < 0) j += nn;
        xr[u][q] = row[min((unsigned)(j - padl), lastv)];  // clamped; positions without data have wg = 0
      }
    }
  };
  long long g = blockIdx.x;
  if (g < ngroups) fetch(g, t, A.in_shift, A.pad_left, last_in, n);

  for (; g < ngroups; g += gridDim.x) {
    int tt = t, osh = A.out_shift, ish = A.in_shift, padl = A.pad_left, nn = n;
    unsigned lastv = last_in;
    asm volatile("" : "+v"(tt));  // keep per-lane address arithmetic inside the loop (see k_zf2)
    asm volatile("" : "+s"(osh));
    asm volatile("" : "+s"(ish));
    asm volatile("" : "+s"(padl));
    asm volatile("" : "+s"(lastv));
    asm volatile("" : "+s"(nn));
    Cx<V> v[P];
#pragma unroll
    for (int q = PH; q < P; ++q) {
      v[q].re = V(0);
      v[q].im = V(0);
    }
#pragma unroll
    for (int q = 0; q < PH; ++q) {
      Cx<V> x;
      if constexpr (PACKED) {
        x.re = V{xr[0][q].re, xr[1][q].re};
        x.im = V{xr[0][q].im, xr[1][q].im};
      } else {
        x = xr[0][q];
      }
      if (A.inverse) x.im = -x.im;
      v[q] = (x * ca[q]) * wg[q];
    }
    if (g + gridDim.x < ngroups) fetch(g + gridDim.x, tt, ish, padl, lastv, nn);

    FFT::run(v, lds, tw, tt);
#pragma unroll
    for (int q = 0; q < P; ++q) {
      v[q] = v[q] * cb[q];
      v[q].im = -v[q].im;  // conj -> inverse transform via the forward one
    }
    FFT::run(v, lds, tw, tt);

    // X[m] = conj(v[m]) a[m] scale for m < n, at output index (m + out_shift) mod n
    T bv[NS];
    int bi[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      bv[u] = T(-1);
      bi[u] = 0x7fffffff;
    }
#pragma unroll
    for (int q = 0; q < PH; ++q) {
      const int m = tt + NT * q;
      int k = m + osh;
      if (k >= nn) k -= nn;
      Cx<V> y = v[q];
      y.im = -y.im;
      y = (y * ca[q]) * A.scale;
      if (A.inverse) y.im = -y.im;
      if constexpr (AMAX) {
        const V m2 = y.re * y.re + y.im * y.im;
        if (m < nn) {
          if constexpr (PACKED) {
            amax_take(bv[0], bi[0], m2.x, k);
            amax_take(bv[NS - 1], bi[NS - 1], m2.y, k);
          } else {
            amax_take(bv[0], bi[0], m2, k);
          }
        }
      }
      if constexpr (WRITE) {
        if (m < nn) {
          if constexpr (PHASE) y = y * A.phase[k];
          if constexpr (PACKED) {
            const long long s0 = NS * g, s1 = NS * g + 1;
            A.out[s0 * (long long)nn + k] = mk<T>(y.re.x, y.im.x);
            if (s1 < A.n_batch) A.out[s1 * (long long)nn + k] = mk<T>(y.re.y, y.im.y);
          } else {
            A.out[g * (long long)nn + k] = y;
          }
        }
      }
    }
    if constexpr (AMAX) {
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const long long s = NS * g + u;
        const bool live = s < A.n_batch;
        amax_reduce_store<T, NT>(bv[u], bi[u], tt, live, live ? s : NS * g, A.absmax2, A.argidx,
                                 red_v + u * (NT / XM_WAVE + 1), red_i + u * (NT / XM_WAVE + 1));
      }
    }
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

// out[j] = (complex128) in[row, j] for the row that holds the global arg-max, row = *flat / n_per_row.
// The row index is read from device memory so that the arg-max spectrum can be recomputed in complex128
// without a host round trip between the arg-max reduction and this gather.
template <class T>
__global__ void k_gather_row(const Cx<T>* __restrict__ in, long long in_stride, int n_in,
                             const long long* __restrict__ flat, int n_per_row, Cx<double>* __restrict__ out) {
  const long long row = flat[0] / n_per_row;  // the launcher's caller guarantees 0 <= flat < n_batch * n_per_row
  const Cx<T>* src = in + row * in_stride;
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_in; j += gridDim.x * blockDim.x)
    out[j] = mk<double>((double)src[j].re, (double)src[j].im);
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

// Windowed L1 norm of every FID: norm[b] = sum_j |in[b, j]| * |window[j + pad_left]|  (window == NULL: 1), over the
// 1-KiB blocks (128 complex64 / 64 complex128 samples) whose index is a multiple of `sub_step` (1 = every sample).
// sum_j |z_j| / sqrt(N) bounds every |X[k]| of the row from above and equals the peak height of a single
// decaying resonance, so the row with the largest norm is the natural GUESS for the row that holds the global
// arg-max (autophase's speculative schedule; the guess is verified against the true maxima afterwards, so a
// ranking statistic is all that is needed: a regular subset of whole cache lines spread over the window's support
// ranks rows like the full sum does and reads 1/sub_step of the bytes).
// Streaming read, one wave per row, persistent grid.
template <class T>
__global__ __launch_bounds__(256) void k_row_l1(const Cx<T>* __restrict__ in, long long in_stride,
                                                 const T* __restrict__ window, long long n_batch, int n_in,
                                                 int pad_left, int sub_step, T* __restrict__ norm,
                                                 unsigned long long* gkey) {
  // one WAVE per row (four rows per workgroup): no LDS, no barriers, a shuffle reduction at the end of the row
  constexpr int UN = 8;  // independent 16-byte loads in flight per lane and round (read-only streaming)
  constexpr int PER = 16 / (int)sizeof(Cx<T>);  // samples per 16-byte load: 2 (complex64) or 1 (complex128)
  const bool wide = PER == 1 || ((in_stride & 1) == 0 && (n_in & 1) == 0 && (reinterpret_cast<size_t>(in) & 15) == 0);
  const int lane = threadIdx.x & (XM_WAVE - 1);
  const long long wave = (long long)blockIdx.x * (256 / XM_WAVE) + (threadIdx.x / XM_WAVE);
  const long long nwaves = (long long)gridDim.x * (256 / XM_WAVE);
  // vector index v (16-byte words of a row) -> the v-th word of the sampled subset: blocks of 64 words
  // (128 complex64 samples), every sub_step-th block
  auto word_of = [&](int v) { return sub_step <= 1 ? v : ((v >> 6) * sub_step << 6) + (v & 63); };
  unsigned long long best = 0;  // gkey: best (norm, row) of this wave's rows, merged once at the end
  const int nvec = n_in / PER;
  const int nblk = (nvec + 63) >> 6;
  const int nsub = sub_step <= 1 ? nvec : (((nblk + sub_step - 1) / sub_step) << 6);  // words of the subset (upper bound)
  // the subset usually fits ONE round of UN words per lane: the lane's window weights are then the same for every
  // row and live in registers (re-reading them per row cost as many L2 bytes as the samples themselves)
  const bool one_round = wide && nsub <= XM_WAVE * UN;
  T wreg[UN][PER];
  if (one_round) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int v = lane + XM_WAVE * u, j = word_of(v) * PER;
#pragma unroll
      for (int e = 0; e < PER; ++e)
        wreg[u][e] = (v < nsub && j + e < n_in) ? (window ? fabs(window[j + e + pad_left]) : T(1)) : T(0);
    }
  }
  for (long long b = wave; b < n_batch; b += nwaves) {
    const Cx<T>* __restrict__ row = in + b * in_stride;
    T acc = T(0);
    if (one_round) {
      Cx<T> x[UN][PER];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int v = lane + XM_WAVE * u;
        const int j = min(word_of(v), nvec - 1);  // clamped: words beyond the subset carry weight 0
        const xm_u4 raw = *reinterpret_cast<const xm_u4*>(row + (long long)j * PER);
        __builtin_memcpy(&x[u][0], &raw, 16);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int e = 0; e < PER; ++e) {  // weight 0 = not part of the sum (a clamped duplicate): never 0 * inf
          const T m = sqrt(x[u][e].re * x[u][e].re + x[u][e].im * x[u][e].im);
          acc += wreg[u][e] != T(0) ? m * wreg[u][e] : T(0);
        }
    } else if (wide) {
      for (int j0 = lane; j0 < nsub; j0 += XM_WAVE * UN) {
        Cx<T> x[UN][PER];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int j = word_of(j0 + XM_WAVE * u);
          if (j0 + XM_WAVE * u < nsub && j < nvec) {
            const xm_u4 raw = *reinterpret_cast<const xm_u4*>(row + (long long)j * PER);
            __builtin_memcpy(&x[u][0], &raw, 16);
          } else {
#pragma unroll
            for (int e = 0; e < PER; ++e) x[u][e] = mk<T>(T(0), T(0));
          }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          const int j = word_of(j0 + XM_WAVE * u) * PER;
#pragma unroll
          for (int e = 0; e < PER; ++e) {
            const T m = sqrt(x[u][e].re * x[u][e].re + x[u][e].im * x[u][e].im);
            acc += (window && j + e < n_in) ? m * fabs(window[j + e + pad_left]) : m;
          }
        }
      }
    } else {
      for (int j = lane; j < n_in; j += XM_WAVE) {
        if (sub_step > 1 && ((j >> 7) % sub_step) != 0) continue;
        const Cx<T> x = row[j];
        const T m = sqrt(x.re * x.re + x.im * x.im);
        acc += window ? m * fabs(window[j + pad_left]) : m;
      }
    }
#pragma unroll
    for (int m = XM_WAVE / 2; m >= 1; m >>= 1) acc += shfl_xor_t(acc, m);
    if (lane == 0 && norm) norm[b] = acc;
    if constexpr (sizeof(T) == 4) {
      if (gkey) {
        const unsigned bits = acc != acc ? 0x7fc00000u : __float_as_uint(acc);  // norms are >= 0: ordered like their bits
        const unsigned long long k = ((unsigned long long)bits << 32) | (unsigned long long)(0xffffffffu - (unsigned)b);
        best = k > best ? k : best;
      }
    }
  }
  if constexpr (sizeof(T) == 4) {
    if (gkey && lane == 0 && best != 0) atomicMax(gkey + (blockIdx.x % XM_KEY_SLOTS) * XM_KEY_STRIDE, best);
  }
}

// Consumer of a global arg-max key (see PipeArgs::gkey): out_max2[0] = the value, out_flat[0] = row * n_per_row, and
// the key is left zero for its next producer.  With `in`: also the winning row, upcast to complex128 (the gather of
// k_gather_row).  ONE workgroup.
template <class T>
__global__ __launch_bounds__(1024) void k_key_take(unsigned long long* key, int n_per_row, float* out_max2, long long* out_flat,
                                                    const Cx<T>* __restrict__ in, long long in_stride, int n_in,
                                                    Cx<double>* __restrict__ out_row) {
  __shared__ unsigned long long part[XM_KEY_SLOTS];
  if (threadIdx.x < XM_KEY_SLOTS)
    part[threadIdx.x] = __hip_atomic_load(key + threadIdx.x * XM_KEY_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  unsigned long long k = 0;
  for (int i = 0; i < XM_KEY_SLOTS; ++i) k = part[i] > k ? part[i] : k;
  const long long row = k ? (long long)(0xffffffffu - (unsigned)(k & 0xffffffffu)) : 0;  // nothing published: row 0
  if (in) {
    const Cx<T>* src = in + row * in_stride;
    for (int j = threadIdx.x; j < n_in; j += blockDim.x) out_row[j] = mk<double>((double)src[j].re, (double)src[j].im);
  }
  if (threadIdx.x < XM_KEY_SLOTS)
    __hip_atomic_store(key + threadIdx.x * XM_KEY_STRIDE, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x == 0) {
    out_max2[0] = __uint_as_float((unsigned)(k >> 32));
    out_flat[0] = row * (long long)n_per_row;
  }
}

// (value a, flat index ia) beats (b, ib): NaN outranks numbers, then the larger value, ties -> the lower index
template <class T>
XM_DEV bool amax_final_better(T a, long long ia, T b, long long ib) {
  const bool an = a != a, bn = b != b;
  if (an != bn) return an;
  if (an) return ia < ib;
  return a > b || (a == b && ia < ib);
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
  // One workgroup, so the scan is latency bound: every thread scans the VALUES of its rows with UN
  // independent loads in flight per round and remembers the first row holding its maximum; the index
  // array is read once per thread, for that row only.
  // np.argmax(np.abs(x)) returns the FIRST NaN when there is one (NaN compares as the maximum there): a NaN row
  // maximum (rows whose kernels propagate NaN) outranks every number, first row wins.  Kernels that build the row
  // maxima with fmax / integer atomics never report NaN; such rows carry their largest finite value instead.
  constexpr int UN = 16;
  bool have_nan = false;
  for (long long b0 = threadIdx.x; b0 < n_batch; b0 += 1024 * UN) {
    T v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long b = b0 + 1024LL * u;
      v[u] = b < n_batch ? absmax2[b] : T(-1);
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long long b = b0 + 1024LL * u;
      const bool is_nan = v[u] != v[u];
      if (is_nan ? !have_nan : (!have_nan && v[u] > bv)) {  // strict: keeps the lowest row among equal values
        bv = v[u];
        bi = b;
      }
      have_nan = have_nan || is_nan;
    }
  }
  if (bi == 0x7fffffffffffffffLL && threadIdx.x == 0 && n_batch > 0) {  // nothing above -1 anywhere: row 0
    bv = absmax2[0];
    bi = 0;
  }
  if (bi != 0x7fffffffffffffffLL) bi = bi * (long long)n + argidx[bi];  // one index load per thread
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    T ov = shfl_xor_t(bv, m);
    long long oi = __shfl_xor(bi, m, XM_WAVE);
    if (amax_final_better(ov, oi, bv, bi)) {
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
      if (amax_final_better(red_v[w], red_i[w], bv, bi)) {
        bv = red_v[w];
        bi = red_i[w];
      }
    }
    out_max2[0] = bv;
    out_flat[0] = bi;
  }
}
