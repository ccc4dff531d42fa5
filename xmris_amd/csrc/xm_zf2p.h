// k_zf2p -- second generation of the packed complex64 ">= 2x end zero-fill" kernel (the hot kernel of the path:
// 65,536 x 4096 -> 8192 on the roofline config).  Same mathematics as k_zf2 (xm_kernels.h): N = 2H, the upper half of
// the FFT input is structurally zero, X[2m] = FFT_H(z)[m] and X[2m+1] = FFT_H(z W_N^k)[m] ride in the two lanes of
// the packed-f32 VALU, persistent workgroups, next FID prefetched.  What changed, each measured on MI355X
// (profiles/r02/):
//   * linear output phase (ZF2_RAMP): the autophase ramp e^{i(a + b k)} is exactly linear in the output index, and
//     k = base_q + 2t (+1) for butterfly q of thread t, so it factorises into (per-thread) x (wave-uniform per q):
//     the per-thread factor e^{i b 2t} is folded into the last stage's register twiddles once per launch (one extra
//     complex multiply per spectrum), the odd bins' extra e^{i b} into the odd half's input rotation W_N^col (free),
//     and the wave-uniform e^{i(a + b base_q)} comes from the kernel arguments (SGPR operands of four packed ops).
//     No phase-table loads in the loop: k_zf2 issued one 16-byte L2 load per output pair (as many bytes L2 -> CU
//     as the whole HBM write stream) and every such load had to wait, in vmcnt order, for the stores issued before it;
//   * ZF2P_LOAD16: 16-byte FID loads.  Lanes 0-31 of a wave load (x[c], x[c+1]) of row 2j, lanes 32-63 the same
//     columns of row 2j+1; one v_permlane32_swap per dword pair hands each lane the two rows of its own column.
//     The stage-0 column of lane l is therefore 64w + 2(l mod 32) + l/32 instead of 64w + l (stage 0 has no
//     twiddles and scatters by address, so any thread -> column bijection works), and with one pad element per
//     2 R0 elements both the stage-0 scatter (ds_write_b128, 8-lane groups) and the strided gathers (ds_read_b128,
//     16-lane groups) are bank-conflict free (k_zf2: 22 % of its LDS cycles were conflicts);
//   * ZF2P_NT: spectrum stores carry the nontemporal hint (streaming copy of this traffic pattern: +3 %);
//   * ZF2P_QUEUE: rows are handed out by a device-scope counter instead of the static stride b + k G.  Workgroups
//     of a streaming kernel run at very different speeds on this chip (a pure copy kernel with this traffic
//     pattern: the median workgroup of a static schedule is done 30 % of the kernel time before the last one,
//     tools/stream_lab.hip), so a static split leaves much of the chip idle in the tail.  Thread 0 claims the row
//     after next right after the prefetch is issued; the ticket comes back during the transform and is published
//     through LDS in front of an existing barrier (BlockFFT's hook), so the claim costs no barrier and no stall.
//     The last workgroup to leave resets the counters for the next launch.
#pragma once
#include "xm_kernels.h"

enum { ZF2P_LOAD16 = 1, ZF2P_NT = 2, ZF2P_QUEUE = 8 };  // OPT bits

constexpr int xm_ilog2(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

template <class PL, int MODE, int OPT>
__global__ __launch_bounds__(PL::NT, (zf2_waves<float, PL>())) void k_zf2p(PipeArgs<float> A) {
  using T = float;
  using V = xm_f2;
  constexpr unsigned N = 2 * PL::N, NT = PL::NT;
  constexpr int P = PL::P;
  constexpr bool WRITE = (MODE & ZF2_WRITE) != 0, PHASE = (MODE & ZF2_PHASE) != 0, AMAX = (MODE & ZF2_AMAX) != 0;
  constexpr bool RAMP = (MODE & ZF2_RAMP) != 0;
  constexpr bool L16 = (OPT & ZF2P_LOAD16) != 0;
  constexpr int AUX = (OPT & ZF2P_NT) ? 2 : 0;
  static_assert(!(PHASE && RAMP), "phase table and ramp are exclusive");
  static_assert(!RAMP || WRITE, "a ramp needs an output");
  static_assert(P % 2 == 0 && NT >= XM_WAVE, "pair loads need an even number of points per thread, whole waves");
  using FFT = BlockFFT<V, PL, L16 ? xm_ilog2(2 * PL::radix(0)) : -1>;
  using HT = HotTw<T, PL, RAMP>;
  extern __shared__ __attribute__((aligned(16))) char xm_smem[];
  Cx<V>* lds = reinterpret_cast<Cx<V>*>(xm_smem);
  Cx<T>* mid = reinterpret_cast<Cx<T>*>(lds + FFT::lds_elems());
  T* red_v = reinterpret_cast<T*>(mid + HT::mid_lds_size());  // half length 8192: the 32 KB of middle twiddles stay in L2
  int* red_i = reinterpret_cast<int*>(red_v + NT / XM_WAVE + 1);
  unsigned* lds_next = reinterpret_cast<unsigned*>(red_i + NT / XM_WAVE + 1);
  constexpr bool QUEUE = (OPT & ZF2P_QUEUE) != 0;
  const unsigned t = threadIdx.x;
  const unsigned lane = t & (XM_WAVE - 1), half = lane >> 5;
  // stage-0 column of this thread
  const unsigned col = L16 ? (t & ~(XM_WAVE - 1u)) + 2u * (lane & 31u) + half : t;

  HT tw;
  tw.mid = HT::mid_in_lds() ? mid : A.tw;
  tw.load(A.tw, (int)t);
  for (unsigned i = t; i < (unsigned)HT::mid_lds_size(); i += NT) mid[i] = A.tw[i];
  if constexpr (RAMP) {  // per-thread part of the output phase, e^{i b 2t}, folded into the last-stage twiddles
    double sn, cs;
    sincos(A.ramp_db * (double)(2u * t), &sn, &cs);
    tw.fold(mk<T>((T)cs, (T)sn));
  }
  Cx<T> rot = A.aux[col];  // W_N^col
  if constexpr (RAMP) rot = rot * mk<T>(A.ramp_e[0], A.ramp_e[1]);  // odd bins: e^{i b (k + 1)} = e^{i b k} e^{i b}
  const unsigned n_in = (unsigned)A.n_in;
  const unsigned coff = col - (unsigned)A.pad_left;  // wraps for col < pad_left -> fails the range test
  T w[P];  // window sample * FFT scale; 0 outside the acquired samples (the zero fill)
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const bool ok = (coff + NT * q) < n_in;
    w[q] = ok ? (A.window ? A.window[col + NT * q] * A.scale : A.scale) : T(0);
  }
  __syncthreads();

  // FID sample j = col + NT*q of a row lives at row[j - pad_left]; positions outside the acquired samples read a
  // clamped (valid) address and are zeroed by their window weight.
  constexpr int NRAW = L16 ? P / 2 : P;
  typedef unsigned raw_t __attribute__((ext_vector_type(L16 ? 4 : 2)));
  raw_t raw[NRAW];
  // L16: element offset of this lane's pair for j = 0: columns (c0, c0 + 1) of row `half`
  const unsigned e0 = (t & ~(XM_WAVE - 1u)) + 2u * (lane & 31u) + NT * half - (unsigned)A.pad_left;
  auto fetch = [&](long long s2, unsigned ee0, unsigned cc, unsigned nin) {
    const Cx<T>* __restrict__ row = A.in + s2 * A.in_stride;
    if constexpr (L16) {
#pragma unroll
      for (int j = 0; j < P / 2; ++j) {
        const unsigned e = min(ee0 + 2u * NT * j, nin - 2u);
        raw[j] = *reinterpret_cast<const raw_t*>(reinterpret_cast<const char*>(row) + (size_t)e * 8u);
      }
    } else {
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const unsigned e = min(cc + NT * q, nin - 1u);
        raw[q] = *reinterpret_cast<const raw_t*>(reinterpret_cast<const char*>(row) + (size_t)e * 8u);
      }
    }
  };
  unsigned best_key = 0, best_row = 0;  // A.gkey: this wave's best (max |X|^2 bits, row), wave-uniform
  // Work is handed out in CHUNKS of `ch` consecutive rows (one ticket per chunk: a single counter takes ~90 tickets
  // per microsecond, short rows would outrun it).  c_cur = the chunk being transformed, c_nxt = the one after it
  // (its first row is prefetched during the last row of c_cur): static stride for the first round, then claimed.
  const long long ch = A.queue_chunk > 0 ? A.queue_chunk : 1;
  long long c_cur = blockIdx.x, c_nxt = c_cur + gridDim.x, c_nn = c_nxt + gridDim.x;
  long long s = c_cur * ch;
  unsigned off = 0;  // row within the chunk
  if (s < A.n_batch) fetch(s, e0, coff, n_in);
  if constexpr (QUEUE) {
    if (t == 0) *lds_next = atomicAdd(A.queue, 1u) + gridDim.x;
    __syncthreads();
    c_nxt = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)*lds_next);
    __syncthreads();
  }

  while (s < A.n_batch) {
    const bool chunk_end = (off + 1 == (unsigned)ch) || (s + 1 >= A.n_batch);
    const long long s_nxt = chunk_end ? c_nxt * ch : s + 1;  // the row prefetched during this transform
    // opaque copies: keep the (loop-invariant) address arithmetic inside the loop instead of in ~100 hoisted registers
    unsigned tt = t, cc = col, sh = (unsigned)A.out_shift, nin = n_in, pl = (unsigned)A.pad_left, ee0 = e0;
    asm volatile("" : "+v"(tt));
    asm volatile("" : "+v"(cc));
    asm volatile("" : "+v"(ee0));
    asm volatile("" : "+s"(sh));
    asm volatile("" : "+s"(nin));
    asm volatile("" : "+s"(pl));

    // the prefetched samples -> (even-bin, odd-bin) half-FFT inputs in the two packed lanes
    Cx<T> xr[P];
    if constexpr (L16) {
#pragma unroll
      for (int j = 0; j < P / 2; ++j) {
        // (x, y) = row of this lane's half at column c0, (z, w) = same row at column c0 + 1: lower lanes keep
        // (x, y) and receive the upper lanes' (x, y) [row 2j+1, column c0]; upper lanes keep (z, w) and
        // receive the lower lanes' (z, w) [row 2j, column c0 + 1]
        const auto re = __builtin_amdgcn_permlane32_swap(raw[j].x, raw[j].z, false, false);
        const auto im = __builtin_amdgcn_permlane32_swap(raw[j].y, raw[j].w, false, false);
        xr[2 * j] = mk<T>(__uint_as_float(re[0]), __uint_as_float(im[0]));
        xr[2 * j + 1] = mk<T>(__uint_as_float(re[1]), __uint_as_float(im[1]));
      }
    } else {
#pragma unroll
      for (int q = 0; q < P; ++q) xr[q] = mk<T>(__uint_as_float(raw[q].x), __uint_as_float(raw[q].y));
    }
    Cx<V> v[P];
    static_for<0, P>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      const Cx<T> e = xr[q] * w[q];
      const Cx<T> o = mul_w<q, 2 * P, T>(e * rot);
      v[q].re = V{e.re, o.re};
      v[q].im = V{e.im, o.im};
    });
    if (s_nxt < A.n_batch) fetch(s_nxt, ee0, cc - pl, nin);
    if constexpr (QUEUE) {
      // first row of a chunk: thread 0 claims the chunk after next right after the prefetch is issued; the ticket
      // is back by the last exchange and goes through LDS in front of one of the transform's own barriers
      const bool claim = off == 0u;  // workgroup-uniform
      unsigned ticket = 0;
      if (claim && tt == 0u) ticket = atomicAdd(A.queue, 1u) + gridDim.x;
      FFT::run_cols(v, lds, tw, (int)tt, (int)cc, [&]() {
        if (claim && tt == 0u) *lds_next = ticket;
      });
      if (claim) c_nn = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)*lds_next);
    } else {
      FFT::run_cols(v, lds, tw, (int)tt, (int)cc);
    }

    const unsigned t2 = 2u * tt;
    if constexpr (AMAX) {  // thread-local max |X|^2 first (packed), then the first index holding it
      T bv = T(-1);
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const V m2 = v[q].re * v[q].re + v[q].im * v[q].im;
        bv = fmax(bv, fmax(m2.x, m2.y));
      }
      bv = amax_nan_if_unset(bv);  // a row of NaNs reports NaN (np.argmax returns the first NaN)
      if (A.amax_value_only) {
        // value only: one atomic max per wave into the row's slot (zeroed by the launcher; squared magnitudes
        // order like their bit patterns) -- no LDS slot, no workgroup barrier
        const unsigned key = wave_reduce_u32<true>(__float_as_uint(bv));
        if (A.gkey) {  // rows come in ascending order: strict > keeps the lowest row among equal values
          const bool take = key > best_key;
          best_key = take ? key : best_key;
          best_row = take ? (unsigned)s : best_row;
        } else {
          if ((tt & (XM_WAVE - 1)) == 0u) atomicMax(reinterpret_cast<unsigned*>(A.absmax2 + s), key);
          if (tt == 0u) A.argidx[s] = 0;
        }
      } else {
        int bi = 0;
        if (!A.amax_value_only) {
          bi = 0x7fffffff;
          const bool all_nan = bv != bv;  // then every index of this thread qualifies
#pragma unroll
          for (int q = 0; q < P; ++q) {
            const int k0 = (int)(((2u * NT * q + sh) & (N - 1u)) + t2);
            const V m2 = v[q].re * v[q].re + v[q].im * v[q].im;
            bi = min(bi, ((m2.x == bv) | all_nan) ? k0 : 0x7fffffff);
            bi = min(bi, ((m2.y == bv) | all_nan) ? k0 + 1 : 0x7fffffff);
          }
        }
        amax_reduce_store<T, (int)NT>(bv, bi, (int)tt, true, s, A.absmax2, A.argidx, red_v, red_i);
      }
    }
    if constexpr (WRITE) {
      Cx<T>* __restrict__ orow = A.out + s * (long long)N;
      const __amdgpu_buffer_rsrc_t rph = xm_rsrc(A.phase, PHASE ? N * 8u : 0u);
      // the wave-uniform ramp factors are re-read from the kernel-argument segment (scalar loads, scalar cache)
      // every spectrum: 2 P SGPRs held across the transform would spill
      typedef const T __attribute__((address_space(4))) * kptr_t;
      kptr_t rc = (kptr_t)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() +
                           __builtin_offsetof(PipeArgs<T>, ramp_c));
      asm volatile("" : "+s"(rc));
      static_for<0, P>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const unsigned base = (2u * NT * q + sh) & (N - 1u);  // wave-uniform
        // this butterfly's 2*NT contiguous outputs: descriptor base = row + base (see buf_store in xm_kernels.h)
        const __amdgpu_buffer_rsrc_t rout = xm_rsrc(orow + base, 2u * NT * 8u);
        Cx<V> y = v[q];
        if constexpr (RAMP) {
          // explicit FMAs: every instantiation (with / without the maxima) rounds the product the same way
          const V cr = V{rc[2 * q], rc[2 * q]}, ci = V{rc[2 * q + 1], rc[2 * q + 1]};
          y.re = __builtin_elementwise_fma(v[q].re, cr, -(v[q].im * ci));
          y.im = __builtin_elementwise_fma(v[q].re, ci, v[q].im * cr);
        } else if constexpr (PHASE) {
          const CxPair<T> ph = buf_load(rph, t2 * 8u, base * 8u, (CxPair<T>*)nullptr);
          const V cr = V{ph.a.re, ph.b.re}, ci = V{ph.a.im, ph.b.im};
          y.re = __builtin_elementwise_fma(v[q].re, cr, -(v[q].im * ci));
          y.im = __builtin_elementwise_fma(v[q].re, ci, v[q].im * cr);
        }
        xm_u4 u;
        u.x = __float_as_uint(y.re.x);
        u.y = __float_as_uint(y.im.x);
        u.z = __float_as_uint(y.re.y);
        u.w = __float_as_uint(y.im.y);
        __builtin_amdgcn_raw_buffer_store_b128(u, rout, t2 * 8u, 0, AUX);
      });
    }
    if (chunk_end) {
      c_cur = c_nxt;
      c_nxt = c_nn;
      if constexpr (!QUEUE) c_nn = c_nxt + gridDim.x;
      off = 0;
    } else {
      ++off;
    }
    s = s_nxt;
  }
  if constexpr (AMAX) {
    if (A.gkey && best_key != 0u && lane == 0u)
      atomicMax(A.gkey + (blockIdx.x % XM_KEY_SLOTS) * XM_KEY_STRIDE,
                ((unsigned long long)best_key << 32) | (unsigned long long)(0xffffffffu - best_row));
  }
  if constexpr (QUEUE) {  // the last workgroup out leaves the counters at zero for the next launch
    if constexpr (AMAX) {
      // ... and, asked to (A.key_result), merges the partial keys into the caller's result record: every wave's
      // atomic is acknowledged (vmcnt) before its workgroup counts itself out, so the last one sees them all
      if (A.gkey && A.key_result) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
    }
    unsigned last = 0;
    if (t == 0) {
      const unsigned d = atomicAdd(A.queue + 1, 1u);
      last = d == gridDim.x - 1u;
      if (last) {
        __hip_atomic_store(A.queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if constexpr (AMAX) {
      if (A.gkey && A.key_result && t < XM_WAVE) {  // first wave (XM_KEY_SLOTS == 64 == one key per lane)
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
            A.key_result->max2 = __uint_as_float((unsigned)(k >> 32));
            A.key_result->flat = (long long)(0xffffffffu - (unsigned)(k & 0xffffffffu)) * (long long)N;
          }
        }
      }
    }
  }
}
