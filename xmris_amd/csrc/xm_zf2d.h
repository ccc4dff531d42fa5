// k_zf2d -- the complex128 twin of k_zf2p for the hot shape (4096 -> 8192, ">= 2x end zero fill"); with the 1024-thread
// plan of the half length 8192 (one workgroup per CU, RotTw below) also 8192 -> 16384, which has no in-LDS transform of
// its full length (measured: 2.28 ms for 16,384 rows instead of 10.2 through the long-transform path).
//
// k_zf2<double> (xm_kernels.h) runs the 512-thread x 8-point plan at ~180 VGPRs: ONE workgroup per CU, whose eight
// waves move in lockstep from barrier to barrier -- counters (profiles/r02/pmc_c128_main.txt): VALU issue 15 % of the
// wave cycles, LDS 28 % busy, 40 % waiting; the arg-max-only pass alone (no stores) takes 1.0 ms of the 1.3-1.4.
// Nothing overlaps.  Here: the 256-thread x 16-point plan (radices 16.16.16: two exchanges instead of three), TWO
// workgroups per CU (<= 256 VGPRs each) that fill each other's barrier and load waits.  To fit the registers
//   * the two half transforms run one after the other through the same 16-byte-element exchange buffer: the even-bin
//     half is transformed first and kept (64 VGPRs), the odd-bin half is then formed from the still-live samples;
//   * the 15 last-stage twiddles W_4096^{r t} of a thread are not held (60 VGPRs) but generated as a running product
//     g, g^2, ... from g = W_4096^t (ChainTw: 15 extra complex multiplies per half, ~1e-15 relative);
//   * the next row is not prefetched (its 64 VGPRs do not fit beside both halves; tried in front of the epilogue:
//     190 spilled registers, and sample by sample between the epilogue's stores, each load taking the registers of
//     the pair just stored: 230-350): the other workgroup computes while this one waits for its samples.
// Measured (32,768 rows, MI355X): write + ramp 1.16 ms = 5.57 TB/s (k_zf2<double>: 1.31 with the ramp, 1.44 with
// the table), with the per-row maxima 1.21 (1.43 / 1.60).  k_zf2p's conflict-free column remap was tried: LDS
// conflicts 21 % -> 0 of the LDS cycles, time unchanged (the remapped 16-byte loads fill half a sector per
// quarter wave) -- the plain columns stay.  Nontemporal stores (k_zf2p: +3 %): 1.16 -> 1.98 ms here -- a lane's 32
// bytes leave as two 16-byte stores, and without the cache to merge them every sector is written in halves; with
// neighbouring lanes trading halves first (whole 32-byte sectors per instruction): 1.18 plain, 1.97 nontemporal.
// Rows come from the device-scope queue (xm_kernels.h: WorkQueue), the output phase is the factorised ramp of
// xm_zf2p.h (ZF2_RAMP) or none; maxima, if asked for, per row and value only (ZF2_AMAX | ZF2_VALUE_ONLY).  Without
// ZF2_WRITE (the classic schedule's arg-max pre-pass) each half is reduced to its maximum as soon as it is transformed
// (prefetching the next row beside the second transform still spilled 130 registers: the 16-point butterfly itself
// needs ~100).
#pragma once
#include "xm_kernels.h"

// Twiddle source for plans whose last stage is ONE butterfly per thread: middle stages from the LDS copy, the last
// stage's r-th twiddle as f g^r (f: the folded per-thread unit factor, R0 only), generated in ascending r
template <class S, class PL, bool R0>
struct ChainTw {
  static constexpr bool kHasR0 = R0;
  static constexpr int K = PL::K;
  static_assert(K > 1 && PL::P == PL::radix(K - 1), "one last-stage butterfly per thread");
  static constexpr int mid_size() { return PL::tw_offset(K - 1); }
  const Cx<S>* mid;
  Cx<S> g;      // W^{t}: the table's r = 1 entry of this thread's butterfly
  Cx<S> r0[1];  // f
  mutable Cx<S> cur;
  XM_DEV void load(const Cx<S>* __restrict__ tw, int t) { g = tw[PL::tw_offset(K - 1) + t]; }
  XM_DEV void fold(Cx<S> f) {
    static_assert(R0, "fold needs the r = 0 slot");
    r0[0] = f;
  }
  template <int ST, int U, int R1>
  XM_DEV Cx<S> get(int k) const {
    if constexpr (ST == K - 1) {
      if constexpr (R1 == 1) {
        if constexpr (R0) cur = r0[0] * g; else cur = g;
      } else {
        cur = cur * g;
      }
      return cur;
    } else {
      return mid[PL::tw_offset(ST) + (R1 - 1) * PL::ns(ST) + k];
    }
  }
};

// waves per SIMD: two workgroups of 256 threads, or one of 1024 (half length 8192: the exchange buffer fills the LDS)
template <class PL>
constexpr int zf2d_waves() {
  return PL::NT >= 1024 ? 4 : 2;
}

// Twiddle source for plans whose last stage is radix 2 (P/2 butterflies per thread): the twiddle of butterfly u is
// W_N^{t + NT u} = g W_P^u with g = W_N^t -- one register pair and compile-time rotations instead of P/2 pairs (and, with
// the folded unit factor f, P/2 more): the 1024-thread plan has 128 VGPRs for everything
template <class S, class PL, bool R0>
struct RotTw {
  static constexpr bool kHasR0 = R0;
  static constexpr int K = PL::K;
  static_assert(K > 1 && PL::radix(K - 1) == 2, "last stage radix 2");
  static constexpr int mid_size() { return PL::tw_offset(K - 1); }
  const Cx<S>* mid;
  Cx<S> fg;               // f g
  Cx<S> r0[PL::P / 2];    // f (all alike: the compiler keeps one)
  XM_DEV void load(const Cx<S>* __restrict__ tw, int t) {
    fg = tw[PL::tw_offset(K - 1) + t];
    if constexpr (R0) {
#pragma unroll
      for (int u = 0; u < PL::P / 2; ++u) r0[u] = mk<S>(S(1), S(0));
    }
  }
  XM_DEV void fold(Cx<S> f) {
    static_assert(R0, "fold needs the r = 0 slot");
    fg = fg * f;
#pragma unroll
    for (int u = 0; u < PL::P / 2; ++u) r0[u] = f;
  }
  template <int ST, int U, int R1>
  XM_DEV Cx<S> get(int k) const {
    if constexpr (ST == K - 1)
      return mul_w<U, PL::P, S>(fg);
    else
      return mid[PL::tw_offset(ST) + (R1 - 1) * PL::ns(ST) + k];
  }
};

template <class PL, int MODE>
__global__ __launch_bounds__(PL::NT, (zf2d_waves<PL>())) void k_zf2d(PipeArgs<double> A) {
  using T = double;
  constexpr unsigned N = 2 * PL::N, NT = PL::NT;
  constexpr int P = PL::P;
  constexpr bool WRITE = (MODE & ZF2_WRITE) != 0, RAMP = (MODE & ZF2_RAMP) != 0, AMAX = (MODE & ZF2_AMAX) != 0;
  static_assert((WRITE || AMAX) && !(MODE & ZF2_PHASE) && (WRITE || !RAMP), "no phase table; a ramp needs an output");
  static_assert(!AMAX || (MODE & ZF2_VALUE_ONLY), "maxima: value only");
  constexpr bool GKEY = (MODE & ZF2_GKEY) != 0;
  // ZF2_DMA: the next row's samples are prefetched by the memory system itself -- `global_load_lds_dwordx4`, no
  // destination registers -- into the exchange buffer, which is idle from the last exchange of the second transform
  // until the first one of the next row: the loads are in flight during the epilogue (maxima + stores).  Sample
  // t + NT q of the row lands at byte 16 (NT q + t): every thread reads back exactly what its own lane fetched, so only
  // the counter wait is needed -- and ONE extra barrier per row, between the last thread's read of its staged samples
  // and the first stage-0 scatter into the same bytes.
  constexpr bool DMA = (MODE & ZF2_DMA) != 0;
  static_assert(!GKEY || AMAX, "the key holds maxima");
  using FFT = BlockFFT<T, PL>;
  // one last-stage butterfly per thread: generated twiddles; P/2 radix-2 ones: one twiddle + compile-time rotations
  using TW = typename std::conditional<P == PL::radix(PL::K - 1), ChainTw<T, PL, RAMP>, RotTw<T, PL, RAMP>>::type;
  constexpr bool MID_LDS = TW::mid_size() * (int)sizeof(Cx<T>) <= 8192;
  extern __shared__ __attribute__((aligned(16))) char xm_smem[];
  Cx<T>* lds = reinterpret_cast<Cx<T>*>(xm_smem);
  Cx<T>* mid = lds + FFT::lds_elems();
  T* red_v = reinterpret_cast<T*>(mid + (MID_LDS ? TW::mid_size() : 0));
  int* red_i = reinterpret_cast<int*>(red_v + NT / XM_WAVE + 1);
  unsigned* wq_slot = reinterpret_cast<unsigned*>(red_i + NT / XM_WAVE + 1);
  const unsigned t = threadIdx.x;

  TW tw;
  tw.mid = MID_LDS ? mid : A.tw;  // half length 8192: 64 KB of middle twiddles stay in L2
  tw.load(A.tw, (int)t);
  for (unsigned i = t; i < (unsigned)(MID_LDS ? TW::mid_size() : 0); i += NT) mid[i] = A.tw[i];
  Cx<T> rot = A.aux[t];  // W_N^t
  if constexpr (RAMP) {  // e^{i b 2t} into the last stage, the odd bins' e^{i b} into their rotation (xm_zf2p.h)
    double sn, cs;
    sincos(A.ramp_db * (double)(2u * t), &sn, &cs);
    tw.fold(mk<T>(cs, sn));
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
  __syncthreads();

  constexpr unsigned CB = sizeof(Cx<T>);
  // XM_AMAX_GLOBAL_KEY: this wave's best (max |X|^2 bits, row) over the rows its workgroup transforms; rows come in
  // ascending order (the queue's tickets do), so a strict > keeps the lowest row among equal maxima.  64 bits of
  // value + a row do not fit one atomic: every wave leaves its pair in a slot of its own and the last workgroup out
  // merges the slots (below).
  unsigned long long best_bits = 0;
  unsigned best_row = 0;
  bool have = false;
  WorkQueue wq;
  wq.init(A.queue, A.queue_chunk, wq_slot, A.n_batch, t);
  Cx<T> xr[P];
  auto fetch = [&](long long s2, unsigned tt, unsigned nin, unsigned pl) {
    const Cx<T>* __restrict__ row = A.in + s2 * A.in_stride;
    if (nin == NT * P && pl == 0u) {  // exactly half full: no clamp, scalar row base + one 32-bit lane offset
      const unsigned lane_off = tt * CB;
#pragma unroll
      for (int q = 0; q < P; ++q)
        xr[q] = *reinterpret_cast<const Cx<T>*>(reinterpret_cast<const char*>(row + NT * q) + lane_off);
    } else {
      const unsigned toff2 = tt - pl;
#pragma unroll
      for (int q = 0; q < P; ++q) xr[q] = row[min(toff2 + NT * q, nin - 1u)];
    }
  };
  // ZF2_DMA: row s2 -> the staging bytes of the exchange buffer
  auto stage_row = [&](long long s2, unsigned tt, unsigned nin, unsigned pl) {
    const Cx<T>* __restrict__ row = A.in + s2 * A.in_stride;
    char* stage = reinterpret_cast<char*>(lds) + (tt & ~(XM_WAVE - 1u)) * CB;  // wave-uniform; + lane * 16 by the hardware
    const unsigned toff2 = tt - pl;
#pragma unroll
    for (int q = 0; q < P; ++q) {
      const Cx<T>* src = row + ((nin == NT * P && pl == 0u) ? (tt + NT * q) : min(toff2 + NT * q, nin - 1u));
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                       (void __attribute__((address_space(3)))*)(stage + (size_t)q * NT * CB), 16, 0, 0);
    }
  };
  if constexpr (DMA) {
    if (wq.item < A.n_batch) stage_row(wq.item, t, n_in, (unsigned)A.pad_left);
  }
  for (; wq.item < A.n_batch; wq.advance()) {
    const long long s = wq.item;
    // opaque copies: keep the (loop-invariant) address arithmetic inside the loop instead of in hoisted registers
    unsigned tt = t, sh = (unsigned)A.out_shift, nin = n_in, pl = (unsigned)A.pad_left;
    asm volatile("" : "+v"(tt));
    asm volatile("" : "+s"(sh));
    asm volatile("" : "+s"(nin));
    asm volatile("" : "+s"(pl));
    if constexpr (DMA) {
      // the compiler does not tie the LDS reads below to the global_load_lds that filled the bytes: wait for the
      // counter explicitly (this also drains the previous row's stores -- as the first load-dependent instruction of
      // the variant without the prefetch does, the counter returns in order)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const Cx<T>* st = lds + tt;  // (staged: plain layout, no pads)
#pragma unroll
      for (int q = 0; q < P; ++q) xr[q] = st[NT * q];
      __syncthreads();
    } else {
      fetch(s, tt, nin, pl);
    }
    const unsigned ticket = wq.claim(tt);  // the chunk after next; back long before the second transform's hook

    Cx<T> e[P], h[P];
#pragma unroll
    for (int q = 0; q < P; ++q) e[q] = xr[q] * w[q];
    FFT::run(e, lds, tw, (int)tt);  // even bins: FFT_H(z)
    T bv = T(-1);
    if constexpr (!WRITE) {
#pragma unroll
      for (int q = 0; q < P; ++q) bv = fmax(bv, e[q].re * e[q].re + e[q].im * e[q].im);
    }
    static_for<0, P>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      h[q] = mul_w<q, 2 * P, T>((xr[q] * w[q]) * rot);
    });
    FFT::run_cols(h, lds, tw, (int)tt, (int)tt, [&]() { wq.publish(tt, ticket); });  // odd bins: FFT_H(z W_N^k)
    wq.collect();
    if constexpr (DMA) {  // the exchange buffer is idle now: the next row streams into it during the epilogue
      const long long s_nxt = wq.next_item();
      if (s_nxt < A.n_batch) stage_row(s_nxt, tt, nin, pl);
    }

    const unsigned t2 = 2u * tt;
    if constexpr (AMAX) {
#pragma unroll
      for (int q = 0; q < P; ++q) {
        if constexpr (WRITE) bv = fmax(bv, e[q].re * e[q].re + e[q].im * e[q].im);
        bv = fmax(bv, h[q].re * h[q].re + h[q].im * h[q].im);
      }
      bv = amax_nan_if_unset(bv);
      if constexpr (GKEY) {
        const unsigned long long kb = wave_reduce_u64_max((unsigned long long)__double_as_longlong(bv));
        const bool take = !have || kb > best_bits;
        best_bits = take ? kb : best_bits;
        best_row = take ? (unsigned)s : best_row;
        have = true;
      } else {
        amax_reduce_store<T, (int)NT>(bv, 0, (int)tt, true, s, A.absmax2, A.argidx, red_v, red_i);
      }
    }
    if constexpr (WRITE) {
      Cx<T>* __restrict__ orow = A.out + s * (long long)N;
      typedef const T __attribute__((address_space(4))) * kptr_t;
      kptr_t rc = (kptr_t)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() +
                           __builtin_offsetof(PipeArgs<T>, ramp_c));
      asm volatile("" : "+s"(rc));
      static_for<0, P>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const unsigned base = (2u * NT * q + sh) & (N - 1u);  // wave-uniform
        const __amdgpu_buffer_rsrc_t rout = xm_rsrc(orow + base, 2u * NT * CB);
        CxPair<T> o;
        o.a = e[q];
        o.b = h[q];
        if constexpr (RAMP) {
          const Cx<T> c = mk<T>(rc[2 * q], rc[2 * q + 1]);
          o.a = o.a * c;
          o.b = o.b * c;
        }
        buf_store(rout, t2 * CB, o);
      });
    }
  }
  if constexpr (GKEY) {
    {
      constexpr unsigned NW = NT / XM_WAVE;
      unsigned long long* slots = A.gkey + XM_KEY_C128_WORD;
      if ((t & (XM_WAVE - 1u)) == 0u) {  // (an idle wave leaves an empty slot: row = all ones loses every tie)
        unsigned long long* sl = slots + 2u * (blockIdx.x * NW + t / XM_WAVE);
        __hip_atomic_store(sl, have ? best_bits : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sl + 1, have ? (unsigned long long)best_row : ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // every wave's slot is acknowledged (vmcnt) before its workgroup counts itself out: the last one out sees them all
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) {
        const unsigned d = atomicAdd(A.queue + 1, 1u);
        const unsigned last = d == gridDim.x - 1u;
        if (last) {
          __hip_atomic_store(A.queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(A.queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *wq_slot = last;
      }
      __syncthreads();
      if (*wq_slot != 0u && A.key_result) {  // the last workgroup out: merge gridDim.x * NW slots
        unsigned long long bb = 0, br = ~0ull;
        for (unsigned i = t; i < gridDim.x * NW; i += NT) {
          const unsigned long long vb = __hip_atomic_load(slots + 2u * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const unsigned long long vr = __hip_atomic_load(slots + 2u * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const bool take = vb > bb || (vb == bb && vr < br);
          bb = take ? vb : bb;
          br = take ? vr : br;
        }
#pragma unroll
        for (int m = XM_WAVE / 2; m >= 1; m >>= 1) {
          const unsigned long long ob = ((unsigned long long)(unsigned)__shfl_xor((int)(bb >> 32), m, XM_WAVE) << 32) |
                                        (unsigned)__shfl_xor((int)(unsigned)bb, m, XM_WAVE);
          const unsigned long long orr = ((unsigned long long)(unsigned)__shfl_xor((int)(br >> 32), m, XM_WAVE) << 32) |
                                         (unsigned)__shfl_xor((int)(unsigned)br, m, XM_WAVE);
          const bool take = ob > bb || (ob == bb && orr < br);
          bb = take ? ob : bb;
          br = take ? orr : br;
        }
        unsigned long long* part = reinterpret_cast<unsigned long long*>(lds);  // (the exchange buffer is idle)
        if ((t & (XM_WAVE - 1u)) == 0u) {
          part[2u * (t / XM_WAVE)] = bb;
          part[2u * (t / XM_WAVE) + 1] = br;
        }
        __syncthreads();
        if (t == 0) {
          for (unsigned w = 1; w < NW; ++w) {
            const unsigned long long ob = part[2u * w], orr = part[2u * w + 1];
            const bool take = ob > bb || (ob == bb && orr < br);
            bb = take ? ob : bb;
            br = take ? orr : br;
          }
          const long long row = br == ~0ull ? 0 : (long long)br;  // nothing published: row 0
          *reinterpret_cast<double*>(A.key_result) = __longlong_as_double((long long)bb);  // (XM_C128: max2 as a double)
          A.key_result->flat = row * (long long)N;
        }
      }
      return;
    }
  }
  wq.finish(t);
}
