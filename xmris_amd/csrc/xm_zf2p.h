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
//   * the guess stage of the speculative schedule runs on the same kernel (round 3):
//       ZF2P_EST  -- "coarse spectra": the first n_in (<= 512) samples of every row through the half-length-512 plan
//                    (one wave per row), max |X|^2 of the 1024-bin transform stored per row (PipeArgs::est) and merged
//                    into the arg-max key.  A truncated, coarsely sampled spectrum underestimates a line's height by a
//                    factor in [~0.8, 1] (scalloping + the missing tail), so every row whose estimate lies within that
//                    band of the largest one is a CANDIDATE for the global arg-max;
//       ZF2P_CAND -- "refine": every workgroup scans its share of the estimates (rows dealt round-robin with a
//                    rotation, so neighbouring bright voxels go to different workgroups), transforms its candidates
//                    exactly (full row, the arithmetic of the main pass) and merges (max |X|^2, row) into a key; the
//                    last workgroup out decodes the key, clears both keys and gathers the winning FID as complex128.
//                    Branch and bound: a workgroup keeps its (at most 16) candidates with the LARGEST estimates and
//                    takes them in descending order; every exact maximum is published (PipeArgs::gbest, atomic max) and
//                    a candidate whose estimate falls below band^2 x the best exact maximum so far is skipped -- on
//                    data whose rows are alike (a phantom: thousands of rows inside the band) the second round already
//                    finds nothing left to do;
//       ZF2P_IN64 -- rows are complex128 in memory, converted to float on load (ranking statistics for the
//                    complex128 schedule: the verification against the fp64 main pass stays exact).
#pragma once
#include "xm_kernels.h"

enum { ZF2P_LOAD16 = 1, ZF2P_NT = 2, ZF2P_QUEUE = 8, ZF2P_EST = 16, ZF2P_CAND = 32, ZF2P_IN64 = 64 };  // OPT bits
constexpr int ZF2P_CAND_CAP = 16;      // candidates one workgroup transforms at most
constexpr int ZF2P_CAND_KMAX = 16384;  // row blocks (of gridDim.x rows each) one workgroup can scan: bits of its LDS bitmap
// LDS words of the refine stage behind the workgroup's ticket word `lds_next`: [1, CAP] candidate rows, (CAP, 2 CAP]
// their estimates (descending), then the count, the largest estimate of the launch (float bits) and the scan's bitmap
constexpr int ZF2P_CAND_ROWS = 1, ZF2P_CAND_EST = 1 + ZF2P_CAND_CAP, ZF2P_CAND_COUNT = 1 + 2 * ZF2P_CAND_CAP,
              ZF2P_CAND_TOP = ZF2P_CAND_COUNT + 1, ZF2P_CAND_BITMAP = ZF2P_CAND_TOP + 1;
constexpr size_t zf2p_cand_lds_bytes() { return (ZF2P_CAND_BITMAP + ZF2P_CAND_KMAX / 32) * sizeof(unsigned); }

constexpr int xm_ilog2(int v) {
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

// pad shift of the exchange buffer (one pad element per 2^shift elements; -1: the block FFT's default)
#ifndef ZF2P_SH8
#define ZF2P_SH8 4
#endif
template <class PL, bool L16>
constexpr int zf2p_pad_shift() {
  if (!L16) return -1;
  return PL::radix(0) == 8 ? ZF2P_SH8 : xm_ilog2(2 * PL::radix(0));
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
  constexpr bool EST = (OPT & ZF2P_EST) != 0, CAND = (OPT & ZF2P_CAND) != 0, IN64 = (OPT & ZF2P_IN64) != 0;
  static_assert(!EST || (NT == XM_WAVE && AMAX && !WRITE), "the per-row estimate is stored by the row's one wave");
  static_assert(!CAND || (AMAX && !WRITE && (OPT & ZF2P_QUEUE) == 0), "candidates: maxima only, static order");
  static_assert(!(IN64 && L16), "pair loads are complex64 loads");
  static_assert(!(PHASE && RAMP), "phase table and ramp are exclusive");
  static_assert(!RAMP || WRITE, "a ramp needs an output");
  static_assert(P % 2 == 0 && NT >= XM_WAVE, "pair loads need an even number of points per thread, whole waves");
  using FFT = BlockFFT<V, PL, zf2p_pad_shift<PL, L16>()>;
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

  // rows this workgroup iterates over: all of the launch's (static stride or the queue), or its own candidates
  long long n_rows = A.n_batch;
  unsigned* cand = lds_next + ZF2P_CAND_ROWS;
  const unsigned n_in = (unsigned)A.n_in;
  if constexpr (CAND) {
    float* cand_e = reinterpret_cast<float*>(lds_next + ZF2P_CAND_EST);
    unsigned* cand_n = lds_next + ZF2P_CAND_COUNT;
    unsigned* top_bits = lds_next + ZF2P_CAND_TOP;
    unsigned* bitmap = lds_next + ZF2P_CAND_BITMAP;
    const unsigned G = gridDim.x, b = blockIdx.x;
    const unsigned kmax = (unsigned)((A.n_batch + G - 1) / G);  // <= ZF2P_CAND_KMAX (checked by the launcher)
    for (unsigned i = t; i < (kmax + 31u) / 32u; i += NT) bitmap[i] = 0u;
    if (t < XM_WAVE) {  // the largest estimate of the launch: the high word of the guess pass's key
      const unsigned long long k = __hip_atomic_load(A.gkey_in + t * XM_KEY_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned hi = wave_reduce_u32<true>((unsigned)(k >> 32));
      if (t == 0) *top_bits = hi;
    }
    __syncthreads();
    const float thr = A.band2 * __uint_as_float(*top_bits);
    // row of (workgroup b, block k): k G + (b + 37 k) mod G -- a bijection per block, rotated from block to block
    for (unsigned k = t; k < kmax; k += NT) {
      const long long r = (long long)k * G + (b + 37u * k) % G;
      if (r < A.n_batch) {
        const float e = A.est[r];
        if (e >= thr || e != e) atomicOr(&bitmap[k >> 5], 1u << (k & 31u));  // a NaN row is always a candidate
      }
    }
    __syncthreads();
    if (t == 0) {  // the ZF2P_CAND_CAP largest estimates (a NaN first), descending; equal estimates: the lower row first
      unsigned n = 0;
      for (unsigned w = 0; w < (kmax + 31u) / 32u; ++w) {
        unsigned bits = bitmap[w];
        while (bits) {
          const unsigned k = w * 32u + (unsigned)__builtin_ctz(bits);
          bits &= bits - 1u;
          const unsigned r = (unsigned)((long long)k * G + (b + 37u * k) % G);
          float e = A.est[r];
          e = e != e ? __builtin_inff() : e;
          unsigned pos = n;  // insertion point: behind everything >= e (rows arrive in ascending order)
          while (pos > 0 && cand_e[pos - 1] < e) --pos;
          if (pos >= (unsigned)ZF2P_CAND_CAP) continue;
          const unsigned top = n < (unsigned)ZF2P_CAND_CAP ? n : (unsigned)ZF2P_CAND_CAP - 1u;
          for (unsigned i = top; i > pos; --i) {
            cand[i] = cand[i - 1];
            cand_e[i] = cand_e[i - 1];
          }
          cand[pos] = r;
          cand_e[pos] = e;
          n = n < (unsigned)ZF2P_CAND_CAP ? n + 1u : n;
        }
      }
      *cand_n = n;
    }
    __syncthreads();
    n_rows = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)*cand_n);
  }
  // a workgroup without candidates skips the table loads (it still counts itself out at the end)
  const bool live = !CAND || n_rows > 0;

  HT tw;
  tw.mid = HT::mid_in_lds() ? mid : A.tw;
  if (live) {
    tw.load(A.tw, (int)t);
    for (unsigned i = t; i < (unsigned)HT::mid_lds_size(); i += NT) mid[i] = A.tw[i];
  }
  if constexpr (RAMP) {  // per-thread part of the output phase, e^{i b 2t}, folded into the last-stage twiddles
    double sn, cs;
    sincos(A.ramp_db * (double)(2u * t), &sn, &cs);
    tw.fold(mk<T>((T)cs, (T)sn));
  }
  Cx<T> rot = live ? A.aux[col] : mk<T>(T(0), T(0));  // W_N^col
  if constexpr (RAMP) rot = rot * mk<T>(A.ramp_e[0], A.ramp_e[1]);  // odd bins: e^{i b (k + 1)} = e^{i b k} e^{i b}
  const unsigned coff = col - (unsigned)A.pad_left;  // wraps for col < pad_left -> fails the range test
  T w[P];  // window sample * FFT scale; 0 outside the acquired samples (the zero fill)
#pragma unroll
  for (int q = 0; q < P; ++q) {
    const bool ok = live && (coff + NT * q) < n_in;
    w[q] = ok ? (A.window ? A.window[col + NT * q] * A.scale : A.scale) : T(0);
  }
  __syncthreads();

  // FID sample j = col + NT*q of a row lives at row[j - pad_left]; positions outside the acquired samples read a
  // clamped (valid) address and are zeroed by their window weight.
  constexpr int NRAW = L16 ? P / 2 : P;
  typedef unsigned raw_t __attribute__((ext_vector_type((L16 || IN64) ? 4 : 2)));
  raw_t raw[NRAW];
  // L16: element offset of this lane's pair for j = 0: columns (c0, c0 + 1) of row `half`
  const unsigned e0 = (t & ~(XM_WAVE - 1u)) + 2u * (lane & 31u) + NT * half - (unsigned)A.pad_left;
  auto fetch = [&](long long s2, unsigned ee0, unsigned cc, unsigned nin) {
    constexpr size_t EB = IN64 ? 16u : 8u;  // bytes per stored sample
    const char* __restrict__ row = reinterpret_cast<const char*>(A.in) + (size_t)(s2 * A.in_stride) * EB;
    if constexpr (L16) {
#pragma unroll
      for (int j = 0; j < P / 2; ++j) {
        const unsigned e = min(ee0 + 2u * NT * j, nin - 2u);
        // (streamed once: nontemporal -- main kernel 1.046-1.054 -> 1.037-1.045 ms in a same-box A/B, within its noise)
        raw[j] = __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(row + (size_t)e * 8u));
      }
    } else {
#pragma unroll
      for (int q = 0; q < P; ++q) {
        const unsigned e = min(cc + NT * q, nin - 1u);
        raw[q] = *reinterpret_cast<const raw_t*>(row + (size_t)e * EB);
      }
    }
  };
  // A.gkey: this wave's best (max |X|^2 bits, row), wave-uniform.  `have`: an all-zero launch still publishes its
  // first row (np.argmax of zeros is 0; a key of value 0 is non-zero through its row word)
  unsigned best_key = 0, best_row = 0;
  bool have = false;
  auto rowid = [&](long long v) -> long long {  // iteration index -> row of the batch
    if constexpr (CAND) return (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)cand[v]);
    else return v;
  };
  unsigned gb_seen = 0;  // the best exact maximum (float bits) this wave has read so far
  // CAND: the first candidate at or behind v that the best exact maximum found so far (by any workgroup) does not rule out
  auto next_live = [&](long long v) -> long long {
    if constexpr (CAND) {
      const float* cand_e = reinterpret_cast<const float*>(lds_next + ZF2P_CAND_EST);
      // (64 partial bounds on cache lines of their own, like the keys: one address takes ~90 atomics per microsecond)
      const unsigned gb = wave_reduce_u32<true>(__hip_atomic_load(A.gbest + lane * (2u * XM_KEY_STRIDE), __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT));
      gb_seen = gb;
      const float lim = A.band2 * __uint_as_float(gb);  // NaN (a NaN row was found): nothing is ruled out
      while (v < n_rows && __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(cand_e[v]))) < lim) ++v;
    }
    return v;
  };
  // Work is handed out in CHUNKS of `ch` consecutive rows (one ticket per chunk: a single counter takes ~90 tickets
  // per microsecond, short rows would outrun it).  c_cur = the chunk being transformed, c_nxt = the one after it
  // (its first row is prefetched during the last row of c_cur): static stride for the first round, then claimed.
  const long long ch = (!CAND && A.queue_chunk > 0) ? A.queue_chunk : 1;
  const long long c_first = CAND ? 0 : blockIdx.x, c_step = CAND ? 1 : gridDim.x;
  long long c_cur = c_first, c_nxt = c_cur + c_step, c_nn = c_nxt + c_step;
  long long s = c_cur * ch;
  unsigned off = 0;  // row within the chunk
  if constexpr (CAND) {
    // A workgroup whose best candidate is not among the front runners (estimate below 0.9^2 of the largest one) naps
    // for about ten microseconds before it starts: by then the front runners have published their exact maxima and
    // the bound usually rules out everything this workgroup holds (rows of one spectral shape: thousands of
    // candidates inside the band, one exact transform needed).  s_sleep idles the wave, it does not poll.
    if (n_rows > 0) {
      const float* cand_e = reinterpret_cast<const float*>(lds_next + ZF2P_CAND_EST);
      const float top = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(cand_e[0])));
      const float emax2 = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)lds_next[ZF2P_CAND_TOP]));
      if (top < 0.81f * emax2) {  // (false for NaN on either side: NaN rows start at once)
#pragma unroll 1
        for (int i = 0; i < 3; ++i) __builtin_amdgcn_s_sleep(127);
        if (t < XM_WAVE) {
          const long long nx = next_live(0);
          if (t == 0u) *lds_next = (unsigned)nx;
        }
        __syncthreads();
        s = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)*lds_next);
        __syncthreads();
      }
    }
  }
  if (s < n_rows) fetch(rowid(s), e0, coff, n_in);
  if constexpr (QUEUE) {
    if (t == 0) *lds_next = atomicAdd(A.queue, 1u) + gridDim.x;
    __syncthreads();
    c_nxt = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)*lds_next);
    __syncthreads();
  }

  while (s < n_rows) {
    const bool chunk_end = (off + 1 == (unsigned)ch) || (s + 1 >= n_rows);
    // the row prefetched during this transform.  CAND: no prefetch -- the successor is chosen late in the transform (the
    // bound is fresher then), by the first wave for the whole workgroup, and handed over through LDS like a queue ticket
    long long s_nxt = chunk_end ? c_nxt * ch : s + 1;
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
    } else if constexpr (IN64) {
#pragma unroll
      for (int q = 0; q < P; ++q) {
        Cx<double> d;
        __builtin_memcpy(&d, &raw[q], 16);
        xr[q] = mk<T>((T)d.re, (T)d.im);
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
    // (round 3, same-box A/B of four builds, main kernel alone: this order 1.067 ms; every wave waiting for the
    // prefetch right here 1.123; the prefetch issued behind the stores 1.19; the previous row's stores drained before
    // the prefetch is issued 1.082)
    if constexpr (!CAND) {
      if (s_nxt < n_rows) fetch(rowid(s_nxt), ee0, cc - pl, nin);
    }
    if constexpr (CAND) {
      FFT::run_cols(v, lds, tw, (int)tt, (int)cc, [&]() {
        if (tt < XM_WAVE) {
          const long long nx = next_live(s + 1);
          if (tt == 0u) *lds_next = (unsigned)nx;
        }
      });
      s_nxt = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)*lds_next);
      if (s_nxt < n_rows) fetch(rowid(s_nxt), ee0, cc - pl, nin);  // (lands while the maxima are reduced)
    } else if constexpr (QUEUE) {
      // first row of a chunk: thread 0 claims the chunk after next right after the prefetch is issued; the ticket
      // is back by the last exchange and goes through LDS in front of one of the transform's own barriers
      const bool claim = off == 0u;  // workgroup-uniform
      // NB the compiler waits for the ticket right here (s_waitcnt vmcnt(0) behind the atomic: its wave-wide rewrite
      // of the atomic reads the value back at once), i.e. the first wave of the workgroup sits out the prefetch just
      // issued.  Moving that wait into the hook (atomic optimiser off, "+ gridDim.x" deferred) made the kernel SLOWER:
      // 1.011 -> 1.110 ms on one box (round 3, same-box A/B of two builds) -- the stall spaces each workgroup's loads
      // from its stores; it stays.
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
          const unsigned r = (unsigned)rowid(s);
          // (candidates come in descending order of their ESTIMATES: equal maxima -> the lower row, explicitly)
          const bool take = !have || key > best_key || (CAND && key == best_key && r < best_row);
          if constexpr (CAND) {
            // a lower bound of this row's maximum: rules out weaker candidates (published only when it raises the bound)
            if (lane == 0u && key > gb_seen) atomicMax(A.gbest + (blockIdx.x % XM_KEY_SLOTS) * (2u * XM_KEY_STRIDE), key);
          }
          best_key = take ? key : best_key;
          best_row = take ? r : best_row;
          have = true;
          if constexpr (EST) {
            if (lane == 0u) A.est[r] = __uint_as_float(key);
          }
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
      if constexpr (!QUEUE) c_nn = c_nxt + c_step;
      off = 0;
    } else {
      ++off;
    }
    s = s_nxt;
  }
  if constexpr (AMAX) {
    if (A.gkey && have && lane == 0u)
      atomicMax(A.gkey + (blockIdx.x % XM_KEY_SLOTS) * XM_KEY_STRIDE,
                ((unsigned long long)best_key << 32) | (unsigned long long)(0xffffffffu - best_row));
  }
  if constexpr (QUEUE || CAND || (AMAX && WRITE)) {  // the last workgroup out leaves the counters at zero for the next launch
    // (static write modes with maxima count out too: their key is decoded here like the queued modes')
    // ... and, asked to (A.key_result / A.take_flat), merges the partial keys into the caller's result record: every
    // wave's atomic is acknowledged (vmcnt) before its workgroup counts itself out, so the last one sees them all
    bool decode = false;
    if constexpr (AMAX) decode = A.gkey && (A.key_result || (CAND && A.take_flat));
    if (decode) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    unsigned last = 0;
    if (t == 0) {
      const unsigned d = atomicAdd(A.queue + 1, 1u);
      last = d == gridDim.x - 1u;
      if (last) {
        __hip_atomic_store(A.queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(A.queue + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if constexpr (CAND) *lds_next = last;
    }
    if constexpr (AMAX) {
      if (decode && t < XM_WAVE) {  // first wave (XM_KEY_SLOTS == 64 == one key per lane)
        last = (unsigned)__builtin_amdgcn_readfirstlane((int)last);
        if (last) {
          unsigned long long k = __hip_atomic_load(A.gkey + t * XM_KEY_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(A.gkey + t * XM_KEY_STRIDE, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if constexpr (CAND) {  // every workgroup has read the guess pass's key by now
            __hip_atomic_store(A.gkey_in + t * XM_KEY_STRIDE, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.gbest + t * (2u * XM_KEY_STRIDE), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
#pragma unroll
          for (int m = XM_WAVE / 2; m >= 1; m >>= 1) {
            const unsigned hi = (unsigned)__shfl_xor((int)(k >> 32), m, XM_WAVE), lo = (unsigned)__shfl_xor((int)(unsigned)k, m, XM_WAVE);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            k = o > k ? o : k;
          }
          if (t == 0) {
            const unsigned row = k ? 0xffffffffu - (unsigned)(k & 0xffffffffu) : 0u;  // nothing published: row 0
            if (A.key_result) {
              A.key_result->max2 = __uint_as_float((unsigned)(k >> 32));
              A.key_result->flat = (long long)row * (long long)N;
            }
            if constexpr (CAND) {
              if (A.take_max2) A.take_max2[0] = __uint_as_float((unsigned)(k >> 32));
              A.take_flat[0] = (long long)row * (long long)N;
              cand[0] = row;
            }
          }
        }
      }
      if constexpr (CAND) {  // ... and gathers the winning FID as complex128 (the whole workgroup)
        if (decode) {
          __syncthreads();
          if (*lds_next != 0u && A.take_row) {
            const size_t r = (size_t)cand[0] * (size_t)A.in_stride;
            for (unsigned j = t; j < n_in; j += NT) {
              if constexpr (IN64) {
                A.take_row[j] = reinterpret_cast<const Cx<double>*>(A.in)[r + j];
              } else {
                const Cx<T> x = A.in[r + j];
                A.take_row[j] = mk<double>((double)x.re, (double)x.im);
              }
            }
          }
        }
      }
    }
  }
}
