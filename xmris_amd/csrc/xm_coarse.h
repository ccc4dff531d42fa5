// k_coarse_mfma -- the coarse spectra of the speculative schedule's guess stage on the matrix cores.
//
// What it computes (reference statement served: phasing.py:229, the ROW of the global arg-max, needed before the spectra
// exist): est[b] = max_k |X_b[k]|^2 * scale^2 with X_b the 1024-bin transform of the first 512 windowed samples of row
// b -- the estimate `xm_guess_refine` ranks its candidates by (it checks every candidate exactly, so the estimate only
// has to be good to a few percent: the band it is compared with has 15 % of margin).
//
// Why matrix cores: the FFT version of this kernel (k_zf2p, OPT ZF2P_EST) is VALU bound -- 398 vector instructions per
// row and wave, 0.075-0.088 ms for 65,536 rows against an HBM floor of 0.045 (it reads 4 KiB of every 32 KiB row).  A
// 1024-point DFT factors into two 32-point stages, and a 32-point DFT of 32 columns at once is a 32 x 32 x 32 matrix
// product -- the native shape of v_mfma_f32_32x32x16_f16:
//     n = 32 n1 + n2 (n1 < 16: only 512 samples are non-zero),  k = k1 + 32 k2
//     D1[n2, k1] = sum_n1 x[32 n1 + n2] W32^(n1 k1)             one k-step:  A = data (row n2, k = n1), B = W32
//     Z [n2, k1] = D1[n2, k1] W1024^(n2 k1)                     element-wise on the accumulator tile
//     D2[k2, k1] = sum_n2 W32^(n2 k2) Z[n2, k1] = X[k1 + 32 k2]  two k-steps: A = W32, B = Z
// The second product sums over Z's ROW index, so the accumulator tile of the first is the B operand of the second
// without any lane movement (cdna_hip_programming.md, "An accumulator tile as the next MFMA's operand"): the k order
// inside a step is permuted, and the constant operand is built with the same permutation.  Complex products are four
// real ones (the sign of -Im rides in a third copy of the constant).  12 MFMAs and ~150 vector instructions per row.
//
// Precision: operands are fp16 (11 bits), sums fp32.  Every row is scaled by a power of two so that its largest
// windowed sample lies in [1, 2) -- raw FIDs can be ADC counts of 1e5 or volts of 1e-6, fp16 spans 6e-8 ... 65504 -- and
// the estimate is scaled back exactly.  Measured against the fp64 DFT: within 2e-3 (tests/test_gpu_kernels.py).
// A NaN sample makes the whole estimate NaN, which outranks every number (np.argmax returns the first NaN).
//
// One wave per row, four waves per workgroup, rows handed out wave-strided in ascending order (ties -> the lower row),
// the next row's samples are loaded while the current one is transformed.  The launch's largest estimate goes into the
// arg-max key exactly as k_zf2p's EST mode leaves it.
#pragma once
#include "xm_kernels.h"

typedef _Float16 xm_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 xm_h2 __attribute__((ext_vector_type(2)));
typedef float xm_f16v __attribute__((ext_vector_type(16)));

struct CoarseArgs {
  const void* in;            // [n_batch, in_stride] complex64 (or complex128: IN64) rows
  const float* window;       // >= 512 weights
  const Cx<float>* w1024;    // W_1024^k, k < 512 (xm_gen_half(1024))
  float* est;                // [n_batch]
  unsigned long long* gkey;  // XM_KEY_SLOTS partial keys
  long long in_stride;
  long long n_batch;
  float scale2;              // the full transform's ortho scale, squared
};

constexpr int kCoarseWaves = 4;  // waves per workgroup

XM_DEV Cx<float> coarse_w1024(const Cx<float>* __restrict__ t, unsigned k) {  // W_1024^k for any k (table holds k < 512)
  k &= 1023u;
  const Cx<float> w = t[k & 511u];
  return (k & 512u) ? mk<float>(-w.re, -w.im) : w;
}

XM_DEV xm_h8 coarse_pack(const float* v) {  // eight floats -> eight halves, round to nearest (v_cvt_pk_f16_f32: one
  union {                                     // instruction per pair; the round-toward-zero form biased the estimates
    xm_h8 v8;                                 // by -0.2 %)
    xm_h2 v2[4];
  } u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const xm_f2 p = {v[2 * j], v[2 * j + 1]};
    u.v2[j] = __builtin_convertvector(p, xm_h2);
  }
  return u.v8;
}

template <bool IN64>  // complex128 rows: converted to float on load (a ranking statistic, like the FFT version's IN64 mode)
__global__ __launch_bounds__(64 * kCoarseWaves, 2) void k_coarse_mfma(CoarseArgs A) {
  const unsigned lane = threadIdx.x & 63u, r = lane & 31u, h = lane >> 5;
  const long long wave = (long long)blockIdx.x * kCoarseWaves + (threadIdx.x >> 6);
  const long long n_waves = (long long)gridDim.x * kCoarseWaves;

  // ---- per-lane constants, held for the whole launch ------------------------------------------------------------
  // B operand of the first product: B1[k = n1 = 8h + j][col = k1 = r] = W32^(n1 k1)
  // A operand of the second, k-step s: A2[row = k2 = r][k = n2], n2 = 16 s + 8 (j >> 2) + 4 h + (j & 3)  (the order in
  // which an accumulator tile presents its rows)
  xm_h8 b1r, b1i, b1n, a2r[2], a2i[2], a2n[2];
  {
    float cr[8], ci[8], cn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const Cx<float> w = coarse_w1024(A.w1024, 32u * ((8u * h + (unsigned)j) * r));
      cr[j] = w.re;
      ci[j] = w.im;
      cn[j] = -w.im;
    }
    b1r = coarse_pack(cr);
    b1i = coarse_pack(ci);
    b1n = coarse_pack(cn);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned n2 = 16u * s + 8u * ((unsigned)j >> 2) + 4u * h + ((unsigned)j & 3u);
        const Cx<float> w = coarse_w1024(A.w1024, 32u * (n2 * r));
        cr[j] = w.re;
        ci[j] = w.im;
        cn[j] = -w.im;
      }
      a2r[s] = coarse_pack(cr);
      a2i[s] = coarse_pack(ci);
      a2n[s] = coarse_pack(cn);
    }
  }
  // twiddles of the accumulator tile: register q is row n2 = (q & 3) + 8 (q >> 2) + 4 h, column k1 = r
  float twr[16], twi[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const unsigned n2 = ((unsigned)q & 3u) + 8u * ((unsigned)q >> 2) + 4u * h;
    const Cx<float> w = coarse_w1024(A.w1024, n2 * r);
    twr[q] = w.re;
    twi[q] = w.im;
  }
  // data as the A operand of the first product: A1[row = n2 = r][k = n1 = 8h + j] = x[32 (8h + j) + r] * window
  float wnd[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) wnd[j] = A.window[32u * (8u * h + (unsigned)j) + r];

  auto fetch = [&](long long s, Cx<float>* x) {
    // (nontemporal: 256 MiB stream through once -- loaded the plain way they push the tables of the kernels that follow
    // out of the L2: the one-workgroup fp64 transform of the selection stage went from 17 to 21 us)
    if constexpr (IN64) {
      const xm_d2* __restrict__ row = reinterpret_cast<const xm_d2*>(A.in) + s * A.in_stride;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const xm_d2 v = __builtin_nontemporal_load(row + 32u * (8u * h + (unsigned)j) + r);
        x[j] = mk<float>((float)v.x, (float)v.y);
      }
      return;
    }
    const xm_f2* __restrict__ row = reinterpret_cast<const xm_f2*>(A.in) + s * A.in_stride;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#if !defined(XM_COARSE_NT) || XM_COARSE_NT
      const xm_f2 v = __builtin_nontemporal_load(row + 32u * (8u * h + (unsigned)j) + r);
#else
      const xm_f2 v = row[32u * (8u * h + (unsigned)j) + r];
#endif
      x[j] = mk<float>(v.x, v.y);
    }
  };

  unsigned best_key = 0u, best_row = 0u;
  bool have = false;
  // rows loaded ahead of the one being transformed (XM_COARSE_DEPTH, XM_COARSE_NT: compile-time A/B switches).
  // Standalone on 65,536 rows, three rounds (profiles/r04/coarse_kernel.txt): depth 1 plain loads 56.6-57.4 us, depth 1
  // nontemporal 60.0-60.8, depth 2 plain 58.4-58.6, depth 2 nontemporal 62.2-63.7 -- a second row in flight buys
  // nothing (8 waves per CU already hold 32 KiB), and the nontemporal hint costs 4 us here but gives the kernels
  // queued behind it their tables back (in the stream: refine 36.1 -> 33.7 us, the fp64 transform 21.2 -> 17.3).
#ifndef XM_COARSE_DEPTH
#define XM_COARSE_DEPTH 1
#endif
  constexpr int DEPTH = XM_COARSE_DEPTH;
  Cx<float> nx[DEPTH][8];
  long long s = wave;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
    if (s + d * n_waves < A.n_batch) fetch(s + d * n_waves, nx[d]);
  auto one_row = [&](Cx<float>* cur, long long s) {
    float re[8], im[8];
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      re[j] = cur[j].re * wnd[j];
      im[j] = cur[j].im * wnd[j];
      m = fmaxf(m, fmaxf(fabsf(re[j]), fabsf(im[j])));  // (fmax drops NaNs: they reach the result through the products)
    }
    if (s + DEPTH * n_waves < A.n_batch) fetch(s + DEPTH * n_waves, cur);
    // power-of-two row scale: the largest sample into [1, 2)
    unsigned e = (wave_reduce_u32<true>(__float_as_uint(m)) >> 23) & 0xffu;
    e = e > 253u ? 253u : (e < 1u ? 1u : e);
    const float up = __uint_as_float((254u - e) << 23);  // 2^(127 - e)
    const float down = __uint_as_float(e << 23);         // 2^(e - 127)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      re[j] *= up;
      im[j] *= up;
    }
    const xm_h8 ar = coarse_pack(re), ai = coarse_pack(im);
    xm_f16v d1r = {0.f}, d1i = {0.f};
    d1r = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar, b1r, d1r, 0, 0, 0);
    d1r = __builtin_amdgcn_mfma_f32_32x32x16_f16(ai, b1n, d1r, 0, 0, 0);
    d1i = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar, b1i, d1i, 0, 0, 0);
    d1i = __builtin_amdgcn_mfma_f32_32x32x16_f16(ai, b1r, d1i, 0, 0, 0);
    float zr[16], zi[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      zr[q] = d1r[q] * twr[q] - d1i[q] * twi[q];
      zi[q] = d1r[q] * twi[q] + d1i[q] * twr[q];
    }
    xm_f16v d2r = {0.f}, d2i = {0.f};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const xm_h8 br = coarse_pack(zr + 8 * k), bi = coarse_pack(zi + 8 * k);
      d2r = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2r[k], br, d2r, 0, 0, 0);
      d2r = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2n[k], bi, d2r, 0, 0, 0);
      d2i = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2r[k], bi, d2i, 0, 0, 0);
      d2i = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2i[k], br, d2i, 0, 0, 0);
    }
    float bv = -1.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) bv = fmaxf(bv, d2r[q] * d2r[q] + d2i[q] * d2i[q]);
    bv = amax_nan_if_unset(bv);
    bv = bv * (down * down) * A.scale2;  // (NaN stays NaN; an all-zero row stays 0)
    const unsigned key = wave_reduce_u32<true>(bv != bv ? 0x7fc00000u : __float_as_uint(bv));
    const unsigned row = (unsigned)s;
    const bool take = !have || key > best_key;  // rows ascend within a wave: strict > keeps the lowest of equals
    best_key = take ? key : best_key;
    best_row = take ? row : best_row;
    have = true;
    if (lane == 0u) A.est[s] = __uint_as_float(key);
  };
  for (; s < A.n_batch; s += DEPTH * n_waves) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
      if (s + d * n_waves < A.n_batch) one_row(nx[d], s + d * n_waves);
  }
  if (A.gkey && have && lane == 0u)
    atomicMax(A.gkey + (blockIdx.x % XM_KEY_SLOTS) * XM_KEY_STRIDE,
              ((unsigned long long)best_key << 32) | (unsigned long long)(0xffffffffu - best_row));
}
