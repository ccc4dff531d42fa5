// The in-LDS FFT plans.  X(N, NT, radices...): N points held by NT threads (NT a power of two,
// P = N/NT points per thread absorbs the odd factors); every radix divides P and they multiply to N.
#pragma once
#include "xm_blockfft.h"

// power-of-two lengths (also used as the half-length of the >=2x zero-fill path and as the
// convolution length of the Bluestein path)
#define XM_PLANS_POW2(X)      \
  X(16, 1, 16)                \
  X(32, 4, 8, 4)              \
  X(64, 8, 8, 8)              \
  X(128, 8, 16, 8)            \
  X(256, 16, 16, 16)          \
  X(512, 64, 8, 8, 8)         \
  X(1024, 64, 16, 16, 4)      \
  X(2048, 128, 16, 16, 8)     \
  X(4096, 256, 16, 16, 16)    \
  X(8192, 512, 16, 16, 16, 2)

// half-length plans of the persistent ">= 2x zero-fill" kernel (k_zf2): 8 points per thread keep the
// whole per-thread state (two half-FFTs, prefetched FID, window, last-stage twiddles) under 128 VGPRs,
// i.e. 4 waves per SIMD
#define XM_PLANS_ZF2(X)       \
  X(512, 64, 8, 8, 8)         \
  X(1024, 128, 8, 8, 4, 4)    \
  X(2048, 256, 8, 8, 8, 4)    \
  X(4096, 512, 8, 8, 8, 8)

// plans of the generic persistent kernel (k_fft2, complex64): power-of-two lengths reuse the 8-point
// shapes above (two spectra per lane pair double the per-thread state), plus 8192
#define XM_PLANS_FFT2_EXTRA(X) X(8192, 1024, 8, 8, 8, 8, 2)

// tiny lengths and 3*2^k / 5*2^k lengths.  Where the odd factor rides in the LAST stage (12 or 20 points per thread,
// whose outputs stay in registers) every stage that scatters through the LDS has power-of-two geometry: shifts and
// masks instead of divisions by 20 / 80 / ..., one per-lane base plus immediate offsets (FftPlan::scatter_pow2) -- at
// the price of 11 / 19 last-stage twiddles held in registers.  Measured both ways on one box (round 3, TB/s of the
// staged seam, complex64 / complex128; odd factor first -> last): 2560 2.9 -> 4.1 / 2.9 -> 4.9, 5120 2.8 -> 3.8 /
// 3.0 -> 4.5, 6144 3.9 -> 4.3 / 4.1 -> 4.8, 384 and 640 +4 %; 768, 1280, 1536 and 3072 lose 2-10 % and keep the
// odd factor first.
#define XM_PLANS_OTHER(X)     \
  X(2, 1, 2)                  \
  X(4, 1, 4)                  \
  X(8, 1, 8)                  \
  X(384, 16, 4, 4, 24)        \
  X(768, 64, 12, 4, 4, 4)     \
  X(1536, 128, 12, 4, 4, 4, 2) \
  X(3072, 256, 12, 4, 4, 4, 4) \
  X(6144, 512, 4, 4, 4, 4, 2, 12) \
  X(640, 32, 4, 4, 2, 20)     \
  X(1280, 64, 20, 4, 4, 4)    \
  X(2560, 128, 4, 4, 4, 2, 20) \
  X(5120, 256, 4, 4, 4, 4, 20)

// 1536 on ONE wave per spectrum pair, 24 points per thread, three stages instead of five -- for the plain transform
// (the staged to_spectrum seam, complex64): same-box A/B against the row above 4.83 -> 5.13-5.16 TB/s.  The fused modes
// keep the 12-point plan: with the ramp, the maxima and the prefetch the 24-point kernel needs ~460 VGPRs and the
// configs[4] main pass LOSES 5 % (0.165 -> 0.174 ms).  The same idea loses on 768 (32 threads: 5.1 -> 3.7 TB/s) and
// 3072 (128 threads x 24 points: 4.6 -> 3.1 odd factor first, 4.3 last).  profiles/r04/ab_plans.txt
struct Plan1536Wide {
  using type = FftPlan<1536, 64, 8, 8, 24>;
};
// 16384, complex64: 128 KiB of exchange buffer
#define XM_PLANS_C64_ONLY(X) X(16384, 1024, 16, 16, 16, 4)
// 16384, complex128: 16384 x 16 B does not fit the 160 KiB LDS -- the exchange goes through one plane of doubles,
// real parts then imaginary parts (BlockFFT::SPLIT); radix 8 keeps 1024 threads within their 128 VGPRs
struct Plan16kD {
  using type = FftPlan<16384, 1024, 8, 8, 8, 8, 4>;
};
template <int N>
struct PlanOf;
#define XM_DEF_PLAN(N, NT, ...)               \
  template <>                                 \
  struct PlanOf<N> {                          \
    using type = FftPlan<N, NT, __VA_ARGS__>; \
  };
XM_PLANS_POW2(XM_DEF_PLAN)
XM_PLANS_OTHER(XM_DEF_PLAN)
XM_PLANS_C64_ONLY(XM_DEF_PLAN)
#undef XM_DEF_PLAN

template <int N>
struct Zf2PlanOf;
#define XM_DEF_PLAN(N, NT, ...)               \
  template <>                                 \
  struct Zf2PlanOf<N> {                       \
    using type = FftPlan<N, NT, __VA_ARGS__>; \
  };
XM_PLANS_ZF2(XM_DEF_PLAN)
XM_PLANS_FFT2_EXTRA(XM_DEF_PLAN)
#undef XM_DEF_PLAN
