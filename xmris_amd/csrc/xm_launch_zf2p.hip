// Instantiation + launch of k_zf2p (xm_zf2p.h), the packed complex64 ">= 2x end zero-fill" kernel of the hot path.
// Its own translation unit so that it compiles in parallel with the other kernels.
#include "xm_host.h"
#include "xm_plans.h"
#include "xm_tables.h"
#include "xm_zf2p.h"
#include "xm_coarse.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>

namespace {

using T = float;

template <class PL, int MODE, int OPT>
int launch_mode(PipeArgs<T> A, hipStream_t st) {
  const void* tw = nullptr;
  int rc = xm_table_get(TK_TWIDDLE, PL::N, PL::signature(), XM_C64, xm_gen_twiddles<PL>, nullptr, &tw);
  if (rc) return rc;
  A.tw = (const Cx<T>*)tw;
  if (A.n_batch <= 0) return XM_OK;
  constexpr bool L16 = (OPT & ZF2P_LOAD16) != 0;
  using FFT = BlockFFT<xm_f2, PL, zf2p_pad_shift<PL, L16>()>;
  constexpr bool CAND = (OPT & ZF2P_CAND) != 0;
  const size_t lds = (size_t)FFT::lds_elems() * sizeof(Cx<xm_f2>) + (size_t)HotTw<T, PL>::mid_lds_size() * sizeof(Cx<T>) +
                     ((size_t)PL::NT / XM_WAVE + 2) * (sizeof(T) + sizeof(int)) + (CAND ? zf2p_cand_lds_bytes() : 0);
  static XmResidency res;
  int resident = 0;
  rc = xm_resident_blocks(res, k_zf2p<PL, MODE, OPT>, PL::NT, lds, &resident, st);
  if (rc) return rc;
  if constexpr ((MODE & ZF2_WRITE) != 0 && PL::N == 2048) {
    // BASELINE configs[1]'s main pass (2048 -> 4096): TWO workgroups per CU, as the 4096-point plan has by its LDS -- with
    // the four that fit 0.170 ms, with three 0.164, with two 0.156 (4.73 -> 5.17 TB/s; profiles/r04/resident_cap.txt).
    // A streaming kernel wants few, deep streams (k_zf_apod: three per CU).
    int cus = 0;
    if (xm_stream_cu_count(st, &cus) == XM_OK && cus > 0 && resident > 2 * cus) resident = 2 * cus;
  }
  // rows per ticket: about 96 KiB of traffic per chunk (the hot shape's row: 1 -- measured: 2 rows per ticket cost it
  // 12 %), so that a launch at full speed draws at most ~60 of the ~90 tickets per microsecond one counter sustains
  const long long row_bytes = (long long)sizeof(Cx<T>) * ((long long)A.n_in + (A.out ? 2 * PL::N : 0));
  long long chunk = (98304 + row_bytes - 1) / row_bytes;
  static const int chunk_env = getenv("XM_QUEUE_CHUNK") ? atoi(getenv("XM_QUEUE_CHUNK")) : 0;  // tuning switch
  if (chunk_env > 0) chunk = chunk_env;
  chunk = chunk < 1 ? 1 : (chunk > 64 ? 64 : chunk);
  if constexpr ((OPT & ZF2P_QUEUE) == 0) {
    // static split (the instruction-bound maxima-only passes): as many workgroups as fit the chip, equal shares --
    // the 96 KiB rule above is the queue's; with 4 KiB rows it left a third of the wave slots empty
    static const bool even_env = getenv("XM_STATIC_EVEN") == nullptr || atoi(getenv("XM_STATIC_EVEN")) != 0;  // tuning switch
    if (even_env && chunk_env <= 0) {
      chunk = (A.n_batch + resident - 1) / resident;
      chunk = chunk < 1 ? 1 : chunk;
    }
  }
  A.queue_chunk = (int)chunk;
  const long long nchunks = (A.n_batch + chunk - 1) / chunk;
  long long blocks = nchunks < resident ? nchunks : resident;
  if constexpr (CAND) {  // every workgroup scans ceil(n_batch / blocks) estimates: a bit each in its LDS bitmap
    blocks = A.n_batch < resident ? A.n_batch : resident;
    if ((A.n_batch + blocks - 1) / blocks > ZF2P_CAND_KMAX)
      return xm_fail(XM_ERR_INVALID_ARG, "guess_refine: too many rows for one launch");
  }
  if constexpr ((MODE & ZF2_AMAX) != 0) {
    // value-only maxima are accumulated with one atomic max per wave: the slots start at +0.0
    if (A.amax_value_only && !A.gkey) HIP_TRY(hipMemsetAsync(A.absmax2, 0, (size_t)A.n_batch * sizeof(T), st));
  }
  if constexpr ((OPT & (ZF2P_QUEUE | ZF2P_CAND)) != 0 || ((MODE & ZF2_AMAX) != 0 && (MODE & ZF2_WRITE) != 0)) {
    rc = xm_queue_slot(&A.queue);
    if (rc) return rc;
  }
  xm_note_kernel("k_zf2p", &typeid(PL), nullptr, MODE, OPT);
  hipLaunchKernelGGL((k_zf2p<PL, MODE, OPT>), dim3((unsigned)blocks), dim3(PL::NT), lds, st, A);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

// Write modes are HBM bound and unevenly so across workgroups: dynamic row hand-out + nontemporal stores.  The
// arg-max-only pre-pass is instruction bound (every workgroup takes the same time per row): static stride.
constexpr int kOptWrite = ZF2P_LOAD16 | ZF2P_NT | ZF2P_QUEUE;
constexpr int kOptAmax = ZF2P_LOAD16;

// Short launches (a few dozen rows per workgroup: BASELINE configs[1], 16,384 x 2048 -> 4096, is 0.17 ms) gain nothing
// from the row queue -- its point is the tail of millisecond launches -- and pay for its tickets: the ramp modes are
// also built with the static split, chosen below `XM_ZF2P_STATIC_ROWS` rows per resident workgroup (tuning switch;
// default set from the same-box A/B in profiles/r04/time_configs.txt).
constexpr int kOptWriteStatic = ZF2P_LOAD16 | ZF2P_NT;

template <class PL, int MODE>
int launch_write(const PipeArgs<T>& A, hipStream_t st) {
  static const long long static_rows = getenv("XM_ZF2P_STATIC_ROWS") ? atoll(getenv("XM_ZF2P_STATIC_ROWS")) : 0;
  int cus = 0;
  if (static_rows > 0 && xm_stream_cu_count(st, &cus) == XM_OK && cus > 0 && A.n_batch < static_rows * 2 * cus)
    return launch_mode<PL, MODE, kOptWriteStatic>(A, st);
  return launch_mode<PL, MODE, kOptWrite>(A, st);
}

template <class PL>
int launch_plan(PipeArgs<T> A, const double* ramp, hipStream_t st) {
  const bool wr = A.out != nullptr, ph = A.phase != nullptr, am = A.absmax2 != nullptr;
  if (ramp) {
    // e^{i (a + b k)}, k = base_q + 2t (+1): the wave-uniform factors, and e^{i b} for the odd bins
    constexpr unsigned N = 2 * PL::N;
    for (int q = 0; q < PL::P; ++q) {
      const unsigned base = (2u * PL::NT * q + (unsigned)A.out_shift) & (N - 1u);
      const double a = ramp[0] + ramp[1] * (double)base;
      A.ramp_c[2 * q] = (T)std::cos(a);
      A.ramp_c[2 * q + 1] = (T)std::sin(a);
    }
    A.ramp_e[0] = (T)std::cos(ramp[1]);
    A.ramp_e[1] = (T)std::sin(ramp[1]);
    A.ramp_db = ramp[1];
    A.phase = nullptr;
    return am ? launch_write<PL, ZF2_WRITE | ZF2_RAMP | ZF2_AMAX>(A, st) : launch_write<PL, ZF2_WRITE | ZF2_RAMP>(A, st);
  }
  if (wr && ph && am) return launch_mode<PL, ZF2_WRITE | ZF2_PHASE | ZF2_AMAX, kOptWrite>(A, st);
  if (wr && ph) return launch_mode<PL, ZF2_WRITE | ZF2_PHASE, kOptWrite>(A, st);
  if (wr && am) return launch_mode<PL, ZF2_WRITE | ZF2_AMAX, kOptWrite>(A, st);
  if (wr) return launch_mode<PL, ZF2_WRITE, kOptWrite>(A, st);
  return launch_mode<PL, ZF2_AMAX, kOptAmax>(A, st);
}

}  // namespace

// ---- guess stage of the speculative schedule ------------------------------------------------------------------
namespace {

constexpr int kGuessHalf = 512;  // coarse spectra: the first <= 512 samples of a row on a 1024-bin grid

template <class PL, bool IN64>
int launch_refine(const PipeArgs<T>& A, hipStream_t st) {
  return launch_mode<PL, ZF2_AMAX, ZF2P_CAND | (IN64 ? ZF2P_IN64 : ZF2P_LOAD16)>(A, st);
}

template <bool IN64>
int refine_plan(int h, const PipeArgs<T>& A, hipStream_t st) {
  switch (h) {
    case 512: return launch_refine<typename Zf2PlanOf<512>::type, IN64>(A, st);
    case 1024: return launch_refine<typename Zf2PlanOf<1024>::type, IN64>(A, st);
    case 2048: return launch_refine<typename Zf2PlanOf<2048>::type, IN64>(A, st);
    case 4096: return launch_refine<typename PlanOf<4096>::type, IN64>(A, st);
    case 8192: return launch_refine<typename Zf2PlanOf<8192>::type, IN64>(A, st);
    default: break;
  }
  return xm_fail(XM_ERR_UNSUPPORTED_N, "no half-length plan for " + std::to_string(h));
}

bool pair_loads_ok(const void* in, int64_t in_stride, int n_in, int pad_left) {
  return (pad_left % 2 == 0) && (n_in % 2 == 0) && (in_stride % 2 == 0) && ((reinterpret_cast<size_t>(in) & 15u) == 0);
}

int half_table(int n, const Cx<T>** out) {
  const void* half = nullptr;
  const int rc = xm_table_get(TK_HALF, n, 0, XM_C64, xm_gen_half, nullptr, &half);
  *out = (const Cx<T>*)half;
  return rc;
}

}  // namespace

int xm_zf2p_guess_supported(const void* in, int64_t in_stride, int n_in, int n_out, int pad_left, unsigned flags, int dtype) {
  const int h = n_out / 2;
  if (n_out % 2 || pad_left != 0 || n_in > h || n_in < 2) return 0;
  if (h != 512 && h != 1024 && h != 2048 && h != 4096 && h != 8192) return 0;
  if (flags & (XM_FFT_SHIFT_IN | XM_FFT_INVERSE)) return 0;
  if (dtype == XM_C128) return (reinterpret_cast<size_t>(in) & 15u) == 0;
  const int ng = n_in < kGuessHalf ? n_in : kGuessHalf;
  return pair_loads_ok(in, in_stride, n_in, 0) && ng % 2 == 0;
}

int xm_zf2p_guess_rows(const void* in, int64_t in_stride, const float* window, int64_t n_batch, int n_in, int n_out,
                       int n_guess, float scale, float* est, unsigned long long* key, int dtype, hipStream_t st) {
  PipeArgs<T> A;
  std::memset(&A, 0, sizeof(A));
  A.in = (const Cx<T>*)in;
  A.window = window;
  A.in_stride = in_stride;
  A.n_batch = n_batch;
  A.n = 2 * kGuessHalf;
  int ng = n_guess > 0 && n_guess < kGuessHalf ? n_guess : kGuessHalf;
  if (ng > n_in) ng = n_in;
  if (dtype == XM_C64) ng &= ~1;  // pair loads
  A.n_in = ng;
  A.amax_value_only = 1;
  A.scale = scale;
  A.gkey = key;
  A.est = est;
  A.absmax2 = est;  // (the kernel's "maxima wanted" marker)
  (void)n_out;
  int rc = half_table(A.n, &A.aux);
  if (rc) return rc;
  // rows of at least 512 samples: the matrix-core version (xm_coarse.h); XM_GUESS_FFT=1 keeps the FFT one
  static const bool fft_only = getenv("XM_GUESS_FFT") != nullptr;  // tuning switch
  if (ng == kGuessHalf && window && !fft_only) {
    if (n_batch <= 0) return XM_OK;
    CoarseArgs C;
    C.in = in;
    C.window = window;
    C.w1024 = A.aux;
    C.est = est;
    C.gkey = key;
    C.in_stride = in_stride;
    C.n_batch = n_batch;
    C.scale2 = scale * scale;
    static XmResidency res32, res64;
    int resident = 0;
    rc = dtype == XM_C64 ? xm_resident_blocks(res32, k_coarse_mfma<false>, 64 * kCoarseWaves, 0, &resident, st)
                         : xm_resident_blocks(res64, k_coarse_mfma<true>, 64 * kCoarseWaves, 0, &resident, st);
    if (rc) return rc;
    const long long want = (n_batch + kCoarseWaves - 1) / kCoarseWaves;
    const long long blocks = want < resident ? want : resident;
    xm_note_kernel("k_coarse_mfma", nullptr, dtype == XM_C64 ? "float" : "double", kGuessHalf, -1);
    if (dtype == XM_C64)
      hipLaunchKernelGGL(k_coarse_mfma<false>, dim3((unsigned)blocks), dim3(64 * kCoarseWaves), 0, st, C);
    else
      hipLaunchKernelGGL(k_coarse_mfma<true>, dim3((unsigned)blocks), dim3(64 * kCoarseWaves), 0, st, C);
    HIP_TRY(hipGetLastError());
    return XM_OK;
  }
  using PL = typename Zf2PlanOf<kGuessHalf>::type;
  return dtype == XM_C64 ? launch_mode<PL, ZF2_AMAX, ZF2P_EST | ZF2P_LOAD16>(A, st)
                         : launch_mode<PL, ZF2_AMAX, ZF2P_EST | ZF2P_IN64>(A, st);
}

int xm_zf2p_guess_refine(const void* in, int64_t in_stride, const float* window, int64_t n_batch, int n_in, int n_out,
                         unsigned flags, float scale, const float* est, unsigned long long* guess_key, float band,
                         unsigned long long* work_key, float* out_max2, long long* out_flat, void* out_row, int dtype,
                         hipStream_t st) {
  PipeArgs<T> A;
  std::memset(&A, 0, sizeof(A));
  A.in = (const Cx<T>*)in;
  A.window = window;
  A.in_stride = in_stride;
  A.n_batch = n_batch;
  A.n = n_out;
  A.n_in = n_in;
  A.out_shift = (flags & XM_FFT_SHIFT_OUT) ? n_out / 2 : 0;
  A.amax_value_only = 1;
  A.scale = scale;
  A.gkey = work_key;
  A.gbest = reinterpret_cast<unsigned*>(work_key) + XM_KEY_GBEST_WORD;
  A.gkey_in = guess_key;
  A.est = const_cast<float*>(est);
  A.absmax2 = const_cast<float*>(est);  // (the kernel's "maxima wanted" marker)
  A.band2 = band * band;
  A.take_max2 = out_max2;
  A.take_flat = out_flat;
  A.take_row = (Cx<double>*)out_row;
  int rc = half_table(n_out, &A.aux);
  if (rc) return rc;
  return dtype == XM_C64 ? refine_plan<false>(n_out / 2, A, st) : refine_plan<true>(n_out / 2, A, st);
}

bool xm_zf2p_eligible(const PipeArgs<float>& A, int64_t in_stride) {
  static const bool gen1 = getenv("XM_ZF2_GEN1") != nullptr;  // tuning switch: the first-generation kernel
  if (gen1) return false;
  // pair loads: every (even, odd) sample pair of a row is one aligned 16-byte word
  return (A.pad_left % 2 == 0) && (A.n_in % 2 == 0) && (in_stride % 2 == 0) && ((reinterpret_cast<size_t>(A.in) & 15u) == 0);
}

int xm_zf2p_launch(int h, const PipeArgs<float>& A, const double* ramp, hipStream_t st) {
  switch (h) {
    case 512: return launch_plan<typename Zf2PlanOf<512>::type>(A, ramp, st);
#ifdef XM_ZF2P_WIDE_SMALL  // build-time A/B switch: the 16-point plans (fewer stages, half the threads) below 4096 too --
                          // round 4, same box: configs[1] main pass 0.1721 vs 0.1724 ms, maxima-only pass 0.113 vs 0.107: left off
    case 1024: return launch_plan<typename PlanOf<1024>::type>(A, ramp, st);
    case 2048: return launch_plan<typename PlanOf<2048>::type>(A, ramp, st);
#else
    case 1024: return launch_plan<typename Zf2PlanOf<1024>::type>(A, ramp, st);
    case 2048: return launch_plan<typename Zf2PlanOf<2048>::type>(A, ramp, st);
#endif
    case 4096: return launch_plan<typename PlanOf<4096>::type>(A, ramp, st);  // 256 threads x 16 points, 16.16.16
    case 8192: return launch_plan<typename Zf2PlanOf<8192>::type>(A, ramp, st);  // 1024 x 8, one workgroup per CU
    default: break;
  }
  return xm_fail(XM_ERR_UNSUPPORTED_N, "no half-length plan for " + std::to_string(h));
}
