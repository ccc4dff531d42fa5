// Instantiation + launch of k_zf2d (xm_zf2d.h), the complex128 kernel of the hot shape (4096 -> 8192).  Its own
// translation unit so that it compiles in parallel with the other kernels.
#include "xm_host.h"
#include "xm_plans.h"
#include "xm_tables.h"
#include "xm_zf2d.h"

#include <cmath>
#include <cstdlib>

namespace {

using T = double;

template <class PL, int MODE>
int launch_mode(PipeArgs<T> A, hipStream_t st) {
  const void* tw = nullptr;
  int rc = xm_table_get(TK_TWIDDLE, PL::N, PL::signature(), XM_C128, xm_gen_twiddles<PL>, nullptr, &tw);
  if (rc) return rc;
  A.tw = (const Cx<T>*)tw;
  if (A.n_batch <= 0) return XM_OK;
  constexpr size_t mid_bytes = (size_t)PL::tw_offset(PL::K - 1) * sizeof(Cx<T>);  // must mirror k_zf2d
  const size_t lds = (size_t)BlockFFT<T, PL>::lds_elems() * sizeof(Cx<T>) + (mid_bytes <= 8192 ? mid_bytes : 0) +
                     ((size_t)PL::NT / XM_WAVE + 2) * (sizeof(T) + sizeof(int));
  static XmResidency res;
  int resident = 0;
  rc = xm_resident_blocks(res, k_zf2d<PL, MODE>, PL::NT, lds, &resident, st);
  if (rc) return rc;
  // one row (64 KiB in + 128 KiB out) per ticket
  A.queue_chunk = 1;
  long long blocks = A.n_batch < resident ? A.n_batch : resident;
  if constexpr ((MODE & ZF2_GKEY) != 0) {
    // the arg-max key holds one 16-byte (value, row) slot per WAVE from word XM_KEY_C128_WORD on: a device with more
    // resident workgroups than the buffer has slots runs with fewer (the rows come from the queue either way) instead
    // of writing past XM_KEY_BYTES (advisor, round 3)
    constexpr long long slots = ((long long)XM_KEY_BYTES - 8ll * XM_KEY_C128_WORD) / 16, waves = PL::NT / XM_WAVE;
    static_assert(slots >= waves, "arg-max key too small for one workgroup");
    if (blocks * waves > slots) blocks = slots / waves;
  }
  rc = xm_queue_slot(&A.queue);
  if (rc) return rc;
  xm_note_kernel("k_zf2d", &typeid(PL), nullptr, MODE, -1);
  hipLaunchKernelGGL((k_zf2d<PL, MODE>), dim3((unsigned)blocks), dim3(PL::NT), lds, st, A);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

template <class PL, int MODE>
int launch_mode_dma(const PipeArgs<T>& A, hipStream_t st) {
  static const bool no_dma = getenv("XM_ZF2D_NODMA") != nullptr;  // tuning switch: the next row is not prefetched
  return no_dma ? launch_mode<PL, MODE>(A, st) : launch_mode<PL, MODE | ZF2_DMA>(A, st);
}

template <class PL>
int launch_plan(PipeArgs<double> A, const double* ramp, hipStream_t st) {
  const bool wr = A.out != nullptr, am = A.absmax2 != nullptr, key = A.gkey != nullptr;
  constexpr int AM = ZF2_AMAX | ZF2_VALUE_ONLY, AMK = AM | ZF2_GKEY;
  if (!wr) return key ? launch_mode_dma<PL, AMK>(A, st) : launch_mode_dma<PL, AM>(A, st);
  if (ramp) {
    // e^{i (a + b k)}, k = base_q + 2t (+1): the wave-uniform factors, and e^{i b} for the odd bins (xm_zf2p.h)
    constexpr unsigned N = 2 * PL::N;
    for (int q = 0; q < PL::P; ++q) {
      const unsigned base = (2u * PL::NT * q + (unsigned)A.out_shift) & (N - 1u);
      const double a = ramp[0] + ramp[1] * (double)base;
      A.ramp_c[2 * q] = std::cos(a);
      A.ramp_c[2 * q + 1] = std::sin(a);
    }
    A.ramp_e[0] = std::cos(ramp[1]);
    A.ramp_e[1] = std::sin(ramp[1]);
    A.ramp_db = ramp[1];
    if (key) return launch_mode_dma<PL, ZF2_WRITE | ZF2_RAMP | AMK>(A, st);
    return am ? launch_mode_dma<PL, ZF2_WRITE | ZF2_RAMP | AM>(A, st) : launch_mode_dma<PL, ZF2_WRITE | ZF2_RAMP>(A, st);
  }
  if (key) return launch_mode_dma<PL, ZF2_WRITE | AMK>(A, st);
  return am ? launch_mode_dma<PL, ZF2_WRITE | AM>(A, st) : launch_mode_dma<PL, ZF2_WRITE>(A, st);
}

}  // namespace

int xm_zf2d_launch(int h, const PipeArgs<double>& A, const double* ramp, hipStream_t st, bool* handled) {
  static const bool gen1 = getenv("XM_ZF2D_GEN1") != nullptr;  // tuning switch: k_zf2<double> / the long-transform path
  const bool wr = A.out != nullptr, ph = A.phase != nullptr, am = A.absmax2 != nullptr;
  *handled = !gen1 && !ph && (wr ? (!am || A.amax_value_only) : (am && A.amax_value_only)) && (h == 4096 || h == 8192);
  if (!*handled) return XM_OK;
  if (h == 4096) return launch_plan<typename PlanOf<4096>::type>(A, ramp, st);  // 256 threads x 16 points, 16.16.16
  return launch_plan<typename Zf2PlanOf<8192>::type>(A, ramp, st);               // 1024 threads x 8 points, 8.8.8.8.2
}
