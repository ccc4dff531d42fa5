// Host-side declarations shared by the translation units of libxmris_hip.so.
#pragma once
#include <cstdlib>
#include "../../include/xmris_hip.h"
#include "xm_common.h"

#include <mutex>
#include <string>
#include <typeinfo>
#include <vector>

int xm_fail(int code, const std::string& msg);
#define HIP_TRY(expr)                                                                \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess) {                                                          \
      (void)hipGetLastError(); /* clear the sticky error: the next call starts clean */ \
      return xm_fail(XM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    }                                                                                \
  } while (0)

enum XmTableKind { TK_TWIDDLE = 0, TK_HALF = 1, TK_CHIRP = 2, TK_CHIRP_FFT = 3, TK_BIG_WN = 4 };

// Cached device table (kind, n, m, dtype, current device).  `gen` fills re/im in fp64 when the table
// does not exist yet; it is rounded once to the storage precision and uploaded.
typedef void (*xm_table_gen)(int n, int m, const void* ctx, std::vector<double>& re, std::vector<double>& im);
int xm_table_get(int kind, int n, int m, int dtype, xm_table_gen gen, const void* ctx, const void** out);

// e^{sign * 2 pi i * num/den} with the range reduction done on the integers
void xm_unit(long long num, long long den, double sign, double& c, double& s);

bool xm_has_direct_plan(int n, int dtype);
bool xm_has_pow2_plan(int n, int dtype);
int xm_bluestein_m(int n);
bool xm_supported(int n, int dtype);

// defined in xm_launch_f32.hip / xm_launch_f64.hip.  `ramp`: nullptr, or {phase0, dphase} in radians -- the output is
// multiplied by e^{i (phase0 + dphase k)}, k = output index (then `phase` must be nullptr)
int xm_pipeline_f32(const void* in, int64_t in_stride, void* out, const void* window, const void* phase,
                    const double* ramp, int64_t n_batch, int n_in, int n_out, int pad_left, unsigned flags,
                    void* absmax2, int32_t* argidx, hipStream_t st);
int xm_pipeline_f64(const void* in, int64_t in_stride, void* out, const void* window, const void* phase,
                    const double* ramp, int64_t n_batch, int n_in, int n_out, int pad_left, unsigned flags,
                    void* absmax2, int32_t* argidx, hipStream_t st);
// 1 when length n has a path beyond the in-LDS plans (four-step over global memory, xm_bigfft.inc)
int xm_big_supported_f32(int n);
int xm_big_supported_f64(int n);
// ... and whether it has an in-LDS path (direct plan or chirp-z inside the LDS)
bool xm_supported_in_lds(int n, int dtype);
// 1 when a geometry has a kernel that applies the ramp natively (no table is built), else 0
int xm_ramp_native_f32(const void* in, int64_t in_stride, int n_in, int n_out, int pad_left, unsigned flags);
int xm_ramp_native_f64(const void* in, int64_t in_stride, int n_in, int n_out, int pad_left, unsigned flags);
// ... and whether it leaves the launch's arg-max in a key (XM_AMAX_GLOBAL_KEY)
int xm_key_native_f32(const void* in, int64_t in_stride, int n_in, int n_out, int pad_left, unsigned flags);
int xm_key_native_f64(const void* in, int64_t in_stride, int n_in, int n_out, int pad_left, unsigned flags);

// xm_launch_zf2p.hip: guess stage of the speculative schedule (xm_guess_* in xmris_hip.h)
int xm_zf2p_guess_supported(const void* in, int64_t in_stride, int n_in, int n_out, int pad_left, unsigned flags, int dtype);
int xm_zf2p_guess_rows(const void* in, int64_t in_stride, const float* window, int64_t n_batch, int n_in, int n_out,
                       int n_guess, float scale, float* est, unsigned long long* key, int dtype, hipStream_t st);
int xm_zf2p_guess_refine(const void* in, int64_t in_stride, const float* window, int64_t n_batch, int n_in, int n_out,
                         unsigned flags, float scale, const float* est, unsigned long long* guess_key, float band,
                         unsigned long long* work_key, float* out_max2, long long* out_flat, void* out_row, int dtype,
                         hipStream_t st);

// {head, done} counter pair (zero) for one launch of a persistent kernel that hands out rows dynamically; the
// kernel's last workgroup leaves it zero again.  Slots come from a per-device ring of 1024.
int xm_queue_slot(unsigned** out);

// e^{i (phase0 + dphase k)}, k < n, into `table` (device, storage precision of `dtype`), fp64 sincos per entry
int xm_ramp_table_async(void* table, int n, double phase0, double dphase, int dtype, hipStream_t st);

// CUs a stream may use: the population count of its CU mask (all CUs for an ordinary stream; xm_stream_create makes
// streams with a partition of the chip).  Cached per stream handle.
extern "C" int xm_stream_cu_count(hipStream_t st, int* cus);  // (C linkage: defined among the ABI functions)

// Grid of a persistent kernel = CUs of its stream x resident workgroups per CU.  The occupancy query and the
// dynamic-LDS opt-in run once per kernel instantiation and device, under a lock (the launchers are re-entrant).
struct XmResidency {
  std::mutex mu;
  int blocks[16] = {0};  // resident workgroups per CU
};
template <class K>
int xm_resident_blocks(XmResidency& r, K kern, int nt, size_t lds, int* out, hipStream_t st = nullptr) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return xm_fail(XM_ERR_INVALID_ARG, "device ordinal out of range");
  int per_cu = 0;
  {
    std::lock_guard<std::mutex> lk(r.mu);
    if (r.blocks[dev] == 0) {
      if (lds > 48 * 1024)
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int q = 0;
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, kern, nt, lds));
      r.blocks[dev] = q < 1 ? 1 : q;
    }
    per_cu = r.blocks[dev];
  }
  // (tuning switch XM_RESIDENT_CAP=<workgroups per CU>: an upper bound for every persistent grid -- the sweep that found
  // k_zf_apod's three per CU, profiles/r04/zf_apod.txt, applied to the other kernels: profiles/r04/resident_cap.txt)
  static const int cap_env = getenv("XM_RESIDENT_CAP") ? atoi(getenv("XM_RESIDENT_CAP")) : 0;
  if (cap_env > 0 && per_cu > cap_env) per_cu = cap_env;
  int cus = 0;
  const int rc = xm_stream_cu_count(st, &cus);
  if (rc) return rc;
  *out = per_cu * cus;
  return XM_OK;
}

// What the dispatcher launched last on this thread, as the profiler names it (`xm_last_kernel_string`): the fused
// launchers note the kernel template, its plan and its mode words right before the launch -- reports quote this
// instead of a string typed by hand.
void xm_note_kernel(const char* base, const std::type_info* plan, const char* scalar, int mode, int opt);
