// Host-side declarations shared by the translation units of libxmris_hip.so.
#pragma once
#include "../../include/xmris_hip.h"
#include "xm_common.h"

#include <string>
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

enum XmTableKind { TK_TWIDDLE = 0, TK_HALF = 1, TK_CHIRP = 2, TK_CHIRP_FFT = 3 };

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

// defined in xm_launch_f32.hip / xm_launch_f64.hip
int xm_pipeline_f32(const void* in, int64_t in_stride, void* out, const void* window, const void* phase,
                    int64_t n_batch, int n_in, int n_out, int pad_left, unsigned flags, void* absmax2,
                    int32_t* argidx, hipStream_t st);
int xm_pipeline_f64(const void* in, int64_t in_stride, void* out, const void* window, const void* phase,
                    int64_t n_batch, int n_in, int n_out, int pad_left, unsigned flags, void* absmax2,
                    int32_t* argidx, hipStream_t st);
