// complex128 (2 x float64) instantiations of the fused kernels.
#define XM_REAL double
#include "xm_launch.inc"

int xm_pipeline_f64(const void* in, int64_t in_stride, void* out, const void* window, const void* phase,
                    const double* ramp, int64_t n_batch, int n_in, int n_out, int pad_left, unsigned flags,
                    void* absmax2, int32_t* argidx, hipStream_t st) {
  return pipeline_typed(in, in_stride, out, window, phase, ramp, n_batch, n_in, n_out, pad_left, flags, absmax2,
                        argidx, st);
}

int xm_ramp_native_f64(const void* in, int64_t in_stride, int n_in, int n_out, int pad_left, unsigned flags) {
  return ramp_native(in, in_stride, n_in, n_out, pad_left, flags) ? 1 : 0;
}

int xm_key_native_f64(const void* in, int64_t in_stride, int n_in, int n_out, int pad_left, unsigned flags) {
  return key_native(in, in_stride, n_in, n_out, pad_left, flags) ? 1 : 0;
}

int xm_big_supported_f64(int n) { return big_supported(n) ? 1 : 0; }
