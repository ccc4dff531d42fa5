/* xmris_hip.h -- C ABI of libxmris_hip.so: the MI355X (gfx950) backend of the xmris `.xmr`
 * spectral hot path  zero_fill -> apodize_exp -> to_spectrum (ortho FFT + fftshift) -> autophase.
 *
 * The reference (andrewendlinger/xmris v0.6.1) is pure Python and has no FFI; the seam these
 * entry points replace is the ndarray boundary inside `src/xmris/processing/*.py`
 * (`da.values` + `da.get_axis_num(dim)` on the way in, `da.copy(data=...)` on the way out).
 * Each function cites the reference statement it stands in for.  All metadata (dims, coords,
 * attrs, validation, exceptions) stays in the Python host layer (`xmris_amd/`).
 *
 * Conventions
 *  - Plain C types only.  All array pointers are DEVICE-ACCESSIBLE pointers owned by the caller (device
 *    memory; small outputs may also live in hipHostMalloc'd host memory, which the selection stage uses
 *    to hand (max, flat index, winning spectrum) to the host without memcpy nodes); the
 *    library never frees or retains them.  `stream` is a hipStream_t passed as void* (NULL =
 *    the default stream).  Calls are asynchronous on that stream; nothing synchronises.
 *  - Layout: `n_batch` spectra, C-contiguous, FID / frequency axis last, interleaved complex
 *    (re, im).  dtype XM_C64 = 2 x float32, XM_C128 = 2 x float64.
 *  - Return value: 0 on success, negative xm_status on failure (no exceptions, no aborts);
 *    `xm_last_error_string()` describes the last failure on the calling thread.
 *  - Twiddle / chirp tables are computed in fp64 on the host, rounded once to the storage
 *    precision and cached per (length, dtype, device) inside the library (mutex-guarded);
 *    `xm_clear_cache()` frees them.  Reentrant otherwise.
 *  - Device: the device that owns the input buffer is made current for the duration of a call (tables, occupancy
 *    caches and scratch are per device); `stream` must belong to that device.
 */
#ifndef XMRIS_HIP_H
#define XMRIS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { XM_C64 = 0, XM_C128 = 1 } xm_dtype;

typedef enum {
  XM_OK = 0,
  XM_ERR_INVALID_ARG = -1,   /* null pointer, negative size, bad dtype/flag combination   */
  XM_ERR_UNSUPPORTED_N = -2, /* transform length has no in-LDS plan (see xm_fft_supported) */
  XM_ERR_HIP = -3,           /* a HIP runtime call failed; see xm_last_error_string()      */
  XM_ERR_NO_DEVICE = -4
} xm_status;

/* xm_fft1d_batched / xm_pipeline_fused flags */
#define XM_FFT_INVERSE 1u   /* e^{+2 pi i km/N}  (np.fft.ifftn, fourier.py:210)             */
#define XM_FFT_ORTHO 2u     /* scale 1/sqrt(N)   (norm="ortho", fourier.py:153)             */
#define XM_FFT_SHIFT_IN 4u  /* roll the INPUT by (N+1)/2 first  (ifftshift, fourier.py:57)  */
#define XM_FFT_SHIFT_OUT 8u /* roll the OUTPUT by N/2 afterwards (fftshift, fourier.py:31)  */
#define XM_AMAX_VALUE_ONLY 16u /* xm_pipeline_fused: absmax2[b] only, argidx[b] is written as 0 (the caller
                                  recovers the index along the axis from the winning spectrum itself)      */

#define XM_AMAX_GLOBAL_KEY 32u /* xm_pipeline_fused(_ramp), geometries of xm_pipeline_ramp_native only: `absmax2` points to
                                  an arg-max KEY BUFFER (XM_KEY_BYTES bytes, zero at launch) that receives the global
                                  arg-max of the launch as partial keys, max |X|^2 float bits << 32 | (0xffffffff - row);
                                  no per-row outputs.  `argidx` == NULL: the key stays for xm_argmax_key_take (merges,
                                  decodes, clears).  `argidx` != NULL: it points to a result record `xm_argmax_result`, device-accessible --
                                  e.g. pinned host memory -- that the kernel's last workgroup fills itself, clearing the key. */
#define XM_KEY_BYTES 131072   /* 64 partial keys on cache lines of their own (8 KiB), scratch words of the consumers, and from
                                 byte 65536 the per-wave (value, row) slots of the complex128 kernels */
typedef struct {
  float max2;   /* max |X|^2 of the launch (XM_C128: these eight bytes hold it as a double) */
  float pad_;
  int64_t flat; /* winning row * n_out (index along the axis: 0) */
} xm_argmax_result;
/* XM_C64 without a >= 2x zero fill (the plans of k_fft2: 512 ... 8192, 3 * 2^k) takes the key in the
 * xm_pipeline_fused_ramp form only, with an output and at least two rows.
 * XM_C128 takes XM_AMAX_GLOBAL_KEY on the geometries of xm_pipeline_key_native (half lengths 4096 and 8192) and only with a
 * result record: 64 bits of value + a row do not fit one atomic, so every wave leaves its (value, row) pair in a slot
 * of the key buffer and the last workgroup out merges them; the key buffer needs no clearing. */

int xm_version(void); /* 10000*major + 100*minor + patch */
const char* xm_last_error_string(void);
int xm_clear_cache(void);
/* The kernel the fused dispatcher (xm_pipeline_fused / _ramp, xm_fft1d_batched, xm_guess_*) launched last ON THE
 * CALLING THREAD, spelled as the profiler spells it, e.g. "k_zf2p<FftPlan<4096,256,16,16,16>, 13, 11>" (template, plan,
 * mode words): reports and counter files are matched against this, not against a string typed by hand. */
const char* xm_last_kernel_string(void);
/* 1 if a length-n transform of `dtype` has an in-LDS plan (direct or Bluestein), else 0. */
int xm_fft_supported(int n, int dtype);
/* Build (and cache) the tables for length n so that later calls do no allocation. */
int xm_plan_prepare(int n, int dtype);

/* A1  zero_fill  (processing/fid.py:251  da.pad(..., constant_values=0)).
 * out[b, j] = in[b, j - pad_left] for pad_left <= j < pad_left + n_in, else 0.  Bit-exact copy. */
int xm_zero_fill(const void* in, void* out, int64_t n_batch, int n_in, int n_out, int pad_left,
                 int dtype, void* stream);

/* A2  apodize  (processing/fid.py:139  da * weight).  out[b, j] = in[b, j] * window[j];
 * `window` is n real values of the storage precision (host computes exp(-pi*lb*t), fid.py:136).
 * in == out allowed. */
int xm_apodize(const void* in, void* out, const void* window, int64_t n_batch, int n, int dtype,
               void* stream);

/* A1 + A2 in ONE pass  (processing/fid.py:251 `da.pad(...)` followed by fid.py:136-139 `da * exp(-pi lb t)`):
 *   out[b, j] = in[b, j - pad_left] * window[j]  for pad_left <= j < pad_left + n_in,  +0 elsewhere.
 * The fused zero-fill + apodisation launch for chains that stop before the FFT: 16-byte loads of the FID axis, the
 * window staged through the LDS, nontemporal 16-byte stores, rows handed out by a device-scope counter.  `window`:
 * n_out reals of the OUTPUT precision.  `out_dtype` = `in_dtype`, or XM_C128 for XM_C64 rows (numpy's promotion when a
 * complex64 FID meets the float64 window, fid.py:136-139: each product is then the exact complex128 product numpy
 * computes).  n_out * sizeof(real) <= 96 KiB.  in != out. */
int xm_zf_apod(const void* in, int64_t in_row_stride, void* out, const void* window, int64_t n_batch, int n_in, int n_out,
               int pad_left, int in_dtype, int out_dtype, void* stream);

/* A3/A4/A5  fft / ifft / fftshift folded  (processing/fourier.py:153, 210, 31, 57).
 * out[b, (m + s_out) mod n] = scale * sum_k in[b, (k - s_in) mod n] * e^{-+2 pi i k m / n}. */
/* 1 when xm_pipeline_fused(_ramp) accepts XM_AMAX_GLOBAL_KEY for this geometry and dtype, else 0. */
int xm_pipeline_key_native(const void* in, int64_t in_row_stride, int n_in, int n_out, int pad_left, unsigned flags,
                           int dtype);

int xm_fft1d_batched(const void* in, void* out, int64_t n_batch, int n, unsigned flags, int dtype,
                     void* stream);

/* A4 alone  (fourier.py:31-32, 57-58  da.roll).  out[b, (j + shift) mod n] = in[b, j]. */
int xm_roll(const void* in, void* out, int64_t n_batch, int n, int shift, int dtype, void* stream);

/* A8  phase apply  (processing/phasing.py:73  da * exp(1j*phase_array)).
 * out[b, j] = in[b, j] * phase_table[j]  (n complex values; host computes e^{i phi}, phasing.py:62-69).
 * in == out allowed. */
int xm_phase_apply(const void* in, void* out, const void* phase_table, int64_t n_batch, int n,
                   int dtype, void* stream);

/* A6  per-spectrum max |X|^2 and its first index  (first half of phasing.py:229).
 * absmax2[b] (real, storage precision) and argidx[b] (int32) for every spectrum of an existing array. */
int xm_absmax_rows(const void* in, int64_t n_batch, int n, void* absmax2, int32_t* argidx, int dtype,
                   void* stream);

/* A6, speculative schedule: norm[b] = sum_j |in[b, j]| * |window[j + pad_left]| (window may be NULL), the windowed
 * L1 norm of every FID.  sum|z| / sqrt(N) bounds every |X[k]| of the row and equals the peak of a single decaying
 * resonance, so the row with the largest norm is the guess for the row through the global maximum
 * (phasing.py:229) that lets the host search (p0, p1) BEFORE any spectrum exists; the fused main pass then
 * returns the true per-row maxima, and a wrong guess is repaired (see xmris_amd/pipeline.py::run_stream).
 * Only a ranking is needed, so the sum may run over a regular subset: the 1-KiB blocks (128 complex64 / 64
 * complex128 samples) whose index is a multiple of `sub_step` (1 = every sample), i.e. whole cache lines spread
 * over the n_in leading samples.
 * `norm`: n_batch reals of the storage precision. */
int xm_row_l1(const void* in, int64_t in_row_stride, const void* window, int64_t n_batch, int n_in, int pad_left,
              int sub_step, void* norm, uint64_t* key, int dtype, void* stream);
/* `key` (complex64 only, may be NULL): an arg-max key buffer (XM_KEY_BYTES, zero at launch) that receives the row with
 * the largest norm as float bits << 32 | (0xffffffff - row) -- the launch then needs no separate arg-max reduction;
 * `norm` may be NULL. */

/* A6: consumer of an arg-max key buffer written by xm_row_l1 / XM_AMAX_GLOBAL_KEY: out_max2[0] (float) = the value,
 * out_flat[0] = row * n_per_row, buffer := 0 for its next producer.  With `in` (n_batch x n_in rows, dtype) also
 * out_row[j] = (complex128) in[row, j] -- xm_gather_row_c128 of the winning row in the same launch.  `in` / `out_row`
 * may be NULL.  out_* are device-accessible (the selection stage passes pinned host memory). */
int xm_argmax_key_take(uint64_t* key, int n_per_row, void* out_max2, int64_t* out_flat, const void* in,
                       int64_t in_row_stride, int n_in, void* out_row, int dtype, void* stream);

/* A6, speculative schedule, guess stage (replaces xm_row_l1 + xm_argmax_key_take where xm_guess_supported() says so).
 * The host searches (p0, p1) on the spectrum of the row that holds the global max |X| (phasing.py:229, 241-242)
 * BEFORE the spectra exist; these two calls find that row without transforming every row in full:
 *   xm_guess_rows    est[b] = max_k |X_c[b, k]|^2 of a COARSE spectrum of row b -- its first n_guess (<= 512; 0 = 512)
 *                    samples times their window weights on a 1024-bin grid (one wave per row; 4 KiB of a 32 KiB row
 *                    read) -- and the largest estimate in `key` (an arg-max key buffer, zero at launch).  A truncated,
 *                    coarsely sampled spectrum underestimates a line's height by a bounded factor (scalloping of the
 *                    grid, the missing tail), so the true arg-max row lies among the rows whose estimate is within
 *                    that band of the largest one.  Rows of 512 samples and more are transformed on the matrix
 *                    cores (fp16 operands behind a per-row power-of-two scale, fp32 sums: est within 2e-3 of the exact
 *                    coarse spectrum's maximum; a NaN sample gives NaN), other rows by the fp32 FFT (2e-5);
 *   xm_guess_refine  transforms every row with est[b] >= band^2 * max(est) exactly (all n_in samples -> n_out bins, the
 *                    arithmetic of xm_pipeline_fused; at most 16 rows per resident workgroup) and leaves the winner
 *                    like xm_argmax_key_take does: out_max2[0] = its max |X|^2 (float), out_flat[0] = row * n_out,
 *                    out_row[j] = (complex128) in[row, j].  `guess_key` is consumed (left zero), `work_key` is a second
 *                    key buffer (zero at launch, left zero).
 * Geometry: "end" zero fill to >= 2x (pad_left = 0, n_out/2 in {512 ... 8192}); `window`: n_out FLOAT32 weights for
 * either dtype; XM_C128 rows are converted to float on load (a ranking; the main pass's own maxima verify the guess,
 * xmris_amd/pipeline.py::run_stream).  `est`: n_batch floats.  out_* device-accessible (pinned host memory is fine). */
int xm_guess_supported(const void* in, int64_t in_row_stride, int n_in, int n_out, int pad_left, unsigned flags, int dtype);
int xm_guess_rows(const void* in, int64_t in_row_stride, const void* window, int64_t n_batch, int n_in, int n_out,
                  int n_guess, unsigned flags, float* est, uint64_t* key, int dtype, void* stream);
int xm_guess_refine(const void* in, int64_t in_row_stride, const void* window, int64_t n_batch, int n_in, int n_out,
                    unsigned flags, const float* est, uint64_t* guess_key, float band, uint64_t* work_key,
                    float* out_max2, int64_t* out_flat, void* out_row, int dtype, void* stream);

/* A6  global arg-max  (phasing.py:229  np.argmax(np.abs(values)), first maximum in C order).
 * Reduces the per-spectrum pairs: out_max2[0] = max_b absmax2[b], out_flat[0] = b*n + argidx[b]
 * of the first such b.  Both outputs are device-accessible scalars. */
int xm_argmax_reduce(const void* absmax2, const int32_t* argidx, int64_t n_batch, int n, void* out_max2,
                     int64_t* out_flat, int dtype, void* stream);

/* A6  the ONE spectrum through the global maximum (phasing.py:241-242 `da.isel(...)`), fetched without a host
 * round trip: out[j] = (complex128) in[row, j], j < n_in, with row = flat_index[0] / n_per_row read from DEVICE
 * memory (the output of xm_argmax_reduce).  Feeds the complex128 recomputation of that spectrum for the solver. */
int xm_gather_row_c128(const void* in, int64_t in_row_stride, int n_in, const int64_t* flat_index, int n_per_row,
                       void* out, int dtype, void* stream);

/* The fused hot path, one launch:
 *   z[j]   = (pad_left <= j < pad_left + n_in) ? in[b, j - pad_left] * window[j] : 0   (A1+A2)
 *   X      = FFT_n_out(z) * scale, optionally rolled                                    (A3+A4)
 *   absmax2[b], argidx[b] = max |X|^2 and its first index (after the roll)              (A6)
 *   out[b, k] = X[k] * phase_table[k]                                                   (A8)
 * `window` (n_out reals), `phase_table` (n_out complex), `out`, `absmax2`, `argidx` may each be
 * NULL to skip that part (out == NULL -> the arg-max pre-pass, nothing is written but the pairs).
 * `in_row_stride` = elements between consecutive input spectra (>= n_in). */
int xm_pipeline_fused(const void* in, int64_t in_row_stride, void* out, const void* window,
                      const void* phase_table, int64_t n_batch, int n_in, int n_out, int pad_left,
                      unsigned flags, void* absmax2, int32_t* argidx, int dtype, void* stream);

/* The fused hot path with the autophase ramp in closed form (A8, processing/phasing.py:62-73: on a uniform axis
 * phi[k] = rad(p0) + rad(p1) * (c[k] - pivot) / range is linear in the output index k):
 *   out[b, k] = X[k] * e^{i (phase0 + dphase * k)},   phase0 / dphase in radians, fp64.
 * Everything else as xm_pipeline_fused (`out` must be given).  On the ">= 2x end zero fill" geometries
 * (xm_pipeline_ramp_native: complex64 with 16-byte aligned rows, complex128 always) the kernel applies the ramp in
 * factorised form -- no table exists, nothing is read per output; other geometries expand the ramp into a
 * stream-ordered scratch table first. */
int xm_pipeline_fused_ramp(const void* in, int64_t in_row_stride, void* out, const void* window, double phase0,
                           double dphase, int64_t n_batch, int n_in, int n_out, int pad_left, unsigned flags,
                           void* absmax2, int32_t* argidx, int dtype, void* stream);
/* 1 if xm_pipeline_fused_ramp applies the ramp natively for this geometry (see above), else 0. */
int xm_pipeline_ramp_native(const void* in, int64_t in_row_stride, int n_in, int n_out, int pad_left, unsigned flags,
                            int dtype);

/* "next" (SURVEY 8f rank 4): asymmetric-least-squares baseline (processing/baseline.py:10-40 `_als_core`
 * applied along the last axis by `xr.apply_ufunc`, :102-110).  For every spectrum: n_iter rounds of
 * (W + lam*D'D) z = W y (pentadiagonal SPD, band LDL' in fp64) and w = p (y > z) + (1 - p) (y < z);
 * out[b, j] = y[b, j] - z[b, j] in float64, y = the REAL part of the input (is_complex) or the input itself
 * (real float32 / float64 for XM_C64 / XM_C128); n >= 4.  `workspace`: device scratch of
 * xm_baseline_als_workspace_bytes(n_batch, n) bytes owned by the caller. */
int64_t xm_baseline_als_workspace_bytes(int64_t n_batch, int n);
int xm_baseline_als(const void* in, int is_complex, int64_t n_batch, int n, double lam, double p, int n_iter, void* out,
                    void* workspace, int64_t workspace_bytes, int dtype, void* stream);

/* ---- A7  host-side autophase search (no GPU involved; O(1) per dataset) ------------------------
 * Objectives of processing/phasing.py:100-157 and the differential-evolution driver the reference
 * reaches through scipy (phasing.py:276-284: best1bin, tol, seed, bounds p0 in [-180,180] deg,
 * p1 in [-4000,4000] deg).  `slice_re_im`: n interleaved complex128 samples of the ONE spectrum through the
 * global maximum; `coords`: its n float64 coordinates; method 0 = acme, 1 = peak_minima, 2 = positivity. */
void* xm_solver_create(const double* slice_re_im, const double* coords, int n, double pivot, int method,
                       int target_idx, int index_width);
void xm_solver_destroy(void* solver);
/* objective value at x = (p0[, p1]) in degrees */
double xm_solver_score(void* solver, const double* x, int nx);
/* `count` parameter vectors, stored 2 doubles apart (p0[, p1] in degrees), evaluated in one hand-off to the worker
 * team -> out[count]; each value equals xm_solver_score's for the same vector */
void xm_solver_score_batch(void* solver, const double* xs, int nx, int count, double* out);
/* objective evaluations performed so far (xm_solver_de evaluates some trials speculatively, so this can exceed the
 * nfev it reports, which counts the evaluations of the sequential algorithm) */
long xm_solver_nfev(void* solver);
/* team size of one objective evaluation inside xm_solver_de (<= 0: min(16, hardware threads / 2); 1 = serial);
 * returns the value set.  Outside xm_solver_de evaluations are always serial. */
int xm_solver_set_threads(void* solver, int threads);
/* f(x) and its forward-difference gradient (n = 1 or 2 parameters in degrees, box [lb, ub]) exactly as scipy's L-BFGS-B
 * requests them for the polish of phasing.py:276-284 (approx_derivative "2-point", abs_step 1e-8, bounds-aware steps);
 * n + 1 evaluations in one batch.  Returns 0, or -1 for bad arguments. */
int xm_solver_fg(void* solver, const double* x, int n, const double* lb, const double* ub, double* f_out, double* g_out);
/* diagnostics: work shares that a search's own thread computed in place of a team member that was not there in time
 * (still asleep, descheduled) -- the value of every evaluation is the same either way */
long xm_solver_pool_backups(void);
/* the differential-evolution generations (no polish); returns 0 = converged, 1 = maxiter reached */
int xm_solver_de(void* solver, int p0_only, unsigned seed, double tol, int maxiter, double* x_out /*[2]*/,
                 double* fun_out, int* nfev_out, int* nit_out);

/* A8, host side: the phase table e^{i phi}, phi[k] = rad(p0) + rad(p1) * (coords[k] - pivot) / (max - min coords)
 * (processing/phasing.py:56-73; scalar phase for a zero range), fp64 arithmetic rounded once to the storage
 * precision.  `out`: n interleaved (re, im) pairs in HOST memory, float32 if as_float != 0 else float64 (the caller's
 * pinned staging buffer for xm_phase_apply / xm_pipeline_fused).  Returns 0, or -1 for bad arguments. */
int xm_phase_table(const double* coords, int n, double p0_deg, double p1_deg, double pivot, void* out, int as_float);

/* ---- A7 on the device: the (p0, p1) search of processing/phasing.py:276-284 for the ACME objective (:100-122) as ONE
 * workgroup beside the streaming kernels (csrc/xm_search.hip): scipy 1.15.3's differential evolution for the
 * reference's configuration (seed -> numpy RandomState stream, latin hypercube, best1bin, dither U[0.5, 1), CR 0.7,
 * immediate updating, tol on std/mean) -- the trial vectors are scipy's bit for bit given equal comparisons of the
 * energies -- followed by the test scipy's L-BFGS-B polish starts with (f and its forward-difference gradient at the
 * best member; projected gradient against pgtol = 1e-5).  `slice`: the n complex128 bins of the spectrum through the
 * global maximum, device-accessible (the selection stage leaves it in pinned host memory); its coordinate axis must
 * be uniform, c[k] = c0 + k cstep, x_range = max c - min c > 0 (phasing.py:56-59).  The pivot is the coordinate of the
 * first arg-max of |slice| (phasing.py:229-238).  `out` (device-accessible, e.g. pinned host memory) is filled when
 * the search ends, `seq` last (system-scope release): poll it with xm_atomic_load_acquire_i64.  Asynchronous on
 * `stream`; the stream's device must be current.  n <= 16576. */
typedef struct {
  double x[2];          /* p0, p1 in degrees (p1 = 0 with p0_only)                                                  */
  double fun;           /* objective at x                                                                            */
  double pg_norm;       /* L-BFGS-B's projected-gradient norm at x (forward differences, approx_derivative's steps) */
  int32_t nfev, nit;    /* evaluations and generations of the differential evolution (the gradient test's n + 1
                           evaluations are not counted)                                                              */
  int32_t status;       /* 0 = converged, 1 = maxiter reached                                                        */
  int32_t needs_polish; /* 0: pg_norm <= pgtol / 2, scipy's polish would return x unchanged; 1: the caller polishes
                           (xmris_amd/autophase_solver.py, the reference's route)                                    */
  int32_t target_idx;   /* first arg-max of |slice| (index of the pivot coordinate)                                  */
  int32_t pad_;
  uint64_t seq;         /* the launch's `seq`, written last                                                          */
  double t_us[8];       /* diagnostics, microseconds as the optimiser's wave saw them: [0] naming the points,
                           [1] phase tables, [2] drawing the next trial's random part (overlaps the workers' sums),
                           [3] waiting for the sums, [4] taking the scores, [5] the whole search                     */
} xm_search_result;
int xm_search_supported(int n, int method, double x_range);
int xm_search_launch(const void* slice, int n, double c0, double cstep, double x_range, int method, int p0_only,
                     unsigned seed, double tol, int maxiter, uint64_t seq, xm_search_result* out, void* stream);
/* The device objective alone (tests; cross-checks against the numpy statement): fs[e] = ACME score at
 * (xs[2e], xs[2e+1]) degrees, e < count, pivot = coordinate of bin `target_idx` (< 0: the first arg-max of |slice|).
 * `xs` / `fs` device-accessible. */
int xm_search_eval(const void* slice, int n, double c0, double cstep, double x_range, int target_idx, int p0_only,
                   const double* xs, int count, double* fs, void* stream);

/* ---- A7 on the host without the interpreter: the same search -- xm_solver_de's generations, then the projected-
 * gradient test of the polish (xm_solver_fg) -- run by a native thread of the library's search service; `out` (HOST
 * memory) is filled like a search kernel fills it, `seq` last (poll it with xm_atomic_load_acquire_i64).  `slice`
 * (n complex128, host-readable) and `coords` (n float64) are copied at submission; `out` must stay valid until the
 * search has ended.  `target_idx` < 0: the first
 * arg-max of |slice| (phasing.py:229); the pivot is coords[target_idx].  `threads`: team size of the generations
 * (<= 0: the library's default).  Returns at once. */
int xm_hostsearch_submit(const void* slice, int n, const double* coords, int method, int target_idx, int index_width,
                         int p0_only, unsigned seed, double tol, int maxiter, int threads, uint64_t seq,
                         xm_search_result* out);

/* searches the service runs side by side (1 ... 8, default 4): further submissions wait in its queue, in order */
int xm_hostsearch_set_workers(int n);

/* ---- streams with a partition of the chip.  A search kernel (above) needs a whole CU's registers for milliseconds,
 * and the streaming kernels are persistent grids sized to fill every CU: sharing one pool of CUs, searches wait for a
 * kernel boundary to start and the streaming kernels then find CUs taken (measured: main pass +7 %, stalls of
 * milliseconds).  So the chip is split: `reserved_cus` CUs -- the first mask bits, i.e. spread round-robin over the
 * eight XCDs -- belong to the search streams (partition 1), the rest to the compute stream (partition 0).  Persistent
 * launches size their grids by the CUs of the stream they are given.  The stream's device is the current one. */
int xm_stream_create(void** stream, int reserved_cus, int partition);
int xm_stream_destroy(void* stream);
/* CUs `stream` may use (all of the device's for an ordinary stream); negative xm_status on failure */
int xm_stream_cus(void* stream);

/* ---- (e) multi-GPU: publication primitives of the one-node O(1) exchange (xmris_amd/sharding.py::ShmExchange; the
 * global arg-max and the one (p0, p1) of phasing.py:229, 276-290 cross the ranks through a shared-memory page).  HOST
 * pointers.  The payload of a slot is written with plain stores, its sequence word with a release store, and readers
 * poll the sequence words with acquire loads. */
int64_t xm_atomic_load_acquire_i64(const int64_t* p);
void xm_atomic_store_release_i64(int64_t* p, int64_t value);
/* 1 once all `count` words (`stride_words` apart) are >= value; 0 if not within `spin_us` microseconds of busy polling */
int xm_atomic_wait_all_ge_i64(const int64_t* p, int stride_words, int count, int64_t value, int spin_us);

#ifdef __cplusplus
}
#endif
#endif /* XMRIS_HIP_H */
