// Host side of libxmris_hip.so: table cache, element-wise kernel launches and the extern "C" ABI
// declared in include/xmris_hip.h.  gfx950 only.  The fused FFT kernels are instantiated per
// storage precision in xm_launch_f32.hip / xm_launch_f64.hip.
#include "xm_host.h"
#include "xm_als.h"
#include "xm_kernels.h"
#include "xm_plans.h"
#include "xm_zfapod.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#define XM_VERSION_NUM 400  // 0.4.0 (round 4: xm_search_*, xm_hostsearch_*, xm_stream_*, xm_zf_apod, xm_atomic_*, xm_last_kernel_string)

static thread_local std::string g_err;
static thread_local std::string g_last_kernel;

#include <cxxabi.h>
void xm_note_kernel(const char* base, const std::type_info* plan, const char* scalar, int mode, int opt) {
  std::string name = base;
  name += "<";
  if (scalar) name += std::string(scalar) + ", ";
  if (plan) {
    int status = 0;
    char* dm = abi::__cxa_demangle(plan->name(), nullptr, nullptr, &status);
    std::string pn = (status == 0 && dm) ? dm : plan->name();
    if (dm) free(dm);
    std::string packed;  // "FftPlan<4096, 256, 16, 16, 16>" -> "FftPlan<4096,256,16,16,16>" (the spelling of the reports)
    for (size_t i = 0; i < pn.size(); ++i)
      if (!(pn[i] == ' ' && i > 0 && pn[i - 1] == ',')) packed += pn[i];
    name += packed + ", ";
  }
  name += std::to_string(mode);
  if (opt >= 0) name += ", " + std::to_string(opt);
  name += ">";
  g_last_kernel = name;
}
int xm_fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
static int fail(int code, const std::string& msg) { return xm_fail(code, msg); }

// ------------------------------------------------------------------------------------------------
// Plan predicates
// ------------------------------------------------------------------------------------------------
bool xm_has_pow2_plan(int n, int dtype) {
  (void)dtype;
  switch (n) {
#define XM_CASE(N, NT, ...) case N:
    XM_PLANS_POW2(XM_CASE)
    return true;
    default:
      return false;
  }
}

bool xm_has_direct_plan(int n, int dtype) {
  switch (n) {
    XM_PLANS_POW2(XM_CASE)
    XM_PLANS_OTHER(XM_CASE)
    return true;
    XM_PLANS_C64_ONLY(XM_CASE)  // 16384: complex128 has a plan of its own (Plan16kD)
    return true;
#undef XM_CASE
    default:
      return false;
  }
}

int xm_bluestein_m(int n) {  // smallest convolution length with a plan >= max(2n-1, 16): 2^k, or 3 * 2^k where built
  int m = 16;
  while (m < 2 * n - 1) m <<= 1;
  static const bool pow2_only = getenv("XM_BLUE_POW2") != nullptr;  // tuning switch
  if (!pow2_only && m == 4096 && 2 * n - 1 <= 3072) return 3072;
  return m;
}

bool xm_supported_in_lds(int n, int dtype) {
  if (n < 2) return false;
  if (xm_has_direct_plan(n, dtype)) return true;
  const int m = xm_bluestein_m(n);
  return xm_has_pow2_plan(m, dtype) || m == 16384 || m == 3072;
}

bool xm_supported(int n, int dtype) {
  if (xm_supported_in_lds(n, dtype)) return true;
  return (dtype == XM_C64 ? xm_big_supported_f32(n) : xm_big_supported_f64(n)) != 0;
}

// ------------------------------------------------------------------------------------------------
// Table cache (per kind, length, dtype, device), mutex-guarded
// ------------------------------------------------------------------------------------------------
struct TableKey {
  int kind, n, m, dtype, device;
  bool operator<(const TableKey& o) const {
    return std::tie(kind, n, m, dtype, device) < std::tie(o.kind, o.n, o.m, o.dtype, o.device);
  }
};
static std::mutex g_mu;
static std::map<TableKey, void*> g_tables;

void xm_unit(long long num, long long den, double sign, double& c, double& s) {
  num %= den;
  if (num < 0) num += den;
  // octant symmetry on the integer fraction keeps the argument of cos/sin in [0, pi/4]
  const long long q = (8 * num) / den;  // octant 0..7
  long long r_num = num;
  double cc, ss;
  switch (q) {
    case 0: { const double a = 2.0 * M_PI * (double)r_num / (double)den; cc = std::cos(a); ss = std::sin(a); break; }
    case 1: { const double a = 2.0 * M_PI * (double)(den - 4 * num) / (double)(4 * den); cc = std::sin(a); ss = std::cos(a); break; }
    case 2: { const double a = 2.0 * M_PI * (double)(4 * num - den) / (double)(4 * den); cc = -std::sin(a); ss = std::cos(a); break; }
    case 3: { const double a = 2.0 * M_PI * (double)(den - 2 * num) / (double)(2 * den); cc = -std::cos(a); ss = std::sin(a); break; }
    case 4: { const double a = 2.0 * M_PI * (double)(2 * num - den) / (double)(2 * den); cc = -std::cos(a); ss = -std::sin(a); break; }
    case 5: { const double a = 2.0 * M_PI * (double)(3 * den - 4 * num) / (double)(4 * den); cc = -std::sin(a); ss = -std::cos(a); break; }
    case 6: { const double a = 2.0 * M_PI * (double)(4 * num - 3 * den) / (double)(4 * den); cc = std::sin(a); ss = -std::cos(a); break; }
    default: { const double a = 2.0 * M_PI * (double)(den - num) / (double)den; cc = std::cos(a); ss = -std::sin(a); break; }
  }
  c = cc;
  s = sign * ss;
}

template <class T>
static int upload(const std::vector<double>& re, const std::vector<double>& im, void** dev) {
  const size_t n = re.size();
  std::vector<Cx<T>> h(n);
  for (size_t i = 0; i < n; ++i) {
    h[i].re = (T)re[i];
    h[i].im = (T)im[i];
  }
  void* d = nullptr;
  HIP_TRY(hipMalloc(&d, (n ? n : 1) * sizeof(Cx<T>)));
  if (n) HIP_TRY(hipMemcpy(d, h.data(), n * sizeof(Cx<T>), hipMemcpyHostToDevice));
  *dev = d;
  return XM_OK;
}

int xm_table_get(int kind, int n, int m, int dtype, xm_table_gen gen, const void* ctx, const void** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  TableKey key{kind, n, m, dtype, dev};
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_tables.find(key);
  if (it == g_tables.end()) {
    std::vector<double> re, im;
    gen(n, m, ctx, re, im);
    void* d = nullptr;
    int rc = dtype == XM_C64 ? upload<float>(re, im, &d) : upload<double>(re, im, &d);
    if (rc) return rc;
    it = g_tables.emplace(key, d).first;
  }
  *out = it->second;
  return XM_OK;
}

// ------------------------------------------------------------------------------------------------
// Row-queue counters of the persistent kernels that hand out rows dynamically: a ring of {head, done} pairs per
// device, zeroed once; every launch takes the next slot and its last workgroup leaves the pair zero again.  A slot
// comes round again after 1024 launches on the device -- far more than can be in flight at once.
// ------------------------------------------------------------------------------------------------
constexpr unsigned kQueueSlots = 1024, kQueueStride = 16;  // 64 bytes per slot
static unsigned* g_queue_base[16] = {nullptr};
static unsigned g_queue_next[16] = {0};

int xm_queue_slot(unsigned** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return fail(XM_ERR_INVALID_ARG, "device ordinal out of range");
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_queue_base[dev]) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, kQueueSlots * kQueueStride * sizeof(unsigned)));
    HIP_TRY(hipMemset(p, 0, kQueueSlots * kQueueStride * sizeof(unsigned)));
    g_queue_base[dev] = (unsigned*)p;
  }
  *out = g_queue_base[dev] + (size_t)(g_queue_next[dev]++ % kQueueSlots) * kQueueStride;
  return XM_OK;
}

// e^{i (phase0 + dphase k)} for the kernels that take the phase as a table
template <class T>
__global__ void k_ramp_table(Cx<T>* table, int n, double phase0, double dphase) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) {
    double sn, cs;
    sincos(phase0 + dphase * (double)k, &sn, &cs);
    table[k] = mk<T>((T)cs, (T)sn);
  }
}

int xm_ramp_table_async(void* table, int n, double phase0, double dphase, int dtype, hipStream_t st) {
  if (dtype == XM_C64)
    hipLaunchKernelGGL(k_ramp_table<float>, dim3((n + 255) / 256), dim3(256), 0, st, (Cx<float>*)table, n, phase0, dphase);
  else
    hipLaunchKernelGGL(k_ramp_table<double>, dim3((n + 255) / 256), dim3(256), 0, st, (Cx<double>*)table, n, phase0, dphase);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

// The library keys its table cache, occupancy caches and scratch on the CURRENT device; the caller's buffers decide
// which device that has to be.  Entry points that launch kernels make the device of their (device-memory) input
// current for the duration of the call.
struct DeviceGuard {
  int prev = -1;
  explicit DeviceGuard(const void* dev_ptr) {
    if (!dev_ptr) return;
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, dev_ptr) != hipSuccess) {
      (void)hipGetLastError();
      return;
    }
    if (at.type != hipMemoryTypeDevice) return;
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return;
    if (cur != at.device && hipSetDevice(at.device) == hipSuccess) prev = cur;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

static int grid_for(long long total, int block) {
  long long g = (total + block - 1) / block;
  const long long cap = 256LL * 16;  // 256 CUs x 16 workgroups, grid-stride the rest
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

static int check_common(const void* in, int64_t n_batch, int n, int dtype) {
  if (!in && n_batch > 0) return fail(XM_ERR_INVALID_ARG, "null input pointer");
  if (n_batch < 0 || n < 1) return fail(XM_ERR_INVALID_ARG, "negative batch or non-positive length");
  if (dtype != XM_C64 && dtype != XM_C128) return fail(XM_ERR_INVALID_ARG, "dtype must be XM_C64 or XM_C128");
  return XM_OK;
}

// ------------------------------------------------------------------------------------------------
// extern "C"
// ------------------------------------------------------------------------------------------------
// A1 + A2 in one pass (fid.py:251 then :136-139), see xm_zfapod.h
template <class TI, class TO>
static int launch_zf_apod(const void* in, int64_t in_stride, void* out, const void* window, int64_t n_batch, int n_in,
                          int n_out, int pad_left, hipStream_t st) {
  ZfApodArgs<TI, TO> A;
  A.in = (const Cx<TI>*)in;
  A.out = (Cx<TO>*)out;
  A.window = (const TO*)window;
  A.in_stride = in_stride;
  A.n_batch = n_batch;
  A.n_in = n_in;
  A.n_out = n_out;
  A.pad_left = pad_left;
  const size_t lds = (size_t)(n_in < n_out ? n_in : n_out) * sizeof(TO);  // (the window over the acquired samples)
  const size_t esz = sizeof(Cx<TI>);
  // 16-byte lanes: rows of both arrays on 16-byte boundaries, and (8-byte elements) whole pairs inside the samples
  const bool vec = (reinterpret_cast<size_t>(in) % 16 == 0) && (reinterpret_cast<size_t>(out) % 16 == 0) &&
                   ((size_t)in_stride * esz) % 16 == 0 && ((size_t)n_out * sizeof(Cx<TO>)) % 16 == 0 &&
                   (esz == 16 || (pad_left % 2 == 0 && n_in % 2 == 0 && n_out % 2 == 0));
  int rc = xm_queue_slot(&A.queue);
  if (rc) return rc;
  int resident = 0;
  if (vec) {
    static XmResidency res;
    rc = xm_resident_blocks(res, k_zf_apod<TI, TO, true>, 256, lds, &resident, st);
    if (rc) return rc;
    // Few workgroups per CU: measured on 65,536 x 4096 -> 8192 complex64 with 1 / 2 / 3 / 4 / 5 per CU: 3.27 / 5.40 /
    // 5.63 / 5.34 / 5.33 TB/s, with ten 4.73; rows of complex128 out (twice the bytes per workgroup and step): 2 / 3 / 4
    // per CU 5.48 / 5.31 / 5.06, promoted complex64 -> complex128 4.77 / 4.21 / 3.96 (profiles/r04/zf_apod.txt) -- a
    // streaming copy wants few, deep streams: three for 8-byte outputs, two for 16-byte ones
    static const int wgs_env = getenv("XM_ZFAPOD_WGS") ? atoi(getenv("XM_ZFAPOD_WGS")) : (sizeof(TO) == 4 ? 3 : 2);  // tuning switch
    int cus = 0;
    if (wgs_env > 0 && xm_stream_cu_count(st, &cus) == XM_OK && cus > 0 && resident > wgs_env * cus) resident = wgs_env * cus;
    const long long blocks = n_batch < resident ? n_batch : resident;
    xm_note_kernel("k_zf_apod", nullptr, sizeof(TI) == 4 ? (sizeof(TO) == 4 ? "float, float" : "float, double") : "double, double", 1, -1);
    hipLaunchKernelGGL((k_zf_apod<TI, TO, true>), dim3((unsigned)blocks), dim3(256), lds, st, A);
  } else {
    static XmResidency res;
    rc = xm_resident_blocks(res, k_zf_apod<TI, TO, false>, 256, lds, &resident, st);
    if (rc) return rc;
    const long long blocks = n_batch < resident ? n_batch : resident;
    xm_note_kernel("k_zf_apod", nullptr, sizeof(TI) == 4 ? (sizeof(TO) == 4 ? "float, float" : "float, double") : "double, double", 0, -1);
    hipLaunchKernelGGL((k_zf_apod<TI, TO, false>), dim3((unsigned)blocks), dim3(256), lds, st, A);
  }
  HIP_TRY(hipGetLastError());
  return XM_OK;
}


extern "C" {

int xm_version(void) { return XM_VERSION_NUM; }

const char* xm_last_error_string(void) { return g_err.c_str(); }

// ------------------------------------------------------------------------------------------------
// Streams with a partition of the chip (hipExtStreamCreateWithCUMask).  Mask bit i is CU slot i / 8 of XCD i % 8
// (probed: tools/cu_mask_probe.hip), so the first R bits are R CUs spread evenly over the eight XCDs.
// ------------------------------------------------------------------------------------------------
static std::mutex g_stream_mu;
static std::map<hipStream_t, int> g_stream_cus;

int xm_stream_cu_count(hipStream_t st, int* cus) {
  {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    auto it = g_stream_cus.find(st);
    if (it != g_stream_cus.end()) {
      *cus = it->second;
      return XM_OK;
    }
  }
  int dev = 0, total = 0;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev));
  int n = total;
  if (st != nullptr) {
    uint32_t mask[16] = {0};
    const int words = (total + 31) / 32 < 16 ? (total + 31) / 32 : 16;
    if (hipExtStreamGetCUMask(st, (uint32_t)words, mask) == hipSuccess) {
      int c = 0;
      for (int w = 0; w < words; ++w) c += __builtin_popcount(mask[w]);
      if (c > 0 && c <= total) n = c;
    } else {
      (void)hipGetLastError();
    }
  }
  std::lock_guard<std::mutex> lk(g_stream_mu);
  g_stream_cus[st] = n;
  *cus = n;
  return XM_OK;
}

extern "C" int xm_stream_create(void** stream, int reserved_cus, int partition) {
  if (!stream || reserved_cus < 0 || (partition != 0 && partition != 1)) return fail(XM_ERR_INVALID_ARG, "xm_stream_create: bad argument");
  int dev = 0, total = 0;
  HIP_TRY(hipGetDevice(&dev));
  HIP_TRY(hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev));
  if (reserved_cus >= total || (partition == 1 && reserved_cus == 0)) return fail(XM_ERR_INVALID_ARG, "xm_stream_create: bad partition");
  const int words = (total + 31) / 32;
  std::vector<uint32_t> mask((size_t)words, 0u);
  for (int i = 0; i < total; ++i) {
    const bool reserved = i < reserved_cus;
    if (reserved == (partition == 1)) mask[i / 32] |= 1u << (i % 32);
  }
  hipStream_t st = nullptr;
  HIP_TRY(hipExtStreamCreateWithCUMask(&st, (uint32_t)words, mask.data()));
  {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_stream_cus[st] = partition == 1 ? reserved_cus : total - reserved_cus;
  }
  *stream = (void*)st;
  return XM_OK;
}

extern "C" int xm_stream_destroy(void* stream) {
  if (!stream) return XM_OK;
  {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_stream_cus.erase((hipStream_t)stream);
  }
  HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return XM_OK;
}

extern "C" int xm_stream_cus(void* stream) {
  int cus = 0;
  const int rc = xm_stream_cu_count((hipStream_t)stream, &cus);
  return rc ? rc : cus;
}

const char* xm_last_kernel_string(void) { return g_last_kernel.c_str(); }

int xm_clear_cache(void) {
  {
    std::lock_guard<std::mutex> lk(g_stream_mu);
    g_stream_cus.clear();
  }
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& kv : g_tables) (void)hipFree(kv.second);
  g_tables.clear();
  for (int d = 0; d < 16; ++d) {
    if (g_queue_base[d]) (void)hipFree(g_queue_base[d]);  // NB: tables and rings are freed with their device current
    g_queue_base[d] = nullptr;
  }
  return XM_OK;
}

int xm_fft_supported(int n, int dtype) { return xm_supported(n, dtype) ? 1 : 0; }

int xm_plan_prepare(int n, int dtype) {
  // run the pipeline on an empty batch: the dispatch makes the same decisions as a real call of that geometry and
  // builds + caches the tables of the kernel it picks (twiddles of that kernel's plan, half-length rotation, chirps)
  if (!xm_supported(n, dtype)) return fail(XM_ERR_UNSUPPORTED_N, "unsupported length " + std::to_string(n));
  int rc = XM_OK;
  // no zero fill, and (even n) the 2x end zero fill of n/2 samples that the hot path runs
  const int n_ins[2] = {n, n / 2};
  for (int i = 0; i < (n % 2 == 0 && n >= 4 ? 2 : 1) && rc == XM_OK; ++i) {
    const unsigned fl = XM_FFT_ORTHO | XM_FFT_SHIFT_OUT;
    rc = dtype == XM_C64
             ? xm_pipeline_f32(nullptr, n_ins[i], nullptr, nullptr, nullptr, nullptr, 0, n_ins[i], n, 0, fl, nullptr, nullptr, nullptr)
             : xm_pipeline_f64(nullptr, n_ins[i], nullptr, nullptr, nullptr, nullptr, 0, n_ins[i], n, 0, fl, nullptr, nullptr, nullptr);
  }
  return rc;
}

int xm_zero_fill(const void* in, void* out, int64_t n_batch, int n_in, int n_out, int pad_left, int dtype,
                 void* stream) {
  int rc = check_common(in, n_batch, n_in, dtype);
  if (rc) return rc;
  if (!out || n_out < n_in || pad_left < 0 || pad_left + n_in > n_out)
    return fail(XM_ERR_INVALID_ARG, "zero_fill: bad output pointer or geometry");
  if (n_batch == 0) return XM_OK;
  const long long total = (long long)n_batch * n_out;
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(in);
  if (dtype == XM_C64)
    hipLaunchKernelGGL(k_zero_fill<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const Cx<float>*)in,
                       (Cx<float>*)out, (long long)n_batch, n_in, n_out, pad_left);
  else
    hipLaunchKernelGGL(k_zero_fill<double>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const Cx<double>*)in,
                       (Cx<double>*)out, (long long)n_batch, n_in, n_out, pad_left);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_apodize(const void* in, void* out, const void* window, int64_t n_batch, int n, int dtype, void* stream) {
  int rc = check_common(in, n_batch, n, dtype);
  if (rc) return rc;
  if (!out || !window) return fail(XM_ERR_INVALID_ARG, "apodize: null output or window");
  if (n_batch == 0) return XM_OK;
  const long long total = (long long)n_batch * n;
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(in);
  if (dtype == XM_C64)
    hipLaunchKernelGGL(k_apodize<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const Cx<float>*)in,
                       (Cx<float>*)out, (const float*)window, (long long)n_batch, n);
  else
    hipLaunchKernelGGL(k_apodize<double>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const Cx<double>*)in,
                       (Cx<double>*)out, (const double*)window, (long long)n_batch, n);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_zf_apod(const void* in, int64_t in_row_stride, void* out, const void* window, int64_t n_batch, int n_in, int n_out,
               int pad_left, int in_dtype, int out_dtype, void* stream) {
  int rc = check_common(in, n_batch, n_in, in_dtype);
  if (rc) return rc;
  if (!out || !window || n_out < n_in || pad_left < 0 || pad_left + n_in > n_out || in_row_stride < n_in || in == out)
    return fail(XM_ERR_INVALID_ARG, "zf_apod: bad pointer or geometry");
  if ((out_dtype != XM_C64 && out_dtype != XM_C128) || (in_dtype == XM_C128 && out_dtype == XM_C64))
    return fail(XM_ERR_INVALID_ARG, "zf_apod: the output precision is the input's or complex128");
  if ((size_t)n_out * (out_dtype == XM_C64 ? 4 : 8) > 96 * 1024)
    return fail(XM_ERR_UNSUPPORTED_N, "zf_apod: the window does not fit the LDS (use xm_zero_fill + xm_apodize)");
  if (n_batch == 0) return XM_OK;
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(in);
  if (in_dtype == XM_C64 && out_dtype == XM_C64)
    return launch_zf_apod<float, float>(in, in_row_stride, out, window, n_batch, n_in, n_out, pad_left, st);
  if (in_dtype == XM_C64)
    return launch_zf_apod<float, double>(in, in_row_stride, out, window, n_batch, n_in, n_out, pad_left, st);
  return launch_zf_apod<double, double>(in, in_row_stride, out, window, n_batch, n_in, n_out, pad_left, st);
}

int xm_phase_apply(const void* in, void* out, const void* phase_table, int64_t n_batch, int n, int dtype,
                   void* stream) {
  int rc = check_common(in, n_batch, n, dtype);
  if (rc) return rc;
  if (!out || !phase_table) return fail(XM_ERR_INVALID_ARG, "phase_apply: null output or table");
  if (n_batch == 0) return XM_OK;
  const long long total = (long long)n_batch * n;
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(in);
  if (dtype == XM_C64)
    hipLaunchKernelGGL(k_phase<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const Cx<float>*)in,
                       (Cx<float>*)out, (const Cx<float>*)phase_table, (long long)n_batch, n);
  else
    hipLaunchKernelGGL(k_phase<double>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const Cx<double>*)in,
                       (Cx<double>*)out, (const Cx<double>*)phase_table, (long long)n_batch, n);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_roll(const void* in, void* out, int64_t n_batch, int n, int shift, int dtype, void* stream) {
  int rc = check_common(in, n_batch, n, dtype);
  if (rc) return rc;
  if (!out || in == out) return fail(XM_ERR_INVALID_ARG, "roll: output must be a distinct buffer");
  if (n_batch == 0) return XM_OK;
  shift %= n;
  if (shift < 0) shift += n;
  const long long total = (long long)n_batch * n;
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(in);
  if (dtype == XM_C64)
    hipLaunchKernelGGL(k_roll<float>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const Cx<float>*)in,
                       (Cx<float>*)out, (long long)n_batch, n, shift);
  else
    hipLaunchKernelGGL(k_roll<double>, dim3(grid_for(total, 256)), dim3(256), 0, st, (const Cx<double>*)in,
                       (Cx<double>*)out, (long long)n_batch, n, shift);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_absmax_rows(const void* in, int64_t n_batch, int n, void* absmax2, int32_t* argidx, int dtype,
                   void* stream) {
  int rc = check_common(in, n_batch, n, dtype);
  if (rc) return rc;
  if (!absmax2 || !argidx) return fail(XM_ERR_INVALID_ARG, "absmax_rows: null outputs");
  if (n_batch == 0) return XM_OK;
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(in);
  const int grid = (int)(n_batch < 65536 * 4 ? n_batch : 65536 * 4);
  if (dtype == XM_C64)
    hipLaunchKernelGGL(k_absmax_rows<float>, dim3(grid), dim3(256), 0, st, (const Cx<float>*)in,
                       (long long)n_batch, n, (float*)absmax2, argidx);
  else
    hipLaunchKernelGGL(k_absmax_rows<double>, dim3(grid), dim3(256), 0, st, (const Cx<double>*)in,
                       (long long)n_batch, n, (double*)absmax2, argidx);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_row_l1(const void* in, int64_t in_row_stride, const void* window, int64_t n_batch, int n_in, int pad_left,
              int sub_step, void* norm, uint64_t* key, int dtype, void* stream) {
  int rc = check_common(in, n_batch, n_in, dtype);
  if (rc) return rc;
  if ((!norm && !key) || pad_left < 0 || in_row_stride < n_in || sub_step < 1)
    return fail(XM_ERR_INVALID_ARG, "row_l1: bad arguments");
  if (key && (dtype != XM_C64 || n_batch > 0xffffffffLL))
    return fail(XM_ERR_INVALID_ARG, "row_l1: the arg-max key needs complex64 and fewer than 2^32 rows");
  if (n_batch == 0) return XM_OK;
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(in);
  const int64_t want = (n_batch + 3) / 4;  // four rows (waves) per workgroup
  const int grid = (int)(want < 256 * 8 ? want : 256 * 8);
  if (dtype == XM_C64)
    hipLaunchKernelGGL(k_row_l1<float>, dim3(grid), dim3(256), 0, st, (const Cx<float>*)in, (long long)in_row_stride,
                       (const float*)window, (long long)n_batch, n_in, pad_left, sub_step, (float*)norm,
                       (unsigned long long*)key);
  else
    hipLaunchKernelGGL(k_row_l1<double>, dim3(grid), dim3(256), 0, st, (const Cx<double>*)in,
                       (long long)in_row_stride, (const double*)window, (long long)n_batch, n_in, pad_left, sub_step,
                       (double*)norm, (unsigned long long*)nullptr);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_argmax_reduce(const void* absmax2, const int32_t* argidx, int64_t n_batch, int n, void* out_max2,
                     int64_t* out_flat, int dtype, void* stream) {
  if (!absmax2 || !argidx || !out_max2 || !out_flat || n_batch < 1 || n < 1)
    return fail(XM_ERR_INVALID_ARG, "argmax_reduce: null pointer or empty batch");
  if (dtype != XM_C64 && dtype != XM_C128) return fail(XM_ERR_INVALID_ARG, "bad dtype");
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(absmax2);
  if (dtype == XM_C64)
    hipLaunchKernelGGL(k_argmax_final<float>, dim3(1), dim3(1024), 0, st, (const float*)absmax2, argidx,
                       (long long)n_batch, n, (float*)out_max2, (long long*)out_flat);
  else
    hipLaunchKernelGGL(k_argmax_final<double>, dim3(1), dim3(1024), 0, st, (const double*)absmax2, argidx,
                       (long long)n_batch, n, (double*)out_max2, (long long*)out_flat);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_argmax_key_take(uint64_t* key, int n_per_row, void* out_max2, int64_t* out_flat, const void* in,
                       int64_t in_row_stride, int n_in, void* out_row, int dtype, void* stream) {
  if (!key || !out_max2 || !out_flat || n_per_row < 1) return fail(XM_ERR_INVALID_ARG, "argmax_key_take: null pointer");
  if (in && (!out_row || n_in < 1 || in_row_stride < n_in)) return fail(XM_ERR_INVALID_ARG, "argmax_key_take: bad row geometry");
  if (dtype != XM_C64 && dtype != XM_C128) return fail(XM_ERR_INVALID_ARG, "bad dtype");
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(key);
  if (dtype == XM_C64)
    hipLaunchKernelGGL(k_key_take<float>, dim3(1), dim3(1024), 0, st, (unsigned long long*)key, n_per_row, (float*)out_max2,
                       (long long*)out_flat, (const Cx<float>*)in, (long long)in_row_stride, n_in, (Cx<double>*)out_row);
  else
    hipLaunchKernelGGL(k_key_take<double>, dim3(1), dim3(1024), 0, st, (unsigned long long*)key, n_per_row, (float*)out_max2,
                       (long long*)out_flat, (const Cx<double>*)in, (long long)in_row_stride, n_in, (Cx<double>*)out_row);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_guess_supported(const void* in, int64_t in_row_stride, int n_in, int n_out, int pad_left, unsigned flags, int dtype) {
  if (!in || (dtype != XM_C64 && dtype != XM_C128) || in_row_stride < n_in) return 0;
  return xm_zf2p_guess_supported(in, in_row_stride, n_in, n_out, pad_left, flags, dtype);
}

static float guess_scale(int n_out, unsigned flags) {
  return (flags & XM_FFT_ORTHO) ? (float)(1.0 / std::sqrt((double)n_out)) : 1.0f;
}

int xm_guess_rows(const void* in, int64_t in_row_stride, const void* window, int64_t n_batch, int n_in, int n_out,
                  int n_guess, unsigned flags, float* est, uint64_t* key, int dtype, void* stream) {
  int rc = check_common(in, n_batch, n_out, dtype);
  if (rc) return rc;
  if (!est || !key || n_guess < 0 || n_batch > 0xffffffffLL)
    return fail(XM_ERR_INVALID_ARG, "guess_rows: null output or bad sizes");
  if (!xm_guess_supported(in, in_row_stride, n_in, n_out, 0, flags, dtype))
    return fail(XM_ERR_INVALID_ARG, "guess_rows: geometry outside xm_guess_supported");
  if (n_batch == 0) return XM_OK;
  DeviceGuard guard(in);
  return xm_zf2p_guess_rows(in, in_row_stride, (const float*)window, n_batch, n_in, n_out, n_guess,
                            guess_scale(n_out, flags), est, (unsigned long long*)key, dtype, (hipStream_t)stream);
}

int xm_guess_refine(const void* in, int64_t in_row_stride, const void* window, int64_t n_batch, int n_in, int n_out,
                    unsigned flags, const float* est, uint64_t* guess_key, float band, uint64_t* work_key,
                    float* out_max2, int64_t* out_flat, void* out_row, int dtype, void* stream) {
  int rc = check_common(in, n_batch, n_out, dtype);
  if (rc) return rc;
  if (!est || !guess_key || !work_key || work_key == guess_key || !out_flat || n_batch < 1 || n_batch > 0xffffffffLL ||
      !(band > 0.0f && band <= 1.0f))
    return fail(XM_ERR_INVALID_ARG, "guess_refine: null pointer, empty batch or band outside (0, 1]");
  if (!xm_guess_supported(in, in_row_stride, n_in, n_out, 0, flags, dtype))
    return fail(XM_ERR_INVALID_ARG, "guess_refine: geometry outside xm_guess_supported");
  DeviceGuard guard(in);
  return xm_zf2p_guess_refine(in, in_row_stride, (const float*)window, n_batch, n_in, n_out, flags,
                              guess_scale(n_out, flags), est, (unsigned long long*)guess_key, band,
                              (unsigned long long*)work_key, out_max2, (long long*)out_flat, out_row, dtype,
                              (hipStream_t)stream);
}

int64_t xm_baseline_als_workspace_bytes(int64_t n_batch, int n) {
  if (n_batch < 0 || n < 0) return 0;
  return 5 * n_batch * (int64_t)n * (int64_t)sizeof(double);
}

int xm_baseline_als(const void* in, int is_complex, int64_t n_batch, int n, double lam, double p, int n_iter, void* out,
                    void* workspace, int64_t workspace_bytes, int dtype, void* stream) {
  if ((!in || !out || !workspace) && n_batch > 0) return fail(XM_ERR_INVALID_ARG, "baseline_als: null pointer");
  // n == 3: the band of D'D has no interior rows and the kernel's hard-coded ends would overlap
  if (n_batch < 0 || n < 4 || n_iter < 1) return fail(XM_ERR_INVALID_ARG, "baseline_als: needs n >= 4, n_iter >= 1");
  if (dtype != XM_C64 && dtype != XM_C128) return fail(XM_ERR_INVALID_ARG, "bad dtype");
  if (workspace_bytes < xm_baseline_als_workspace_bytes(n_batch, n))
    return fail(XM_ERR_INVALID_ARG, "baseline_als: workspace too small (see xm_baseline_als_workspace_bytes)");
  if (n_batch == 0) return XM_OK;
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(in);
  const long long plane = (long long)n_batch * n;
  double* yt = (double*)workspace;
  double *zt = yt + plane, *l1t = yt + 2 * plane, *l2t = yt + 3 * plane, *vt = yt + 4 * plane;
  const long long by = (n_batch + 31) / 32;
  if (by > 65535) return fail(XM_ERR_INVALID_ARG, "baseline_als: n_batch too large for one launch (> 2M spectra)");
  dim3 tgrid((n + 31) / 32, (unsigned)by), tblock(32, 8);
  if (dtype == XM_C64) {
    if (is_complex)
      hipLaunchKernelGGL((k_als_transpose_in<float, true>), tgrid, tblock, 0, st, (const float*)in, (long long)n_batch, n, yt);
    else
      hipLaunchKernelGGL((k_als_transpose_in<float, false>), tgrid, tblock, 0, st, (const float*)in, (long long)n_batch, n, yt);
  } else {
    if (is_complex)
      hipLaunchKernelGGL((k_als_transpose_in<double, true>), tgrid, tblock, 0, st, (const double*)in, (long long)n_batch, n, yt);
    else
      hipLaunchKernelGGL((k_als_transpose_in<double, false>), tgrid, tblock, 0, st, (const double*)in, (long long)n_batch, n, yt);
  }
  hipLaunchKernelGGL(k_als_solve, dim3((unsigned)((n_batch + 63) / 64)), dim3(64), 0, st, yt, zt, l1t, l2t, vt,
                     (long long)n_batch, n, lam, p, n_iter);
  hipLaunchKernelGGL(k_als_transpose_out, tgrid, tblock, 0, st, yt, zt, (long long)n_batch, n, (double*)out);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

int xm_gather_row_c128(const void* in, int64_t in_row_stride, int n_in, const int64_t* flat_index, int n_per_row,
                       void* out, int dtype, void* stream) {
  if (!in || !flat_index || !out || n_in < 1 || n_per_row < 1 || in_row_stride < n_in)
    return fail(XM_ERR_INVALID_ARG, "gather_row: null pointer or bad geometry");
  if (dtype != XM_C64 && dtype != XM_C128) return fail(XM_ERR_INVALID_ARG, "bad dtype");
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(in);
  const int grid = (n_in + 255) / 256;
  if (dtype == XM_C64)
    hipLaunchKernelGGL(k_gather_row<float>, dim3(grid), dim3(256), 0, st, (const Cx<float>*)in,
                       (long long)in_row_stride, n_in, (const long long*)flat_index, n_per_row, (Cx<double>*)out);
  else
    hipLaunchKernelGGL(k_gather_row<double>, dim3(grid), dim3(256), 0, st, (const Cx<double>*)in,
                       (long long)in_row_stride, n_in, (const long long*)flat_index, n_per_row, (Cx<double>*)out);
  HIP_TRY(hipGetLastError());
  return XM_OK;
}

static int pipeline_common(const void* in, int64_t in_row_stride, void* out, const void* window,
                           const void* phase_table, const double* ramp, int64_t n_batch, int n_in, int n_out,
                           int pad_left, unsigned flags, void* absmax2, int32_t* argidx, int dtype, void* stream) {
  int rc = check_common(in, n_batch, n_out, dtype);
  if (rc) return rc;
  if (n_in < 1 || pad_left < 0 || pad_left + n_in > n_out || in_row_stride < n_in)
    return fail(XM_ERR_INVALID_ARG, "pipeline: bad zero-fill geometry or row stride");
  if (flags & XM_AMAX_GLOBAL_KEY) {
    if (!absmax2 || n_batch > 0xffffffffLL ||
        !xm_pipeline_key_native(in, in_row_stride, n_in, n_out, pad_left, flags & ~(XM_AMAX_GLOBAL_KEY | XM_AMAX_VALUE_ONLY), dtype))
      return fail(XM_ERR_INVALID_ARG, "pipeline: XM_AMAX_GLOBAL_KEY needs a key and a geometry of xm_pipeline_key_native");
    if (dtype == XM_C128 && !argidx)
      return fail(XM_ERR_INVALID_ARG, "pipeline: complex128 arg-max keys are decoded by the launch itself (result record needed)");
  } else if ((absmax2 == nullptr) != (argidx == nullptr))
    return fail(XM_ERR_INVALID_ARG, "pipeline: absmax2 and argidx must be given together");
  if (!out && !absmax2) return fail(XM_ERR_INVALID_ARG, "pipeline: nothing to produce");
  if (out == in) return fail(XM_ERR_INVALID_ARG, "pipeline: in-place operation is not supported");
  if (ramp && !out) return fail(XM_ERR_INVALID_ARG, "pipeline: a phase ramp needs an output");
  if (flags & ~(XM_FFT_INVERSE | XM_FFT_ORTHO | XM_FFT_SHIFT_IN | XM_FFT_SHIFT_OUT | XM_AMAX_VALUE_ONLY | XM_AMAX_GLOBAL_KEY))
    return fail(XM_ERR_INVALID_ARG, "pipeline: unknown flag bits");
  if (!xm_supported(n_out, dtype))
    return fail(XM_ERR_UNSUPPORTED_N, "no plan for length " + std::to_string(n_out));
  hipStream_t st = (hipStream_t)stream;
  DeviceGuard guard(in);
  if (dtype == XM_C64)
    return xm_pipeline_f32(in, in_row_stride, out, window, phase_table, ramp, n_batch, n_in, n_out, pad_left, flags,
                           absmax2, argidx, st);
  return xm_pipeline_f64(in, in_row_stride, out, window, phase_table, ramp, n_batch, n_in, n_out, pad_left, flags,
                         absmax2, argidx, st);
}

int xm_pipeline_fused(const void* in, int64_t in_row_stride, void* out, const void* window,
                      const void* phase_table, int64_t n_batch, int n_in, int n_out, int pad_left, unsigned flags,
                      void* absmax2, int32_t* argidx, int dtype, void* stream) {
  return pipeline_common(in, in_row_stride, out, window, phase_table, nullptr, n_batch, n_in, n_out, pad_left, flags,
                         absmax2, argidx, dtype, stream);
}

int xm_pipeline_fused_ramp(const void* in, int64_t in_row_stride, void* out, const void* window, double phase0,
                           double dphase, int64_t n_batch, int n_in, int n_out, int pad_left, unsigned flags,
                           void* absmax2, int32_t* argidx, int dtype, void* stream) {
  const double ramp[2] = {phase0, dphase};
  return pipeline_common(in, in_row_stride, out, window, nullptr, ramp, n_batch, n_in, n_out, pad_left, flags, absmax2,
                         argidx, dtype, stream);
}

int xm_pipeline_ramp_native(const void* in, int64_t in_row_stride, int n_in, int n_out, int pad_left, unsigned flags,
                            int dtype) {
  if ((dtype != XM_C64 && dtype != XM_C128) || n_in < 1 || pad_left < 0 || pad_left + n_in > n_out) return 0;
  return dtype == XM_C64 ? xm_ramp_native_f32(in, in_row_stride, n_in, n_out, pad_left, flags)
                         : xm_ramp_native_f64(in, in_row_stride, n_in, n_out, pad_left, flags);
}

int xm_pipeline_key_native(const void* in, int64_t in_row_stride, int n_in, int n_out, int pad_left, unsigned flags,
                           int dtype) {
  if ((dtype != XM_C64 && dtype != XM_C128) || n_in < 1 || pad_left < 0 || pad_left + n_in > n_out) return 0;
  if (dtype == XM_C64) return xm_key_native_f32(in, in_row_stride, n_in, n_out, pad_left, flags);
  static const bool gen1 = getenv("XM_ZF2D_GEN1") != nullptr;  // (tuning switch: k_zf2<double> has no key)
  return !gen1 && xm_key_native_f64(in, in_row_stride, n_in, n_out, pad_left, flags);
}

int xm_fft1d_batched(const void* in, void* out, int64_t n_batch, int n, unsigned flags, int dtype, void* stream) {
  if (!out) return fail(XM_ERR_INVALID_ARG, "fft: null output");
  return xm_pipeline_fused(in, n, out, nullptr, nullptr, n_batch, n, n, 0, flags, nullptr, nullptr, dtype, stream);
}

}  // extern "C"
