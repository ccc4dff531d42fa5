// Development bench for the hot kernel (4096 -> 8192 complex64): times k_zf2 (first generation) and the k_zf2p
// variants back to back on one box and cross-checks their outputs (against each other on every row, against an
// fp64 host DFT on two rows).  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ixmris_amd/csrc tools/zf2_lab.hip -o tools/zf2_lab
#include "xm_zf2p.h"
#include "xm_plans.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static Cx<float> hc(float re, float im) {
  Cx<float> c;
  c.re = re;
  c.im = im;
  return c;
}

using PL = PlanOf<4096>::type;  // 256 threads x 16 points, radices 16.16.16
constexpr int H = 4096, N = 8192;

// the kernel's memory pattern without its arithmetic (tools/write_patterns.hip, "interleaved"): persistent 256-thread
// workgroups, 32 KiB row in (8 B/lane x 16, next row prefetched), 64 KiB row out (16 B/lane x 16)
template <int AUX>
__global__ __launch_bounds__(256, 2) void k_rows(PipeArgs<float> A) {
  const unsigned t = threadIdx.x;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 x[16];
  auto fetch = [&](long long r) {
    const f2* row = reinterpret_cast<const f2*>(A.in) + r * 4096;
#pragma unroll
    for (int q = 0; q < 16; ++q) x[q] = row[t + 256 * q];
  };
  long long s = blockIdx.x;
  if (s < A.n_batch) fetch(s);
  for (; s < A.n_batch; s += gridDim.x) {
    xm_u4 y[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) y[q] = xm_u4{__float_as_uint(x[q].x), __float_as_uint(x[q].y), __float_as_uint(-x[q].y), __float_as_uint(x[q].x)};
    if (s + gridDim.x < A.n_batch) fetch(s + gridDim.x);
    const __amdgpu_buffer_rsrc_t rs = xm_rsrc(A.out + s * 8192, 65536u);
#pragma unroll
    for (int q = 0; q < 16; ++q) __builtin_amdgcn_raw_buffer_store_b128(y[q], rs, (256u * q + t) * 16u, 0, AUX);
  }
}

template <class PLAN>
std::vector<Cx<float>> twiddles() {
  std::vector<Cx<float>> tw(PLAN::tw_size());
  for (int s = 1; s < PLAN::K; ++s) {
    const int R = PLAN::radix(s), Ns = PLAN::ns(s), off = PLAN::tw_offset(s);
    for (int r = 1; r < R; ++r)
      for (int k = 0; k < Ns; ++k) {
        const double a = -2.0 * M_PI * (double)((long long)r * k % ((long long)Ns * R)) / (double)((long long)Ns * R);
        tw[off + (r - 1) * Ns + k] = hc((float)std::cos(a), (float)std::sin(a));
      }
  }
  return tw;
}

int main(int argc, char** argv) {
  const long rows = argc > 1 ? atol(argv[1]) : 65536;
  const int reps = argc > 2 ? atoi(argv[2]) : 10;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;

  std::vector<Cx<float>> hx((size_t)rows * H);
  {
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd;
    // only the first 64 and last 64 rows are random on the host; the rest is filled on the device from them
    for (size_t i = 0; i < (size_t)std::min<long>(rows, 128) * H; ++i) hx[i] = hc(nd(rng), nd(rng));
    for (long r = 128; r < rows; ++r) std::copy(hx.begin() + (size_t)(r % 128) * H, hx.begin() + (size_t)(r % 128 + 1) * H, hx.begin() + (size_t)r * H);
  }
  Cx<float>*dx, *dout, *dref, *dtw, *dhalf, *dphase;
  float *dwin, *dmax;
  int* didx;
  CK(hipMalloc(&dx, (size_t)rows * H * 8));
  CK(hipMalloc(&dout, (size_t)rows * N * 8));
  CK(hipMalloc(&dref, (size_t)rows * N * 8));
  CK(hipMemcpy(dx, hx.data(), (size_t)rows * H * 8, hipMemcpyHostToDevice));
  auto tw = twiddles<PL>();
  CK(hipMalloc(&dtw, tw.size() * 8));
  CK(hipMemcpy(dtw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
  std::vector<Cx<float>> half(H), phase(N);
  std::vector<float> win(N);
  for (int k = 0; k < H; ++k) half[k] = hc((float)std::cos(-2 * M_PI * k / N), (float)std::sin(-2 * M_PI * k / N));
  for (int k = 0; k < N; ++k) win[k] = (float)std::exp(-M_PI * 5.0 * k / 5000.0);
  const double pa = 0.7, pb = 0.0085;  // ramp: phase[k] = e^{i (pa + pb k)}
  for (int k = 0; k < N; ++k) phase[k] = hc((float)std::cos(pa + pb * k), (float)std::sin(pa + pb * k));
  CK(hipMalloc(&dhalf, H * 8));
  CK(hipMemcpy(dhalf, half.data(), H * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dphase, N * 8));
  CK(hipMemcpy(dphase, phase.data(), N * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dwin, N * 4));
  CK(hipMemcpy(dwin, win.data(), N * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dmax, rows * 4));
  CK(hipMalloc(&didx, rows * 4));

  PipeArgs<float> A;
  memset(&A, 0, sizeof(A));
  A.in = dx;
  A.out = dout;
  A.window = dwin;
  A.phase = dphase;
  A.tw = dtw;
  A.aux = dhalf;
  A.absmax2 = dmax;
  A.argidx = didx;
  A.in_stride = H;
  A.n_batch = rows;
  A.n = N;
  A.n_in = H;
  A.out_shift = N / 2;
  A.amax_value_only = 1;
  A.scale = (float)(1.0 / std::sqrt((double)N));
  unsigned* dqueue;
  CK(hipMalloc(&dqueue, 256));
  CK(hipMemset(dqueue, 0, 256));
  A.queue = dqueue;
  A.ramp_db = pb;
  for (int q = 0; q < PL::P; ++q) {
    const unsigned base = (2u * PL::NT * q + N / 2) & (N - 1u);
    A.ramp_c[2 * q] = (float)std::cos(pa + pb * base);
    A.ramp_c[2 * q + 1] = (float)std::sin(pa + pb * base);
  }
  A.ramp_e[0] = (float)std::cos(pb);
  A.ramp_e[1] = (float)std::sin(pb);

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double bytes = (double)rows * (H + N) * 8;
  struct Variant {
    std::string name;
    std::function<void()> launch;
    int per_cu;
    std::vector<float> ts;
  };
  std::vector<Variant> vars;
  auto add = [&](const std::string& name, auto kern, size_t lds, int nt, PipeArgs<float> A) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, nt, lds));
    const long blocks = std::min<long>(rows, (long)per_cu * cus);
    vars.push_back({name, [=] { hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(nt), lds, 0, A); }, per_cu, {}});
  };
  auto time_all = [&] {
    for (auto& v : vars) {
      CK(hipMemsetAsync(dmax, 0, rows * 4, 0));
      v.launch();
    }
    CK(hipDeviceSynchronize());
    // steady state: BATCH back-to-back launches of one variant are timed as a whole (what a kernel leaves in the
    // L2 / Infinity Cache -- dirty output lines -- is paid by the next launch, so single interleaved launches of
    // different variants measure each other), `reps` rounds over the variants, median of the per-launch averages
    constexpr int BATCH = 8;
    for (int r = 0; r < reps; ++r)
      for (auto& v : vars) {
        v.launch();  // untimed: brings the caches into this variant's steady state
        CK(hipEventRecord(e0));
        for (int b = 0; b < BATCH; ++b) v.launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        v.ts.push_back(ms / BATCH);
      }
    CK(hipGetLastError());
    for (auto& v : vars) {
      std::sort(v.ts.begin(), v.ts.end());
      const float med = v.ts[v.ts.size() / 2];
      printf("%-56s %d WG/CU  median %8.4f ms  %7.1f GB/s  (min %8.4f  p90 %8.4f)\n", v.name.c_str(), v.per_cu, med,
             bytes / med / 1e6, v.ts[0], v.ts[(v.ts.size() * 9) / 10]);
    }
  };
  // row-wise comparison of dout against dref on a sample (first 64, last 64 rows)
  auto compare = [&](const char* what) {
    const long ns = std::min<long>(rows, 64);
    std::vector<Cx<float>> a((size_t)2 * ns * N), b((size_t)2 * ns * N);
    CK(hipMemcpy(a.data(), dout, (size_t)ns * N * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(a.data() + (size_t)ns * N, dout + (size_t)(rows - ns) * N, (size_t)ns * N * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), dref, (size_t)ns * N * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data() + (size_t)ns * N, dref + (size_t)(rows - ns) * N, (size_t)ns * N * 8, hipMemcpyDeviceToHost));
    double worst = 0;
    for (long r = 0; r < 2 * ns; ++r) {
      double mx = 0, df = 0;
      for (int k = 0; k < N; ++k) {
        const auto &p = a[(size_t)r * N + k], &q = b[(size_t)r * N + k];
        mx = std::max(mx, (double)std::hypot(q.re, q.im));
        df = std::max(df, (double)std::hypot(p.re - q.re, p.im - q.im));
      }
      worst = std::max(worst, df / mx);
    }
    printf("    %-44s max rel diff vs k_zf2 mode 3 (128 rows): %.3e %s\n", what, worst, worst < 2e-6 ? "ok" : "MISMATCH");
  };
  auto host_check = [&](const char* what, bool with_phase) {
    double worst = 0;
    for (long r : {0L, rows - 1}) {
      std::vector<Cx<float>> got(N);
      CK(hipMemcpy(got.data(), dout + (size_t)r * N, N * 8, hipMemcpyDeviceToHost));
      std::vector<std::complex<double>> z(H);
      for (int j = 0; j < H; ++j) z[j] = std::complex<double>(hx[(size_t)r * H + j].re, hx[(size_t)r * H + j].im) * (double)win[j];
      double mx = 0, df = 0;
      for (int m = 0; m < N; m += 37) {
        std::complex<double> acc = 0;
        for (int j = 0; j < H; ++j) acc += z[j] * std::polar(1.0, -2.0 * M_PI * (double)(((long long)j * m) % N) / N);
        acc /= std::sqrt((double)N);
        const int k = (m + N / 2) % N;
        if (with_phase) acc *= std::polar(1.0, pa + pb * k);
        mx = std::max(mx, std::abs(acc));
        df = std::max(df, std::abs(acc - std::complex<double>(got[k].re, got[k].im)));
      }
      worst = std::max(worst, df / mx);
    }
    printf("    %-44s max rel err vs fp64 DFT (2 rows, strided bins): %.3e %s\n", what, worst, worst < 2e-6 ? "ok" : "MISMATCH");
  };
  std::vector<float> refmax;
  auto check_max = [&](const char* what) {
    std::vector<float> m(rows);
    CK(hipMemcpy(m.data(), dmax, rows * 4, hipMemcpyDeviceToHost));
    if (refmax.empty()) { refmax = m; return; }
    double worst = 0;
    for (long r = 0; r < rows; ++r) worst = std::max(worst, std::fabs((double)m[r] - refmax[r]) / refmax[r]);
    printf("    %-44s per-row max |X|^2 vs k_zf2 mode 7: %.3e %s\n", what, worst, worst < 2e-6 ? "ok" : "MISMATCH");
  };
  auto once = [&](auto kern, size_t lds, int nt, PipeArgs<float> A) {  // one checked launch into dout
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, nt, lds));
    CK(hipMemsetAsync(dmax, 0, rows * 4, 0));
    CK(hipMemsetAsync(dout, 0xff, (size_t)rows * N * 8, 0));
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long>(rows, (long)per_cu * cus)), dim3(nt), lds, 0, A);
    CK(hipDeviceSynchronize());
    CK(hipGetLastError());
  };

  using P8 = Zf2PlanOf<4096>::type;  // 512 threads x 8 points, radices 8.8.8.8
  PipeArgs<float> A8 = A;
  {
    auto tw8 = twiddles<P8>();
    Cx<float>* dtw8;
    CK(hipMalloc(&dtw8, tw8.size() * 8));
    CK(hipMemcpy(dtw8, tw8.data(), tw8.size() * 8, hipMemcpyHostToDevice));
    A8.tw = dtw8;
    for (int q = 0; q < P8::P; ++q) {
      const unsigned base = (2u * P8::NT * q + N / 2) & (N - 1u);
      A8.ramp_c[2 * q] = (float)std::cos(pa + pb * base);
      A8.ramp_c[2 * q + 1] = (float)std::sin(pa + pb * base);
    }
  }
  PipeArgs<float> Aref = A;
  Aref.out = dref;
  const size_t lds1 = (size_t)BlockFFT<xm_f2, PL>::lds_elems() * 16 + (size_t)HotTw<float, PL>::mid_size() * 8 + (PL::NT / 64 + 2) * 8;
  const size_t lds2 = (size_t)BlockFFT<xm_f2, PL, xm_ilog2(2 * PL::radix(0))>::lds_elems() * 16 + (size_t)HotTw<float, PL>::mid_size() * 8 + (PL::NT / 64 + 2) * 8;
  const size_t lds81 = (size_t)BlockFFT<xm_f2, P8>::lds_elems() * 16 + (size_t)HotTw<float, P8>::mid_size() * 8 + (P8::NT / 64 + 2) * 8;
  const size_t lds82 = (size_t)BlockFFT<xm_f2, P8, xm_ilog2(2 * P8::radix(0))>::lds_elems() * 16 + (size_t)HotTw<float, P8>::mid_size() * 8 + (P8::NT / 64 + 2) * 8;
  printf("%d CUs, %ld rows x %d -> %d complex64, %d rounds x 8 back-to-back launches per variant; LDS %zu / %zu / %zu / %zu B\n", cus, rows, H, N, reps, lds1, lds2, lds81, lds82);

  // ---- correctness -----------------------------------------------------------------------------
  once(k_zf2<float, PL, 3>, lds1, PL::NT, Aref);
  once(k_zf2<float, PL, 7>, lds1, PL::NT, A);
  check_max("");
  host_check("k_zf2 mode 7 (table)", true);
#define CHECK(NAME, KERN, LDS, NTH, PH, MX) once(KERN, LDS, NTH, (NTH == 512 ? A8 : A)); if (PH) compare(NAME); host_check(NAME, PH); if (MX) check_max(NAME);
  CHECK("k_zf2p 1/0", (k_zf2p<PL, 1, 0>), lds1, PL::NT, false, false)
  CHECK("k_zf2p 3/0 table", (k_zf2p<PL, 3, 0>), lds1, PL::NT, true, false)
  CHECK("k_zf2p 9/0 ramp", (k_zf2p<PL, 9, 0>), lds1, PL::NT, true, false)
  CHECK("k_zf2p 9/NT", (k_zf2p<PL, 9, 2>), lds1, PL::NT, true, false)
  CHECK("k_zf2p 9/L16", (k_zf2p<PL, 9, 1>), lds2, PL::NT, true, false)
  CHECK("k_zf2p 13/NT", (k_zf2p<PL, 13, 2>), lds1, PL::NT, true, true)
  CHECK("k_zf2p 13/L16+NT", (k_zf2p<PL, 13, 3>), lds2, PL::NT, true, true)
  CHECK("k_zf2p 13/NT+Q", (k_zf2p<PL, 13, 10>), lds1, PL::NT, true, true)
  CHECK("k_zf2p 13/L16+NT+Q", (k_zf2p<PL, 13, 11>), lds2, PL::NT, true, true)
  CHECK("k_zf2p P8 13/L16+NT+Q", (k_zf2p<P8, 13, 11>), lds82, P8::NT, true, true)
  CHECK("k_zf2p P8 9/NT", (k_zf2p<P8, 9, 2>), lds81, P8::NT, true, false)
  CHECK("k_zf2p P8 13/L16+NT", (k_zf2p<P8, 13, 3>), lds82, P8::NT, true, true)
  CHECK("k_zf2p P8 13/NT", (k_zf2p<P8, 13, 2>), lds81, P8::NT, true, true)

  // ---- timing ------------------------------------------------------------------------------------
  add("k_zf2 mode 1 (write)", k_zf2<float, PL, 1>, lds1, PL::NT, A);
  add("k_zf2 mode 3 (write+table)", k_zf2<float, PL, 3>, lds1, PL::NT, A);
  add("k_zf2 mode 7 (write+table+max)", k_zf2<float, PL, 7>, lds1, PL::NT, A);
  add("k_zf2 mode 4 (max only, folded window)", k_zf2<float, PL, 4>, lds1, PL::NT, A);
  add("k_zf2p mode 9 (write+ramp), NT", k_zf2p<PL, 9, 2>, lds1, PL::NT, A);
  add("k_zf2p mode 9 (write+ramp), L16+NT", k_zf2p<PL, 9, 3>, lds2, PL::NT, A);
  add("k_zf2p mode 13 (write+ramp+max), NT", k_zf2p<PL, 13, 2>, lds1, PL::NT, A);
  add("k_zf2p mode 13 (write+ramp+max), L16+NT", k_zf2p<PL, 13, 3>, lds2, PL::NT, A);
  add("k_zf2p mode 4 (max only)", k_zf2p<PL, 4, 0>, lds1, PL::NT, A);
  add("k_zf2p mode 4 (max only), L16", k_zf2p<PL, 4, 1>, lds2, PL::NT, A);
  add("k_zf2p P8 (512 thr x 8) mode 9, NT", k_zf2p<P8, 9, 2>, lds81, P8::NT, A8);
  add("k_zf2p P8 mode 13, NT", k_zf2p<P8, 13, 2>, lds81, P8::NT, A8);
  add("k_zf2p P8 mode 13, L16+NT", k_zf2p<P8, 13, 3>, lds82, P8::NT, A8);
  add("k_zf2p P8 mode 4, L16", k_zf2p<P8, 4, 1>, lds82, P8::NT, A8);
  add("k_zf2p mode 9, NT+Q", k_zf2p<PL, 9, 10>, lds1, PL::NT, A);
  add("k_zf2p mode 13, NT+Q", k_zf2p<PL, 13, 10>, lds1, PL::NT, A);
  add("k_zf2p mode 13, L16+NT+Q", k_zf2p<PL, 13, 11>, lds2, PL::NT, A);
  add("k_zf2p mode 13, L16+Q", k_zf2p<PL, 13, 9>, lds2, PL::NT, A);
  add("k_zf2p mode 4, L16+Q", k_zf2p<PL, 4, 9>, lds2, PL::NT, A);
  add("k_zf2p P8 mode 9, NT+Q", k_zf2p<P8, 9, 10>, lds81, P8::NT, A8);
  add("k_zf2p P8 mode 13, NT+Q", k_zf2p<P8, 13, 10>, lds81, P8::NT, A8);
  add("k_zf2p P8 mode 13, L16+NT+Q", k_zf2p<P8, 13, 11>, lds82, P8::NT, A8);
  add("k_zf2p P8 mode 13, L16+Q", k_zf2p<P8, 13, 9>, lds82, P8::NT, A8);
  add("k_zf2p P8 mode 4, L16+Q", k_zf2p<P8, 4, 9>, lds82, P8::NT, A8);
  add("streaming copy, same pattern, no arithmetic", k_rows<0>, 0, 256, A);
  add("streaming copy, same pattern, nt stores", k_rows<2>, 0, 256, A);
  time_all();
  return 0;
}
