// Steady-state streaming ceiling of the main pass's traffic (32 KiB row in -> 64 KiB row out, 65,536 rows) and what
// shapes it: workgroups per CU, store pacing, nontemporal hints, load/store interleaving, a VALU stand-in for the FFT.
// Every variant is timed as 8 back-to-back launches (steady state of the L2 / Infinity Cache), several rounds.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/stream_lab.hip -o tools/stream_lab
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

// NT: threads per workgroup (256: 16 points per thread, 512: 8).  AUX: store cache policy (2 = nt).  PACE: after
// every store wait until at most PACE vector-memory operations of the wave are outstanding (-1: no waits).
// DELAY: rounds of 4*P packed FMAs between load and store (stand-in for the FFT's VALU time; 12 ~ 770 packed ops
// at P = 16).  ILV: stores of row i interleaved one by one with the prefetch loads of row i+1.
template <int NT, int AUX, int PACE, int DELAY, bool ILV, int LB = 2>
__global__ __launch_bounds__(NT, LB) void k_rows(const f2* __restrict__ in, f4* __restrict__ out, long R) {
  constexpr int P = 4096 / NT;
  const unsigned t = threadIdx.x;
  f2 x[P];
  auto fetch1 = [&](long r, int q) { x[q] = in[r * 4096 + t + NT * q]; };
  long s = blockIdx.x;
  if (s < R) {
#pragma unroll
    for (int q = 0; q < P; ++q) fetch1(s, q);
  }
  for (; s < R; s += gridDim.x) {
    f4 y[P];
#pragma unroll
    for (int q = 0; q < P; ++q) y[q] = f4{x[q].x, x[q].y, -x[q].y, x[q].x};
    const bool more = s + gridDim.x < R;
    if constexpr (!ILV) {
      if (more) {
#pragma unroll
        for (int q = 0; q < P; ++q) fetch1(s + gridDim.x, q);
      }
    }
    if constexpr (DELAY > 0) {
#pragma unroll 1
      for (int d = 0; d < DELAY; ++d) {
#pragma unroll
        for (int q = 0; q < P; ++q) y[q] = y[q] * 1.0000001f + y[(q + 1) % P] * 1e-9f;
      }
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + s * 4096, 0, 65536, 0x00020000);
#pragma unroll
    for (int q = 0; q < P; ++q) {
      u4 u;
      __builtin_memcpy(&u, &y[q], 16);
      __builtin_amdgcn_raw_buffer_store_b128(u, rs, (NT * q + t) * 16u, 0, AUX);
      if constexpr (ILV) {
        if (more) fetch1(s + gridDim.x, q);
      }
      if constexpr (PACE >= 0) {
        if constexpr (PACE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (PACE == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        if constexpr (PACE == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        if constexpr (PACE == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        if constexpr (PACE == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
    }
  }
}

// Row hand-out by a device-scope counter instead of the static stride (the next row index is fetched one iteration ahead, so
// the atomic's latency is hidden); stamps[b] = s_memrealtime when workgroup b has issued its last store.
// QMODE 0: static stride b + k G; 1: one global counter; 2: one counter per XCD (blockIdx % 8) over that XCD's rows
template <int AUX, int DELAY, int QMODE>
__global__ __launch_bounds__(256, 2) void k_rows_q(const f2* __restrict__ in, f4* __restrict__ out, long R, unsigned* heads,
                                                    unsigned long long* stamps) {
  constexpr int P = 16, NT = 256;
  const unsigned t = threadIdx.x;
  __shared__ long nxt;
  f2 x[P];
  const long G = gridDim.x;
  auto claim = [&](long prev) -> long {  // next row of this workgroup (uniform), or R when the work is done
    if constexpr (QMODE == 0) return prev + G;
    if (t == 0) {
      if constexpr (QMODE == 1) {
        nxt = (long)atomicAdd(heads, 1u) + G;  // the first G rows are the static first round
      } else {
        const unsigned x8 = blockIdx.x & 7;
        const long per = R / 8, got = (long)atomicAdd(heads + 32 * x8, 1u) + G / 8;
        nxt = got < per ? x8 * per + got : R;
      }
    }
    __syncthreads();
    const long v = nxt;
    __syncthreads();
    return v;
  };
  long s = QMODE == 2 ? (long)(blockIdx.x & 7) * (R / 8) + (blockIdx.x >> 3) : (long)blockIdx.x;
  if (s < R) {
#pragma unroll
    for (int q = 0; q < P; ++q) x[q] = in[s * 4096 + t + NT * q];
  }
  while (s < R) {
    f4 y[P];
#pragma unroll
    for (int q = 0; q < P; ++q) y[q] = f4{x[q].x, x[q].y, -x[q].y, x[q].x};
    const long s2 = claim(s);
    if (s2 < R) {
#pragma unroll
      for (int q = 0; q < P; ++q) x[q] = in[s2 * 4096 + t + NT * q];
    }
    if constexpr (DELAY > 0) {
#pragma unroll 1
      for (int d = 0; d < DELAY; ++d) {
#pragma unroll
        for (int q = 0; q < P; ++q) y[q] = y[q] * 1.0000001f + y[(q + 1) % P] * 1e-9f;
      }
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out + s * 4096, 0, 65536, 0x00020000);
#pragma unroll
    for (int q = 0; q < P; ++q) {
      u4 u;
      __builtin_memcpy(&u, &y[q], 16);
      __builtin_amdgcn_raw_buffer_store_b128(u, rs, (NT * q + t) * 16u, 0, AUX);
    }
    s = s2;
  }
  if (t == 0) stamps[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

int main(int argc, char** argv) {
  const long rows = argc > 1 ? atol(argv[1]) : 65536;
  const int rounds = argc > 2 ? atoi(argv[2]) : 5;
  const long in_bytes = rows * 4096 * 8, out_bytes = rows * 8192 * 8;
  void *in, *out;
  CK(hipMalloc(&in, in_bytes));
  CK(hipMalloc(&out, out_bytes));
  CK(hipMemset(in, 1, in_bytes));
  CK(hipMemset(out, 0, out_bytes));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%d CUs; %ld rows: %.2f GiB in, %.2f GiB out; %d rounds x 8 back-to-back launches (median / min per launch)\n", cus, rows,
         in_bytes / 1073741824.0, out_bytes / 1073741824.0, rounds);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Var { std::string name; std::function<void()> launch; std::vector<float> ts; };
  std::vector<Var> vars;
  auto add = [&](const std::string& name, auto kern, int nt, int wg_per_cu) {
    const long blocks = (long)wg_per_cu * cus;
    vars.push_back({name + ", " + std::to_string(wg_per_cu) + " WG/CU", [=] { hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(nt), 0, 0, (const f2*)in, (f4*)out, rows); }, {}});
  };
  add("256 thr, plain", k_rows<256, 0, -1, 0, false>, 256, 2);
  add("256 thr, plain", k_rows<256, 0, -1, 0, false, 1>, 256, 1);
  add("256 thr, plain", k_rows<256, 0, -1, 0, false>, 256, 3);
  add("256 thr, plain", k_rows<256, 0, -1, 0, false>, 256, 4);
  add("256 thr, nt", k_rows<256, 2, -1, 0, false>, 256, 2);
  add("256 thr, nt", k_rows<256, 2, -1, 0, false>, 256, 4);
  add("256 thr, nt, pace 0", k_rows<256, 2, 0, 0, false>, 256, 2);
  add("256 thr, nt, pace 1", k_rows<256, 2, 1, 0, false>, 256, 2);
  add("256 thr, nt, pace 2", k_rows<256, 2, 2, 0, false>, 256, 2);
  add("256 thr, nt, pace 4", k_rows<256, 2, 4, 0, false>, 256, 2);
  add("256 thr, nt, pace 8", k_rows<256, 2, 8, 0, false>, 256, 2);
  add("256 thr, plain, pace 2", k_rows<256, 0, 2, 0, false>, 256, 2);
  add("256 thr, nt, interleaved ld/st", k_rows<256, 2, -1, 0, true>, 256, 2);
  add("256 thr, nt, interleaved ld/st, pace 2", k_rows<256, 2, 2, 0, true>, 256, 2);
  add("256 thr, nt, delay 12", k_rows<256, 2, -1, 12, false>, 256, 2);
  add("256 thr, nt, delay 12, pace 2", k_rows<256, 2, 2, 12, false>, 256, 2);
  add("256 thr, plain, delay 12", k_rows<256, 0, -1, 12, false>, 256, 2);
  add("256 thr, plain, delay 12, pace 2", k_rows<256, 0, 2, 12, false>, 256, 2);
  add("256 thr, nt, delay 24", k_rows<256, 2, -1, 24, false>, 256, 2);
  add("256 thr, nt, delay 24, pace 2", k_rows<256, 2, 2, 24, false>, 256, 2);
  add("512 thr, plain", k_rows<512, 0, -1, 0, false>, 512, 2);
  add("512 thr, nt", k_rows<512, 2, -1, 0, false>, 512, 2);
  add("512 thr, nt", k_rows<512, 2, -1, 0, false, 1>, 512, 1);
  add("512 thr, nt, pace 2", k_rows<512, 2, 2, 0, false>, 512, 2);
  add("512 thr, nt, delay 24", k_rows<512, 2, -1, 24, false>, 512, 2);
  add("512 thr, nt, delay 24, pace 2", k_rows<512, 2, 2, 24, false>, 512, 2);
  add("512 thr, nt, delay 24, pace 0", k_rows<512, 2, 0, 24, false>, 512, 2);
  unsigned* heads;
  unsigned long long* stamps;
  CK(hipMalloc(&heads, 4096));
  CK(hipMalloc(&stamps, 8 * 4096));
  auto addq = [&](const std::string& name, auto kern) {
    const long blocks = 2L * cus;
    vars.push_back({name + ", 2 WG/CU", [=] {
      CK(hipMemsetAsync(heads, 0, 4096, 0));
      hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, 0, (const f2*)in, (f4*)out, rows, heads, stamps); }, {}});
  };
  addq("256 thr, nt, static (stamped)", k_rows_q<2, 0, 0>);
  addq("256 thr, nt, one queue", k_rows_q<2, 0, 1>);
  addq("256 thr, nt, queue per XCD", k_rows_q<2, 0, 2>);
  addq("256 thr, nt, delay 24, static (stamped)", k_rows_q<2, 24, 0>);
  addq("256 thr, nt, delay 24, one queue", k_rows_q<2, 24, 1>);
  addq("256 thr, nt, delay 24, queue per XCD", k_rows_q<2, 24, 2>);
  // finish-time spread of the static schedule
  for (int which = 0; which < 2; ++which) {
    auto& v = vars[vars.size() - 6 + 3 * which];
    v.launch();
    v.launch();
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(2 * cus);
    CK(hipMemcpy(st.data(), stamps, 16 * cus, hipMemcpyDeviceToHost));
    std::vector<double> us;
    const unsigned long long mx = *std::max_element(st.begin(), st.end());
    for (auto x : st) us.push_back((double)(mx - x) / 100.0);  // 100 MHz
    std::sort(us.begin(), us.end());
    printf("static schedule%s: workgroup finish times before the last one (us): median %.1f  p10 %.1f  p90 %.1f  earliest %.1f\n",
           which ? " (delay 24)" : "", us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10], us.back());
    double per_xcd[8] = {0};
    for (int b = 0; b < 2 * cus; ++b) per_xcd[b & 7] += (double)(mx - st[b]) / 100.0 / (2 * cus / 8);
    printf("   mean per blockIdx %% 8:");
    for (int x = 0; x < 8; ++x) printf(" %.1f", per_xcd[x]);
    printf("\n");
  }
  for (auto& v : vars) v.launch();
  CK(hipDeviceSynchronize());
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vars) {
      v.launch();
      CK(hipEventRecord(e0));
      for (int b = 0; b < 8; ++b) v.launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      v.ts.push_back(ms / 8);
    }
  CK(hipGetLastError());
  const double bytes = (double)in_bytes + out_bytes;
  for (auto& v : vars) {
    std::sort(v.ts.begin(), v.ts.end());
    const float med = v.ts[v.ts.size() / 2];
    printf("%-52s %8.4f ms  %7.1f GB/s   (min %8.4f)\n", v.name.c_str(), med, bytes / med / 1e6, v.ts[0]);
  }
  return 0;
}
