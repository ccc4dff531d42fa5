// Which subset of a 32 KiB row can the guess kernel read fastest?  One wave per row, `np` pieces of `pw` 16-byte
// words each, piece i at word offset i * ps.  Build: hipcc -O3 --offload-arch=gfx950 tools/guess_patterns.hip -o tools/guess_patterns
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int UN>
__global__ __launch_bounds__(256) void k_sub(const f4* __restrict__ in, long rows, int pw, int ps, int np, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long)gridDim.x * 4;
  const int nsub = pw * np;
  for (long b = wave; b < rows; b += nw) {
    const f4* row = in + b * 2048;
    float acc = 0;
    for (int j0 = lane; j0 < nsub; j0 += 64 * UN) {
      f4 x[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int v = j0 + 64 * u;
        x[u] = v < nsub ? row[(v / pw) * ps + (v % pw)] : f4{0, 0, 0, 0};
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) acc += __builtin_sqrtf(x[u].x * x[u].x + x[u].y * x[u].y) + __builtin_sqrtf(x[u].z * x[u].z + x[u].w * x[u].w);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if (lane == 0) out[b] = acc;
  }
}

int main() {
  const long rows = 65536;
  f4* in;
  float* out;
  CK(hipMalloc(&in, rows * 32768));
  CK(hipMalloc(&out, rows * 4));
  CK(hipMemset(in, 1, rows * 32768));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct P { const char* name; int pw, ps, np; };
  const P pats[] = {{"first 18 KiB contiguous (round 1)", 1152, 1152, 1}, {"first 4 KiB contiguous", 256, 256, 1}, {"first 8 KiB contiguous", 512, 512, 1},
                    {"1 KiB of every 4 KiB over 18 KiB", 64, 256, 5}, {"2 KiB of every 8 KiB over 18 KiB", 128, 512, 3},
                    {"4 KiB at 0 and 12 KiB", 256, 768, 2}, {"2 KiB at 0, 6, 12 KiB", 128, 384, 3}, {"512 B of every 2 KiB over 18 KiB", 32, 128, 9},
                    {"first 2 KiB contiguous", 128, 128, 1}};
  for (int grid : {2048, 4096, 8192, 16384}) {
    printf("grid %d workgroups of 256 (4 rows each)\n", grid);
    for (const P& p : pats) {
      std::vector<float> ts;
      for (int r = 0; r < 12; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_sub<8>, dim3(grid), dim3(256), 0, 0, in, rows, p.pw, p.ps, p.np, out);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) ts.push_back(ms);
      }
      std::sort(ts.begin(), ts.end());
      const double bytes = (double)rows * p.pw * p.np * 16;
      printf("  %-40s %6.1f MB  %8.4f ms  %7.1f GB/s\n", p.name, bytes / 1e6, ts[ts.size() / 2], bytes / ts[ts.size() / 2] / 1e6);
    }
  }
  return 0;
}
