// Probe: which physical CUs does a stream created with hipExtStreamCreateWithCUMask get, for a few masks?
// Prints, per mask, the set of (xcc, se, sh, cu) that workgroups of a small kernel ran on.  gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void where(unsigned* out, int spin) {
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
    unsigned xcc = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (3 << 11));
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {
  }
}

int main() {
  int cus = 0;
  CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  printf("CUs %d\n", cus);
  const int blocks = 2048;
  unsigned* d;
  CHECK(hipMalloc(&d, blocks * 8));
  std::vector<unsigned> h(2 * blocks);
  struct M { const char* name; std::vector<uint32_t> bits; };
  std::vector<M> masks;
  masks.push_back({"all", std::vector<uint32_t>(8, 0xffffffffu)});
  masks.push_back({"bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0}});
  masks.push_back({"bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0}});
  masks.push_back({"all but bits 0..7", {0xffffff00u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}});
  masks.push_back({"bits 248..255", {0, 0, 0, 0, 0, 0, 0, 0xff000000u}});
  for (auto& m : masks) {
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, (uint32_t)m.bits.size(), m.bits.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", m.name, hipGetErrorString(e)); continue; }
    CHECK(hipMemsetAsync(d, 0xff, blocks * 8, st));
    hipLaunchKernelGGL(where, dim3(blocks), dim3(64), 0, st, d, 2000);
    CHECK(hipStreamSynchronize(st));
    CHECK(hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int b = 0; b < blocks; ++b) {
      unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
      unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
      per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    size_t total = 0;
    printf("%-20s:", m.name);
    for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += kv.second.size(); }
    printf("  -> %zu distinct CUs\n", total);
    uint32_t got[8] = {0};
    if (hipExtStreamGetCUMask(st, 8, got) == hipSuccess) { printf("   GetCUMask:"); for (int i = 0; i < 8; ++i) printf(" %08x", got[i]); printf("\n"); }
    CHECK(hipStreamDestroy(st));
  }
  return 0;
}
