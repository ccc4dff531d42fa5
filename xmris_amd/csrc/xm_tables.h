// Host-side generators of the cached device tables (fp64, rounded once by xm_table_get), shared by the
// translation units that launch the fused kernels.
#pragma once
#include "xm_host.h"

#include <vector>

// stage twiddles of a plan: stage s >= 1, input r >= 1, column k < Ns  ->  W_{Ns R}^{r k} at [off_s + (r-1) Ns + k]
template <class PL>
inline void xm_gen_twiddles(int, int, const void*, std::vector<double>& re, std::vector<double>& im) {
  re.assign(PL::tw_size(), 0.0);
  im.assign(PL::tw_size(), 0.0);
  for (int s = 1; s < PL::K; ++s) {
    const int R = PL::radix(s), Ns = PL::ns(s), off = PL::tw_offset(s);
    for (int r = 1; r < R; ++r)
      for (int k = 0; k < Ns; ++k)
        xm_unit((long long)r * k, (long long)Ns * R, -1.0, re[off + (r - 1) * Ns + k], im[off + (r - 1) * Ns + k]);
  }
}

// W_n^k, k < n/2 (the odd-bin rotation of the ">= 2x zero fill" kernels)
inline void xm_gen_half(int n, int, const void*, std::vector<double>& re, std::vector<double>& im) {
  re.resize(n / 2);
  im.resize(n / 2);
  for (int k = 0; k < n / 2; ++k) xm_unit(k, n, -1.0, re[k], im[k]);
}
