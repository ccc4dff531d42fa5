// AsLS baseline kernels (included by xm_api.hip only: they are not templates).
#pragma once
#include "xm_common.h"

// =================================================================================================
// Asymmetric least squares baseline (reference processing/baseline.py:10-40).
// Per spectrum and iteration: solve (W + lam * D'D) z = W y, D = second difference, i.e. a symmetric
// positive definite PENTADIAGONAL system, then re-weight w = p (y > z) + (1 - p) (y < z).
// The band LDL' recurrence is sequential along the spectrum, so the parallelism is across spectra:
// one thread per spectrum on TRANSPOSED data [n, n_batch] (lane-contiguous loads and stores), fp64
// (the condition number reaches ~1e9 for lam = 1e5, p = 1e-3).
// =================================================================================================
template <class TIN, bool COMPLEX_IN>
__global__ __launch_bounds__(256) void k_als_transpose_in(const TIN* __restrict__ in, long long n_batch, int n,
                                                          double* __restrict__ yt) {
  __shared__ double tile[32][33];
  const long long b0 = (long long)blockIdx.y * 32;
  const int j0 = blockIdx.x * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {  // read rows of `in` (contiguous along j)
    const long long b = b0 + r;
    const int j = j0 + threadIdx.x;
    double v = 0.0;
    if (b < n_batch && j < n) v = (double)(COMPLEX_IN ? in[2 * (b * n + j)] : in[b * n + j]);  // real part
    tile[r][threadIdx.x] = v;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {  // write rows of `yt` (contiguous along b)
    const int j = j0 + r;
    const long long b = b0 + threadIdx.x;
    if (b < n_batch && j < n) yt[(long long)j * n_batch + b] = tile[threadIdx.x][r];
  }
}

__global__ __launch_bounds__(256) void k_als_transpose_out(const double* __restrict__ yt, const double* __restrict__ zt,
                                                           long long n_batch, int n, double* __restrict__ out) {
  __shared__ double tile[32][33];
  const long long b0 = (long long)blockIdx.y * 32;
  const int j0 = blockIdx.x * 32;
  for (int r = threadIdx.y; r < 32; r += 8) {
    const int j = j0 + r;
    const long long b = b0 + threadIdx.x;
    double v = 0.0;
    if (b < n_batch && j < n) v = yt[(long long)j * n_batch + b] - zt[(long long)j * n_batch + b];  // y - baseline
    tile[r][threadIdx.x] = v;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += 8) {
    const long long b = b0 + r;
    const int j = j0 + threadIdx.x;
    if (b < n_batch && j < n) out[b * n + j] = tile[threadIdx.x][r];
  }
}

// yt, zt, l1t, l2t, vt: [n][n_batch] doubles.  zt receives the baseline.
__global__ __launch_bounds__(64) void k_als_solve(const double* __restrict__ yt, double* __restrict__ zt,
                                                  double* __restrict__ l1t, double* __restrict__ l2t,
                                                  double* __restrict__ vt, long long n_batch, int n, double lam,
                                                  double p, int n_iter) {
  const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_batch) return;
  const long long nb = n_batch;
  for (int it = 0; it < n_iter; ++it) {
    // forward: factor A = L D L' (unit lower band L: l1 = L[i][i-1], l2 = L[i][i-2]) and solve L D v = W y
    double d1 = 1.0, d2 = 1.0, l1p = 0.0, u1 = 0.0, u2 = 0.0;
    for (int i = 0; i < n; ++i) {
      const long long o = (long long)i * nb + s;
      const double y = yt[o];
      double w = 1.0;
      if (it > 0) {
        const double z = zt[o];
        w = p * (double)(y > z) + (1.0 - p) * (double)(y < z);
      }
      // D'D: diagonal 1,5,6,...,6,5,1; first off-diagonal -2,-4,...,-4,-2; second off-diagonal 1
      const double c0 = (i == 0 || i == n - 1) ? 1.0 : ((i == 1 || i == n - 2) ? 5.0 : 6.0);
      const double a_im1 = (i == 1 || i == n - 1) ? -2.0 * lam : -4.0 * lam;  // A[i][i-1]
      const double di = w + lam * c0;
      const double l2 = i >= 2 ? lam / d2 : 0.0;
      const double l1 = i >= 1 ? (a_im1 - l2 * l1p * d2) / d1 : 0.0;
      const double dd = di - l1 * l1 * d1 - l2 * l2 * d2;
      const double u = w * y - l1 * u1 - l2 * u2;
      l1t[o] = l1;
      l2t[o] = l2;
      vt[o] = u / dd;
      d2 = d1;
      d1 = dd;
      l1p = l1;
      u2 = u1;
      u1 = u;
    }
    // backward: L' z = v
    double z1 = 0.0, z2 = 0.0, l1n = 0.0, l2n = 0.0, l2nn = 0.0;  // l1[i+1], l2[i+1], l2[i+2]
    for (int i = n - 1; i >= 0; --i) {
      const long long o = (long long)i * nb + s;
      const double z = vt[o] - l1n * z1 - l2nn * z2;
      zt[o] = z;
      l2nn = l2n;
      l2n = l2t[o];
      l1n = l1t[o];
      z2 = z1;
      z1 = z;
    }
  }
}

