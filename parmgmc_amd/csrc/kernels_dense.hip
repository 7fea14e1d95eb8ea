// Coarse-grid exact sampler kernels (gfx950): y = W^T (W b + xi) with W = L^-1, L the lower Cholesky factor.
//
// Replaces the two BLAS trsv calls of PCApply_CholSampler's dense path (reference src/pc_chols.c:220-260,
// :284-288: v = L^-1 x; v += xi; y = L^-T v).  A triangular solve is a chain of N dependent steps -- on a GPU
// that is N kernel-wide synchronisations for a few-thousand-row system -- so the inverse factor is formed once at
// set-up and each sample is two triangular matrix-vector products, one wavefront per row, reading W (stored
// twice, row-major lower and row-major upper=W^T) exactly once: 2 * N(N+1)/2 * 8 bytes, HBM/L2 bound.
#include <hip/hip_runtime.h>
#include "pmg_kernels.h"

namespace {

// out[i] = sum_{k in [lo_i, hi_i)} M[i*n + k] * x[k] (+ add[i]); lower: [0, i+1), upper: [i, n)
template <bool UPPER>
__global__ __launch_bounds__(256) void tri_gemv_kernel(int32_t n, const double *__restrict__ M, const double *__restrict__ x, const double *__restrict__ add, double *__restrict__ out)
{
  const int lane = threadIdx.x & 63;
  const int i    = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  const int     k0  = UPPER ? i : 0, k1 = UPPER ? n : i + 1;
  const double *row = M + (int64_t)i * n;
  double        s   = 0.0;
  for (int k = k0 + lane; k < k1; k += 64) s = fma(row[k], x[k], s);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if (lane == 0) out[i] = add ? s + add[i] : s;
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? 0 : 1; }

} // namespace

extern "C" int pmgk_tri_gemv(int32_t n, int upper, const double *M, const double *x, const double *add, double *out, void *stream)
{
  if (n <= 0) return 0;
  const dim3 block(256), grid((n + 3) / 4);
  if (upper) hipLaunchKernelGGL((tri_gemv_kernel<true>), grid, block, 0, (hipStream_t)stream, n, M, x, add, out);
  else hipLaunchKernelGGL((tri_gemv_kernel<false>), grid, block, 0, (hipStream_t)stream, n, M, x, add, out);
  return launch_status();
}
